# Dev measurement (GPU): the two FAST schedulers (MCGPU_FAST_SCHED 0 = per-wave pools, 1 = workgroup-level pool) on ONE box:
# pinned tallies first, then kernel ms per 1e8-history launch on the three bench workloads.  Extra environment per variant after "--".
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for s in 0 1; do echo -n "sched $s pins: "; MCGPU_FAST_SCHED=$s timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "(pinned and not workgroup) or ragged or sharding_and_determinism or beyond_32" 2>&1 | tail -1; done
for rep in 1 2; do for s in 0 1; do for wl in ${WORKLOADS:-catphan cirs thorax}; do MCGPU_FAST_SCHED=$s timeout 300 $B --workload $wl 2>/dev/null | x "sched$s $wl"; done; done; done
if [ -n "${VARIANTS:-}" ]; then
  IFS=';' read -ra V <<< "$VARIANTS"
  for v in "${V[@]}"; do for wl in ${WORKLOADS:-catphan cirs thorax}; do env MCGPU_FAST_SCHED=1 $v timeout 300 $B --workload $wl 2>/dev/null | x "[$v] $wl"; done; done
fi
