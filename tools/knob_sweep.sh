# Dev measurement (GPU): one engine build ($1 = build/ab/<name>.so), scheduler knobs swept through the environment on the given
# workloads ($WORKLOADS).  Each line: "tC,tR,tN,flyable_low,swap_batch,hold_q".
lib=$1; shift
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for cfg in "$@"; do IFS=, read tc tr tn fl sb hq <<< "$cfg"
  for wl in ${WORKLOADS:-catphan thorax}; do
    MCGPU_THRESH_COMPTON=$tc MCGPU_THRESH_RAYLEIGH=$tr MCGPU_THRESH_NEW=$tn MCGPU_FLYABLE_LOW=$fl MCGPU_SWAP_BATCH=$sb MCGPU_HOLD_Q=$hq MCGPU_AMD_LIB=$PWD/$lib $B --workload $wl 2>/dev/null | x "$cfg $wl"
  done
done
