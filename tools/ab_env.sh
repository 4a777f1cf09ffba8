# Dev measurement (GPU): one library, several environments (knobs that pick a kernel variant at run time), on ONE box.
# usage: VARIANTS="A=1;B=2 C=3" [LIB=build/ab/x.so] [WORKLOADS="catphan cirs"] bash tools/ab_env.sh
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
IFS=';' read -ra V <<< "${VARIANTS:-}"
for rep in 1 2; do for v in "default" "${V[@]}"; do for wl in ${WORKLOADS:-catphan cirs thorax}; do
  if [ "$v" = default ]; then e=""; else e="$v"; fi
  env ${LIB:+MCGPU_AMD_LIB=$PWD/$LIB} $e $B --workload $wl 2>/dev/null | x "[$v] $wl"
done; done; done
