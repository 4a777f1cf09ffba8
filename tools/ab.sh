# Dev measurement (GPU): the bench workloads with each engine build under build/ab/*.so, on ONE box (boxes differ by up to 10 %)
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for rep in 1 2; do for lib in build/ab/*.so; do for wl in ${WORKLOADS:-catphan cirs thorax}; do MCGPU_AMD_LIB=$PWD/$lib $B --workload $wl 2>/dev/null | x "$(basename $lib) $wl"; done; done; done
