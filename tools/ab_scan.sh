# Dev measurement (GPU): the 894-projection scan with the MetaImage stacks on disk (bench.py's end_to_end leg) with each engine build
# under build/ab/*.so, on ONE box: total seconds, kernel ms per projection, what is left after the last kernel.
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-compat --no-workloads --ascii-projections 0"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read())['end_to_end']; print('$1', 'scan', round(d['seconds_total'],3), 's, kernels', round(d['ms_per_projection_kernels'],3), 'ms/projection, after the last kernel', round(d['drain_after_last_kernel_ms'],1), 'ms,', round(d['histories_per_s_with_stacks']/1e9,3), 'Ghist/s')"; }
for rep in 1 2 3; do for lib in build/ab/*.so; do MCGPU_AMD_LIB=$PWD/$lib $B 2>/dev/null | x "$(basename $lib)"; done; done
