# Dev check (GPU): every engine build under build/ab/*.so must keep the COMPAT kernel bit-exact against the oracle (small cases),
# then its COMPAT throughput on the bench workloads is timed (bench.py's `compat` leg), twice, on ONE box.
for lib in build/ab/*.so; do echo -n "$(basename $lib) bit-exact: "; MCGPU_AMD_LIB=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "compat_kernel_bit_exact or batching" 2>&1 | tail -1; done
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-workloads --no-fdk"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['compat']['value']/1e9,3), 'Ghist/s compat', round(d['compat']['ms_per_launch'],2), 'ms')"; }
for rep in 1 2; do for lib in build/ab/*.so; do for wl in ${WORKLOADS:-catphan cirs thorax}; do MCGPU_AMD_LIB=$PWD/$lib $B --workload $wl 2>/dev/null | x "$(basename $lib) $wl"; done; done; done
