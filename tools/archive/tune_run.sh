# Dev measurement (GPU): sweep of the FAST scheduler knobs (thresholds compton, rayleigh, new, flyable_low, swap_batch; HOLD_QS = values of MCGPU_HOLD_Q)
for wl in ${WORKLOADS:-catphan cirs}; do python bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1; done
CFGS=${CFGS:-"24,8,36,12,24 24,8,36,12,32 24,8,36,12,40 32,8,36,12,32 32,8,44,12,32 40,8,44,12,32 32,8,40,12,28 32,12,40,16,32 28,8,40,12,36 32,8,36,20,32 32,8,36,12,48 48,8,48,12,32"}
for q in ${HOLD_QS:-6}; do for wl in ${WORKLOADS:-catphan cirs}; do echo "$wl hold_q=$q"; MCGPU_HOLD_Q=$q TUNE_INPUT=/tmp/mcgpu_bench_${wl}_512_894/input.in python tools/tune.py $CFGS 2>&1 | grep -v amdgpu | tail -14; done; done
