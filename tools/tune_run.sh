# Dev measurement (GPU): sweep of the FAST scheduler knobs (thresholds compton, rayleigh, new, flyable_low, swap_batch) on two workloads
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1
python bench.py --workload thorax --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1
CFGS=${CFGS:-"24,8,36,12,24 16,8,36,12,24 8,8,36,12,24 32,8,36,12,24 24,4,36,12,24 24,16,36,12,24 24,8,28,12,24 24,8,44,12,24 24,8,52,12,24 24,8,36,12,16 24,8,36,12,32 24,8,36,12,8 24,8,36,24,24 24,8,36,6,24 16,6,44,12,16 32,12,48,16,32"}
echo CATPHAN; TUNE_INPUT=/tmp/mcgpu_bench_catphan_512_894/input.in python tools/tune.py $CFGS 2>&1 | tail -18
echo THORAX; TUNE_INPUT=/tmp/mcgpu_bench_thorax_512_894/input.in TUNE_HIST=5e7 python tools/tune.py $CFGS 2>&1 | tail -18
echo NOTRADE; MCGPU_NO_SLOT_TRADE=1 TUNE_INPUT=/tmp/mcgpu_bench_catphan_512_894/input.in python tools/tune.py 24,8,36,12,24 2>&1 | tail -2;  MCGPU_NO_SLOT_TRADE=1 TUNE_INPUT=/tmp/mcgpu_bench_thorax_512_894/input.in TUNE_HIST=5e7 python tools/tune.py 24,8,36,12,24 2>&1 | tail -2
