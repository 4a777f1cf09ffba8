python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1
python bench.py --workload thorax --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat > /dev/null 2>&1
CFGS="24,8,36,12,24 16,8,36,12,24 12,8,36,12,24 8,8,36,12,24 32,8,36,12,24 16,4,36,12,24 16,12,36,12,24 16,8,28,12,24 16,8,44,12,24 16,8,52,12,24 16,8,36,12,16 16,8,36,12,32 16,8,36,12,40 16,8,36,20,24 16,8,36,6,24 12,6,40,12,32"
echo CATPHAN; TUNE_INPUT=/tmp/mcgpu_bench_catphan_512_894/input.in python tools/tune.py $CFGS 2>&1 | tail -18
echo THORAX; TUNE_INPUT=/tmp/mcgpu_bench_thorax_512_894/input.in TUNE_HIST=5e7 python tools/tune.py $CFGS 2>&1 | tail -18
