# A/B on one GPU box: every engine build under build/ab/*.so on the three bench workloads (tools/ab.sh), then the second brick
# level on / off (MCGPU_SUB_BRICKS) with the default library.  Usage: bash tools/ab_second_level.sh
mkdir -p gpurun_out/r03e
bash tools/ab.sh > gpurun_out/r03e/ab.txt 2>&1
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for rep in 1 2; do for wl in catphan cirs thorax; do for sb in 0 1; do MCGPU_SUB_BRICKS=$sb $B --workload $wl 2>/dev/null | x "sub_bricks=$sb $wl"; done; done; done >> gpurun_out/r03e/ab.txt 2>&1
cat gpurun_out/r03e/ab.txt
