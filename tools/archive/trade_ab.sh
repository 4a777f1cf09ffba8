B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for rep in 1 2; do for t in 0 1 3; do for wl in catphan cirs thorax; do MCGPU_SLOT_TRADE=$t $B --workload $wl 2>/dev/null | x "trade=$t $wl"; done; done; done
