# Dev measurement (GPU): FAST kernel ms per 1e8-history launch on the three bench workloads (6 launches each)
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for wl in catphan cirs thorax; do $B --workload $wl 2>/dev/null | x $wl; done
