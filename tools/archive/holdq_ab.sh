# Dev measurement (GPU): flight-segment end rule (MCGPU_HOLD_Q, sixteenths of the flying lanes) on the three bench workloads
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-compat"
x() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), 'Ghist/s', round(d['roofline']['kernel_ms_avg'],3), 'ms')"; }
for rep in 1 2; do for q in ${QS:-0 2 4 6 8}; do for wl in ${WORKLOADS:-catphan cirs thorax}; do MCGPU_HOLD_Q=$q $B --workload $wl 2>/dev/null | x "hold_q=$q $wl"; done; done; done
