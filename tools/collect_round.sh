#!/bin/bash
# End-of-round measurement set (GPU box, repo root): PMC passes first (their summaries are what the bench lines quote as
# roofline.traffic, stamped with the kernel's source hash), then bench lines, rocprofv3 kernel stats -> gpurun_out/<tag>/
# Usage: bash tools/collect_round.sh r02
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r06}; OUT=gpurun_out/$TAG
export TMPDIR=/tmp
cd "$ROOT"
mkdir -p "$OUT"
LIGHT="--steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads --no-reference-arithmetic"
bash tools/pmc_collect.sh $OUT/pmc > /dev/null 2>&1; cp $OUT/pmc/summary.json $OUT/pmc_summary_catphan.json; cp $OUT/pmc/summary.json profiles/pmc_summary_latest.json
for wl in thorax cirs thorax_textured; do bash tools/pmc_collect.sh $OUT/pmc_$wl --workload $wl $LIGHT > /dev/null 2>&1; cp $OUT/pmc_$wl/summary.json $OUT/pmc_summary_$wl.json; cp $OUT/pmc_$wl/summary.json profiles/pmc_summary_$wl.json; done
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
for wl in cirs thorax; do python bench.py --workload $wl --no-workloads > $OUT/bench_line_$wl.json 2> $OUT/bench_line_$wl.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads --no-reference-arithmetic > $OUT/bench_line_under_rocprof.json 2> $OUT/prof.err
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
head -1 $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/bench_kernel_trace_track.csv; grep track_ $(find $OUT/prof -name "*kernel_trace.csv" | head -1) >> $OUT/bench_kernel_trace_track.csv
# the drop-in default: ASCII projection files formatted on the device (kernel stats of a 24-projection scan)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ascii -- python3 tools/scan_ascii.py 24 > $OUT/scan_ascii.log 2> $OUT/scan_ascii.err
cp $(find $OUT/prof_ascii -name "*kernel_stats.csv" | head -1) $OUT/scan_ascii_kernel_stats.csv
python3 tools/scan_ascii.py 200 > $OUT/scan_ascii_200.log 2>&1
BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 8 --warmup 2 > $OUT/bench_line_2_ranks_sharing_one_gpu.json 2> $OUT/two_ranks.err
# the third route of the N > 1 path: projection sharding (no exchange, no collective); RCCL itself refuses two ranks on one device
BENCH_SHARE_GPU=1 BENCH_EXCHANGE=none python bench.py --gpus 2 --steps 8 --warmup 2 > $OUT/bench_line_2_ranks_projection_sharded_one_gpu.json 2> $OUT/two_ranks_none.err
# six ranks on the one GPU: as many processes as the pool admits on a card (eight ranks: in-process, tests/test_exchange.py, tests/test_gpu_dropin.py)
BENCH_SHARE_GPU=1 python bench.py --gpus 6 --steps 8 --warmup 2 > $OUT/bench_line_6_ranks_sharing_one_gpu.json 2> $OUT/six_ranks.err
# the FAST kernel in single precision against the variant with the reference's double-precision sub-steps (mode fast64), one box
python tools/arith_ab.py > $OUT/fast_vs_fast64_ab.txt 2> $OUT/fast_vs_fast64_ab.err
# kernel stats of the tissue workloads, the FAST section statistics (diagnostic library) and the long version of the RNG test
for wl in cirs thorax; do rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -- python3 bench.py --workload $wl --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-compat --no-workloads --no-reference-arithmetic > /dev/null 2> $OUT/prof_$wl.err; cp $(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_$wl.csv; rm -rf $OUT/prof_$wl; done
python tools/fast_stats.py catphan cirs thorax thorax_textured > $OUT/fast_section_stats.txt 2> $OUT/fast_section_stats.err
bash tools/compat_stats.sh > /dev/null 2>&1; cp gpurun_out/compat_stats.txt $OUT/compat_section_stats.txt
MCGPU_RNG_TEST_LOG2=24 python -m pytest tests/test_fast_rng.py -q -m gpu -s 2>&1 | grep -A3 "history ids" > $OUT/fast_rng_statistics_2p24_ids.txt
rm -rf $OUT/prof $OUT/prof_ascii $OUT/pmc/pass* $OUT/pmc_thorax/pass* $OUT/pmc_cirs/pass* $OUT/pmc_thorax_textured/pass*
ls -la $OUT
