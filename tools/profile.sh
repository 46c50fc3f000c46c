#!/bin/bash
# rocprofv3 evidence for the bench command, in one GPU call:   tools/profile.sh <tag>   (run on the GPU box, repo root)
#   gpurun_out/prof_<tag>/kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --steps 10`
#   gpurun_out/prof_<tag>/bench.json         the same command without the profiler (with the CPU baseline)
#   gpurun_out/prof_<tag>/pmc.json           per-kernel means of the counters, one rocprofv3 --pmc pass per group
# Counters are collected in their own passes with nothing but --pmc (no tracing domains), as the pool requires.
set -e
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
python bench.py --steps 10 --warmup 2 --no-other-configs > "$OUT/bench.json" 2> "$OUT/bench.log"
[ -f bench_details.json ] && cp bench_details.json "$OUT/bench_details.json"
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $group -d "$OUT/pmc_$i" -o pmc --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2> "$OUT/pmc_$i.log"
    echo "pmc pass $i ($group) done"
done
python tools/parse_pmc.py "$OUT" > "$OUT/pmc.json"
head -4 "$OUT/kernel_stats.csv"
cat "$OUT/bench.json"
