#!/bin/bash
# tools/profile_kpt.sh <tag>: ns/step + rocprofv3 --kernel-trace --stats of ekf_for_kpt, tile-layout kernels beside the generic one
set -e
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/kpt_$TAG; mkdir -p $OUT
python tools/kpt_time.py 2> /dev/null | tee $OUT/time.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python tools/kpt_time.py --no-mle > /dev/null 2> $OUT/trace.log
cp "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -8 $OUT/kernel_stats.csv | cut -c1-200
