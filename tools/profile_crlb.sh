#!/bin/bash
# tools/profile_crlb.sh <tag>: rocprofv3 evidence for the large-batch EKF (262 144 x 500, the CRLB job's shape) in the default launch shape
# (one lane per trial at this size) and with four trials per wavefront forced (ekf4_mfma_x4_kernel<true>):
# kernel-trace stats, then one --pmc pass per counter group (nothing else enabled) -> gpurun_out/crlb_<tag>/{shape}/pmc.json
set -e
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$ROOT"
for SHAPE in ${CRLB_SHAPES:-"default:0" "four_trials_per_wave:0x200"}; do
    NAME=${SHAPE%%:*}; FLAGS=${SHAPE##*:}
    for WANT in full means; do
        OUT=$ROOT/gpurun_out/crlb_$TAG/${NAME}_$WANT
        mkdir -p "$OUT"
        python tools/crlb_probe.py 262144 500 $FLAGS $WANT 5 | tee "$OUT/time.txt"
        rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python tools/crlb_probe.py 262144 500 $FLAGS $WANT 3 > /dev/null 2> "$OUT/trace.log"
        cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
        i=0
        for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
            i=$((i + 1))
            rocprofv3 --pmc $group -d "$OUT/pmc_$i" -o pmc --output-format csv -- python tools/crlb_probe.py 262144 500 $FLAGS $WANT 2 > /dev/null 2> "$OUT/pmc_$i.log" || echo "pmc group $i failed"
        done
        python tools/parse_pmc.py "$OUT" > "$OUT/pmc.json"
        echo "$NAME $WANT done"
    done
done
