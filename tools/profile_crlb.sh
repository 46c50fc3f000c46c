#!/bin/bash
# tools/profile_crlb.sh <tag>: rocprofv3 evidence for the large-batch filters at the CRLB jobs' shape (262 144 x 500, one lane per trial:
# cgp_lane4.hpp) -- EKF with full outputs and with the means alone, Gauss-Hermite sigma-point filter with the means alone -- and the
# round-4 lane kernel (flags 0x14 = one lane per trial | generic kernel) on the same box: kernel-trace stats, then one --pmc pass per
# counter group (tools/pmc_run.sh) -> gpurun_out/crlb_<tag>/<config>/{kernel_stats.csv,pmc.json,time.txt}
set -e
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for CFG in ${CRLB_CONFIGS:-"ekf_large_full:0:full:ekf" "ekf_large_means:0:means:ekf" "ghf_large_means:0:means:ghf" "ekf_large_full_r04kernel:0x14:full:ekf" "ekf_large_means_r04kernel:0x14:means:ekf"}; do
    IFS=: read -r NAME FLAGS WANT METHOD <<< "$CFG"
    OUT=$ROOT/gpurun_out/crlb_$TAG/$NAME
    mkdir -p "$OUT"
    python tools/crlb_probe.py 262144 500 $FLAGS $WANT 5 $METHOD 2> /dev/null | tee "$OUT/time.txt"
    bash tools/pmc_run.sh "$OUT" python tools/crlb_probe.py 262144 500 $FLAGS $WANT 3 $METHOD
done
