#!/bin/bash
# tools/profile_select.sh <tag>: rocprofv3 evidence for the large-batch eks with selected outputs (262 144 x 500, cgp_lane4.hpp):
# full rows, the marginal's mean + variance alone, E[g(V)] alone -> gpurun_out/select_<tag>/<variant>/{kernel_stats.csv,pmc.json,time.txt}
set -e
TAG=${1:-run}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
for V in full meanvar expect; do
    OUT=$ROOT/gpurun_out/select_$TAG/$V
    mkdir -p "$OUT"
    python tools/select_probe.py $V 5 2> /dev/null | tee "$OUT/time.txt"
    bash tools/pmc_run.sh "$OUT" python tools/select_probe.py $V 3
done
