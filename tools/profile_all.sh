#!/bin/bash
# tools/profile_all.sh <round-tag>: rocprofv3 stats + PMC for the six bench workloads (tools/profile_workload.sh), then the
# per-trial-step issue table.  Run on the GPU box from the repo root; results under gpurun_out/prof_<tag>_<workload>/.
TAG=${1:-r03}
for W in ekf sgp harmonic harmonic_ekf cd_ekf cd_sgp; do
    echo "== $W"
    tools/profile_workload.sh ${TAG}_$W --workload $W > gpurun_out/prof_${TAG}_$W.log 2>&1 || echo "profile of $W failed"
    tail -2 gpurun_out/prof_${TAG}_$W.log | cut -c1-300
done
python tools/issue_table.py gpurun_out/prof_${TAG}_ekf gpurun_out/prof_${TAG}_sgp gpurun_out/prof_${TAG}_cd_sgp gpurun_out/prof_${TAG}_cd_ekf gpurun_out/prof_${TAG}_harmonic gpurun_out/prof_${TAG}_harmonic_ekf > gpurun_out/${TAG}_issue_table.json
