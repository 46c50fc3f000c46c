#!/bin/bash
# tools/collect_profiles.sh <tag>: copy what tools/profile_all.sh, profile_crlb.sh and profile_kpt.sh left under gpurun_out/ (scratch) into
# profiles/ (tracked) under the names bench.py and profiles/README.md use: <tag>_<workload>_{kernel_stats.csv,bench.json,pmc.json},
# <tag>_issue_table.json, <tag>_<crlb config>_{kernel_stats.csv,pmc.json}, <tag>_kpt_{kernel_stats.csv,time.txt}.
set -e
TAG=${1:-r05}
cd "$(dirname "$0")/.."
for W in ekf sgp harmonic harmonic_ekf cd_ekf cd_sgp; do
    D=gpurun_out/prof_${TAG}_$W
    [ -d $D ] || continue
    cp $D/kernel_stats.csv profiles/${TAG}_${W}_kernel_stats.csv
    cp $D/bench.json profiles/${TAG}_${W}_bench.json
    [ -f $D/bench_details.json ] && cp $D/bench_details.json profiles/${TAG}_${W}_bench_details.json
    cp $D/pmc.json profiles/${TAG}_${W}_pmc.json
done
[ -f gpurun_out/${TAG}_issue_table.json ] && cp gpurun_out/${TAG}_issue_table.json profiles/${TAG}_issue_table.json
for D in gpurun_out/crlb_$TAG/*/ gpurun_out/select_$TAG/*/; do
    [ -d "$D" ] || continue                      # (the glob stays literal where the profile script has not been run: ADVICE r5)
    N=$(basename "$D")
    case "$D" in *select_*) N=select_$N;; esac
    cp "$D/kernel_stats.csv" profiles/${TAG}_${N}_kernel_stats.csv
    cp "$D/pmc.json" profiles/${TAG}_${N}_pmc.json
    [ -f "$D/time.txt" ] && cp "$D/time.txt" profiles/${TAG}_${N}_time.txt
done
if [ -d gpurun_out/kpt_$TAG ]; then
    cp gpurun_out/kpt_$TAG/kernel_stats.csv profiles/${TAG}_kpt_kernel_stats.csv
    cp gpurun_out/kpt_$TAG/time.txt profiles/${TAG}_kpt_time.txt
fi
grep -l _library_sha256 profiles/${TAG}_*.json | while read f; do python3 -c "import json,sys; print(json.load(open('$f')).get('_library_sha256','')[:16], '$f')"; done
sha256sum chirpgp_amd/libchirpgp_hip.so | cut -c1-16
