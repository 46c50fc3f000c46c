#!/bin/bash
# tools/ab_workloads.sh <variant.so> ...: filter / smoother kernel times of the six bench workloads with each variant library swapped in for the
# product library (two rounds, one box, one call); the product library is put back on any exit.
LIB=chirpgp_amd/libchirpgp_hip.so
ORIG=$(mktemp /tmp/cgp_orig_XXXXXX.so)
cp "$LIB" "$ORIG"
trap 'cp "$ORIG" "$LIB"; rm -f "$ORIG"' EXIT
for round in 1 2; do
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" "$LIB"; else cp "$ORIG" "$LIB"; fi
    for W in ${WORKLOADS:-ekf sgp harmonic cd_sgp cd_ekf harmonic_ekf}; do
        timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --workload $W --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V round $round $W', 'pass %.3f filter %.3f smoother %.3f' % (r['ms_per_step'], r['kernels']['filter_ms'], r['kernels']['smoother_ms']))" || exit 1
    done
done
done
