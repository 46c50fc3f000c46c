#!/bin/bash
# tools/ab.sh <variant.so> ...: the bench line's kernel times with each variant library swapped in for the product library,
# all on the same box in one call (build/ab/*.so are built in the CPU container; timings only).  The product library is put
# back on ANY exit (trap), so an interrupted run never leaves a variant installed.
LIB=chirpgp_amd/libchirpgp_hip.so
ORIG=$(mktemp /tmp/cgp_orig_XXXXXX.so)
cp "$LIB" "$ORIG"
trap 'cp "$ORIG" "$LIB"; rm -f "$ORIG"' EXIT
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" "$LIB"; else cp "$ORIG" "$LIB"; fi
    python tools/ab_check.py ${CHECK_KIND:-ekf} 2>&1 | tail -1
    for i in 1 2; do
        python bench.py --no-cpu-baseline --no-other-configs ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', 'pass %.3f filter %.3f smoother %.3f' % (r['ms_per_step'], r['kernels']['filter_ms'], r['kernels']['smoother_ms']))"
    done
done
