#!/bin/bash
# tools/ab.sh <variant.so> ...: the bench line's kernel times with each variant library swapped in for the product library,
# all on the same box in one call (build/ab/*.so are built in the CPU container; timings only)
cp chirpgp_amd/libchirpgp_hip.so /tmp/orig.so
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" chirpgp_amd/libchirpgp_hip.so; fi
    for i in 1 2; do
        python bench.py --no-cpu-baseline --no-other-configs ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$V', 'pass %.3f filter %.3f smoother %.3f' % (r['ms_per_step'], r['kernels']['filter_ms'], r['kernels']['smoother_ms']))"
    done
done
cp /tmp/orig.so chirpgp_amd/libchirpgp_hip.so
