#!/bin/bash
# tools/ab_lane.sh <variant.so> ...: the large-batch EKF (262 144 x 500, full outputs and means only) with each variant library swapped in
# for the product library, on one box in one call; the product library is put back on any exit.
LIB=chirpgp_amd/libchirpgp_hip.so
ORIG=$(mktemp /tmp/cgp_orig_XXXXXX.so)
cp "$LIB" "$ORIG"
trap 'cp "$ORIG" "$LIB"; rm -f "$ORIG"' EXIT
for round in 1 2; do
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" "$LIB"; else cp "$ORIG" "$LIB"; fi
    echo "== $V"
    if [ $round = 1 ]; then python tools/lane4_check.py ekf 4096 2>&1 | grep -a "worst\|Error\|error" | cut -c1-200; fi
    python tools/crlb_probe.py 262144 500 0x4 means 5 || exit 1
    python tools/crlb_probe.py 262144 500 0x4 full 5 || exit 1
done
done
