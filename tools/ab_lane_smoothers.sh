#!/bin/bash
# tools/ab_lane_smoothers.sh <variant.so> ...: the large-batch smoothers (tools/lane_smoothers.py) with each variant library swapped in, one box, one call
LIB=chirpgp_amd/libchirpgp_hip.so
ORIG=$(mktemp /tmp/cgp_orig_XXXXXX.so)
cp "$LIB" "$ORIG"
trap 'cp "$ORIG" "$LIB"; rm -f "$ORIG"' EXIT
for round in 1 2; do
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" "$LIB"; else cp "$ORIG" "$LIB"; fi
    echo "== $V"
    python tools/lane_smoothers.py 262144 500 2>&1 | grep -a "eks" | grep -v sgp || exit 1
done
done
