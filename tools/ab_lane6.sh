#!/bin/bash
# tools/ab_lane6.sh <variant.so> ...: every one-lane-per-trial launch at 262 144 x 500 -- EKF (means, full), Gauss-Hermite filter (means), eks / cd_eks (full rows),
# eks with selected outputs -- with each variant library swapped in for the product library, one box, one call; the product library is put back on any exit.
LIB=chirpgp_amd/libchirpgp_hip.so
ORIG=$(mktemp /tmp/cgp_orig_XXXXXX.so)
cp "$LIB" "$ORIG"
trap 'cp "$ORIG" "$LIB"; rm -f "$ORIG"' EXIT
for round in 1 2; do
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" "$LIB"; else cp "$ORIG" "$LIB"; fi
    echo "== $V (round $round)"
    timeout -k 10 120 python tools/crlb_probe.py 262144 500 0x4 means 5 2>/dev/null || exit 1
    timeout -k 10 120 python tools/crlb_probe.py 262144 500 0x4 full 5 2>/dev/null || exit 1
    timeout -k 10 120 python tools/crlb_probe.py 262144 500 0 means 3 ghf 2>/dev/null || exit 1
    timeout -k 10 200 python tools/lane_smoothers.py 262144 500 2>/dev/null | grep -a "eks" | grep -a lane | grep -v sgp || exit 1
    timeout -k 10 200 python tools/select_bench.py 2>/dev/null | grep -a "lane" || exit 1
done
done
