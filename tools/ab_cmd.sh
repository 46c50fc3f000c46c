#!/bin/bash
# tools/ab_cmd.sh "<command>" <variant.so> ...: run <command> with the product library and with each variant library swapped in
# for it (all on one box in one call); the product library is put back on any exit.
LIB=chirpgp_amd/libchirpgp_hip.so
CMD=$1; shift
ORIG=$(mktemp /tmp/cgp_orig_XXXXXX.so)
cp "$LIB" "$ORIG"
trap 'cp "$ORIG" "$LIB"; rm -f "$ORIG"' EXIT
for V in base "$@"; do
    if [ "$V" != base ]; then cp "$V" "$LIB"; else cp "$ORIG" "$LIB"; fi
    echo "== $V"
    bash -c "$CMD" 2>&1 | grep -v amdgpu.ids
done
