#!/bin/bash
# timing ablations of the walk kernel: each variant library is swapped in for the product library (timings only)
cp chirpgp_amd/libchirpgp_hip.so /tmp/orig.so
for A in "$@"; do
    cp build/abl/libabl$A.so chirpgp_amd/libchirpgp_hip.so
    echo "ablate=$A"; tools/ts_prof.sh 1000 ekf 1
done
cp /tmp/orig.so chirpgp_amd/libchirpgp_hip.so
