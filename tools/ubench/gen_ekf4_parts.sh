#!/bin/bash
# tools/ubench/gen_ekf4_parts.sh: (re)generate the straight-line blocks ekf4_parts.hip includes (untracked: they are outputs of
# tools/sched/ekf4_sched.py), then build the microbenchmark.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
mkdir -p tools/ubench/ekf4_parts_gen
for p in all head tail; do
    for o in "" --source-order; do
        python tools/sched/ekf4_sched.py --part $p --no-stores $o --emit tools/ubench/ekf4_parts_gen/$p${o:+_src}.inc
    done
done
hipcc --offload-arch=gfx950 -O3 -fno-fast-math -I chirpgp_amd/csrc -I include -mllvm -amdgpu-mfma-vgpr-form tools/ubench/ekf4_parts.hip -o tools/ubench/ekf4_parts
