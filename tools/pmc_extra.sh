#!/bin/bash
# Extra SQ counters for the bench kernels (one rocprofv3 --pmc pass per group, nothing else enabled).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_extra
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
i=0
for group in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
    i=$((i + 1))
    rocprofv3 --pmc $group -d "$OUT/pmc_$i" -o pmc --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$i.log" || echo "group $i failed"
done
python tools/parse_pmc.py "$OUT"
