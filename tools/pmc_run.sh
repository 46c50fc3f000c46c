#!/bin/bash
# tools/pmc_run.sh <outdir> <program> [args...]: rocprofv3 evidence for ONE command (run on the GPU box from the repo root):
# kernel-trace stats, then one --pmc pass per counter group with nothing else enabled -> <outdir>/{kernel_stats.csv,pmc.json}.
# PMC_GROUPS (";"-separated) overrides the counter groups.  The program goes after "--" itself (no env / bash -c hop: the
# profiler's preloaded library has initialised the GPU by then).  pmc.json records the sha256 of the library that ran (tools/parse_pmc.py).
set -e
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- "$@" > "$OUT/trace.out" 2> "$OUT/trace.log"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
DEFAULT_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;GRBM_GUI_ACTIVE"
IFS=';' read -ra GROUPS_ARR <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
i=0
for group in "${GROUPS_ARR[@]}"; do
    i=$((i + 1))
    rocprofv3 --pmc $group -d "$OUT/pmc_$i" -o pmc --output-format csv -- "$@" > /dev/null 2> "$OUT/pmc_$i.log" || echo "pmc group $i ($group) failed"
done
python "$ROOT/tools/parse_pmc.py" "$OUT" > "$OUT/pmc.json"
rm -rf "$OUT"/trace "$OUT"/pmc_[0-9]*/
echo "pmc_run: $OUT done"
