#!/bin/bash
# Quick timing sweep over the BASELINE config shapes (reduced T) and batch sizes; one JSON line per run.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/sweep.jsonl
: > $OUT
run() { echo "# $*" >> $OUT; python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" >> $OUT 2>/dev/null || echo '{"error": true}' >> $OUT; }
run --workload sgp --T 2000
run --workload harmonic --T 2000
run --workload cd_ekf --T 2000
run --workload cd_sgp --batch 512 --T 1000
run --workload ekf --batch 8192 --T 2000
run --workload ekf --batch 65536 --T 500
run --workload ekf --batch 65536 --T 500 --flags 2
python - <<'PY'
import json
for l in open('gpurun_out/sweep.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    d = json.loads(l)
    if 'error' in d: print('   ERROR'); continue
    k = d['kernels']
    print(f"   {d['value']:.3e} steps/s  filter {k['filter_ms']:.2f} ms ({k['filter_GBs']:.0f} GB/s)  smoother {k['smoother_ms']:.2f} ms ({k['smoother_GBs']:.0f} GB/s)")
PY
