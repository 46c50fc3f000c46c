#!/bin/bash
# Launch-shape crossover: EKF + EKS at several batch sizes, forced wave-per-trial (flags 2) vs lane-per-trial (flags 4).
cd "$(dirname "$0")/.."
OUT=gpurun_out/sweep_b.jsonl
: > $OUT
for B in 1000 2048 3072 4096 6144 8192 16384; do
  for F in 2 4; do
    echo "# B=$B flags=$F" >> $OUT
    python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload ekf --batch $B --T 2000 --flags $F >> $OUT 2>/dev/null || echo '{"error": true}' >> $OUT
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/sweep_b.jsonl'):
    if l.startswith('#'): print(l.strip(), end='  '); continue
    d = json.loads(l)
    if 'error' in d: print('ERROR'); continue
    k = d['kernels']
    print(f"filter {k['filter_ms']:.2f} ms ({k['filter_GBs']:.0f} GB/s)  smoother {k['smoother_ms']:.2f} ms ({k['smoother_GBs']:.0f} GB/s)")
PY
