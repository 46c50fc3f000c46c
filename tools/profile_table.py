"""tools/profile_table.py <tag>: the per-kernel table of profiles/README.md from profiles/<tag>_{workload}_{kernel_stats.csv,bench.json,pmc.json}
and profiles/<tag>_issue_table.json (rocprofv3 averages, bench.py's HIP-event times, instructions and cycles per trial-step, HBM traffic)."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
P = lambda name: os.path.join(ROOT, 'profiles', f'{tag}_{name}')
t = json.load(open(P('issue_table.json')))
rows = (('ekf', '`ekf` = **C2** (1000 x 10^4)', 1e7, 4), ('sgp', '`sgp` = **C3** (1000 x 10^4)', 1e7, 4), ('cd_sgp', '`cd_sgp` = **C4** (512 x 5e4)', 512 * 5e4, 4),
        ('harmonic', '`harmonic` = **C5** (1000 x 10^4, d = 8)', 1e7, 8), ('cd_ekf', '`cd_ekf` (1000 x 10^4)', 1e7, 4), ('harmonic_ekf', '`harmonic_ekf` (1000 x 10^4, d = 8)', 1e7, 8))
print('| workload (B x T) | kernel | rocprofv3 avg (calls; min) | bench.py HIP events | VALU (incl. MFMA) / MFMA / LDS instr per trial-step | cycles per step | HBM traffic per launch / algorithmic |')
print('|---|---|---|---|---|---|---|')
for w, label, units, d in rows:
    st = {r['Name']: (float(r['AverageNs']) / 1e6, float(r['MinNs']) / 1e6, r['Calls']) for r in csv.DictReader(open(P(f'{w}_kernel_stats.csv')))}
    tr = {k: v.get('hbm_bytes_per_launch') for k, v in json.load(open(P(f'{w}_pmc.json'))).items() if not k.startswith('_')}
    b = json.load(open(P(f'{w}_bench.json')))
    for kind, by in (('filter', 8 + 8 * d + 8 * d * d + 8), ('smoother', 2 * (8 * d + 8 * d * d))):
        v = t[w][kind]
        base = v['kernel'].split('<')[0]
        avg, mn, calls = st[[n for n in st if ('::' + base + '(' in n) or ('::' + base + '<' in n)][0]]
        tb = tr[[k for k in tr if k.split('<')[0] == base][0]]
        print(f"| {label if kind == 'filter' else ''} | `{v['kernel']}` | {avg:.3f} ms ({calls} calls; min {mn:.3f}) | {b['kernels'][kind + '_ms']:.3f} ms | "
              f"{v['valu']:.0f} / {v['mfma_f64']:.0f} / {v['lds']:.0f} | {v['cycles']:.0f} | {tb / 1e9:.3f} GB / {by * units / 1e9:.3f} GB = {tb / (by * units):.2f} |")
