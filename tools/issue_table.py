"""profiles/r02_issue_table.json from the per-workload PMC profiles of tools/profile_workload.sh:

    python tools/issue_table.py gpurun_out/prof_<tag>_{ekf,sgp,cd_sgp,cd_ekf,harmonic} ... > profiles/r02_issue_table.json

Per workload and kernel role (filter / smoother): wave-instructions per trial-step by class (SQ_INSTS_VALU, *_FMA_F64, *_MUL_F64,
*_ADD_F64, *_MFMA_F64), cycles per trial-step (4 x SQ_WAVE_CYCLES / waves-steps).  bench.py turns them into executed
float64 FLOP/s for the roofline of the VALU-bound workloads (all 64 lanes counted, FMA = 2)."""
import json
import os
import sys


def main(dirs):
    out = {}
    shas = set()
    for d in dirs:
        bench = json.loads(open(os.path.join(d, 'bench.json')).read().strip().splitlines()[-1])
        pmc = json.load(open(os.path.join(d, 'pmc.json')))
        wl = {v[0]: k for k, v in __import__('bench').WORKLOADS.items()}[bench['config']['workload']]
        units = bench['config']['batch_per_gpu'] * bench['config']['T']
        roles = {}
        shas.add(pmc.get('_library_sha256'))
        for name, c in pmc.items():
            if name.startswith('_'):
                continue
            role = 'smoother' if 'smooth' in name or 'eks' in name or 'sgps' in name.lower() or 'split_kernel' in name else 'filter'
            g = lambda k: c.get(k, {}).get('mean', 0.0) / units
            roles[role] = {'kernel': name, 'valu': g('SQ_INSTS_VALU'), 'salu': g('SQ_INSTS_SALU'), 'lds': g('SQ_INSTS_LDS'),
                           'fma_f64': g('SQ_INSTS_VALU_FMA_F64'), 'mul_f64': g('SQ_INSTS_VALU_MUL_F64'), 'add_f64': g('SQ_INSTS_VALU_ADD_F64'),
                           'trans_f64': g('SQ_INSTS_VALU_TRANS_F64'), 'mfma_f64': g('SQ_INSTS_VALU_MFMA_F64'),
                           'cycles': 4 * g('SQ_WAVE_CYCLES'), 'waves': c.get('SQ_WAVES', {}).get('mean'),
                           'source': 'profiles/' + os.path.basename(d.rstrip('/')).replace('prof_', '') + '_pmc.json'}
        out[wl] = roles
    # one library for the whole table, or none: bench.py quotes the table only for the library that produced it
    out['_library_sha256'] = shas.pop() if len(shas) == 1 else None
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main(sys.argv[1:])
