"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: VGPR / scratch / LDS / occupancy per kernel."""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), capture_output=True, text=True).stdout
        return out.strip().split('\n')
    except Exception:
        return names


def main(path):
    txt = open(path).read()
    blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
    rows = []
    for b in blocks:
        name = b.split('\n')[0].strip()

        def get(k):
            m = re.search(re.escape(k) + r': (\d+)', b)
            return int(m.group(1)) if m else -1
        rows.append((name, get('VGPRs'), get('AGPRs'), get('ScratchSize [bytes/lane]'), get('SGPRs'),
                     get('LDS Size [bytes/block]'), get('Occupancy [waves/SIMD]')))
    names = demangle([r[0] for r in rows])
    for r, n in zip(rows, names):
        n = n.replace('cgp::', '').replace('void ', '')
        print(f'{r[1]:4d} vgpr {r[2]:3d} agpr {r[3]:6d} scratch {r[4]:3d} sgpr {r[5]:6d} lds occ {r[6]}  {n[:120]}')


if __name__ == '__main__':
    main(sys.argv[1])
