"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files under <dir>/pmc_*/ -> JSON on stdout.
Adds hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md "HBM": both counters are in KiB and on
gfx950 FETCH_SIZE reports half of a coalesced stream's bytes), and "_library_sha256" = the sha256 of chirpgp_amd/libchirpgp_hip.so as it
lay in the tree the counters were collected from: bench.py quotes a committed profile only for the library that produced it."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys


def short(name):
    name = re.sub(r'\bcgp::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*\)$', '', name)
    return name.replace('> >', '>>')


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row['Kernel_Name'])
                if k.startswith('__amd') or 'at::native' in k:
                    continue
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {}
    for k, counters in acc.items():
        out[k] = {c: {'dispatches': len(v), 'mean': sum(v) / len(v)} for c, v in sorted(counters.items())}
        if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
            out[k]['hbm_bytes_per_launch'] = (2 * out[k]['FETCH_SIZE']['mean'] + out[k]['WRITE_SIZE']['mean']) * 1024
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'chirpgp_amd', 'libchirpgp_hip.so')
    out['_library_sha256'] = hashlib.sha256(open(so, 'rb').read()).hexdigest() if os.path.exists(so) else None
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main(sys.argv[1])
