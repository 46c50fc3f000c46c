"""Instruction histogram of every loop in one kernel of a gfx950 .s file: tools/loopstat.py file.s kernel_substring"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r'^(\.LBB\d+_\d+):', l))}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
def hist(a, b):
    c = collections.Counter()
    for l in body[a:b + 1]:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if m and not m.group(1).startswith('s_nop'): c[m.group(1)] += 1
    return c
print('kernel lines', len(body))
for a, b, lab in sorted(set(loops)):
    c = hist(a, b)
    tot = sum(c.values())
    f64 = sum(v for k, v in c.items() if k.endswith('f64') or 'f64_e' in k)
    print(f'loop {lab}: lines {a}-{b}: {tot} instr, f64 {f64}, dpp {sum(v for k,v in c.items() if "dpp" in k)}, ds {sum(v for k,v in c.items() if k.startswith("ds_"))}, vmem {sum(v for k,v in c.items() if k.startswith("global_"))}, salu {sum(v for k,v in c.items() if k.startswith("s_"))}')
