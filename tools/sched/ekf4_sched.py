"""Static schedule of the d = 4 EKF step (cgp_mfma4.hpp, H = e_1 form, one trial per wavefront) for a wavefront that has
its SIMD to itself.  The step is written down once as a list of single-instruction operations with their operands; a list
scheduler with the issue costs and result latencies measured on MI355X (tools/ubench/chain_links.hip, mfma_valu_overlap.hip)
orders N consecutive steps as ONE straight-line block, longest remaining path first, and the order is emitted as C++ with a
scheduling barrier after every statement (the compiler keeps the order, allocates registers and inserts the hazard
wait states).  python tools/sched/ekf4_sched.py [--steps 4] [--emit FILE] prints the simulated cycles per step."""
import argparse
import sys

# kind -> (issue cycles, result latency from issue start for a VALU consumer)
COST = {
    'valu': (4, 7),        # v_fma_f64 / v_mul_f64 / v_add_f64: 6.5 cycles dependent
    'cvt': (4, 8),         # v_rndne_f64, v_ldexp_f64, v_cvt_i32_f64: +4 in a chain
    'int': (4, 5),         # 32-bit integer vector op
    'rcp': (8, 19),
    'dpp': (8, 15),        # two v_mov_b32_dpp: 15 after a vector op, 8 after a matrix op (its wait is in the matrix latency)
    'rdl': (8, 18),        # two v_readlane_b32 into an SGPR pair
    'mfma': (16, 35),      # v_mfma_f64_4x4x4: result to a vector op after 35, to another matrix op after 28
    'st': (4, 0), 'lds': (4, 0), 'salu': (4, 4),
}
MFMA_TO_MFMA = 28


class Op:
    def __init__(self, dst, kind, expr, srcs, decl='double'):
        self.dst, self.kind, self.expr, self.srcs, self.decl = dst, kind, expr, srcs, decl


def step_ops(k, part='all'):
    """Operations of step k.  State names carry the step they are valid AFTER: P{k}, ur{k}, uq{k}, th{k}, c{k}, s{k};
    step k reads the *{k-1} values.  `y{k}` is the measurement (SGPR pair), `slotk` its index in the chunk."""
    p, n = k - 1, k
    v = lambda s: s.replace('@', str(n)).replace('#', str(p))
    O = []
    def op(dst, kind, expr, srcs='', decl='double'):
        O.append(Op(v(dst), kind, v(expr), [v(s) for s in srcs.split()], decl))
    op('y@', 'rdl', 'readlane_f64(ychunk, slot + %d)' % (k - 1), '')
    op('u2_@', 'dpp', 'dpp_f64<kQuadBcast2>(uq#)', 'uq#')
    op('m1_@', 'valu', '(-u2_@) * R.log2e', 'u2_@')
    op('kf_@', 'cvt', '__builtin_rint(m1_@)', 'm1_@')
    op('r1_@', 'valu', 'fma(-kf_@, R.ln2hi, -u2_@)', 'kf_@ u2_@')
    op('r_@', 'valu', 'fma(-kf_@, R.ln2lo, r1_@)', 'kf_@ r1_@')
    op('lin_@', 'valu', 'fma(K.ang, u2_@, -th#)', 'u2_@ th#')
    op('hx_@', 'int', '(unsigned)__double2hiint(u2_@) - 0x3FF80000u', 'u2_@', 'unsigned')
    op('accU@', 'int', '(accU# > hx_@ ? accU# : hx_@)', 'accU# hx_@', 'unsigned')
    op('ki_@', 'cvt', '(int)kf_@', 'kf_@', 'int')
    op('r2_@', 'valu', 'r_@ * r_@', 'r_@')
    for i in range(4):
        op(f'a{i}_@', 'valu', f'horner(R.ex[{2 * i + 1}], r_@, R.ex[{2 * i}])', 'r_@')
    op('r4_@', 'valu', 'r2_@ * r2_@', 'r2_@')
    op('b0_@', 'valu', 'horner(a1_@, r2_@, a0_@)', 'a1_@ r2_@ a0_@')
    op('b1_@', 'valu', 'horner(a3_@, r2_@, a2_@)', 'a3_@ r2_@ a2_@')
    op('e_@', 'valu', 'horner(b1_@, r4_@, b0_@)', 'b1_@ r4_@ b0_@')
    op('t_@', 'cvt', '__builtin_amdgcn_ldexp(e_@, ki_@)', 'e_@ ki_@')
    op('t2_@', 'valu', 't_@ * t_@', 't_@')
    for i in range(4):
        op(f'l{i}_@', 'valu', f'horner(R.lq[{2 * i + 1}], t_@, R.lq[{2 * i}])', 't_@')
    op('t4_@', 'valu', 't2_@ * t2_@', 't2_@')
    op('lb0_@', 'valu', 'horner(l1_@, t2_@, l0_@)', 'l1_@ t2_@ l0_@')
    op('lb1_@', 'valu', 'horner(l3_@, t2_@, l2_@)', 'l3_@ t2_@ l2_@')
    op('qa_@', 'valu', 'horner(lb1_@, t4_@, lb0_@)', 'lb1_@ t4_@ lb0_@')
    op('opt_@', 'valu', '1.0 + t_@', 't_@')
    op('rc_@', 'rcp', '__builtin_amdgcn_rcp(opt_@)', 'opt_@')
    op('re_@', 'valu', 'fma(-opt_@, rc_@, 1.0)', 'opt_@ rc_@')
    op('dsp_@', 'valu', 'fma(rc_@, re_@, rc_@)', 'rc_@ re_@')
    op('d_@', 'valu', 'fma(qa_@, t_@, lin_@)', 'qa_@ t_@ lin_@')
    op('d2_@', 'valu', 'd_@ * d_@', 'd_@')
    op('d3_@', 'valu', 'd_@ * d2_@', 'd_@ d2_@')
    op('d4_@', 'valu', 'd2_@ * d2_@', 'd2_@')
    op('cl_@', 'valu', 'fma(d2_@, -0.5, 1.0)', 'd2_@')
    op('sd_@', 'valu', 'fma(d3_@, R.s3, d_@)', 'd3_@ d_@')
    op('cd_@', 'valu', 'fma(d4_@, R.c4, cl_@)', 'd4_@ cl_@')
    op('accD@', 'valu', 'fmax(accD#, fabs(d_@))', 'accD# d_@')
    op('th@', 'valu', 'th# + d_@', 'th# d_@')
    op('ms_@', 'valu', 'sd_@ * -s#', 'sd_@ s#')
    op('mc_@', 'valu', 'c# * sd_@', 'sd_@ c#')
    op('c@', 'valu', 'fma(c#, cd_@, ms_@)', 'c# cd_@ ms_@')
    op('s@', 'valu', 'fma(s#, cd_@, mc_@)', 's# cd_@ mc_@')
    op('j0_@', 'valu', 'fma(K.ksr, s@, K.kk)', 's@')
    op('J_@', 'valu', 'fma(K.kcr, c@, j0_@)', 'c@ j0_@')
    op('kjd_@', 'valu', 'K.kja * dsp_@', 'dsp_@')
    if part == 'head':      # microbenchmark closure: the next step's inputs depend on this head's results
        op('uq@', 'valu', 'fma(J_@, 1e-9, uq#)', 'J_@ uq#')
        op('ur@', 'valu', 'fma(kjd_@, 1e-9, ur#)', 'kjd_@ ur#')
        op('P@', 'valu', 'P# + 0.0', 'P#')
        return O
    if part == 'tail':      # the head's results replaced by cheap stand-ins that still depend on the previous step
        del O[:]
        op('y@', 'rdl', 'readlane_f64(ychunk, slot + %d)' % (k - 1), '')
        op('J_@', 'valu', 'fma(uq#, 1e-9, K.kk)', 'uq#')
        op('kjd_@', 'valu', 'K.kja * 0.5', '')
        op('th@', 'valu', 'th# + 0.0', 'th#')
        op('c@', 'valu', 'c# + 0.0', 'c#')
        op('s@', 'valu', 's# + 0.0', 's#')
        op('accU@', 'int', 'accU# + 0u', 'accU#', 'unsigned')
        op('accD@', 'valu', 'accD# + 0.0', 'accD#')
    op('fr_@', 'mfma', 'mfma4(J_@, ur#, 0.0)', 'J_@ ur#')
    op('fq_@', 'mfma', 'mfma4(ur#, J_@, 0.0)', 'J_@ ur#')
    op('fsw_@', 'dpp', 'dpp_f64<kQuadSwap1>(fq_@)', 'fq_@')
    op('RJ_@', 'valu', 'fma(kjd_@, fsw_@, J_@)', 'kjd_@ fsw_@ J_@')
    op('Q_@', 'mfma', 'mfma4(P#, RJ_@, 0.0)', 'P# RJ_@')
    op('Pa_@', 'dpp', 'dpp_f64<kQuadBcast1>(Q_@)', 'Q_@')
    op('Pp_@', 'mfma', 'mfma4(RJ_@, Q_@, K.Sig)', 'RJ_@ Q_@')
    op('PHq_@', 'mfma', 'mfma4(Pa_@, RJ_@, K.SigHq)', 'Pa_@ RJ_@')
    op('PHr_@', 'dpp', 'dpp_f64<kQuadBcast1>(Pp_@)', 'Pp_@')
    op('sS_@', 'rdl', 'readlane_f64(Pp_@, 17)', 'Pp_@')
    op('S_@', 'valu', 'sS_@ + K.Xi', 'sS_@')
    op('f1_@', 'dpp', 'dpp_f64<kQuadBcast1>(fq_@)', 'fq_@')
    op('in_@', 'valu', 'y@ - f1_@', 'y@ f1_@')
    op('q0_@', 'rcp', '__builtin_amdgcn_rcp(S_@)', 'S_@')
    op('qe_@', 'valu', 'fma(-S_@, q0_@, 1.0)', 'S_@ q0_@')
    op('rS_@', 'valu', 'fma(q0_@, qe_@, q0_@)', 'q0_@ qe_@')
    op('g_@', 'valu', 'rS_@ * in_@', 'rS_@ in_@')
    op('Kn_@', 'valu', 'PHr_@ * -rS_@', 'PHr_@ rS_@')
    op('P@', 'valu', 'fma(Kn_@, PHq_@, Pp_@)', 'Kn_@ PHq_@ Pp_@')
    op('ur@', 'valu', 'fma(PHr_@, g_@, fr_@)', 'PHr_@ g_@ fr_@')
    op('uq@', 'valu', 'fma(PHq_@, g_@, fq_@)', 'PHq_@ g_@ fq_@')
    op('', 'lds', 'park[(slot + %d) * kParkStride] = make_double2(S_@, in_@)' % (k - 1), 'S_@ in_@', None)
    op('', 'st', 'Pfs.store_s(P@, p_off, (unsigned)(t0 + slot + %d) * 128u)' % (k - 1), 'P@', None)
    op('', 'st', 'mfs.store_s(uq@, m_off, (unsigned)(t0 + slot + %d) * 32u)' % (k - 1), 'uq@', None)
    return O


def schedule(ops, live_in, order=None):
    """In-order single issue.  Returns (order, total cycles).  With `order` None: greedy list scheduling, critical path first."""
    prod = {o.dst: o for o in ops if o.dst}
    users = {}
    for o in ops:
        for s in o.srcs:
            users.setdefault(s, []).append(o)
    # longest path to any sink (latency-weighted)
    height = {}
    def h(o):
        if id(o) in height:
            return height[id(o)]
        lat = COST[o.kind][1] if o.dst else COST[o.kind][0]
        best = 0
        for u in users.get(o.dst, []) if o.dst else []:
            best = max(best, h(u))
        height[id(o)] = lat + best
        return height[id(o)]
    for o in ops:
        h(o)
    ready_at = {n: 0 for n in live_in}
    done, out, t = set(), [], 0
    remaining = list(ops)
    def avail(o):
        rt = 0
        for s in o.srcs:
            if s not in ready_at:
                return None
            r = ready_at[s]
            if o.kind == 'mfma' and s in prod and prod[s].kind == 'mfma':
                r -= COST['mfma'][1] - MFMA_TO_MFMA
            if o.kind == 'dpp' and s in prod and prod[s].kind == 'mfma':
                pass
            rt = max(rt, r)
        return rt
    if order is not None:
        for o in order:
            a = avail(o)
            t = max(t, a)
            if o.dst:
                ready_at[o.dst] = t + COST[o.kind][1]
            t += COST[o.kind][0]
        return order, t
    while remaining:
        cands = [(o, avail(o)) for o in remaining]
        cands = [(o, a) for o, a in cands if a is not None]
        now = [(o, a) for o, a in cands if a <= t]
        if now:
            o, a = max(now, key=lambda oa: height[id(oa[0])])
        else:
            o, a = min(cands, key=lambda oa: (oa[1], -height[id(oa[0])]))
            t = a
        if o.dst:
            ready_at[o.dst] = t + COST[o.kind][1]
        t += COST[o.kind][0]
        out.append(o)
        remaining.remove(o)
    return out, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--emit', default=None)
    ap.add_argument('--part', default='all', choices=['all', 'head', 'tail'])
    ap.add_argument('--no-stores', action='store_true')
    ap.add_argument('--source-order', action='store_true', help='emit the operations as written (baseline of the model)')
    a = ap.parse_args()
    ops = []
    for k in range(1, a.steps + 1):
        ops += step_ops(k, a.part)
    if a.no_stores:
        ops = [o for o in ops if o.kind not in ('st', 'lds')]
    live_in = ['P0', 'ur0', 'uq0', 'th0', 'c0', 's0', 'accU0', 'accD0']
    src_order, t_src = schedule(ops, live_in, order=ops)
    order, t = schedule(ops, live_in)
    n = len(ops)
    issue = sum(COST[o.kind][0] for o in ops)
    print(f'{n / a.steps:.0f} operations a step, issue alone {issue / a.steps:.0f} cycles; source order {t_src / a.steps:.0f}, '
          f'list schedule {t / a.steps:.0f} cycles a step', file=sys.stderr)
    if a.emit:
        with open(a.emit, 'w') as f:
            f.write('// generated by tools/sched/ekf4_sched.py --steps %d: do not edit\n' % a.steps)
            for o in (src_order if a.source_order else order):
                if o.dst:
                    f.write(f'const {o.decl} {o.dst} = {o.expr}; SB;\n')
                else:
                    f.write(f'{o.expr}; SB;\n')


if __name__ == '__main__':
    main()
