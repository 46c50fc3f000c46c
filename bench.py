#!/usr/bin/env python3
"""bench.py -- filter + smoother throughput of the MI355X engine on BASELINE.json's headline configuration.

A "step" is one pass of the hot path over one batch of synthetic input: the filter launch (ys -> mfs, Pfs, nll) and
the smoother launch (mfs, Pfs -> mss, Pss), inputs already resident in HBM.  Default workload = BASELINE config C2:
discrete EKF + EKS of the demos' chirp model (demos/ekfs_mle.py), d = 4, T = 10 000, B = 1000 Monte-Carlo trials per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload kf|ekf|sgp|cd_sgp|cd_ekf|harmonic] [--batch B] [--T T]
                    [--force-dist] [--no-other-configs] [--no-cpu-baseline]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (torch.distributed.run as a child
process, one rank per GPU over RCCL; the parent never touches the GPU); under an external torchrun it checks that
WORLD_SIZE == N.  Trials shard across ranks with no data-path collective; one all_gather of the per-trial final NLL
happens after the timed region and is reported as gather_ms.  Scaling: C2 (`ekf`) is weak by default (B = 1000 per
rank) and also reports the strong figure (B = 1000 in total) under "strong"; C3 / C5 (`sgp`, `harmonic`) are strong
by default ("batch=1000 sharded 8 GPUs", BASELINE.json); `--scaling weak|strong` overrides.  Rank 0 prints ONE JSON line.

After the timed loop of the default workload (C2) the same process runs a few passes of the OTHER BASELINE configurations
(C1 kf + rts, C3 sgp, C4 cd_sgp at 512 x 50 000 per GPU, C5 harmonic) and attaches their kernel times, throughput and
roofline fractions under "other_configs" -- `value` stays C2's.  `--force-dist` takes the RCCL path (init_process_group
('nccl'), barrier, all_reduce, all_gather on device tensors) even with one rank.  The host-CPU baseline (the oracle's C port
built with -march=native and the state dimension fixed, one core AND all cores, best of 3) is reported by rank 0 for every N.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# rocprofv3 --pmc passes of the default command (tools/profile.sh), committed; bench.py quotes its traffic / issue figures
PMC_PROFILE = 'profiles/r06_ekf_pmc.json'
ISSUE_TABLE = 'profiles/r06_issue_table.json'
CRLB_PROFILES = {'ekf_full': 'profiles/r06_ekf_large_full_pmc.json', 'ekf_means': 'profiles/r06_ekf_large_means_pmc.json',
                 'ghf_means': 'profiles/r06_ghf_large_means_pmc.json'}
_SHA = {}


def library_sha256():
    """sha256 of the library this process loads (chirpgp_amd/libchirpgp_hip.so)."""
    if 'lib' not in _SHA:
        import hashlib
        try:
            _SHA['lib'] = hashlib.sha256(open(os.path.join(ROOT, 'chirpgp_amd', 'libchirpgp_hip.so'), 'rb').read()).hexdigest()
        except OSError:
            _SHA['lib'] = None
    return _SHA['lib']


def load_profile(rel):
    """A committed counter profile (tools/parse_pmc.py, tools/issue_table.py) -> (dict or None, reason): the static figures of a profile
    are quoted only for the very library that produced them ("_library_sha256" recorded when the counters were collected); a kernel
    change without a re-profile leaves them null on the bench line instead of stale."""
    try:
        prof = json.load(open(os.path.join(ROOT, rel)))
    except OSError:
        return None, f"{rel}: not committed"
    sha = prof.get('_library_sha256')
    if sha is None or sha != library_sha256():
        return None, (f"{rel}: collected from library {str(sha)[:12]}, this run loads {str(library_sha256())[:12]} -- counters withheld "
                      f"(re-profile with tools/profile_all.sh)")
    return prof, None


def chirp_batch(B, T, seed, dt=1e-3, Xi=0.1, num_harmonics=0, offset=8.0, meow=500.0):
    """Synthetic toy chirp of SURVEY.md 8d: meow frequency law (toymodels.py:226-268) tiled in 3141-step windows,
    constant magnitude 1, y = chirp + sqrt(Xi) N(0, 1); trial i uses numpy default_rng(seed + i).  (offset, meow) = (8, 500) is
    the reference's law (8 - 13 Hz); (3.5, 100) the low-frequency record set of "C2_low_freq" (3.5 - 4.5 Hz)."""
    k = np.arange(T)
    window = 3141
    local = (k % window + 1) * dt
    phase = (k // window) * (offset * window * dt) + meow * np.exp(-5.0 / np.sin(local)) + offset * local
    if num_harmonics == 0:
        clean = np.sin(2 * math.pi * phase)
    else:
        clean = sum(np.sin((h + 1) * 2 * math.pi * phase) for h in range(num_harmonics))
    ys = np.empty((B, T))
    for i in range(B):
        ys[i] = clean + math.sqrt(Xi) * np.random.default_rng(seed + i).standard_normal(T)
    return ys


def make_workload(B, T, seed=0, kind='ekf'):
    """Model at the demos' MLE start point [lam, b, delta, ell, sigma, m0_v] = [0.1, 0.1, 0.1, 1, 1, 7]
    (demos/ekfs_mle.py:39) and B x T synthetic measurements."""
    from chirpgp_amd import models as pm
    from chirpgp_amd.quadratures import SigmaPoints
    params = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
    low = kind == 'ekf_low'                  # C2 on a 3.5 - 4.5 Hz record set, initial frequency state 3.5: every chunk in the common regime
    if low:
        kind, params[5] = 'ekf', 3.5
    wl = dict(kind=kind, dt=1e-3, Xi=0.1, B=B, T=T, low=low)
    if kind == 'kf':
        wl['F'], wl['Sigma'] = frozen_frequency_linear_model(params, wl['dt'])
    if kind in ('harmonic', 'harmonic_ekf'):
        drift, disp, disc, m0, P0, H = pm.build_harmonic_chirp_model(params, 3)
        wl.update(ys=chirp_batch(B, T, seed, num_harmonics=3), sgps=SigmaPoints.cubature(8), d=8)
    else:
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        wl.update(ys=chirp_batch(B, T, seed, offset=3.5, meow=100.0) if low else chirp_batch(B, T, seed),
                  sgps=SigmaPoints.gauss_hermite(4, 3), d=4)
    wl.update(drift=drift, disp=disp, disc=disc, m0=m0, P0=P0, H=H)
    return wl


def frozen_frequency_linear_model(params, dt):
    """BASELINE config C1's linear model: the chirp LCD discretisation with the frequency frozen at its initial value,
    F = blockdiag(exp(-lam dt) Rot(2 pi g(m0_v) dt), M32_F) (models.py:296-301), Sigma = blockdiag(q, q, M32_Sigma)
    (models.py:302-308) -- what `kf` / `rts` run on in the plumbing configuration."""
    from chirpgp_amd import models as pm
    lam, b, delta, ell, sigma, m0_v = (float(x) for x in params)
    th = dt * 2 * math.pi * float(pm.g(m0_v))
    rot = math.exp(-lam * dt) * np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
    Fm, Sm = pm._m32(ell, sigma, dt)
    q = b ** 2 * dt if lam == 0. else b ** 2 / (2 * lam) * (1 - math.exp(-2 * lam * dt))
    return pm._blkdiag([rot, Fm]), pm._blkdiag([q, q, Sm])


def bytes_per_trial_step(d):
    """filter reads y (8 B), writes mf, Pf, nll (8d + 8d^2 + 8); smoother reads and writes 8d + 8d^2."""
    filt = 8 + 8 * d + 8 * d * d + 8
    smooth = 2 * (8 * d + 8 * d * d)
    return filt, smooth


def pmc_traffic(kernel_key, profile=None):
    """(HBM bytes per launch of the named kernel, why not) from a committed PMC profile of the same command (rocprofv3 --pmc FETCH_SIZE /
    --pmc WRITE_SIZE in separate passes; KiB units, FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM")."""
    prof, why = load_profile(profile or PMC_PROFILE)
    if prof is None:
        return None, why
    for name, counters in prof.items():
        if not name.startswith('_') and kernel_key in name and 'hbm_bytes_per_launch' in counters:
            return counters['hbm_bytes_per_launch'], None
    return None, f"{profile or PMC_PROFILE}: no kernel named *{kernel_key}*"


def pmc_issue(kernel_key, units):
    """Instructions and cycles per trial-step of the dominant kernel from the same committed profile (SQ_INSTS_VALU,
    SQ_INSTS_SALU, SQ_WAVE_CYCLES x 4): why a T-serial kernel sits far below the HBM roof at B = 1000."""
    prof, why = load_profile(PMC_PROFILE)
    if prof is None:
        return {"withheld": why}
    for name, c in prof.items():
        if not name.startswith('_') and kernel_key in name and all(k in c for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_WAVES')):
            return {"valu_per_step": c['SQ_INSTS_VALU']['mean'] / units, "salu_per_step": c['SQ_INSTS_SALU']['mean'] / units,
                    "cycles_per_step": 4 * c['SQ_WAVE_CYCLES']['mean'] / units, "waves": c['SQ_WAVES']['mean'],
                    "source": PMC_PROFILE, "library_sha256": prof['_library_sha256']}
    return None


def cpu_baseline(wl, target_seconds=10.0):
    """The oracle's C port timed on the host cores on a bounded sample of the SAME workload: the timed build
    (oracle/port.py: build_native -- `gcc -O3 -march=native -DFIXED_D=d -ffp-contract=fast` of oracle/c/port.c, compiled on this
    machine; the checker build is a different object) on ONE core and on ALL cores, best of 3 after a calibration run each."""
    from oracle import port
    import copy
    k = wl['kind']
    T, d, ys = wl['T'], wl['d'], wl['ys']
    nat = port.native(d)
    all_threads = port.num_threads(nat)
    drift_g = copy.copy(wl['drift'])
    drift_g.gamma = wl['disp'].outer()
    label = {'kf': 'kf+rts', 'ekf': 'EKF+EKS', 'harmonic_ekf': 'EKF+EKS (d=8)', 'sgp': 'sgp_filter+sgp_smoother',
             'harmonic': 'sgp_filter+sgp_smoother (cubature, d=8)', 'cd_sgp': 'cd_sgp_filter+cd_sgp_smoother', 'cd_ekf': 'cd_ekf+cd_eks'}[k]

    bufs = {}

    def once(n, steps, reps=1):
        """`reps` back-to-back filter + smoother passes over n trials x steps, into buffers that are allocated and touched once
        per shape (a fresh np.empty per call would time the kernel's first-touch page faults: 3.4 GB per 1000 x 10 000 pass)."""
        y = np.ascontiguousarray(ys[:n, :steps])
        if (n, steps) not in bufs:
            bufs.clear()
            bufs[(n, steps)] = [np.zeros((n, steps, d)), np.zeros((n, steps, d, d)), np.zeros((n, steps)),
                                np.zeros((n, steps, d)), np.zeros((n, steps, d, d))]
        mfs, Pfs, nl, mss, Pss = bufs[(n, steps)]
        a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], y)
        if k == 'kf':
            from chirpgp_amd import models as pm
            model, sg, fm, sm = pm.linear_cond_m_cov(wl['F'], wl['Sigma']), None, port.F_EKF, port.S_EKS
        elif k in ('ekf', 'harmonic_ekf'):
            model, sg, fm, sm = wl['disc'], None, port.F_EKF, port.S_EKS
        elif k in ('sgp', 'harmonic'):
            model, sg, fm, sm = wl['disc'], wl['sgps'], port.F_SGP, port.S_SGP
        elif k == 'cd_sgp':
            model, sg, fm, sm = drift_g, wl['sgps'], port.F_CD_SGP, port.S_CD_SGP
        else:
            model, sg, fm, sm = drift_g, None, port.F_CD_EKF, port.S_CD_EKS
        t0 = time.perf_counter()
        for _ in range(reps):
            port.filter(fm, model, sg, *a, use=nat, out=(mfs, Pfs, nl))
            port.smoother(sm, model, sg, wl['dt'], mfs, Pfs, use=nat, out=(mss, Pss))
        return time.perf_counter() - t0

    def timed(threads):
        """best of 3 on `threads` cores; the sample is sized from a calibration run to ~target_seconds / 6 per repetition"""
        port.set_num_threads(threads, nat)
        n0 = min(ys.shape[0], threads)
        t_cal = min(T, 2000)
        once(n0, t_cal)                                                # allocates and touches the buffers
        per_unit = once(n0, t_cal) / (n0 * t_cal)                      # seconds per trial-step at this thread count
        budget = (target_seconds / 6.0) / max(per_unit, 1e-12)         # trial-steps per repetition
        reps = 1
        if budget >= n0 * T:                                           # whole records: as many trials as fit, full T,
            n = int(min(ys.shape[0], max(n0, (int(budget / T) // threads) * threads)))
            steps = T
            reps = max(1, int(budget / (n * T)))                       # and the whole sample again while the budget lasts
        else:                                                          # a record is longer than the budget: its first steps
            n, steps = n0, max(64, int(budget / n0))
        once(n, steps)
        best = min(once(n, steps, reps) for _ in range(3))
        # a calibration that caught the thread pool cold undersizes the sample (seen: 0.09 s for 128 cores -> 1.4e7 instead of 5e7):
        # grow it until one repetition lasts a good fraction of the budget
        for _ in range(4):
            if best >= target_seconds / 12.0:
                break
            grow = int(min(8, max(2, (target_seconds / 6.0) / max(best, 1e-6))))
            if steps == T and n < ys.shape[0]:
                n2 = int(min(ys.shape[0], n * grow))
                grow = max(1, grow // max(1, n2 // n))
                n = n2
                once(n, steps)
            reps *= grow
            best = min(once(n, steps, reps) for _ in range(3))
        return n * steps * reps / best, (f"{n} of {ys.shape[0]} trials x {steps} of T={T} steps" + (f" x {reps} passes" if reps > 1 else "")
                                          + f", best of 3 ({best:.2f} s)")

    v1, s1 = timed(1)
    va, sa = timed(all_threads)
    port.set_num_threads(all_threads, nat)
    return {"value": va, "unit": "trial-steps/s", "cores": all_threads, "kind": "port",
            "sample": f"{label}, oracle/c/port.c: {sa}",
            "one_core": {"value": v1, "unit": "trial-steps/s", "cores": 1, "sample": s1},
            "build": "gcc " + " ".join(port.NATIVE_FLAGS) + f" -DFIXED_D={d} (timed build; the checker build is -march=x86-64-v3, runtime d)",
            "best_of": 3}


LINE_LIMIT = 4096          # bytes of the ONE contract line on stdout (round 5's 24.7 KB line was not parsed by the driver)
DETAILS_NAME = 'bench_details.json'


def _finite(x):
    """JSON has no NaN / Infinity: every non-finite float becomes null (recursively), so a strict parser accepts the output."""
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, (np.floating,)):
        return _finite(float(x))
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def _sig(x, digits=6):
    """floats rounded to `digits` significant figures (the line is a report, not a checkpoint)"""
    if isinstance(x, float) and math.isfinite(x) and x != 0.0:
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, list):
        return [_sig(v, digits) for v in x]
    return x


def write_details(result):
    """Everything bench.py measured (other_configs, the 45 C2_spread records, time_split_filters, issue_profile, the long provenance
    strings) as ONE strict-JSON file next to bench.py -- or in the temp directory when the tree is read-only -- and never on stdout.
    Returns the path (None if it could not be written anywhere)."""
    import tempfile
    try:
        text = json.dumps(_finite(result), allow_nan=False, indent=1, default=lambda o: o.item() if hasattr(o, 'item') else str(o))
    except (TypeError, ValueError) as exc:                # never lose the contract line over the side file
        text = json.dumps({"error": f"details not serialisable: {exc}"})
    for folder in (ROOT, os.path.join(ROOT, 'gpurun_out'), tempfile.gettempdir()):
        try:
            path = os.path.join(folder, DETAILS_NAME)
            with open(path, 'w') as f:
                f.write(text)
            return path
        except OSError:
            continue
    return None


def contract_line(result, details_path=None):
    """The ONE line of stdout: the contract's fields, `roofline` and `cpu_baseline` trimmed to their numbers and a short provenance, a
    summary of the regimes and of the other configurations -- bounded by LINE_LIMIT whatever the run measured.  Everything else is in the
    details file."""
    r = _finite(result)
    keep = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "collectives", "hbm_gbs_total", "hbm_frac_per_gpu", "gather_ms")
    line = {k: r[k] for k in keep if k in r}
    rf = r.get("roofline") or {}
    roof = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                   "algorithmic_bytes_per_launch", "waves_per_simd")}
    src = rf.get("traffic_source")
    roof["traffic_source"] = (src[:117] + '...') if isinstance(src, str) and len(src) > 120 else src
    if isinstance(rf.get("hbm"), dict):
        roof["hbm"] = {k: rf["hbm"].get(k) for k in ("achieved", "peak", "unit", "frac")}
    line["roofline"] = roof
    k = r.get("kernels") or {}
    line["kernels"] = {x: k.get(x) for x in ("filter_ms", "smoother_ms", "filter_ms_max_over_ranks", "smoother_ms_max_over_ranks")}
    cb = r.get("cpu_baseline")
    if cb is not None:
        c = {x: cb.get(x) for x in ("value", "unit", "cores", "kind") if x in cb}
        sample = cb.get("sample")
        c["sample"] = (sample[:157] + '...') if isinstance(sample, str) and len(sample) > 160 else sample
        if "error" in cb:
            c["error"] = str(cb["error"])[:160]
        if isinstance(cb.get("one_core"), dict):
            c["one_core_value"] = cb["one_core"].get("value")
        line["cpu_baseline"] = c
        for x in ("gpu_over_cpu", "gpu_over_cpu_one_core"):
            if x in r:
                line[x] = r[x]
    rg = r.get("regimes")
    if rg:
        line["regimes"] = {x: rg.get(x) for x in ("chunks", "high", "common", "low", "mid", "wide", "redone", "checked", "high_share")}
    if isinstance(r.get("strong"), dict):
        line["strong"] = {x: r["strong"].get(x) for x in ("value", "global_batch", "batch_per_gpu", "ms_per_step", "gather_ms")}
    oc = r.get("other_configs")
    if oc:
        summ = {}
        for tag in ("C1", "C2_low_freq", "C3", "C4", "C5"):
            if tag in oc:
                summ[tag] = {"value": oc[tag].get("value"), "filter_ms": oc[tag].get("filter_ms"), "smoother_ms": oc[tag].get("smoother_ms")}
        ts = oc.get("time_split_filters") or {}
        if "C4_per_gpu" in ts and "C4_per_gpu_smoother" in ts:      # C4's pass with BOTH launches time-split with burn-in (explicit calls; approximate to the junctions' mismatch)
            summ["C4_time_split"] = {"filter_ms": ts["C4_per_gpu"].get("time_split_filter_ms"), "smoother_ms": ts["C4_per_gpu_smoother"].get("time_split_smoother_ms"),
                                     "junction_mismatch": max(ts["C4_per_gpu"].get("junction_mismatch") or 0.0, ts["C4_per_gpu_smoother"].get("junction_mismatch") or 0.0)}
        sp = oc.get("C2_spread")
        if sp:
            summ["C2_spread"] = {x: sp.get(x) for x in ("combinations", "value_min", "value_median", "value_max")}
        line["other_configs"] = summ
    line["details"] = details_path
    line = _sig(line)
    text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    for drop in ("other_configs", "strong", "regimes", "kernels", "collectives", "hbm_gbs_total", "hbm_frac_per_gpu", "gather_ms"):
        if len(text) < LINE_LIMIT:                                      # (never reached with today's fields; a guarantee, not a plan)
            break
        line.pop(drop, None)
        text = json.dumps(line, allow_nan=False, separators=(',', ':'))
    if len(text) >= LINE_LIMIT:
        # last resort (a config or provenance string of absurd length): the contract's scalar fields and the two objects' numbers alone --
        # a run never ends without its line
        core = {k: line.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                         "vs_baseline", "dtype", "data")}
        core["config"] = {"workload": str((line.get("config") or {}).get("workload"))[:200]}
        core["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        if "cpu_baseline" in line:
            core["cpu_baseline"] = {k: line["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")}
        core["details"] = str(details_path)[:300]
        text = json.dumps(_sig(core), allow_nan=False, separators=(',', ':'))
    return text.replace('\n', ' ')


WORKLOADS = {
    # kind: (label, default trials, default T, default scaling, which roof bounds it in the large-batch limit)
    'kf': ("C1: linear KF+RTS, frozen-frequency chirp (plumbing)", 1, 1000, 'weak', 'hbm'),
    'ekf': ("C2: discrete EKF+EKS (demos/ekfs_mle.py model)", 1000, 10000, 'weak', 'hbm'),
    'sgp': ("C3: Gauss-Hermite order-3 sgp_filter+sgp_smoother", 1000, 10000, 'strong', 'valu_f64'),
    'cd_sgp': ("C4: cd_sgp_filter+cd_sgp_smoother RK4", 512, 50000, 'weak', 'valu_f64'),
    'cd_ekf': ("cd_ekf+cd_eks RK4", 1000, 10000, 'weak', 'valu_f64'),
    'harmonic': ("C5: 3-harmonic chirp, cubature sgp_filter+sgp_smoother", 1000, 10000, 'strong', 'valu_f64'),
    'harmonic_ekf': ("3-harmonic chirp (d = 8), ekf+eks", 1000, 10000, 'weak', 'valu_f64'),
}
# float64 vector peak: 256 CUs x 4 SIMDs x 16 lanes per cycle x 2 flop x 2.4 GHz (a wave64 v_fma_f64 occupies its SIMD for
# 4 cycles: measured 4.0 cycles per independent instruction, one wave per SIMD, tools/ubench/f64_issue.hip)
F64_VALU_PEAK_TFLOPS = 78.6
N_SIMDS = 1024
# ALGORITHMIC float64 flop per trial-step of filter + smoother (transcendental ~ 30 flop-equivalents), counted
# op by op on the reference's formulas -- what a non-redundant evaluation has to do.  The executed-FLOP figure of the
# valu_f64 roofline counts all 64 lanes of every instruction, i.e. also the work replicated across lanes / MFMA blocks.
ALGORITHMIC_FLOP = {'kf': 1.2e3, 'ekf': 1.5e3, 'sgp': 25e3, 'cd_sgp': 110e3, 'harmonic': 25e3}
# C2_low_freq: C2 on records of 3.5 - 4.5 Hz (frequency state between 1.8 and 4.9 throughout): the headline kernel with every chunk in its common regime
OTHER_CONFIGS = (('C1', 'kf'), ('C2_low_freq', 'ekf_low'), ('C3', 'sgp'), ('C4', 'cd_sgp'), ('C5', 'harmonic'))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='ekf', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None,
                    help='trials: per GPU under weak scaling, in total under strong scaling (default: BASELINE config)')
    ap.add_argument('--T', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the passes of the other BASELINE configurations that follow the timed loop of the default workload')
    ap.add_argument('--other-steps', type=int, default=3, help='timed passes of each other configuration')
    ap.add_argument('--no-spread', action='store_true', help='skip other_configs.C2_spread (45 record sets of the headline shape)')
    ap.add_argument('--force-dist', action='store_true',
                    help='take the RCCL path even with one rank: init_process_group("nccl"), barrier, all_reduce and the final '
                         'all_gather on device tensors at world size 1')
    ap.add_argument('--flags', type=int, default=0, help='CGP_* flag bits forwarded to the engine (e.g. 2 = wave per trial)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default=None,
                    help='weak: --batch trials per rank; strong: --batch trials in total, sharded (default per workload: '
                         'ekf / cd_* weak, sgp / harmonic strong as BASELINE.json words them)')
    ap.add_argument('--strong', action='store_true', help='same as --scaling strong')
    ap.add_argument('--rehearse', action='store_true',
                    help='multi-process dry run on fewer GPUs than ranks: ranks share devices and the collectives go over '
                         'gloo on host copies (RCCL refuses two ranks on one device); timings are then meaningless')
    return ap.parse_args(argv)


def launcher_command(gpus, argv, port):
    """The child process that runs this script on `gpus` ranks of one node (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={int(gpus)}',
            '--master-addr', '127.0.0.1', '--master-port', str(int(port)), os.path.abspath(__file__), *argv]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def valu_issue(workload, kernel, units, ms):
    """Executed float64 vector work of one launch from the committed PMC profile of this workload
    (ISSUE_TABLE, written by tools/issue_table.py from rocprofv3 --pmc passes): wave-instructions by
    class per trial-step, hence executed FLOP (all 64 lanes counted, FMA = 2) and the share of VALU issue slots used."""
    prof, _ = load_profile(ISSUE_TABLE)
    try:
        tab = prof[workload][kernel]
    except (TypeError, KeyError):
        return None
    flop_per_step = 64 * (2 * tab['fma_f64'] + tab['mul_f64'] + tab['add_f64']) + 2 * 256 * tab.get('mfma_f64', 0)
    return {"executed_tflops": flop_per_step * units / (ms * 1e-3) / 1e12, "valu_per_step": tab['valu'],
            "f64_per_step": tab['fma_f64'] + tab['mul_f64'] + tab['add_f64'], "cycles_per_step": tab.get('cycles'),
            "source": tab.get('source', ISSUE_TABLE)}


def roofline_of(kind, B, T, d, filt_ms, smooth_ms, bench_shape=False):
    """The roofline object of one workload from rank 0's own kernel times (HIP events around the C-ABI calls)."""
    label, _, _, _, bound = WORKLOADS[kind]
    bf, bs = bytes_per_trial_step(d)
    units = B * T
    dom = ('filter', filt_ms, bf) if filt_ms >= smooth_ms else ('smoother', smooth_ms, bs)
    achieved = dom[2] * units / (dom[1] * 1e-3) / 1e9 if dom[1] else None
    traffic, why = pmc_traffic('ekf4_mfma' if dom[0] == 'filter' else 'walk4_smoother') if bench_shape else (None, None)
    hbm = {"bound": "hbm", "kernel": dom[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
           "traffic_source": (PMC_PROFILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not re-measured by "
                              "this run; collected from the library this run loads: sha256 " + str(library_sha256())[:16] + ")") if traffic is not None else why,
           "algorithmic_bytes_per_launch": dom[2] * units, "avg_launch_ms": dom[1],
           # one wavefront per trial below ~2.5 trials per SIMD (cgp_api.hip:choose_wave): occupancy of the 1024 SIMDs
           "waves_per_simd": round(B / N_SIMDS, 3) if B < 2560 else None}
    # useful work of the whole pass against the float64 vector peak (SURVEY.md 8d's op counts), next to any executed figure
    algo = ALGORITHMIC_FLOP.get(kind)
    pass_ms = filt_ms + smooth_ms
    algo_tflops = algo * units / (pass_ms * 1e-3) / 1e12 if (algo and pass_ms) else None
    hbm["algorithmic_tflops"] = algo_tflops
    hbm["algorithmic_frac"] = algo_tflops / F64_VALU_PEAK_TFLOPS if algo_tflops is not None else None
    hbm["algorithmic_convention"] = ("SURVEY.md 8d's op count of the REFERENCE's formulas per trial-step, a transcendental priced at 30 flop-equivalents: a "
                                     "convention for useful work that is comparable across rounds and kernels -- an upper bound on it (the engine's polynomials "
                                     "execute fewer operations per transcendental), not a measurement; `frac` beside it counts executed instructions")
    if bound != 'valu_f64':
        return hbm
    # sigma-point / RK4 workloads: 80-220 flop per byte against a machine balance of ~10, so the
    # float64 vector pipe is the roof; the HBM fraction stays beside it.  `frac` = EXECUTED flop (all 64 lanes of every
    # instruction: issue occupancy, replicated work included); `algorithmic_frac` = useful flop of SURVEY.md 8d.
    # (the committed instruction counts were profiled at the workload's DEFAULT batch per GPU: a shard of another size takes other
    # launch shapes -- the time-split smoother doubles the executed work -- so the executed figures are withheld there)
    profiled_shape = B == WORKLOADS[kind][1]
    iss = valu_issue(kind, dom[0], units, dom[1]) if (dom[1] and profiled_shape) else None
    return {"bound": "valu_f64", "kernel": dom[0],
            "achieved": iss["executed_tflops"] if iss else None, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": iss["executed_tflops"] / F64_VALU_PEAK_TFLOPS if iss else None,
            "frac_counts": "executed wave-instructions x 64 lanes (replicated lanes included): issue occupancy, not useful work",
            "algorithmic_tflops": algo_tflops, "algorithmic_frac": hbm["algorithmic_frac"], "algorithmic_convention": hbm["algorithmic_convention"],
            "algorithmic_flop_per_trial_step": algo,
            "executed_flop_source": iss["source"] if iss else (None if profiled_shape else f"withheld: profiled at B = {WORKLOADS[kind][1]} per GPU, this launch has B = {B}"),
            "valu_per_step": iss["valu_per_step"] if iss else None, "f64_per_step": iss["f64_per_step"] if iss else None,
            "cycles_per_step": iss["cycles_per_step"] if iss else None,
            "waves_per_simd": hbm["waves_per_simd"], "avg_launch_ms": dom[1], "traffic": None, "traffic_source": None,
            "hbm": {k: hbm[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}}


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Start the ranks as a child process BEFORE anything here imports torch or touches the GPU; never exec.
        import subprocess
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        sys.exit(subprocess.run(launcher_command(args.gpus, sys.argv[1:], free_port()), env=env).returncode)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch it as '
                         f'`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` '
                         f'or let `python bench.py --gpus {args.gpus}` start the ranks itself')

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL; must be set before HIP initialises
    # stdout carries ONE line, the contract line: whatever a library prints on file descriptor 1 meanwhile (gloo announces its peers
    # there) goes to stderr; the line itself is written to the saved descriptor at the end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU')
    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:                                            # --force-dist: a one-rank group on the loopback
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(free_port()))
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if args.rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    ranks_seen = dist.get_world_size() if use_dist else 1

    def coll(t):
        """Tensor as the process group wants it: HBM for RCCL, a host copy for the gloo rehearsal."""
        return t.cpu() if args.rehearse else t

    from chirpgp_amd import filters_smoothers as fs
    from chirpgp_amd import _engine
    from chirpgp_amd.parallel import shard_bounds

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(kind, mode, batch, T, steps, warmup, kw):
        """`warmup` untimed and `steps` timed passes of workload `kind` under `mode` scaling -> measurements (wall time and
        kernel times are the maximum over ranks)."""
        if mode == 'strong':                          # contiguous shard of the total batch
            lo, hi = shard_bounds(batch, rank, world)
            B, B_total = hi - lo, batch
        else:
            B, B_total = batch, batch * world
        wl = make_workload(max(B, 1), T, seed=1000003 * rank, kind=kind)
        if B == 0:                                    # more ranks than trials: this rank idles through the barriers
            wl['ys'] = wl['ys'][:0]
        ys_dev = torch.from_numpy(wl['ys']).cuda()

        def step():
            k = wl['kind']
            if k == 'kf':
                f = fs.kf(wl['F'], wl['Sigma'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], ys_dev, **kw)
                s = fs.rts(wl['F'], wl['Sigma'], f[0], f[1], **kw)
            elif k in ('ekf', 'harmonic_ekf'):
                f = fs.ekf(wl['disc'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.eks(wl['disc'], f[0], f[1], wl['dt'], **kw)
            elif k in ('sgp', 'harmonic'):
                f = fs.sgp_filter(wl['disc'], wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.sgp_smoother(wl['disc'], wl['sgps'], f[0], f[1], wl['dt'], **kw)
            elif k == 'cd_sgp':
                f = fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], f[0], f[1], wl['dt'], **kw)
            else:
                f = fs.cd_ekf(wl['drift'], wl['disp'], wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev, **kw)
                s = fs.cd_eks(wl['drift'], wl['disp'], f[0], f[1], wl['dt'], **kw)
            return f, s

        # Allocator priming (untimed, before the warm-up steps): the timed loop keeps one result alive while the next
        # step allocates its outputs, so it alternates between two sets of output buffers; two passes whose results are
        # held together make PyTorch's caching allocator own both sets, and no hipMalloc of GB-sized buffers lands in the
        # timed region whatever --warmup is.  Device memory management is not part of the path being measured.
        prime = [step(), step()]
        torch.cuda.synchronize()
        del prime
        for _ in range(warmup):
            step()
        sync()
        # HIP events on the launch stream, recorded immediately around each C-ABI call (kernel duration, not host time)
        events = _engine.kernel_events = []
        t0 = time.perf_counter()
        for _ in range(steps):
            f, s = step()
        sync()
        elapsed = time.perf_counter() - t0
        _engine.kernel_events = None
        regimes = None
        if wl['kind'] == 'ekf' and B > 0:
            # which regimes of its speculative step the filter ran the 64-step chunks of THESE records in: one more, untimed,
            # launch with the context's counters switched on (include/chirpgp_hip.h: cgp_debug_set / cgp_debug_counters)
            _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1)
            _engine.debug_counters(reset=True)
            step()
            regimes = _engine.debug_counters(reset=True)
            _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
            chunks = B * ((T + 63) // 64)
            regimes['chunks'] = chunks
            regimes['high_share'] = regimes['high'] / chunks if chunks else None
            regimes['kernel'] = 'ekf4_mfma_kernel (one trial per wavefront)' if (regimes['high'] + regimes['common'] + regimes['low'] + regimes['mid'] + regimes['redone'] + regimes['checked'] + regimes['wide']) else 'not counted by this launch shape'
        filt = [a.elapsed_time(b) for n, a, b in events if n == 'filter']
        smooth = [a.elapsed_time(b) for n, a, b in events if n == 'smoother']
        mine = [elapsed, float(np.mean(filt)) if filt else 0.0, float(np.mean(smooth)) if smooth else 0.0]
        tmax = coll(torch.tensor(mine, dtype=torch.float64, device='cuda'))
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                # wall and kernel times: the slowest rank's
        elapsed, filt_max, smooth_max = (float(x) for x in tmax.tolist())

        # the single collective of the path: gather the per-trial final NLL -- after the timed region
        gather_ms = None
        if use_dist:
            from chirpgp_amd import parallel
            last = coll(f[2][:, -1].contiguous())
            torch.cuda.synchronize()
            g0 = time.perf_counter()
            out = parallel.all_gather_trials(last, B_total) if mode == 'strong' else parallel.all_gather_trials(last, B * world)
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - g0) * 1e3
            assert out.shape[0] == B_total
        del f, s, ys_dev
        torch.cuda.empty_cache()
        return dict(wl=wl, kind=wl['kind'], T=T, steps=steps, B=B, B_total=B_total, elapsed=elapsed, gather_ms=gather_ms,
                    filt_ms=mine[1], smooth_ms=mine[2], filt_ms_max=filt_max, smooth_ms_max=smooth_max, regimes=regimes)

    def measure_time_split(steps):
        """The time-split filters with burn-in (cgp_filter_time_split) at the shard sizes BASELINE's 8-GPU configurations leave on one
        GPU -- C2 / C3 / C5: 125 x 10 000, C4: 512 x 50 000 -- next to the sequential launch of the same kernel: kernel time, the
        launch's own junction mismatch, the worst difference of its outputs from the sequential ones.  Not the default anywhere."""
        out = {"what": "cgp_filter_time_split: segments = SIMDs // trials wavefronts per trial, each starting `burn_in` steps before its "
                       "piece from (m0, P0); results are as close to the sequential filter's as junction_mismatch says (include/chirpgp_hip.h)"}
        for tag, kind, B, T, segs, burn in (("C2_shard", 'ekf', 125, 10000, 8, 4096), ("C3_shard", 'sgp', 125, 10000, 8, 3008),
                                            ("C5_shard", 'harmonic', 125, 10000, 8, 3008), ("C4_per_gpu", 'cd_sgp', 512, 50000, 2, 3008)):
            wl = make_workload(B, T, seed=1000003 * rank, kind=kind)
            ys_dev = torch.from_numpy(wl['ys']).cuda()
            a = (wl['H'], wl['Xi'], wl['m0'], wl['P0'], wl['dt'], ys_dev)
            if kind == 'ekf':
                run = lambda **kw: fs.ekf(wl['disc'], *a, **kw)
            elif kind == 'cd_sgp':
                run = lambda **kw: fs.cd_sgp_filter(wl['drift'], wl['disp'](None), wl['sgps'], *a, **kw)
            else:
                run = lambda **kw: fs.sgp_filter(wl['disc'], wl['sgps'], *a, **kw)
            res = {}
            for name, kw in (("sequential", {}), ("time_split", dict(time_split=(segs, burn)))):
                r = run(**kw)
                sync()
                events = _engine.kernel_events = []
                for _ in range(steps):
                    r = run(**kw)
                sync()
                _engine.kernel_events = None
                res[name] = (r, float(np.mean([x.elapsed_time(y) for n, x, y in events if n == 'filter'])))
            err = float(_engine.last_junction_error.max())
            worst = max(float((g - s).abs().max() / s.abs().max()) for g, s in zip(res["time_split"][0], res["sequential"][0]))
            out[tag] = {"workload": WORKLOADS[kind][0], "batch_per_gpu": B, "T": T, "segments": segs, "burn_in": burn,
                        "sequential_filter_ms": res["sequential"][1], "time_split_filter_ms": res["time_split"][1],
                        "speedup": res["sequential"][1] / res["time_split"][1], "junction_mismatch": err, "worst_output_difference": worst,
                        "accepted_at_1e-5": bool(err <= 1e-5)}
            if kind == 'cd_sgp':
                # ... and the smoother's counterpart (cgp_smoother_time_split, round 6) on the sequential filter's rows
                fm, fP = res["sequential"][0][0], res["sequential"][0][1]
                sres = {}
                for name, kw in (("sequential", {}), ("time_split", dict(time_split=(segs, burn)))):
                    sr = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], fm, fP, wl['dt'], **kw)
                    sync()
                    events = _engine.kernel_events = []
                    for _ in range(steps):
                        sr = fs.cd_sgp_smoother(wl['drift'], wl['disp'](None), wl['sgps'], fm, fP, wl['dt'], **kw)
                    sync()
                    _engine.kernel_events = None
                    sres[name] = (sr, float(np.mean([x.elapsed_time(y) for n, x, y in events if n == 'smoother'])))
                serr = float(_engine.last_junction_error.max())
                sworst = max(float((g - s).abs().max() / s.abs().max()) for g, s in zip(sres["time_split"][0], sres["sequential"][0]))
                out[tag + "_smoother"] = {"workload": "cd_sgp_smoother (cgp_smoother_time_split)", "batch_per_gpu": B, "T": T, "segments": segs, "burn_in": burn,
                                          "sequential_smoother_ms": sres["sequential"][1], "time_split_smoother_ms": sres["time_split"][1],
                                          "speedup": sres["sequential"][1] / sres["time_split"][1], "junction_mismatch": serr,
                                          "worst_output_difference": sworst, "accepted_at_1e-5": bool(serr <= 1e-5)}
                del sres, sr
            del res, r, ys_dev
            torch.cuda.empty_cache()
        return out

    def measure_crlb(B, T, steps):
        """The reference's only batched use of the path (tetralith/jobs/crlb_ekf.py:59-79, crlb_ghf.py:64-75): filters over B simulated
        chirp-SDE records of T steps, dt = 0.01, filter only -- the EKF with the means alone (what the job keeps) and with full outputs,
        and the Gauss-Hermite order-3 sigma-point filter with the means alone.  One lane per trial (cgp_lane4.hpp); counter traffic from
        the committed rocprofv3 --pmc passes of tools/profile_crlb.sh, quoted only for the library this run loads."""
        from chirpgp_amd import tools
        from chirpgp_amd.models import model_chirp, disc_chirp_lcd
        from chirpgp_amd.quadratures import SigmaPoints
        _, _, m0, P0, H = model_chirp(0.1, 0.1, 1.0, 1.0, 0.1)
        mc = disc_chirp_lcd(0.1, 0.1, 1.0, 1.0)
        gh3 = SigmaPoints.gauss_hermite(4, 3)
        _, yss = tools.simulate_measurements(mc, H, 0.1, m0, P0, 0.01, T, 666 + rank, batch=B, states=False)
        base = {"d": 4, "T": T, "batch_per_gpu": B, "steps": steps, "data": "simulated on the device (cgp_simulate)"}
        ekf = dict(base, workload="CRLB job shape: ekf filter only, chirp LCD model, dt = 0.01 (tetralith/jobs/crlb_ekf.py:59-79)")
        ghf = dict(base, workload="CRLB job shape: sgp_filter (Gauss-Hermite order 3, 81 points) filter only, chirp LCD model, dt = 0.01 "
                                  "(tetralith/jobs/crlb_ghf.py:64-75)", sigma_points=int(gh3.n_points))
        runs = (("means_only", ekf, 'ekf_means', (True, False, False), 8 + 32, lambda w: fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss, want=w)),
                ("full_outputs", ekf, 'ekf_full', (True, True, True), 176, lambda w: fs.ekf(mc, H, 0.1, m0, P0, 0.01, yss, want=w)),
                ("means_only", ghf, 'ghf_means', (True, False, False), 8 + 32, lambda w: fs.sgp_filter(mc, gh3, H, 0.1, m0, P0, 0.01, yss, want=w)))
        for tag, out, prof, want, nbytes, run in runs:
            for _ in range(2):
                r = run(want)
            sync()
            events = _engine.kernel_events = []
            for _ in range(steps):
                r = run(want)
            sync()
            _engine.kernel_events = None
            ms = float(np.mean([a.elapsed_time(b) for n, a, b in events if n == 'filter']))
            t = coll(torch.tensor([ms], dtype=torch.float64, device='cuda'))
            if use_dist:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t[0])
            algo = nbytes * B * T
            traffic, why = pmc_traffic('lane4_filter_kernel', CRLB_PROFILES[prof]) if (B, T) == (262144, 500) else (None, "profiled at 262144 x 500")
            out[tag] = {"filter_ms": ms, "value": B * world * T / (ms * 1e-3), "unit": "trial-steps/s",
                        "algorithmic_bytes_per_trial_step": nbytes, "achieved_GBs": algo / (ms * 1e-3) / 1e9,
                        "hbm_frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_over_algorithmic": traffic / algo if traffic else None,
                        "traffic_source": CRLB_PROFILES[prof] if traffic else why}
            del r
            torch.cuda.empty_cache()
        del yss
        torch.cuda.empty_cache()
        return ekf, ghf

    def measure_spread(steps):
        """How much the headline depends on its data: the C2 pass (ekf + eks, 1000 x 10^4) over 5 base seeds x measurement
        noise Xi in {0.01, 0.1, 1} x frequency offset in {5.5, 8, 20} Hz of the reference's toy chirp (demos/ekfs_mle.py:16-39 with
        toymodels.py:226-268's law; (seed 0, 0.1, 8) is the headline's record set) -- value = trial-steps/s of each combination from the
        wall time of `steps` passes, and the regime counters of its filter launch: chunks kept from the HIGH / common regime of the
        speculative step, repeated, run on the WIDE step (branch-free full-accuracy softplus: records whose frequency state wanders
        below 1.5) and on the checked step."""
        from chirpgp_amd import models as pm
        params = np.array([0.1, 0.1, 0.1, 1., 1., 7.])
        drift, disp, disc, m0, P0, H = pm.build_chirp_model(params)
        B, T, dt = 1000, 10000, 1e-3
        chunks = B * ((T + 63) // 64)
        steps = max(steps, 10)              # (timed like the headline: the one synchronisation amortised over ten passes or more, not three)
        rows = []
        for seed in (0, 1000, 2000, 3000, 4000):
            for Xi in (0.01, 0.1, 1.0):
                for offset in (5.5, 8.0, 20.0):
                    ys = torch.from_numpy(chirp_batch(B, T, seed, dt=dt, Xi=Xi, offset=offset)).cuda()

                    def step():
                        f = fs.ekf(disc, H, Xi, m0, P0, dt, ys)
                        return f, fs.eks(disc, f[0], f[1], dt)
                    keep = [step(), step()]
                    torch.cuda.synchronize()
                    del keep
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        r = step()
                    torch.cuda.synchronize()
                    dt_pass = (time.perf_counter() - t0) / steps
                    _engine.debug_set(_engine.DBG_COUNT_REGIMES, 1)
                    _engine.debug_counters(reset=True)
                    step()
                    rg = _engine.debug_counters(reset=True)
                    _engine.debug_set(_engine.DBG_COUNT_REGIMES, 0)
                    rows.append({"seed": seed, "Xi": Xi, "offset_hz": offset, "value": B * T / dt_pass, "ms_per_pass": dt_pass * 1e3,
                                 "high": rg['high'], "common": rg['common'], "low": rg['low'], "mid": rg['mid'], "redone": rg['redone'], "wide": rg['wide'], "checked": rg['checked'],
                                 "high_left": rg['high_left']})
                    del r, ys
        torch.cuda.empty_cache()
        vals = np.array([r['value'] for r in rows])
        med = float(np.median(vals))
        slow = min(rows, key=lambda r: r['value'])
        return {"what": "C2 pass over 5 seeds x Xi {0.01, 0.1, 1} x frequency offset {5.5, 8, 20} Hz; value = trial-steps/s (wall of the passes)",
                "combinations": len(rows), "steps": steps, "chunks_per_launch": chunks,
                "value_min": float(vals.min()), "value_median": med, "value_max": float(vals.max()),
                "slowest_over_median_time": med / float(vals.min()), "slowest": slow,
                "value_at_headline_set": float(np.median([r['value'] for r in rows if (r['seed'], r['Xi'], r['offset_hz']) == (0, 0.1, 8.0)])),
                "note": "median and slowest are to be read against the line's own value (the reference's records, seed 0, steps passes)",
                "redone_plus_checked_share_max": max((r['redone'] + r['checked']) / chunks for r in rows),
                "redone_plus_checked_share_at_reference_inputs": max((r['redone'] + r['checked']) / chunks for r in rows if (r['Xi'], r['offset_hz']) == (0.1, 8.0)),
                "wide_share_max": max(r['wide'] / chunks for r in rows), "low_share_max": max(r['low'] / chunks for r in rows), "mid_share_max": max(r['mid'] / chunks for r in rows),
                "by_offset_median": {str(o): float(np.median([r['value'] for r in rows if r['offset_hz'] == o])) for o in (5.5, 8.0, 20.0)},
                "by_Xi_median": {str(x): float(np.median([r['value'] for r in rows if r['Xi'] == x])) for x in (0.01, 0.1, 1.0)},
                "rows": rows}

    label, B_default, T_default, scaling_default, bound = WORKLOADS[args.workload]
    scaling = 'strong' if args.strong else (args.scaling or scaling_default)
    batch = args.batch or B_default
    T = args.T or T_default
    kw = dict(flags=args.flags) if args.flags else {}

    m = measure(args.workload, scaling, batch, T, args.steps, args.warmup, kw)
    # C2's north star also asks for the strong figure (1000 trials in total): measured right after, reported beside
    other = measure(args.workload, 'strong', batch, T, args.steps, args.warmup, kw) if (world > 1 and scaling == 'weak' and args.workload == 'ekf') else None
    # The other BASELINE configurations under the same clock: a few passes each after the headline loop, their own default
    # sizes and scaling (C3 / C5: 1000 trials sharded; C4: 512 x 50 000 per GPU; C1: one record per rank).
    others = {}
    crlb = tsplit = spread = None
    default_shape = args.workload == 'ekf' and args.batch is None and args.T is None and not args.flags
    if default_shape and not args.no_other_configs:
        for tag, kind in OTHER_CONFIGS:
            _, Bo, To, mode_o, _ = WORKLOADS[kind if kind != 'ekf_low' else 'ekf']
            others[tag] = measure(kind, mode_o, Bo, To, args.other_steps, 1, {})
        # (a rehearsal's ranks share ONE device's HBM: the CRLB batch is divided among them there)
        crlb = measure_crlb(262144 // world if args.rehearse else 262144, 500, args.other_steps)
        tsplit = measure_time_split(args.other_steps)          # (every rank runs it -- the shard sizes ARE an 8-GPU run's -- rank 0 reports its own)
        if world == 1 and not args.no_spread:
            spread = measure_spread(args.other_steps)

    if use_dist:
        dist.destroy_process_group()       # ranks other than 0 are done: rank 0 times the host CPU with nobody spinning beside it

    if rank == 0:
        wl, B, B_total, elapsed = m['wl'], m['B'], m['B_total'], m['elapsed']
        d = wl['d']
        filt_ms, smooth_ms = m['filt_ms'], m['smooth_ms']
        bf, bs = bytes_per_trial_step(d)
        units = B * T                                   # rank 0's own share (kernel figures are rank 0's)
        total_units = B_total * T                       # all ranks
        total_gbs = (bf + bs) * total_units / (elapsed / args.steps) / 1e9
        bench_shape = args.workload == 'ekf' and B == 1000 and T == 10000 and not args.flags
        roofline = roofline_of(args.workload, B, T, d, filt_ms, smooth_ms, bench_shape)
        result = {
            "metric": "filter+smoother trial-steps/s (batch x T / wall)",
            "value": total_units * args.steps / elapsed,
            "unit": "trial-steps/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": label, "d": d, "T": T, "batch_per_gpu": B, "global_batch": B_total,
                       "parallelism": f"trials sharded x{world}",
                       "sigma_points": int(wl['sgps'].n_points) if args.workload not in ('kf', 'ekf', 'cd_ekf', 'harmonic_ekf') else None},
            "collectives": ("rccl" if not args.rehearse else "gloo (rehearsal)") if use_dist else None,
            "hbm_gbs_total": total_gbs, "hbm_frac_of_peak_total": total_gbs / (HBM_PEAK_GBS * world),
            "hbm_frac_per_gpu": total_gbs / world / HBM_PEAK_GBS,
            "roofline": roofline,
            "kernels": {"filter_ms": filt_ms, "smoother_ms": smooth_ms,
                        "filter_ms_max_over_ranks": m['filt_ms_max'], "smoother_ms_max_over_ranks": m['smooth_ms_max'],
                        "filter_GBs": bf * units / (filt_ms * 1e-3) / 1e9 if filt_ms else None,
                        "smoother_GBs": bs * units / (smooth_ms * 1e-3) / 1e9 if smooth_ms else None},
            "gather_ms": m['gather_ms'],
            # chunks of 64 steps by the regime the filter's speculative step ran them in (counted by one untimed launch on the
            # timed records): the HIGH regime is 6 % faster, so the headline depends on the records' frequency range --
            # other_configs["C2_low_freq"] is the same pass on records that never enter it
            "regimes": m['regimes'],
        }
        if other is not None:
            result["strong"] = {"value": other['B_total'] * T * args.steps / other['elapsed'], "unit": "trial-steps/s",
                                "global_batch": other['B_total'], "batch_per_gpu": other['B'],
                                "ms_per_step": other['elapsed'] / args.steps * 1e3, "gather_ms": other['gather_ms'],
                                "filter_ms": other['filt_ms_max'], "smoother_ms": other['smooth_ms_max']}
        if bench_shape:
            result["issue_profile"] = pmc_issue('ekf4_mfma' if filt_ms >= smooth_ms else 'walk4_smoother', units)
        if others:
            oc = {}
            for tag, o in others.items():
                lab, _, _, mode_o, _ = WORKLOADS[o['kind']]
                do = o['wl']['d']
                bfo, bso = bytes_per_trial_step(do)
                tot = o['B_total'] * o['T']
                per_pass = o['elapsed'] / o['steps']
                if o['wl'].get('low'):
                    lab += " -- record set of 3.5 - 4.5 Hz, initial frequency state 3.5 (all chunks in the common regime)"
                oc[tag] = {"workload": lab, "d": do, "T": o['T'], "batch_per_gpu": o['B'], "global_batch": o['B_total'], "scaling": mode_o,
                           "steps": o['steps'], "filter_ms": o['filt_ms_max'], "smoother_ms": o['smooth_ms_max'],
                           "ms_per_pass": per_pass * 1e3, "value": tot / per_pass, "unit": "trial-steps/s",
                           "hbm_frac_per_gpu": (bfo + bso) * tot / per_pass / 1e9 / world / HBM_PEAK_GBS,
                           "roofline": roofline_of(o['kind'], o['B'], o['T'], do, o['filt_ms'], o['smooth_ms'])}
                if o.get('regimes'):
                    oc[tag]["regimes"] = o['regimes']
            if crlb:
                oc["CRLB_ekf"], oc["CRLB_ghf"] = crlb
            if spread:
                oc["C2_spread"] = spread
            if tsplit:
                oc["time_split_filters"] = tsplit
            result["other_configs"] = oc
        if not args.no_cpu_baseline:
            try:                                        # the GPU figures above are printed whatever happens to the host-side build / run
                result["cpu_baseline"] = cpu_baseline(wl)
            except Exception as exc:                    # noqa: BLE001 -- gcc missing, OpenMP failing, ...: report it, keep the line
                result["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}", "kind": "port", "value": None}
            if result["cpu_baseline"].get("value"):
                result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
                result["gpu_over_cpu_one_core"] = result["value"] / result["cpu_baseline"]["one_core"]["value"]
        details = write_details(result)
        os.write(line_fd, (contract_line(result, details) + '\n').encode())
    os.close(line_fd)


if __name__ == '__main__':
    main()
