"""AUTO's dispatch table, row by row, on both sides of every threshold ("Next round" 2).

`qoc_create` (csrc/qoc_engine.hip, `plan_for`) picks path, kernel family, chunk count and sweep kernels from the problem shape and the
batch; the thresholds are measured numbers.  A threshold edit must not silently route a shape to a kernel no test runs, so this file
  (1) restates the table of DESIGN.md section 4 as ordered RULES in Python (`expected_plan`), independent of the C++ text,
  (2) walks batch sizes {1, 2, 4, 5, 6, 7, 8, 9, 16, 17, 31, 32, 63, 64, 111, 112} x n in {16, 17, 32, 33, 48, 49, 64} x k in {4, 5, 6, 8} x
      {no state regulariser, forbidden level} in unitary mode (pulse lengths chosen so that seeds x slices falls on either side of the
      latency-mode limits -- 4608 / 4096 for n <= 16, 512 ceil(n / 4) / min(4096, 768 ceil(n / 4)) for 16 < n <= 32, 16384, 4096 --), and the state-transfer routes either side of 48 / 112 control sets,
  (3) creates the AUTO engine for each (1240 unitary rows), asserts the plan it reports (`qoc_plan_describe`) is the expected one, and -- for the
      smallest and the largest batch that resolve to each distinct plan of a shape, i.e. on both sides of every threshold that changes the
      kernels -- checks the first and the last control set of the batch against the CPU oracle (reference: core/tensorflow_state.py:25-65,
      204-261, 323-356).
"""
import numpy as np
import pytest

from tests.golden import cases
from tests.helpers import oracle_system
from tests.test_hip_parity import check_eval

pytestmark = pytest.mark.gpu

import os
import re


def _plan_limits():
    """The measured numbers of the table, from the ONE header the engine compiles them from (csrc/qoc_plan_limits.h: `#define QOC_PLAN_<NAME> <integer>`):
    the rules below are an independent restatement of plan_for's control flow, the numbers exist once."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'quantum-optimal-control_amd', 'csrc', 'qoc_plan_limits.h')
    out = {}
    for line in open(path):
        mt = re.match(r'#define\s+QOC_PLAN_(\w+)\s+(\d+)\b', line)
        if mt:
            out[mt.group(1)] = int(mt.group(2))
    assert len(out) >= 42, 'qoc_plan_limits.h: only %d limits parsed' % len(out)
    return out


LIM = _plan_limits()
LAT_WORK, LAT_WORK_SRC, LAT_WORK_NT3, LAT_WORK_NT4 = LIM['LAT_WORK'], LIM['LAT_WORK_SRC'], LIM['LAT_WORK_NT3'], LIM['LAT_WORK_NT4']   # seeds x slices up to which the latency mode is taken


FORBID = lambda n: {'dwdt': 0.1, 'forbidden_coeff_list': [3.0], 'states_forbidden_list': [n - 1]}      # noqa: E731


def ceil_div(a, b):
    return -(-a // b)


def lat_limit_nt2(n, state_reg):
    """16 < n <= 32: the batch kernels work on the active 4-row strips qa = ceil(n / 4) and take over earlier the smaller n is."""
    if n <= 16:
        return LAT_WORK_SRC if state_reg else LAT_WORK
    qa = max(5, ceil_div(n, 4))
    return min(LAT_WORK_SRC, LIM['LAT_WORK_PER_STRIP_SRC'] * qa) if state_reg else LIM['LAT_WORK_PER_STRIP'] * qa
def _small_instances():
    """(n_pad, slices per row, rows per workgroup, with-state-regulariser-too) of k_small_iter, in table order (csrc/qoc_small_instances.h: X(N, L, R, S))."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'quantum-optimal-control_amd', 'csrc', 'qoc_small_instances.h')
    text = open(path).read()
    out = []
    for name in ('A1', 'B1', 'A', 'B', 'C'):                    # the order of the engine's table; list A1 = builds for ONE workgroup per control set
        body = text[text.index('#define QOC_SMALL_INSTANCES_%s(X)' % name):]
        body = body[:body.index('\n//') if '\n//' in body else len(body)]
        out += [tuple(int(v) for v in mt.groups()) + (name in ('A1', 'B1'),) for mt in re.finditer(r'X\((\d+), (\d+), (\d+), ([01])\)', body)]
    assert len(out) >= 30
    return out


SMALL_INSTANCES = _small_instances()


def small_plan(n, k, m, steps, T, s, B, state_transfer=False, n_forb=0, speed_up=False, bandpass=False, hermitian=True):
    """Row "workgroup-resident path" (csrc/qoc_small.hip: qoc_small_supported, choose, model_us, qoc_small_auto), restated: which instance AUTO takes for one or a
    few control sets of n <= 12 levels, or None.  LDS carve: csrc/qoc_small.h (units of 16 bytes)."""
    Teff = T - 1 if state_transfer else T
    s = 0 if state_transfer else s
    src = n_forb > 0 or speed_up
    if n > 12 or m > n or k > 8 or not (0 <= Teff <= 30) or n_forb > 4 or (state_transfer and not hermitian):
        return None
    N = [v for v in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12) if v >= n][0]
    clog2 = lambda v: max(0, (v - 1).bit_length())                                   # noqa: E731
    best = None
    for (Ni, L, R, S, single) in SMALL_INSTANCES:
        if Ni != N or (src and not S):
            continue
        G = ceil_div(steps, R * L)
        if G > 32 or (single and G > 1) or (bandpass and G > 1) or (G > 1 and B * G > 128):
            continue
        Gp, NN, RL = 1 << clog2(G), N * N, R * L
        NP = NN                                                # qoc_small_node(N)
        lds = ((k + 1) * NN + k * NN + (4 * N if src else 0) + 2 * NN + (3 * NN if src else 0) + (2 * R - 1) * NP + ((2 * Gp - 1) if (src or Gp == 1) else (Gp + Gp // 2 + Gp // 4 + 4)) * NP
               + ((2 * R - 1 + 2 * Gp - 1) * m * N if src else 0) + k * RL + (k * (RL + 4) + 1) // 2 + 64 + (2 * Gp if Gp > 1 else 0)
               + ((k + 1) * RL if bandpass else 0))
        if lds * 16 > 160 * 1024:
            continue
        prod = (2.6 if 5 <= N <= 8 else 4.0) * N * N * 5.9 / 2400.0
        share = ((1.15 if N <= 2 else 1.7) if R >= 32 else 0.9) if N <= 4 else max(1.0, R / 16.0)
        per_slice = (max(Teff - 1, 0) + s + (6.0 if src else 4.0)) * prod + 0.15 + ((0.6 if N <= 4 else 0.3 if N <= 8 else 0.0) if src else 0.0)
        us = share * (L * per_slice + (4.0 if src else 2.0) * clog2(R) * (prod + 0.1))
        if G > 1:
            us += (4.0 if src else 2.0) * 1.5 + (4.0 if src else 1.0) * clog2(G) * (prod + 0.1)
        if bandpass:
            us += 1.2e-4 * k * steps * steps
        us = 1.45 * (us + 1.5)
        if best is None or us < best[0]:
            best = (us, N, R, L, G)
    limit = LIM['SMALL_MAX_MODEL_US_SRC'] if src else max(LIM['SMALL_MAX_MODEL_US'], LIM['SMALL_MODEL_US_BASE'] + 1e-3 * LIM['SMALL_MODEL_NS_PER_SLICE'] * steps)
    if best is None or best[0] > limit or B > LIM['SMALL_MAX_SETS']:
        return None
    return {'path': 'small', 'n_pad': best[1], 'rows': best[2], 'slices_per_row': best[3], 'workgroups': best[4], 'state_sources': 1 if src else 0}


def expected_plan(n, k, m, steps, T, B, state_transfer=False, state_reg=False, hermitian=True, s=2):
    """DESIGN.md section 4, the AUTO table, as ordered rules -> the dict HipEngine.plan reports."""
    st = state_transfer
    small = small_plan(n, k, m, steps, T, s, B, state_transfer=st, n_forb=(2 if st else 1) if state_reg else 0, hermitian=hermitian)
    if small is not None:                                 # row "n <= 12, one or a few control sets": the whole iteration inside one launch
        return small
    deg = T - 1 if st else T                              # matvecexp sums j < T: the propagator is the Taylor polynomial of degree T - 1
    direct_ok = st and n <= 64 and m <= 8
    dpp = n > 32 and m == 1                               # the direct route's k_gemm_taylor_chain_dpp

    def gemm_state_transfer():
        # rows "state transfer ...": GEMM path; direct route for large batches (n <= 32: from 112 control sets, n <= 64: from 48 -- with ONE state
        # vector, where the direct route has the DPP Taylor chain since round 4: from 12, with a state regulariser from 22) and for
        # non-Hermitian generators; the propagator route otherwise; m > 8 or n > 64 cannot run direct
        if m > 32 or not (hermitian or direct_ok):
            return {'path': 'st_fused' if (n <= 64 and m <= 4 and k <= 8) else 'generic'}
        direct = direct_ok and (not hermitian or B >= (LIM['ST_DIRECT_N32'] if n <= 32 else ((LIM['ST_DIRECT_DPP_SRC'] if state_reg else LIM['ST_DIRECT_DPP']) if dpp else LIM['ST_DIRECT_N64'])))
        out = {'path': 'gemm', 'route': 'direct' if direct else 'propagator', 'chains': 'persistent' if (n <= 64 and m <= 8) else 'launches'}
        if direct:      # the kernel of the Taylor chains: the DPP chain at 33 .. 64 levels with one vector -- on packed generators when they are anti-Hermitian
            # ... except padded problems (n <= 56) of up to 128 control sets (or that cannot be packed): the first 40 / 48 / 56 columns of the full image
            cols = 40 if n <= 40 else 48 if n <= 48 else 56 if n <= 56 else 64
            if dpp and cols < 64 and (not hermitian or B <= 128 or cols == 40):
                out['taylor_chain'] = 'columns%d' % cols
            else:
                out['taylor_chain'] = ('packed' if hermitian else 'full') if dpp else 'butterfly'
        return out
    mfma_ok = n <= 64 and m <= 16 and k <= 8 and 1 <= deg <= 22 and hermitian
    if st and not (mfma_ok and (n <= 32 or (n <= 48 and k <= 4))):
        return gemm_state_transfer()                      # row "state transfer, n > 48 (or 32 < n <= 48 with k > 4), or generators that are not anti-Hermitian"
    if not mfma_ok:
        return {'path': 'gemm', 'route': 'unitary'} if m <= 32 else {'path': 'generic'}
    work = B * steps
    # row "latency mode": one or a few control sets
    if deg >= 2 and steps >= LIM['LAT_MIN_SLICES']:
        if n > 48 or (n > 32 and k > 4):
            lat = work <= LAT_WORK_NT4 and B <= LIM['LAT_SETS_NT4']
        elif n > 32:
            lat = work <= LAT_WORK_NT3 and B <= LIM['LAT_SETS_NT3']
        else:
            lat = work <= lat_limit_nt2(n, state_reg) and B <= ((LIM['LAT_SETS_N32_ST_WIDE'] if (st and ceil_div(n, 4) >= 7) else LIM['LAT_SETS_N32']) if n > 16
                                                              else (LIM['LAT_SETS_N16_ST'] if st else LIM['LAT_SETS_N16']))
        lat = lat or (B == 1 and steps <= LIM['LAT_SINGLE_MAX_SLICES'])
    else:
        lat = False
    if lat:
        nt = 2 if n <= 32 else (3 if (n <= 48 and k <= 4) else 4)
        L = ceil_div(steps, ceil_div(steps, 8))
        return {'path': 'mfma', 'nt': nt, 'expm': 5, 'chunks': ceil_div(steps, L), 'sweeps': 'latency_sources' if state_reg else 'latency'}
    # rows "GEMM": 48 < n <= 64 below the NT = 4 batch sizes; 32 < n <= 48 with fewer than 8 control sets; 16 < n <= 32 with a few control sets
    # (2 / 3 / 5 / 7 for ceil(n / 4) = 5 / 6 / 7 / 8; 5 / 6 / 8 / 8 with a state regulariser; state transfer: 8 from 25 levels on); state transfer: the
    # large batches that the direct Taylor chains win (n <= 32: from 112 control sets of more than 20 levels -- 28 with a state regulariser; n > 32: from 48 -- 112; with one state vector from 20 -- 48: csrc/qoc_plan_limits.h)
    nt4_batch = n > 48 and ((k <= 4 and B >= LIM['NT4_MIN_SETS_K4']) or B >= LIM['NT4_MIN_SETS'])
    qa = ceil_div(n, 4)
    gemm_small = ({5: LIM['GEMM_SMALL_SRC_Q5'], 6: LIM['GEMM_SMALL_SRC_Q6']}.get(qa, LIM['GEMM_SMALL_SRC_Q78']) if state_reg
                  else {5: LIM['GEMM_SMALL_Q5'], 6: LIM['GEMM_SMALL_Q6'], 7: LIM['GEMM_SMALL_Q7']}.get(qa, LIM['GEMM_SMALL_Q8'])) if 16 < n <= 32 else 0
    if st and qa >= 7 and 16 < n <= 32:
        gemm_small = LIM['GEMM_SMALL_ST_WIDE']
    st_big = direct_ok and ((B >= LIM['ST_BIG_N32'] and n > (LIM['ST_BIG_N32_MIN_LEVELS_SRC'] if state_reg else LIM['ST_BIG_N32_MIN_LEVELS'])) if n <= 32
                            else B >= (((LIM['ST_BIG_DPP_SRC'] if state_reg else LIM['ST_BIG_DPP']) if dpp else (LIM['ST_BIG_N64_SRC'] if state_reg else LIM['ST_BIG_N64']))))
    if (n > 48 and not nt4_batch) or (32 < n <= 48 and B < LIM['NT3_MIN_SETS']) or \
            (16 < n <= 32 and B <= gemm_small and m <= 8 and steps >= LIM['GEMM_SMALL_MIN_SLICES']) or st_big:
        if st:
            return gemm_state_transfer()
        return {'path': 'gemm', 'route': 'unitary', 'chains': 'persistent' if m <= 8 else 'launches'}
    # rows "MFMA batch kernels"
    nt = 1 if n <= 16 else 2 if n <= 32 else 3 if n <= 48 else 4
    C = min(LIM['CHUNK_ITEMS'] // B if nt == 2 else ceil_div(LIM['CHUNK_ITEMS'], B), LIM['CHUNKS_MAX_NT2'] if nt == 2 else LIM['CHUNKS_MAX'])
    C = max(1, min(C, steps))
    C = ceil_div(steps, ceil_div(steps, C))
    if nt == 2:
        expm = 8 if deg >= 3 else ((4 if deg == 2 else 3) if B * C >= 512 else 1)
        sweeps = 'row_tile_gradient' if k >= 6 else ('downup' if (m <= 8 and not state_reg) else 'pair')
    elif nt == 1:
        expm, sweeps = 1, 'one_wave'
    else:
        expm, sweeps = 7, 'row_tile_gradient'
    return {'path': 'mfma', 'nt': nt, 'expm': expm, 'chunks': C, 'sweeps': sweeps}


def unitary_rows():
    rows = []
    for n in (16, 17, 32, 33, 48, 49, 64):
        for k in (4, 5, 6, 8):
            for reg in (False, True):
                for B in (1, 2, 4, 5, 6, 7, 8, 9, 16, 17, 31, 32, 63, 64, 111, 112):
                    # every (n, k, regulariser, batch) once at 130 slices; the (k = 4 / k = 5) columns also at the pulse lengths that put
                    # seeds x slices on either side of the latency mode's limit for this class of n
                    lens = {130}
                    if k in (4, 5):
                        limit = (LAT_WORK_NT4 if (n > 48 or (n > 32 and k > 4)) else LAT_WORK_NT3 if n > 32 else lat_limit_nt2(n, reg))
                        if 2 <= B <= 16 and 64 <= limit // B <= 1200:
                            lens |= {limit // B, limit // B + 1}
                    if B >= 63 and (n > 32 or k >= 6):
                        lens = {70}                       # large batches of the large shapes: short pulses keep the oracle side cheap
                    for steps in sorted(lens):
                        rows.append((n, k, reg, B, steps))
    # Taylor orders below the in-place kernel's range, and an order-1 problem the latency mode cannot take
    extra = [(32, 4, False, 64, 130, 2), (32, 4, False, 3, 130, 2), (32, 4, False, 64, 130, 1), (32, 4, False, 1, 130, 1), (20, 3, False, 40, 96, 2)]
    return rows, extra


def _problem(n, k, steps, m, T, s, reg, seed):
    c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(T, s), seed=seed)
    c['total_time'] = 20.0 * steps / 500.0                # short pulses: a well-conditioned gradient at every size
    if reg:
        c['reg_coeffs'] = FORBID(n)
    return c


def _run(c, B, expect, seed, check=True):
    from quantum_optimal_control.core import hip_engine
    sp = oracle_system(c)
    # rows whose plan alone is asserted hold ONE control set and PLAN for B (qoc_config.plan_seeds: AUTO decides for the planned batch) -- no
    # gigabyte of propagators is allocated 744 times; rows that are checked against the oracle hold the real batch
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                               state_transfer=sp.state_transfer, reg_coeffs=sp.reg_coeffs, one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs,
                               n_seeds=B if check else 1, plan_seeds=B)
    try:
        got = {key: (int(v) if v.lstrip('-').isdigit() else v) for key, v in eng.plan.items()}
        for key, want in expect.items():
            assert got.get(key) == want, (key, want, got)
        if not check:
            return
        rng = np.random.default_rng(seed)
        bases = rng.normal(0, 1 / np.sqrt(sp.steps), (B, sp.k, sp.steps))
        eng.set_base(bases)
        picked = [B - 1] if (sp.n > 32 and sp.steps > 400) else sorted({0, B - 1})     # (long pulses of the large shapes: one control set -- the oracle side costs seconds there)

        class TwoOf(object):                               # check_eval walks `bases`: hand it the first and the last control set only
            def __init__(self, eng):
                self.eng = eng

            def evaluate(self):
                r = self.eng.evaluate()
                return {key: np.asarray(v)[picked] for key, v in r.items()}

            def get_inter_vecs(self):
                return self.eng.get_inter_vecs()[picked]

            def get_final_unitary(self):
                return self.eng.get_final_unitary()[picked]
        check_eval(TwoOf(eng), sp, [bases[b] for b in picked])
    finally:
        eng.close()


ROWS, EXTRA = unitary_rows()


@pytest.mark.parametrize('n', [16, 17, 32, 33, 48, 49, 64])
@pytest.mark.parametrize('k', [4, 5, 6, 8])
def test_auto_plan_unitary_rows(n, k):
    m = 8
    mine = [r for r in ROWS if (r[0], r[1]) == (n, k)]
    assert len(mine) >= 28
    # oracle parity at the smallest and the largest batch of every distinct (regulariser, plan-without-chunk-count) group of this shape
    groups = {}
    for i, (_, _, reg, B, steps) in enumerate(mine):
        plan = expected_plan(n, k, m, steps, 5, B, state_reg=reg)
        sig = (reg,) + tuple(sorted((key, v) for key, v in plan.items() if key != 'chunks'))
        groups.setdefault(sig, []).append((B, steps, i))
    checked = set()
    for rows in groups.values():
        checked.add(min(rows)[2])
        checked.add(max(rows)[2])
    for i, (_, _, reg, B, steps) in enumerate(mine):
        expect = expected_plan(n, k, m, steps, 5, B, state_reg=reg)
        _run(_problem(n, k, steps, m, 5, 2, reg, seed=100 + n), B, expect, seed=B, check=i in checked)


@pytest.mark.parametrize('n,k,reg,B,steps,T', EXTRA, ids=['T2_batch', 'T2_latency', 'T1_batch', 'T1_single_no_latency', 'n20_T2_small_launch'])
def test_auto_plan_low_taylor_orders(n, k, reg, B, steps, T):
    expect = expected_plan(n, k, 8, steps, T, B, state_reg=reg)
    _run(_problem(n, k, steps, 8, T, 2, reg, seed=7), B, expect, seed=B)


def test_auto_plan_wide_and_odd_shapes():
    """m beyond the fused sweep (m > 8), beyond the MFMA path (m > 16), and beyond every fast path (m > 32)."""
    for (n, k, m, B, steps) in ((32, 4, 13, 64, 130), (32, 4, 16, 20, 130), (40, 3, 20, 16, 96), (36, 2, 34, 3, 40)):
        expect = expected_plan(n, k, m, steps, 5, B)
        _run(_problem(n, k, steps, m, 5, 2, False, seed=3), B, expect, seed=B)


# (n, k, m, control sets, anti-Hermitian generators, forbidden levels): the GEMM-path routes either side of 48 / 112 control sets, the shapes the MFMA path
# takes since round 4 (n <= 32; 32 < n <= 48 with k <= 4) on both sides of ITS limits (latency mode up to 8 control sets at n <= 16, the GEMM route up to 8
# from 25 levels on, the direct Taylor chains for large batches), lossy generators, wide and large problems
ST_ROWS = [(32, 4, 1, 4, True, True), (32, 4, 1, 5, True, True), (25, 4, 1, 5, True, False), (24, 4, 1, 5, True, False), (64, 6, 1, 47, True, True), (64, 6, 1, 48, True, True), (64, 6, 1, 21, True, True), (64, 6, 1, 22, True, True), (64, 6, 1, 11, True, False), (64, 6, 1, 12, True, False), (64, 6, 2, 47, True, True), (64, 6, 2, 48, True, True), (64, 6, 1, 1, True, True), (33, 6, 2, 47, True, True), (33, 6, 2, 48, True, True),
           (33, 4, 2, 47, True, False), (33, 4, 2, 48, True, False), (48, 4, 1, 1, True, True), (48, 4, 1, 8, True, True), (48, 4, 1, 9, True, True),
           (48, 4, 1, 47, True, True), (48, 4, 1, 48, True, True), (48, 4, 1, 31, True, False), (48, 4, 1, 32, True, False), (48, 4, 2, 111, True, True), (48, 4, 2, 112, True, True),
           (32, 4, 1, 1, True, True), (32, 4, 1, 16, True, False), (32, 4, 1, 17, True, False), (32, 4, 1, 64, True, True), (32, 4, 1, 111, True, True), (32, 4, 1, 112, True, True),
           (28, 4, 1, 112, True, True), (28, 4, 1, 112, True, False), (20, 4, 1, 112, True, False), (21, 4, 1, 112, True, False), (21, 4, 1, 111, True, False),
           (16, 3, 4, 8, True, True), (16, 3, 4, 9, True, True), (16, 3, 4, 111, True, True), (16, 3, 4, 112, True, True), (8, 2, 1, 64, True, False),
           (64, 6, 1, 3, False, True), (20, 3, 2, 1, False, True), (24, 3, 12, 4, True, True), (70, 3, 2, 50, True, True)]
ST_LONG = [(27, 4, 1, 8, 640, False), (27, 4, 1, 9, 640, False), (32, 4, 1, 8, 600, True), (24, 4, 1, 8, 600, False), (16, 4, 1, 8, 600, False)]      # beyond the latency mode's reach: the GEMM route / the batch kernels


def _st_problem(n, k, m, steps, hermitian, reg):
    c = cases.case_c3(n=n, k=k, steps=steps, taylor=(8, 0), seed=5)
    c['total_time'] = 4.0 * steps / 100.0
    if not reg:
        c['reg_coeffs'] = {'dwdt': 1e-3}
    rng = np.random.default_rng(n + m)
    if m > 1:
        def vecs():
            Q, _ = np.linalg.qr(rng.normal(size=(n, m)) + 1j * rng.normal(size=(n, m)))
            return [Q[:, j] for j in range(m)]
        c['states_concerned_list'], c['U'] = vecs(), vecs()
    if not hermitian:
        c['H0'] = c['H0'] + 0.05j * np.diag(np.arange(n) / n)          # a lossy drift: the generators are no longer anti-Hermitian
    return c


@pytest.mark.parametrize('n,k,m,B,hermitian,reg', ST_ROWS, ids=['n%d_k%d_m%d_B%d_%s%s' % (r[0], r[1], r[2], r[3], 'herm' if r[4] else 'nonherm', '_forb' if r[5] else '') for r in ST_ROWS])
def test_auto_plan_state_transfer_rows(n, k, m, B, hermitian, reg):
    steps = 100
    expect = expected_plan(n, k, m, steps, 8, B, state_transfer=True, state_reg=reg, hermitian=hermitian)
    _run(_st_problem(n, k, m, steps, hermitian, reg), B, expect, seed=B)


@pytest.mark.parametrize('n,k,m,B,steps,reg', ST_LONG, ids=['n%d_B%d_%d%s' % (r[0], r[3], r[4], '_forb' if r[5] else '') for r in ST_LONG])
def test_auto_plan_state_transfer_long_pulses(n, k, m, B, steps, reg):
    expect = expected_plan(n, k, m, steps, 8, B, state_transfer=True, state_reg=reg)
    _run(_st_problem(n, k, m, steps, True, reg), B, expect, seed=B)


# (n, k, m, slices, control sets, forbidden level): the workgroup-resident path either side of its limits -- 12 / 13 levels, the modelled 45 us (85 with a state regulariser: long pulses of the
# larger sizes), 128 workgroups spinning on each other (control sets x workgroups per set), 256 control sets
SMALL_ROWS = [(2, 1, 2, 100, 1, False), (2, 1, 2, 100, 64, False), (2, 1, 2, 100, 256, False), (2, 1, 2, 100, 257, False), (4, 2, 4, 200, 1, False), (4, 2, 4, 200, 16, True),
              (8, 4, 8, 500, 1, False), (8, 4, 8, 500, 4, False), (8, 4, 8, 500, 5, False), (8, 4, 8, 100, 16, False), (8, 4, 8, 100, 64, False), (9, 4, 4, 300, 1, True),
              (12, 3, 8, 100, 1, False), (12, 3, 8, 400, 1, False), (13, 3, 8, 100, 1, False), (10, 2, 5, 64, 2, True), (10, 2, 5, 700, 1, True), (6, 8, 6, 130, 3, False),
              (5, 2, 2, 1100, 1, False), (3, 1, 3, 4000, 1, False), (3, 1, 3, 4200, 1, False),
              (8, 4, 8, 1000, 1, False), (9, 4, 8, 1000, 1, False), (10, 4, 8, 500, 1, False), (10, 4, 8, 100, 1, False), (12, 4, 8, 50, 1, False)]      # long pulses: the limit grows with the slices


@pytest.mark.parametrize('n,k,m,steps,B,reg', SMALL_ROWS, ids=['n%d_k%d_m%d_%dslices_B%d%s' % (r[0], r[1], r[2], r[3], r[4], '_forb' if r[5] else '') for r in SMALL_ROWS])
def test_auto_plan_small_rows(n, k, m, steps, B, reg):
    expect = expected_plan(n, k, m, steps, 5, B, state_reg=reg)
    _run(_problem(n, k, steps, m, 5, 2, reg, seed=200 + n), B, expect, seed=B, check=B <= 64 and steps <= 1200)


def test_auto_plan_small_rows_cover_both_sides():
    plans = [expected_plan(n, k, m, steps, 5, B, state_reg=reg) for (n, k, m, steps, B, reg) in SMALL_ROWS]
    small = [p for p in plans if p.get('path') == 'small']
    assert 8 <= len(small) <= len(plans) - 5
    assert {p['workgroups'] > 1 for p in small} == {True, False} and {p['rows'] for p in small} >= {16, 32}


def test_auto_plan_small_excluded_shapes():
    """A bandpass regulariser on a pulse that needs several workgroups (its DFT wants the whole pulse in one) or whose 2 k steps^2 terms cost more than the other
    paths, and more than four forbidden levels, stay on the other paths; a bandpass regulariser on a short pulse is taken."""
    c = _problem(4, 2, 64, 3, 5, 2, False, seed=9)
    c['reg_coeffs'] = {'bandpass': 0.1, 'band': [0.5, 2.0]}
    assert small_plan(4, 2, 3, 64, 5, 2, 1, bandpass=True) is not None
    _run(c, 1, {'path': 'small', 'workgroups': 1}, seed=1)
    c = _problem(4, 2, 700, 3, 5, 2, False, seed=9)
    c['reg_coeffs'] = {'bandpass': 0.1, 'band': [0.5, 2.0]}
    assert small_plan(4, 2, 3, 700, 5, 2, 1, bandpass=True) is None
    _run(c, 1, {'path': 'mfma'}, seed=1, check=False)
    c = _problem(8, 2, 64, 3, 5, 2, False, seed=9)
    c['reg_coeffs'] = {'forbidden_coeff_list': [1.0] * 5, 'states_forbidden_list': [7, 6, 5, 4, 3]}
    _run(c, 1, {'path': 'mfma'}, seed=1)
