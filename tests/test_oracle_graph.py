"""Independent pins for oracle rows a6-a14 (the TF graph, which cannot run here -- SURVEY.md 8c):
(1) node-for-node real-embedded graph emulation with the reference's custom gradient functions,
(2) scipy.linalg.expm, (3) unitarity, (4) finite differences as a sanity bound, (5) TF1 Adam semantics."""
import numpy as np
import pytest
import scipy.linalg as la

from oracle import grape_oracle as go
from oracle import tf_graph_emulation as tfe
from tests.golden import cases
from tests.helpers import oracle_system

FULL_REG = {'amplitude': 0.3, 'envelope': 0.2, 'dwdt': 0.1, 'd2wdt2': 0.05, 'forbidden_coeff_list': [3.0, 2.0],
            'states_forbidden_list': [3, 2], 'speed_up': 0.7, 'bandpass': 0.4, 'band': [0.5, 2.0]}


def _cases():
    out = []
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(5, 2), seed=1); out.append(('unitary_plain', c))
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(6, 1), seed=2); c['reg_coeffs'] = dict(FULL_REG)
    c['total_time'] = 2.0; out.append(('unitary_allreg', c))
    c = cases.case_dressed(); out.append(('dressed_forbidden', c))
    c = cases.case_c3(n=6, k=3, steps=15, taylor=(8, 0)); c['reg_coeffs'] = {
        'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [4, 5], 'speed_up': 0.3}
    out.append(('state_transfer_reg', c))
    out.append(('state_small', cases.case_state_small()))
    out.append(('small_auto_U0', cases.case_small_auto()))
    return out


@pytest.mark.parametrize('name,c', _cases(), ids=[n for n, _ in _cases()])
def test_closed_form_equals_graph_emulation(name, c):
    sp = oracle_system(c)
    base = 3.0 * sp.base0
    r = go.evaluate(sp, base, want_inter=True)
    e = tfe.evaluate_graph(sp, base)
    assert abs(r['loss'] - e['loss']) < 1e-13 * max(1, abs(e['loss']))
    assert abs(r['reg_loss'] - e['reg_loss']) < 1e-12 * max(1, abs(e['reg_loss']))
    assert abs(r['unitary_scale'] - e['unitary_scale']) < 1e-12 * max(1, abs(e['unitary_scale']))
    gmax = max(1.0, np.max(np.abs(e['grad'])))
    np.testing.assert_allclose(r['grad'], e['grad'], rtol=0, atol=1e-12 * gmax)
    assert abs(r['grad_squared'] - e['grad_squared']) < 1e-11 * max(1, e['grad_squared'])
    if not sp.state_transfer:
        np.testing.assert_allclose(r['U_final'], go.r_to_c_mat(e['final_state'], sp.n), atol=1e-13)
    # inter_vecs_packed (2n, steps+1, m) <-> (steps+1, n, m) complex
    iv = e['inter_vecs_packed']
    ivc = np.transpose(iv[:sp.n] + 1j * iv[sp.n:], (1, 0, 2))
    np.testing.assert_allclose(r['inter_vecs'], ivc, atol=1e-13 * max(1, np.max(np.abs(ivc))))


def test_faithful_fp32_mode_bounds_reference_parity():
    """Tier 2: the reference runs in float32; fp64 results agree with it only to fp32 round-off."""
    import torch
    c = cases.case_c2(n=4, k=2, steps=12, m=3, taylor=(5, 2), seed=1)
    sp = oracle_system(c)
    r = go.evaluate(sp, sp.base0)
    e32 = tfe.evaluate_graph(sp, sp.base0, dtype=torch.float32)
    assert abs(r['loss'] - e32['loss']) < 1e-5
    np.testing.assert_allclose(r['grad'], e32['grad'], atol=2e-5 * max(1, np.max(np.abs(r['grad']))))


def test_matexp_against_scipy_expm():
    rng = np.random.default_rng(0)
    A = -1j * 0.2 * cases.herm(rng, 8)
    K = go.matexp(A / 2 ** 3, 8, 3)
    np.testing.assert_allclose(K, la.expm(A), atol=1e-13)
    # truncation error of the order-T series is bounded by the next term
    K2 = go.matexp(A, 3, 0)
    assert np.linalg.norm(K2 - la.expm(A), 2) < 1.05 * np.linalg.norm(A, 2) ** 4 / 24 * np.exp(np.linalg.norm(A, 2))


def test_matvecexp_order_is_T_minus_1():
    rng = np.random.default_rng(0)
    B = -1j * 0.3 * cases.herm(rng, 5)
    psi = rng.normal(size=(5, 2)) + 0j
    T = 6
    ref = sum(np.linalg.matrix_power(B, j) @ psi / np.math.factorial(j) if hasattr(np, 'math') else
              np.linalg.matrix_power(B, j) @ psi / float(np.prod(np.arange(1, j + 1))) for j in range(T))
    np.testing.assert_allclose(go.matvecexp(B, psi, T), ref, atol=1e-14)


def test_final_unitary_is_unitary_and_scale_is_one():
    sp = oracle_system(cases.case_c2(n=8, k=2, steps=20, m=4, taylor=(14, 4), seed=3))
    r = go.evaluate(sp, sp.base0, want_grad=False)
    U = r['U_final']
    np.testing.assert_allclose(U.conj().T @ U, np.eye(8), atol=1e-12)
    # unitary_scale sums ALL entries of X^dagger X (not the trace): equals 1 for a unitary X
    assert abs(r['unitary_scale'] - 1.0) < 1e-12


def test_first_order_gradient_approaches_finite_difference_as_dt_to_zero():
    """The reference gradient is the first-order GRAPE approximation (tensorflow_state.py:49-65): it equals the true
    derivative only up to O(dt ||[H_k,H]||).  Sanity bound: error shrinks ~linearly with dt."""
    errs = []
    for steps in (80, 320):
        c = cases.case_c2(n=4, k=2, steps=steps, m=4, taylor=(14, 1), seed=4)
        c['total_time'] = 1.0
        sp = oracle_system(c)
        base = sp.base0
        g = go.evaluate(sp, base)['grad']
        fd = np.zeros_like(g)
        h = 1e-6
        for idx in [(0, 3), (1, steps // 2), (0, steps - 1)]:
            bp = base.copy(); bp[idx] += h
            bm = base.copy(); bm[idx] -= h
            fd[idx] = (go.evaluate(sp, bp, want_grad=False)['reg_loss']
                       - go.evaluate(sp, bm, want_grad=False)['reg_loss']) / (2 * h)
            errs.append(abs(fd[idx] - g[idx]) / (abs(fd[idx]) + 1e-12))
    coarse, fine = max(errs[:3]), max(errs[3:])
    assert fine < coarse * 0.5 and fine < 0.06


def test_tf1_adam_semantics():
    opt = go.Adam((2,))
    x = np.array([1.0, -2.0]); g = np.array([0.5, -0.25])
    x1 = opt.step(x, g, 0.1)
    # t=1: lr_t = lr*sqrt(1-b2)/(1-b1); m=(1-b1)g; v=(1-b2)g^2
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = x - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(x1, exp, rtol=1e-15)


def test_run_adam_loop_semantics():
    """iterations is incremented before the LR is computed; loop stops on max_iterations (run_session.py:47-69)."""
    sp = oracle_system(cases.case_c1())
    res = go.run_adam(sp, dict(rate=0.05, max_iterations=30, learning_rate_decay=100, conv_target=1e-12), history=True)
    assert res['iterations'] == 30
    assert len(res['history']) == 31                     # 30 updates + the final evaluation
    assert res['history'][-1, 0] < res['history'][0, 0]  # loss decreased
    np.testing.assert_allclose(res['uks'], sp.maxA[:, None] * np.sin(res['base']))


def test_d2wdt2_without_dwdt_raises_like_reference():
    c = cases.case_c1(); c['reg_coeffs'] = {'d2wdt2': 0.1}
    sp = oracle_system(c)
    with pytest.raises(NameError):
        go.evaluate(sp, sp.base0)


def test_c_port_matches_numpy_oracle():
    """oracle/qoc_oracle.c (cpu_baseline leg of bench.py) restates the same unitary-mode iteration."""
    import os
    from oracle import c_port
    if not os.path.exists(c_port.LIB):
        pytest.skip('oracle/_build/libqoc_oracle.so not built (run __graft_entry__.build())')
    c = cases.case_c2(n=12, k=3, steps=25, m=5, taylor=(6, 2), seed=8)
    sp = oracle_system(c)
    bases = np.stack([sp.base0, -2.0 * sp.base0])
    r = c_port.evaluate(sp, bases, nthreads=2)
    for b in range(2):
        o = go.evaluate(sp, bases[b])
        assert abs(r['loss'][b] - o['loss']) < 1e-13
        assert abs(r['unitary_scale'][b] - o['unitary_scale']) < 1e-12
        np.testing.assert_allclose(r['grad'][b], o['grad'], atol=1e-13 * max(1, np.max(np.abs(o['grad']))))
        np.testing.assert_allclose(r['U_final'][b], o['U_final'], atol=1e-13)
    # the loop: same TF1-Adam / LR schedule as run_adam
    conv = dict(rate=0.02, max_iterations=7, learning_rate_decay=50, conv_target=-1.0, min_grad=-1.0)
    ref = go.run_adam(sp, conv, base=sp.base0)
    out, loss = c_port.iterate(sp, sp.base0[None], 7, rate=0.02, decay=50.0, nthreads=1)
    np.testing.assert_allclose(out[0], ref['base'], atol=1e-12)
