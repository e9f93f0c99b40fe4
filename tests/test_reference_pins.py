"""Pins that close gaps named by the round-1 review: (a) the scipy drivers (run_session.py:151-196) against the same
scipy.optimize.minimize driven by the CPU oracle, (b) the reference's float32 arithmetic at full C2 size, (c) the first-order
gradient against a derivation that shares no code with the oracle."""
import contextlib
import io
import os

import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import grape_oracle as go
from tests import independent_gradient as ig
from tests.golden import cases
from tests.helpers import grape_kwargs, oracle_system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp32_reference_arithmetic_at_full_c2_size():
    """Tier 2 at n = 32, steps = 500: the node-for-node float32 emulation of the TF graph agrees with the fp64
    closed forms to float32 round-off accumulated over 500 slices -- the best any comparison with the real reference could do."""
    import torch
    from oracle import tf_graph_emulation as tfe
    sp = oracle_system(cases.case_c2())
    r = go.evaluate(sp, sp.base0)
    e32 = tfe.evaluate_graph(sp, sp.base0, dtype=torch.float32)
    assert abs(r['loss'] - e32['loss']) < 1e-5
    assert abs(r['unitary_scale'] - e32['unitary_scale']) < 1e-4
    gmax = np.max(np.abs(r['grad']))
    assert np.max(np.abs(r['grad'] - e32['grad'])) < 2e-4 * gmax
    assert np.max(np.abs(r['grad'] - e32['grad'])) > 1e-9 * gmax          # and it really is a float32 run


def _gradient_problem():
    c = cases.case_c2(n=6, k=3, steps=14, m=4, taylor=(6, 2), seed=41)
    sp = oracle_system(c)
    base = 2.5 * sp.base0 + 0.3
    pairs = [(k, t) for k in range(sp.k) for t in (0, 1, 6, 12, 13)]
    ref = ig.first_order_gradient(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, base, sp.exp_terms, sp.scaling, pairs)
    return sp, base, pairs, ref


def test_oracle_gradient_against_independent_first_order_derivation():
    sp, base, pairs, ref = _gradient_problem()
    g = go.evaluate(sp, base)['grad']
    scale = max(abs(v) for v in ref.values())
    for (k, t) in pairs:
        assert abs(g[k, t] - ref[(k, t)]) < 1e-8 * scale, (k, t, g[k, t], ref[(k, t)])


@pytest.mark.gpu
def test_hip_gradient_against_independent_first_order_derivation():
    from quantum_optimal_control.core import hip_engine
    sp, base, pairs, ref = _gradient_problem()
    scale = max(abs(v) for v in ref.values())
    for path in (hip_engine.PATH_GENERIC, hip_engine.PATH_MFMA, hip_engine.PATH_GEMM):
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling,
                                   reg_coeffs={}, n_seeds=1, path=path)
        eng.set_base(base[None])
        g = eng.evaluate()['grad'][0]
        eng.close()
        for (k, t) in pairs:
            assert abs(g[k, t] - ref[(k, t)]) < 1e-8 * scale, (path, k, t, g[k, t], ref[(k, t)])


def _oracle_scipy_run(sp, method, conv):
    """run_session.bfgs_optimize (run_session.py:151-196) with the CPU oracle in place of the TF session."""
    state = {'end': False}

    def fun(x):
        r = go.evaluate(sp, np.reshape(x, (sp.k, sp.steps)))
        g = np.reshape(r['grad'], -1)
        if r['loss'] < conv['conv_target']:
            state['end'] = True
            g = 0 * g
        return np.float64(r['reg_loss']), np.asarray(g, dtype=np.float64)

    if method == 'L-BFGS-B':
        options = {'maxfun': conv['max_iterations'], 'gtol': conv['min_grad'], 'disp': False, 'maxls': 40}
    else:
        options = {'gtol': conv['min_grad'], 'disp': False, 'maxiter': conv['max_iterations']}
    res = minimize(fun, np.reshape(sp.base0, -1), method=method, jac=True, options=options)
    base = np.reshape(res['x'], (sp.k, sp.steps))
    return np.asarray(sp.maxA)[:, None] * np.sin(base), go.evaluate(sp, base)


@pytest.mark.gpu
@pytest.mark.parametrize('method', ['L-BFGS-B', 'BFGS'])
def test_scipy_drivers_follow_the_oracle_driven_optimiser(method):
    """The same optimiser fed by the HIP engine and by the CPU oracle walks the same path: >= 10 function evaluations, final
    pulses equal to 1e-8 (evaluations agree to 1e-12; a quasi-Newton path amplifies that only mildly over a dozen steps)."""
    from quantum_optimal_control.main_grape.grape import Grape
    c = cases.case_c2(n=5, k=2, steps=16, m=3, taylor=(5, 2), seed=17)
    conv = {'rate': 0.01, 'update_step': 5, 'max_iterations': 14, 'conv_target': 1e-12, 'learning_rate_decay': 100, 'min_grad': 1e-25}
    np.random.seed(c['np_seed'])
    with contextlib.redirect_stdout(io.StringIO()):
        uks, Uf = Grape(convergence=dict(conv), method=method, **grape_kwargs(c))
    sp = oracle_system(c)
    calls = {'n': 0}
    orig = go.evaluate

    def counted(*a, **k):
        calls['n'] += 1
        return orig(*a, **k)
    go.evaluate = counted
    try:
        uks_o, r_o = _oracle_scipy_run(sp, method, conv)
    finally:
        go.evaluate = orig
    assert calls['n'] >= 10
    np.testing.assert_allclose(uks, uks_o, rtol=0, atol=1e-8 * np.max(np.abs(uks_o)))
    np.testing.assert_allclose(Uf, r_o['U_final'], rtol=0, atol=1e-8)


@pytest.mark.skipif(not os.path.isdir('/root/reference/quantum_optimal_control'), reason='needs the reference tree (build container only)')
def test_committed_fixtures_are_what_the_reference_code_produces(tmp_path):
    """Integrity of tests/golden/: re-run both generators against /root/reference in a scratch copy of tests/golden/ and compare every array with
    the committed file -- sysparams_* / helpers (the reference's NumPy code) bit for bit, graph_* (its graph code on the TF1 stand-in) to 1e-14
    (torch reductions may reassociate between runs of different thread counts; the float32 set graph32_* to 1e-5 for the same reason)."""
    import shutil
    import subprocess
    import sys
    golden = os.path.join(ROOT, 'tests', 'golden')
    work = tmp_path / 'repo'
    (work / 'tests').mkdir(parents=True)
    shutil.copytree(golden, work / 'tests' / 'golden')
    for f in ('__init__.py', 'helpers.py'):
        shutil.copy(os.path.join(ROOT, 'tests', f), work / 'tests' / f)
    shutil.copy(os.path.join(ROOT, 'bench.py'), work / 'bench.py')          # the full-size C2 cases take bench.py's problem and restart seeds
    os.symlink(os.path.join(ROOT, 'quantum-optimal-control_amd'), work / 'quantum-optimal-control_amd')
    os.symlink(os.path.join(ROOT, 'oracle'), work / 'oracle')
    for script in (['make_golden.py'], ['make_graph_golden.py'], ['make_graph_golden.py', '--fp32']):
        r = subprocess.run([sys.executable, str(work / 'tests' / 'golden' / script[0])] + script[1:], capture_output=True, text=True, timeout=900, cwd=str(work))
        assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    checked = 0
    for name in sorted(os.listdir(golden)):
        if not name.endswith('.npz') or name.startswith('c2_bench'):
            continue
        old, new = np.load(os.path.join(golden, name)), np.load(work / 'tests' / 'golden' / name)
        assert sorted(old.files) == sorted(new.files), name
        for k in old.files:
            if name.startswith('graph32_') and k == 'base_after_adam':
                continue                                   # float32 sign flips of the first Adam step (tests/test_oracle_golden.py)
            if name.startswith('graph') and old[k].dtype.kind in 'fc':
                tol = 1e-5 if name.startswith('graph32_') else 1e-14        # float32 runs: reductions reassociate at float32 round-off
                np.testing.assert_allclose(new[k], old[k], rtol=0, atol=tol * max(1.0, float(np.max(np.abs(old[k])))), err_msg='%s:%s' % (name, k))
            else:
                assert np.array_equal(old[k], new[k]), (name, k)
            checked += 1
    assert checked > 200
