"""Randomised differential test: random small problems (mode, sizes, Taylor order, regularisers, seed count) evaluated
by every engine path that accepts them, each compared with the NumPy oracle.  Seeds are fixed, so a failure reproduces.
Sizes stay small enough for the oracle to take milliseconds; the structured full-size checks live in test_hip_parity.py."""
import os

import numpy as np
import pytest

from tests.golden import cases
from tests.helpers import oracle_system
from tests.test_hip_parity import check_eval, make_engine

pytestmark = pytest.mark.gpu


def random_problem(seed):
    rng = np.random.default_rng(10_000 + seed)
    st = bool(rng.integers(0, 2))
    n = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24, 31, 32, 33, 40, 47, 48, 63, 64, 65, 70]))
    k = int(rng.integers(1, 4))
    steps = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65, 100, 130]))
    if n > 40:
        steps = min(steps, 33)                                    # keep the oracle fast
    T = int(rng.integers(1, 9))
    s = 0 if st else int(rng.integers(0, 4))
    if st:
        m = int(rng.integers(1, min(n, 9) + 1))
        c = cases.case_c3(n=max(n, 2), k=k, steps=steps, taylor=(T, 0), seed=seed)
        n = max(n, 2)
        vs = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(2 * m)]
        c['states_concerned_list'] = [v / np.linalg.norm(v) for v in vs[:m]]
        c['U'] = [v / np.linalg.norm(v) for v in vs[m:]]
        c['reg_coeffs'] = {}
    else:
        m = int(rng.integers(1, min(n, 12) + 1))
        c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(T, s), seed=seed)
    c['total_time'] = float(rng.uniform(0.2, 0.8)) * steps / 20.0          # |A_t| <= ~0.5: low Taylor orders stay bounded
    reg = {}
    if rng.random() < 0.4:
        reg['amplitude'] = float(rng.uniform(0.05, 0.5))
    if rng.random() < 0.4:
        reg['dwdt'] = float(rng.uniform(0.01, 0.2))
        if rng.random() < 0.5:
            reg['d2wdt2'] = float(rng.uniform(0.01, 0.1))
    if rng.random() < 0.3:
        reg['envelope'] = float(rng.uniform(0.05, 0.3))
    if rng.random() < 0.4 and n >= 3:
        f = rng.choice(n, size=min(2, n - 1), replace=False)
        reg['forbidden_coeff_list'] = [float(x) for x in rng.uniform(1, 5, size=len(f))]
        reg['states_forbidden_list'] = [int(x) for x in f]
    if rng.random() < 0.3:
        reg['speed_up'] = float(rng.uniform(0.1, 0.8))
    c['reg_coeffs'] = reg
    B = int(rng.choice([1, 2, 3]))
    return c, B, rng


@pytest.mark.parametrize('seed', range(128))
def test_random_problem_all_paths(seed):
    from quantum_optimal_control.core import hip_engine
    import oracle.grape_oracle as go
    for attempt in range(16):
        # an ill-conditioned draw (the truncated series blows up: relative parity is meaningless) is re-drawn, not skipped
        c, B, rng = random_problem(seed + 1000 * attempt)
        sp = oracle_system(c)
        us = go.evaluate(sp, sp.base0)['unitary_scale']
        if np.isfinite(us) and abs(us) <= 1e6:
            break
    else:
        raise AssertionError('no well-conditioned draw in 16 attempts from seed %d' % seed)
    bases = [sp.base0] + [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1 * (i + 1) for i in range(B - 1)]
    tried = 0
    for path, chunks, kernel in ((0, 0, 0), (1, 0, 0), (2, 0, 1), (2, 3, 2), (2, 2, 3), (2, 5, 4), (2, 0, 5), (2, 4, 5), (2, 3, 6), (2, 3, 7), (2, 4, 8), (3, 0, 0), (4, 0, 0), (4, 1, 0), (4, 2, 0), (5, 0, 0), (5, 3, 0)):
        try:
            eng = make_engine(sp, n_seeds=B, path=path, chunks=chunks, variant=kernel)
        except hip_engine.QocError:
            continue                                               # this path does not take this problem
        try:
            eng.set_base(np.stack(bases))
            check_eval(eng, sp, bases)
            tried += 1
        except AssertionError as exc:
            raise AssertionError('seed %d path %d chunks %d (n=%d k=%d steps=%d m=%d T=%d s=%d st=%s regs=%s): %s' % (
                seed, path, chunks, sp.n, sp.k, sp.steps, sp.m, sp.exp_terms, sp.scaling, sp.state_transfer,
                sorted(sp.reg_coeffs), exc))
        finally:
            eng.close()
    assert tried >= 2                                              # AUTO + generic at the very least


@pytest.mark.parametrize('seed', range(48))
def test_random_direct_route_one_vector(seed):
    """The Taylor-chain family of the direct state-transfer route with ONE state vector at 33 .. 64 levels (csrc/qoc_gemm_chain_dpp.h, qoc_gemm_chain_sq.h):
    random sizes (every class of active columns: <= 40 / 48 / 56 / 64), pulse lengths on every residue of the prefetch rotation, Taylor orders 1 .. 14,
    Hermitian (packed / active-column images) and lossy (full image) drifts, with and without sources, 1 .. 3 control sets -- the default chain and, where
    it applies, the squared-generator chain (variant 2), each against the oracle."""
    import oracle.grape_oracle as go
    rng = np.random.default_rng(77_000 + seed)
    for attempt in range(16):
        n = int(rng.integers(33, 65))
        steps = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 23, 31, 40]))
        T = int(rng.integers(1, 15))
        k = int(rng.integers(1, 5))
        c = cases.case_c3(n=n, k=k, steps=steps, taylor=(T, 0), seed=seed + 100 * attempt)
        c['total_time'] = float(rng.uniform(0.1, 0.5)) * steps / 10.0
        reg = {}
        if rng.random() < 0.5:
            f = rng.choice(n, size=2, replace=False)
            reg['forbidden_coeff_list'] = [float(x) for x in rng.uniform(1, 5, size=2)]
            reg['states_forbidden_list'] = [int(x) for x in f]
        if rng.random() < 0.4:
            reg['dwdt'] = float(rng.uniform(0.01, 0.2))
        if rng.random() < 0.25:
            reg['speed_up'] = float(rng.uniform(0.1, 0.8))
        c['reg_coeffs'] = reg
        lossy = rng.random() < 0.3
        if lossy:
            c['H0'] = c['H0'] + 0.05j * np.diag(np.arange(n) / n)            # generators no longer anti-Hermitian: the full image
        sp = oracle_system(c)
        us = go.evaluate(sp, sp.base0)['unitary_scale']
        if np.isfinite(us) and abs(us) <= 1e6:
            break
    else:
        raise AssertionError('no well-conditioned draw in 16 attempts from seed %d' % seed)
    B = int(rng.choice([1, 2, 3]))
    bases = [sp.base0] + [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1 * (i + 1) for i in range(B - 1)]
    kinds = set()
    for variant in (0, 2):
        eng = make_engine(sp, n_seeds=B, path=4, chunks=1, variant=variant)
        assert eng.plan.get('route') == 'direct', eng.plan
        kinds.add(eng.plan.get('taylor_chain'))
        eng.set_base(np.stack(bases))
        check_eval(eng, sp, bases)
        eng.close()
    expect = 'full' if (lossy and n > 56) else ('columns%d' % (40 if n <= 40 else 48 if n <= 48 else 56) if n <= 56 else 'packed')
    assert expect in kinds, (kinds, n, lossy)
    if not lossy and 3 <= T <= 14:
        assert 'squared' in kinds, (kinds, T)


@pytest.mark.parametrize('seed', range(int(os.environ.get('QOC_FUZZ_SMALL', '64'))))        # (QOC_FUZZ_SMALL=512: the long run after a change of the product primitive)
def test_random_small_path(seed):
    """The workgroup-resident path (QOC_PATH_SMALL, csrc/qoc_small_kernel.h) on random problems of 2 .. 12 levels: both modes, every regulariser it takes (all but the
    bandpass), 1 .. 3 control sets, pulses of 1 .. 300 slices over AUTO's / 1 / 2 / 3 / 7 workgroups per control set and 16 / 32 rows per workgroup -- one evaluation
    against the oracle, then five iterations of the loop inside ONE launch against the oracle's loop (run_session.py:47-69)."""
    from quantum_optimal_control.core import hip_engine
    import oracle.grape_oracle as go
    rng = np.random.default_rng(91_000 + seed)
    for attempt in range(16):
        st = bool(rng.integers(0, 2))
        n = int(rng.integers(2, 13))
        k = int(rng.integers(1, 6))
        steps = int(rng.choice([1, 2, 5, 16, 17, 31, 33, 64, 100, 129, 200, 300]))
        T = int(rng.integers(1, 10))
        if st:
            m = int(rng.integers(1, min(n, 4) + 1))
            c = cases.case_c3(n=n, k=k, steps=steps, taylor=(T, 0), seed=seed + 100 * attempt)
            vs = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(2 * m)]
            c['states_concerned_list'] = [v / np.linalg.norm(v) for v in vs[:m]]
            c['U'] = [v / np.linalg.norm(v) for v in vs[m:]]
        else:
            m = int(rng.integers(1, n + 1))
            c = cases.case_c2(n=n, k=k, steps=steps, m=m, taylor=(T, int(rng.integers(0, 4))), seed=seed + 100 * attempt)
        c['total_time'] = float(rng.uniform(0.2, 0.8)) * steps / 20.0
        reg = {}
        if rng.random() < 0.4:
            reg['amplitude'] = float(rng.uniform(0.05, 0.5))
        if rng.random() < 0.4:
            reg['dwdt'] = float(rng.uniform(0.01, 0.2))
            if rng.random() < 0.5:
                reg['d2wdt2'] = float(rng.uniform(0.01, 0.1))
        if rng.random() < 0.3:
            reg['envelope'] = float(rng.uniform(0.05, 0.3))
        if rng.random() < 0.45 and n >= 3:
            f = rng.choice(n, size=int(rng.integers(1, min(4, n - 1) + 1)), replace=False)
            reg['forbidden_coeff_list'] = [float(x) for x in rng.uniform(1, 5, size=len(f))]
            reg['states_forbidden_list'] = [int(x) for x in f]
        if rng.random() < 0.3:
            reg['speed_up'] = float(rng.uniform(0.1, 0.8))
        if rng.random() < 0.25 and steps >= 5:
            reg['bandpass'] = float(rng.uniform(0.05, 0.5))
            reg['band'] = [float(rng.uniform(0.1, 0.4)) * steps / c['total_time'] / 2, float(rng.uniform(0.5, 0.9)) * steps / c['total_time'] / 2]
        c['reg_coeffs'] = reg
        sp = oracle_system(c)
        us = go.evaluate(sp, sp.base0)['unitary_scale']
        if np.isfinite(us) and abs(us) <= 1e6:
            break
    else:
        raise AssertionError('no well-conditioned draw in 16 attempts from seed %d' % seed)
    B = int(rng.choice([1, 2, 3]))
    bases = [sp.base0] + [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1 * (i + 1) for i in range(B - 1)]
    tried = 0
    for groups, rows in ((0, 0), (1, 0), (2, 16), (3, 0), (7, 16), (0, 32)):
        try:
            eng = make_engine(sp, n_seeds=B, path=5, chunks=groups, variant=rows)
        except hip_engine.QocError as exc:
            assert 'a pulse that fits' in str(exc), exc           # (a pinned workgroup count / row count that has no instance for this pulse)
            continue
        what = 'seed %d groups %d rows %d plan %s (n=%d k=%d steps=%d m=%d T=%d s=%d st=%s regs=%s)' % (
            seed, groups, rows, eng.plan, sp.n, sp.k, sp.steps, sp.m, sp.exp_terms, sp.scaling, sp.state_transfer, sorted(sp.reg_coeffs))
        try:
            eng.set_base(np.stack(bases))
            check_eval(eng, sp, bases)
            if tried == 0 and sp.steps <= 130:
                conv = dict(rate=0.02, max_iterations=5, learning_rate_decay=50, conv_target=-1.0, min_grad=-1.0)
                ref = go.run_adam(sp, conv, base=bases[-1])
                eng.set_base(np.stack(bases))
                eng.iterate(eng.adam_params(**conv), 5)
                eng.sync()
                np.testing.assert_allclose(eng.get_base()[-1], ref['base'], rtol=0, atol=1e-10)
            tried += 1
        except AssertionError as exc:
            raise AssertionError('%s: %s' % (what, exc))
        finally:
            eng.close()
    # (long pulses of the largest sizes have no instance: their product trees do not fit 160 KB of LDS -- AUTO keeps those on the MFMA path)
    src = 'forbidden_coeff_list' in sp.reg_coeffs or 'speed_up' in sp.reg_coeffs      # (n > 10 with a state regulariser: two trees of 2304-byte nodes, <= ~30 slices)
    assert tried >= 1 or (sp.n >= 9 and sp.steps >= 64) or (sp.n >= 11 and src and sp.steps > 30) or ('bandpass' in sp.reg_coeffs and sp.steps > (128 if sp.n <= 4 else 32 if sp.n <= 8 else 16)), (
        sp.n, sp.steps, sorted(sp.reg_coeffs))       # (a bandpass regulariser: the pulse must fit ONE workgroup of the instance)
