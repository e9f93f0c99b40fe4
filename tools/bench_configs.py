#!/usr/bin/env python
"""Secondary configs of BASELINE.json (not the bench.py line): C1, C3 (state transfer), forbidden-regularised C2, n=64 unitary.
Prints iterations/s per config through the same C ABI."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'quantum-optimal-control_amd'))
import numpy as np  # noqa: E402

from quantum_optimal_control.core import hip_engine  # noqa: E402
from tests.golden import cases  # noqa: E402
from tests.helpers import oracle_system  # noqa: E402


def run(name, c, n_seeds, iters, path=0, variant=0, chunks=0):
    sp = oracle_system(c)
    eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms,
                               sp.scaling, state_transfer=sp.state_transfer, reg_coeffs=sp.reg_coeffs,
                               one_minus_gauss=sp.one_minus_gauss, Vs=sp.Vs, n_seeds=n_seeds, path=path, variant=variant, chunks=chunks)
    rng = np.random.default_rng(0)
    eng.set_base(rng.normal(0, 1 / np.sqrt(sp.steps), (n_seeds, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    # warm up for >= 0.3 s and time >= 0.5 s: a launch-bound run of a few iterations right after the engine was created can come out 2-3x
    # slow (GPU still at its idle clocks; the 17.76 ms line of profiles/r02_secondary_configs.txt was such a run)
    t0 = time.perf_counter()
    eng.iterate(p, 2); eng.sync()
    per = max((time.perf_counter() - t0) / 2, 1e-5)
    eng.iterate(p, max(1, min(2000, int(0.3 / per)))); eng.sync()
    iters = max(iters, min(5000, int(0.5 / per)))
    t0 = time.perf_counter()
    eng.iterate(p, iters); eng.sync()
    el = time.perf_counter() - t0
    s = eng.scalars()
    print('%-34s n=%-3d k=%d steps=%-4d m=%d seeds=%-3d path=%d chunks=%-2d : %9.1f it/s aggregate, %8.3f ms/iteration-batch, loss[0]=%.6f'
          % (name, sp.n, sp.k, sp.steps, sp.m, n_seeds, eng.path, eng.chunks, n_seeds * iters / el, el / iters * 1e3, s['loss'][0]))
    eng.close()


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'c3':           # profiling hook: C3 single trajectory only
        run('C3 state transfer (propagator route)', cases.case_c3(), 1, 20)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'c3x64':        # profiling hook: batched state transfer only
        run('C3 state transfer x64 seeds', cases.case_c3(), 64, 5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'c1':           # profiling hook: BASELINE config 1 on the workgroup-resident path (one control set, then 64)
        run('C1 single qubit', cases.case_c1(), 1, 2000)
        run('C1 single qubit x64 seeds', cases.case_c1(), 64, 2000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'n8small':      # profiling hook: two transmons (n = 8, k = 4, 500 slices), one control set
        run('n=8 one control set', cases.case_c2(n=8, k=4, steps=500, m=8, taylor=(5, 3), seed=2), 1, 2000)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'c5':           # profiling hook: large Hilbert space only
        run('C5 n=512 k=8 steps=2000 (GEMM path)', cases.case_c2(n=512, k=8, steps=2000, m=8, taylor=(5, 3), seed=2), 1, 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1][0] == 'n' and sys.argv[1][1:].isdigit():  # profiling hooks (n48, n64, n36, ...): NT = 3 / NT = 4 unitary gates x 64 seeds on the MFMA path
        nn = int(sys.argv[1][1:])
        run('n=%d unitary (MFMA NT=%d) x64' % (nn, (nn + 15) // 16), cases.case_c2(n=nn, k=4, steps=500, m=8, taylor=(5, 3), seed=2), 64, 5, path=2, variant=int(os.environ.get('QOC_VARIANT', '0')))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'c2reg':        # profiling hook: regularised C2 x 64 only
        c = cases.case_c2(); c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [30, 31]}
        run('C2 x64 + dwdt + forbidden', c, 64, 20)
        sys.exit(0)
    run('C1 single qubit', cases.case_c1(), 1, 50)
    run('C1 single qubit x64 seeds', cases.case_c1(), 64, 50)
    run('C2 single trajectory (AUTO: latency mode)', cases.case_c2(), 1, 20)
    run('C2 single trajectory (GEMM route)', cases.case_c2(), 1, 20, path=4)
    run('C2 single trajectory (MFMA kernels)', cases.case_c2(), 1, 20, path=2)
    run('C2 x64', cases.case_c2(), 64, 20)
    c = cases.case_c2(); c['reg_coeffs'] = {'dwdt': 1e-3, 'forbidden_coeff_list': [10.0, 10.0], 'states_forbidden_list': [30, 31]}
    run('C2 x64 + dwdt + forbidden', c, 64, 20)
    run('C3 state transfer (propagator route)', cases.case_c3(), 1, 5)
    run('C3 state transfer (fused mat-vec)', cases.case_c3(), 1, 5, path=3)
    run('C3 state transfer x64 seeds', cases.case_c3(), 64, 5)
    run('C3 state transfer x256 seeds', cases.case_c3(), 256, 5)
    run('n=64 unitary (GEMM path) x16', cases.case_c2(n=64, k=6, steps=200, m=8, taylor=(6, 3), seed=2), 16, 5, path=4)
    run('n=64 unitary (MFMA NT=4) x16', cases.case_c2(n=64, k=6, steps=200, m=8, taylor=(6, 3), seed=2), 16, 5, path=2)
    run('n=48 unitary (GEMM path) x64', cases.case_c2(n=48, k=4, steps=500, m=8, taylor=(5, 3), seed=2), 64, 3, path=4)
    run('n=48 unitary (MFMA NT=3) x64', cases.case_c2(n=48, k=4, steps=500, m=8, taylor=(5, 3), seed=2), 64, 3, path=2)
    run('n=128 unitary (GEMM path) x4', cases.case_c2(n=128, k=6, steps=500, m=8, taylor=(5, 3), seed=2), 4, 3)
    run('C5 n=512 k=8 steps=2000 (GEMM path)', cases.case_c2(n=512, k=8, steps=2000, m=8, taylor=(5, 3), seed=2), 1, 2)
