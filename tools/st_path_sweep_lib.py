"""Timing helper shared by the state-transfer sweeps: ms per iteration of a batch on a given path / variant / chunk request."""
import time
import numpy as np
from quantum_optimal_control.core import hip_engine


def ms(sp, B, path, variant, chunks=0):
    try:
        eng = hip_engine.HipEngine(sp.Hs, sp.U0, sp.V, sp.W, sp.maxA, sp.dt, sp.total_time, sp.steps, sp.exp_terms, sp.scaling, state_transfer=True,
                                   reg_coeffs=sp.reg_coeffs, n_seeds=B, path=path, variant=variant, chunks=chunks)
    except hip_engine.QocError:
        return float('nan'), -1
    eng.set_base(np.random.default_rng(0).normal(0, 1 / np.sqrt(sp.steps), (B, sp.k, sp.steps)))
    p = eng.adam_params(max_iterations=10 ** 9, conv_target=-1.0, min_grad=-1.0)
    t0 = time.perf_counter(); eng.iterate(p, 2); eng.sync()
    per = max((time.perf_counter() - t0) / 2, 1e-5)
    eng.iterate(p, max(1, min(2000, int(0.3 / per)))); eng.sync()
    it = max(5, min(3000, int(0.4 / per)))
    t0 = time.perf_counter(); eng.iterate(p, it); eng.sync()
    el = (time.perf_counter() - t0) / it * 1e3
    path_used = eng.path
    eng.close()
    return el, path_used
