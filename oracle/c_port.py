"""ctypes loader of oracle/_build/libqoc_oracle.so (C restatement of the unitary-mode iteration).  TEST INFRASTRUCTURE
ONLY: imported by tests and by bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, '_build', 'libqoc_oracle.so')
_DP = C.POINTER(C.c_double)


def load():
    lib = C.CDLL(LIB)
    lib.qoc_oracle_eval.restype = C.c_int
    lib.qoc_oracle_eval.argtypes = [C.c_int] * 7 + [_DP] * 10 + [C.c_int]
    lib.qoc_oracle_iterate.restype = C.c_int
    lib.qoc_oracle_iterate.argtypes = [C.c_int] * 7 + [_DP] * 6 + [C.c_int, C.c_double, C.c_double, _DP, C.c_int]
    lib.qoc_oracle_max_threads.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(_DP)


def _c(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.complex128))


def evaluate(sp, bases, nthreads=0):
    """sp: oracle.grape_oracle.OracleSystem (unitary mode, no regularisers); bases: (n_seeds, k, steps)."""
    lib = load()
    bases = np.ascontiguousarray(np.asarray(bases, dtype=np.float64))
    B = bases.shape[0]
    Hs, U0, V, W = _c(sp.Hs), _c(sp.U0), _c(sp.V), _c(sp.W)
    maxA = np.ascontiguousarray(sp.maxA, dtype=np.float64)
    loss, us = np.empty(B), np.empty(B)
    grad = np.empty_like(bases)
    Uf = np.empty((B, sp.n, sp.n), dtype=np.complex128)
    lib.qoc_oracle_eval(sp.n, sp.k, sp.steps, sp.m, sp.exp_terms, sp.scaling, B, _p(Hs.view(np.float64)),
                        _p(U0.view(np.float64)), _p(V.view(np.float64)), _p(W.view(np.float64)), _p(maxA), _p(bases),
                        _p(loss), _p(us), _p(grad), _p(Uf.view(np.float64)), int(nthreads))
    return dict(loss=loss, unitary_scale=us, grad=grad, U_final=Uf)


def iterate(sp, bases, iters, rate=0.01, decay=2500.0, nthreads=0):
    lib = load()
    bases = np.ascontiguousarray(np.array(bases, dtype=np.float64))
    B = bases.shape[0]
    Hs, U0, V, W = _c(sp.Hs), _c(sp.U0), _c(sp.V), _c(sp.W)
    maxA = np.ascontiguousarray(sp.maxA, dtype=np.float64)
    loss = np.empty(B)
    lib.qoc_oracle_iterate(sp.n, sp.k, sp.steps, sp.m, sp.exp_terms, sp.scaling, B, _p(Hs.view(np.float64)),
                           _p(U0.view(np.float64)), _p(V.view(np.float64)), _p(W.view(np.float64)), _p(maxA),
                           _p(bases), int(iters), float(rate), float(decay), _p(loss), int(nthreads))
    return bases, loss


def max_threads():
    return load().qoc_oracle_max_threads()
