"""Synthetic benchmark systems of SURVEY.md 8(d): seeded random Hamiltonians with the reference's problem shapes.

The reference ships no benchmark inputs (its examples are notebooks); these recipes produce the dicts of `Grape()` keyword
inputs that bench.py, tools/ and the parity tests all share.  Only INPUT construction lives here.
"""
import numpy as np


def herm(rng, n):
    A = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    H = (A + A.conj().T) / 2
    return H / np.linalg.norm(H, 2)


def random_unitary(rng, n):
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n)))
    return Q


def case_c2(n=32, k=4, steps=500, m=8, taylor=(5, 3), seed=0):
    """SURVEY 8(d) C2 recipe (also used, scaled down, for quick parity cases)."""
    rng = np.random.default_rng(seed)
    H0 = 2 * np.pi * 2 * herm(rng, n)
    Hops = [0.2 * 2 * np.pi * 2 * herm(rng, n) for _ in range(k)]
    U = random_unitary(rng, n)
    return dict(H0=H0, Hops=Hops, Hnames=['h%d' % i for i in range(k)], U=U, total_time=100.0 * steps / 500.0,
                steps=steps, states_concerned_list=list(range(m)), maxA=[4.0] * k, reg_coeffs={},
                Taylor_terms=list(taylor) if taylor is not None else None, state_transfer=False, initial_guess=None,
                dressed_info=None, U0=None, np_seed=seed)


def case_c3(n=64, k=6, steps=1000, taylor=(10, 0), seed=3):
    """state transfer e_0 -> e_1 with dwdt + forbidden regularisers."""
    rng = np.random.default_rng(seed)
    H0 = 2 * np.pi * 2 * herm(rng, n)
    Hops = [0.2 * 2 * np.pi * 2 * herm(rng, n) for _ in range(k)]
    e0 = np.zeros(n, dtype=complex); e0[0] = 1
    e1 = np.zeros(n, dtype=complex); e1[1] = 1
    return dict(H0=H0, Hops=Hops, Hnames=['h%d' % i for i in range(k)], U=[e1], total_time=200.0 * steps / 1000.0,
                steps=steps, states_concerned_list=[e0], maxA=[4.0] * k,
                reg_coeffs={'dwdt': 1e-3, 'forbidden_coeff_list': [100, 100], 'states_forbidden_list': [n - 2, n - 1]},
                Taylor_terms=list(taylor) if taylor is not None else None, state_transfer=True, initial_guess=None,
                dressed_info=None, U0=None, np_seed=seed)
