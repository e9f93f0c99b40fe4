"""Seeded input recipes shared by the golden-fixture generator (needs /root/reference, build container only) and by
the tests (no reference needed).  Only INPUT construction lives here; expected outputs are in the .npz fixtures."""
import numpy as np


from quantum_optimal_control.helper_functions.synthetic_systems import herm, random_unitary, case_c2, case_c3  # noqa: F401


def case_c1():
    """single-qubit pi pulse."""
    sz = np.array([[1, 0], [0, -1]], dtype=complex)
    sx = np.array([[0, 1], [1, 0]], dtype=complex)
    return dict(H0=2 * np.pi * 0.05 * sz / 2, Hops=[2 * np.pi * sx / 2], Hnames=['x'], U=sx, total_time=10.0,
                steps=100, states_concerned_list=[0, 1], maxA=[0.2], reg_coeffs={}, Taylor_terms=None,
                state_transfer=False, initial_guess=None, dressed_info=None, U0=None, np_seed=11)


def case_small_auto(seed=5):
    """n<10 branch of the Taylor heuristic (matrix metric), U0 != I, automatic (T,s)."""
    rng = np.random.default_rng(seed)
    n, k, steps = 4, 2, 40
    return dict(H0=2 * np.pi * 0.4 * herm(rng, n), Hops=[2 * np.pi * 0.1 * herm(rng, n) for _ in range(k)],
                Hnames=['a', 'b'], U=random_unitary(rng, n), total_time=8.0, steps=steps,
                states_concerned_list=[0, 1, 2], maxA=[2.0, 3.0], reg_coeffs={'amplitude': 0.1},
                Taylor_terms=None, state_transfer=False, initial_guess=None, dressed_info=None,
                U0=random_unitary(rng, n), np_seed=seed)


def case_big_auto(seed=6):
    """n>=10 branch of the Taylor heuristic (scalar surrogate), automatic (T,s)."""
    rng = np.random.default_rng(seed)
    n, k, steps = 12, 3, 60
    return dict(H0=2 * np.pi * 1.5 * herm(rng, n), Hops=[2 * np.pi * 0.3 * herm(rng, n) for _ in range(k)],
                Hnames=['a', 'b', 'c'], U=random_unitary(rng, n), total_time=20.0, steps=steps,
                states_concerned_list=[0, 1, 2, 3], maxA=[4.0, 4.0, 2.0], reg_coeffs={},
                Taylor_terms=None, state_transfer=False, initial_guess=None, dressed_info=None, U0=None,
                np_seed=seed)


def case_guess(seed=7):
    """Explicit initial_guess (arcsin transform) + maxA default 1.5*max|guess| (grape.py:95-101)."""
    rng = np.random.default_rng(seed)
    n, k, steps = 3, 2, 25
    guess = 0.8 * rng.uniform(-1, 1, size=(k, steps))
    return dict(H0=2 * np.pi * 0.2 * herm(rng, n), Hops=[2 * np.pi * 0.1 * herm(rng, n) for _ in range(k)],
                Hnames=['a', 'b'], U=random_unitary(rng, n), total_time=5.0, steps=steps,
                states_concerned_list=[0, 1], maxA=None, reg_coeffs={}, Taylor_terms=None, state_transfer=False,
                initial_guess=guess, dressed_info=None, U0=None, np_seed=seed)


def case_dressed(seed=8):
    """dressed_info path: initial vectors are dressed eigenvectors (system_parameters.py:178-179)."""
    rng = np.random.default_rng(seed)
    n, k, steps = 6, 2, 30
    H0 = np.diag(np.arange(n) * 1.0).astype(complex) + 0.05 * herm(rng, n)
    return dict(H0=H0, Hops=[2 * np.pi * 0.1 * herm(rng, n) for _ in range(k)], Hnames=['a', 'b'],
                U=random_unitary(rng, n), total_time=6.0, steps=steps, states_concerned_list=[0, 1, 3],
                maxA=[1.0, 2.0],
                reg_coeffs={'forbidden_coeff_list': [10.0], 'states_forbidden_list': [5], 'forbid_dressed': True},
                Taylor_terms=None, state_transfer=False, initial_guess=None, dressed_info='from_H0', U0=None,
                np_seed=seed)


def case_state_small(seed=9):
    """Small state-transfer case, automatic Taylor order (forces scaling 0, one candidate)."""
    rng = np.random.default_rng(seed)
    n, k, steps, m = 5, 2, 30, 2
    init = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(m)]
    init = [v / np.linalg.norm(v) for v in init]
    tg = [rng.normal(size=n) + 1j * rng.normal(size=n) for _ in range(m)]
    tg = [v / np.linalg.norm(v) for v in tg]
    return dict(H0=2 * np.pi * 0.2 * herm(rng, n), Hops=[2 * np.pi * 0.1 * herm(rng, n) for _ in range(k)],
                Hnames=['a', 'b'], U=tg, total_time=4.0, steps=steps, states_concerned_list=init, maxA=[2.0, 2.0],
                reg_coeffs={'dwdt': 0.01}, Taylor_terms=None, state_transfer=True, initial_guess=None,
                dressed_info=None, U0=None, np_seed=seed)


ALL_CASES = dict(c1=case_c1, c2=case_c2, c3_small=lambda: case_c3(n=16, k=3, steps=50, taylor=(10, 0)),
                 small_auto=case_small_auto, big_auto=case_big_auto, guess=case_guess, dressed=case_dressed,
                 state_small=case_state_small)
