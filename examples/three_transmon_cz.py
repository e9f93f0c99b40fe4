#!/usr/bin/env python
"""CZ between the first two of THREE coupled 3-level transmons (n = 27, six controls: a drive and a frequency knob per transmon), the third one
left alone -- the reference's transmon set-up (unitary mode, forbidden second-excited levels, dwdt regulariser) with the `restarts` extension:
a batch of random restarts is optimised at once and the best one returned.  On the MI355X a batch of n = 27 runs the batch kernels of the MFMA
path on the 7 active 4-row strips of the padded 32 x 32 matrices; a single control set (restarts = 1) runs the latency mode.

    python examples/three_transmon_cz.py [--iterations N] [--restarts R]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'quantum-optimal-control_amd'))
from quantum_optimal_control.main_grape.grape import Grape  # noqa: E402


def main(iterations=800, restarts=16, quiet=False):
    lv, alpha, J = 3, -0.25, 0.015                     # levels per transmon, anharmonicity and nearest-neighbour coupling in GHz
    a = np.diag(np.sqrt(np.arange(1, lv)), 1).astype(complex)
    I = np.eye(lv, dtype=complex)
    kron3 = lambda x, y, z: np.kron(np.kron(x, y), z)  # noqa: E731
    A = [kron3(a, I, I), kron3(I, a, I), kron3(I, I, a)]
    N = [x.conj().T @ x for x in A]
    H0 = 2 * np.pi * (sum((alpha / 2) * (n @ n - n) for n in N) + J * sum(A[i].conj().T @ A[i + 1] + A[i + 1].conj().T @ A[i] for i in range(2)))
    Hops = [2 * np.pi * n for n in N] + [2 * np.pi * (x + x.conj().T) / 2 for x in A]
    Hnames = ['z1', 'z2', 'z3', 'x1', 'x2', 'x3']
    idx = lambda i, j, k: (i * lv + j) * lv + k        # noqa: E731
    comp = [idx(i, j, k) for i in (0, 1) for j in (0, 1) for k in (0, 1)]           # the eight computational states
    U = np.eye(lv ** 3, dtype=complex)
    for k in (0, 1):
        U[idx(1, 1, k), idx(1, 1, k)] = -1.0           # CZ on transmons 1, 2; identity on the third
    gate_levels = {idx(0, 2, k) for k in (0, 1)} | {idx(2, 0, k) for k in (0, 1)}    # |02k>, |20k> take part in the CZ
    leak = [s for s in range(lv ** 3) if s not in comp and s not in gate_levels]
    steps, total_time = 400, 60.0
    convergence = {'rate': 0.02, 'update_step': 100, 'max_iterations': iterations, 'conv_target': 1e-4, 'learning_rate_decay': 2000}
    reg = {'dwdt': 0.01, 'forbidden_coeff_list': [2.0] * len(leak), 'states_forbidden_list': leak}
    np.random.seed(7)
    uks, U_final = Grape(H0, Hops, Hnames, U, total_time=total_time, steps=steps, states_concerned_list=comp, convergence=convergence,
                         reg_coeffs=reg, maxA=[0.3] * 3 + [0.05] * 3, method='Adam', show_plots=not quiet, save=False, restarts=restarts)
    from scipy.linalg import expm                       # re-simulate the returned pulse with exact slice propagators
    dt, X = total_time / steps, np.eye(lv ** 3, dtype=complex)
    for t in range(steps):
        X = expm(-1j * dt * (H0 + sum(uks[k, t] * Hops[k] for k in range(len(Hops))))) @ X
    overlap = sum(np.vdot(U[:, i], X[:, i]) for i in comp)
    fidelity = abs(overlap) ** 2 / len(comp) ** 2
    print('pulse shape %s, best of %d restarts: CZ (x) 1 fidelity on the computational subspace = %.6f' % (uks.shape, restarts, fidelity))
    return fidelity


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iterations', type=int, default=800)
    ap.add_argument('--restarts', type=int, default=16)
    args = ap.parse_args()
    main(args.iterations, args.restarts)
