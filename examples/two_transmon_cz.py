#!/usr/bin/env python
"""CZ gate on the computational subspace of two coupled 3-level transmons (unitary mode, n = 9, forbidden-state and dwdt
regularisers): ONE control set, the reference's own way of calling Grape().  On the MI355X this shape runs in the latency mode of
the MFMA path (n padded to 32; the forbidden levels make the costate affine, so the backward half is the two-level affine recursion).

    python examples/two_transmon_cz.py [--iterations N]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'quantum-optimal-control_amd'))
from quantum_optimal_control.main_grape.grape import Grape  # noqa: E402


def main(iterations=600, quiet=False):
    lv, alpha, J = 3, -0.25, 0.02                      # levels per transmon, anharmonicity and coupling in GHz
    a = np.diag(np.sqrt(np.arange(1, lv)), 1).astype(complex)
    I = np.eye(lv, dtype=complex)
    a1, a2 = np.kron(a, I), np.kron(I, a)
    n1, n2 = a1.conj().T @ a1, a2.conj().T @ a2
    H0 = 2 * np.pi * ((alpha / 2) * (n1 @ n1 - n1) + (alpha / 2) * (n2 @ n2 - n2) + J * (a1.conj().T @ a2 + a2.conj().T @ a1))
    Hops = [2 * np.pi * n1, 2 * np.pi * n2, 2 * np.pi * (a1 + a1.conj().T) / 2, 2 * np.pi * (a2 + a2.conj().T) / 2]
    Hnames = ['z1', 'z2', 'x1', 'x2']
    comp = [0, 1, lv, lv + 1]                          # |00>, |01>, |10>, |11>
    U = np.eye(lv * lv, dtype=complex)
    U[lv + 1, lv + 1] = -1.0                           # CZ on the computational subspace (identity elsewhere: not judged)
    leak = [i for i in range(lv * lv) if i not in comp and i not in (2, 2 * lv)]    # |02>, |20> take part in the gate; the rest is leakage
    steps, total_time = 400, 60.0
    convergence = {'rate': 0.02, 'update_step': 100, 'max_iterations': iterations, 'conv_target': 1e-4, 'learning_rate_decay': 1500}
    reg = {'dwdt': 0.01, 'forbidden_coeff_list': [5.0] * len(leak), 'states_forbidden_list': leak}
    np.random.seed(4)
    uks, U_final = Grape(H0, Hops, Hnames, U, total_time=total_time, steps=steps, states_concerned_list=comp,                 # basis-state indices, as the reference takes them in unitary mode
                         convergence=convergence, reg_coeffs=reg, maxA=[0.3, 0.3, 0.05, 0.05], method='Adam',
                         show_plots=not quiet, save=False)
    # re-simulate the returned pulse with exact slice propagators
    from scipy.linalg import expm
    dt, X = total_time / steps, np.eye(lv * lv, dtype=complex)
    for t in range(steps):
        X = expm(-1j * dt * (H0 + sum(uks[k, t] * Hops[k] for k in range(len(Hops))))) @ X
    overlap = sum(np.vdot(U[:, i], X[:, i]) for i in comp)
    fidelity = abs(overlap) ** 2 / len(comp) ** 2
    print('pulse shape %s, CZ fidelity on the computational subspace = %.6f' % (uks.shape, fidelity))
    return fidelity


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--iterations', type=int, default=600)
    args = ap.parse_args()
    main(args.iterations)
