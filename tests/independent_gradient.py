"""A pin of the reference's FIRST-ORDER gradient (core/tensorflow_state.py:49-65) that shares no code with
oracle/grape_oracle.py: the custom gradient of matexp_op hands control k of slice t the inner product
<dL/dK_t, H_k' K_t>, i.e. the directional derivative of the loss along K_t -> K_t + eps H_k' K_t with every other slice
propagator frozen.  Here that derivative is taken numerically (central differences) through a plain NumPy loss."""
import numpy as np


def slice_propagators(Hs, u, T, s):
    """K_t = (sum_{j<=T} A^j/j!)^(2^s), A = (H0' + sum_k u_k,t H_k') / 2^s      (tensorflow_state.py:25-46)."""
    steps = u.shape[1]
    n = Hs.shape[1]
    Ks = np.empty((steps, n, n), dtype=complex)
    for t in range(steps):
        A = (Hs[0] + np.tensordot(u[:, t], Hs[1:], axes=1)) / 2.0 ** s
        term, acc, fact = np.eye(n, dtype=complex), np.eye(n, dtype=complex), 1.0
        for j in range(1, T + 1):
            term = term @ A
            fact *= j
            acc = acc + term / fact
        Ks[t] = np.linalg.matrix_power(acc, 2 ** s)
    return Ks


def loss_from_propagators(Ks, psi0, W):
    """1 - |sum_j <w_j, psi_j(T)>|^2 / m^2      (tensorflow_state.py:282-333)."""
    psi = psi0
    for K in Ks:
        psi = K @ psi
    z = np.sum(np.conj(W) * psi)
    return 1.0 - abs(z) ** 2 / W.shape[1] ** 2


def first_order_gradient(Hs, U0, V, W, maxA, base, T, s, pairs, h=1e-5):
    """d loss / d base[k, t] for the listed (k, t) pairs, reference semantics (first order in the slice generator)."""
    u = np.asarray(maxA)[:, None] * np.sin(base)
    Ks = slice_propagators(Hs, u, T, s)
    psi0 = U0 @ V
    out = {}
    for (k, t) in pairs:
        vals = []
        for eps in (h, -h):
            Kp = Ks.copy()
            Kp[t] = Ks[t] + eps * (Hs[k + 1] @ Ks[t])
            vals.append(loss_from_propagators(Kp, psi0, W))
        dLdu = (vals[0] - vals[1]) / (2 * h)
        out[(k, t)] = dLdu * maxA[k] * np.cos(base[k, t])            # u = maxA sin(base)      tensorflow_state.py:176-178
    return out
