"""CPU ORACLE (TEST INFRASTRUCTURE ONLY) -- op-by-op emulation of the reference's TensorFlow graph in torch-CPU.

Purpose: an *independent* pin for ``grape_oracle.py``.  TensorFlow cannot run in the build container, so the graph
of core/tensorflow_state.py is re-stated here node for node in its own representation -- real-embedded
2n x 2n matrices / 2n vectors -- and differentiated with torch autograd.  The two ``Defun`` operators carry
hand-written gradient functions in the reference (they are NOT the exact derivative); they are mirrored with
``torch.autograd.Function`` so that autograd reproduces exactly what TF would back-propagate:

  matexp_op / matexp_op_grad        tensorflow_state.py:25-46, 49-65, 70-75
  matvecexp_op / matvecexp_op_grad  tensorflow_state.py:77-97, 100-133, 137-142

``dtype=torch.float32`` gives the faithful (reference-precision) mode; ``torch.float64`` isolates the algebra.
Only tests and the bench's cpu_baseline leg may import this file.
"""
import numpy as np
import torch

from .grape_oracle import c_to_r_mat, c_to_r_vec, sort_ev


def _get_matexp(uks, H_all, input_num, taylor_terms, scaling):
    """tensorflow_state.py:25-46."""
    I = H_all[input_num]
    out = I
    H = sum((uks[ii] / (2. ** scaling)) * H_all[ii] for ii in range(input_num))
    Hn = H
    fact = 1.
    for ii in range(1, taylor_terms + 1):
        fact = fact * ii
        out = out + Hn / fact
        if ii != taylor_terms:
            Hn = H @ Hn
    for _ in range(scaling):
        out = out @ out
    return out


def _get_matvecexp(uks, H_all, psi, input_num, taylor_terms, sign=1.0):
    """tensorflow_state.py:77-97 (sign=+1) and the vec_grad loop :118-131 (sign=-1)."""
    out = psi
    H = sum((sign * uks[ii]) * H_all[ii] for ii in range(input_num))
    pn = psi
    fact = 1.
    for ii in range(1, taylor_terms):
        fact = fact * ii
        pn = H @ pn
        out = out + pn / fact
    return out


class MatExpOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uks, H_all, input_num, taylor_terms, scaling):
        ctx.save_for_backward(uks, H_all)
        ctx.cfg = (input_num, taylor_terms, scaling)
        return _get_matexp(uks, H_all, input_num, taylor_terms, scaling)

    @staticmethod
    def backward(ctx, grad):
        uks, H_all = ctx.saved_tensors
        input_num, taylor_terms, scaling = ctx.cfg
        matexp = _get_matexp(uks, H_all, input_num, taylor_terms, scaling)       # recompute, :58
        coeff = [torch.zeros((), dtype=grad.dtype)]                              # :54
        for ii in range(1, input_num):
            coeff.append(torch.sum(grad * (H_all[ii] @ matexp)))                 # :61-63
        return torch.stack(coeff), torch.zeros_like(H_all), None, None, None     # :65


class MatVecExpOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uks, H_all, psi, input_num, taylor_terms):
        ctx.save_for_backward(uks, H_all, psi)
        ctx.cfg = (input_num, taylor_terms)
        return _get_matvecexp(uks, H_all, psi, input_num, taylor_terms)

    @staticmethod
    def backward(ctx, grad):
        uks, H_all, psi = ctx.saved_tensors
        input_num, taylor_terms = ctx.cfg
        mve = _get_matvecexp(uks, H_all, psi, input_num, taylor_terms)           # :107
        coeff = [torch.zeros((), dtype=grad.dtype)]
        for ii in range(1, input_num):
            coeff.append(torch.sum(grad * (H_all[ii] @ mve)))                    # :112-114
        vec_grad = _get_matvecexp(uks, H_all, grad, input_num, taylor_terms, sign=-1.0)   # :118-131
        return torch.stack(coeff), torch.zeros_like(H_all), vec_grad, None, None


def _l2(x):
    return torch.sum(x * x) / 2                                                  # tf.nn.l2_loss


def _inner_product_2D(psi1, psi2, n, m):
    """tensorflow_state.py:282-300."""
    a, b = psi1[0:n, :], psi1[n:2 * n, :]
    c, d = psi2[0:n, :], psi2[n:2 * n, :]
    ac = torch.sum(a * c, 0)
    bd = torch.sum(b * d, 0)
    bc = torch.sum(b * c, 0)
    ad = torch.sum(a * d, 0)
    reals = torch.sum(ac + bd) ** 2
    imags = torch.sum(bc - ad) ** 2
    return (reals + imags) / (m ** 2)


def _inner_product_3D(psi1, psi2, n, m):
    """tensorflow_state.py:302-321; psi: (2n, steps+1, m)."""
    a, b = psi1[0:n], psi1[n:2 * n]
    c, d = psi2[0:n], psi2[n:2 * n]
    ac = torch.sum(a * c, 0)
    bd = torch.sum(b * d, 0)
    bc = torch.sum(b * c, 0)
    ad = torch.sum(a * d, 0)
    reals = torch.sum(torch.sum(ac + bd, 1) ** 2)
    imags = torch.sum(torch.sum(bc - ad, 1) ** 2)
    return (reals + imags) / (m ** 2)


def evaluate_graph(sysp, base, dtype=torch.float64):
    """Build + evaluate the reference graph at ``base``; returns the run_session fetch set
    (grad_pack, loss, reg_loss, unitary_scale, grad_squared, final_state real-embedded, inter_vecs_packed)."""
    n, k, steps, m = sysp.n, sysp.k, sysp.steps, sysp.m
    T, s = sysp.exp_terms, sysp.scaling
    input_num = k + 1
    tt = lambda x: torch.tensor(np.asarray(x), dtype=dtype)
    matrix_list = tt(sysp.matrix_list())                                         # :205
    packed_V = tt(np.stack([c_to_r_vec(sysp.V[:, j]) for j in range(m)], axis=1))     # :150-156 (2n x m)
    base_t = tt(base).clone().requires_grad_(True)                               # ops_weight_base :174
    ops_weight = torch.sin(base_t)                                               # :176
    maxA = tt(sysp.maxA)
    H_weights = torch.cat([torch.ones(1, steps, dtype=dtype), maxA[:, None] * ops_weight], 0)   # :172-181

    if not sysp.state_transfer:
        target_vecs = tt(c_to_r_mat(sysp.U_target)) @ packed_V                   # :164
        ops = [MatExpOp.apply(H_weights[:, t], matrix_list, input_num, T, s) for t in range(steps)]   # :209-211
        X = ops[0] @ tt(c_to_r_mat(sysp.U0))                                     # :214
        inter_states = [X]
        for t in range(1, steps):
            X = ops[t] @ inter_states[t - 1]                                     # :218-220
            inter_states.append(X)
        final_state = inter_states[steps - 1]
        unitary_scale = (0.5 / n) * torch.sum(final_state.t() @ final_state)     # :225
        vec_list = [packed_V] + [inter_states[t] @ packed_V for t in range(steps)]    # :229-238
        inter_packed = torch.stack(vec_list, dim=1)                              # (2n, steps+1, m)
        final_vecs = final_state @ packed_V                                      # :326
        loss = 1 - _inner_product_2D(final_vecs, target_vecs, n, m)
    else:
        target_vecs = tt(np.stack([c_to_r_vec(sysp.W[:, j]) for j in range(m)], axis=1))   # :161
        vec = packed_V
        vec_list = [vec]
        for t in range(steps):                                                   # :253-256
            vec = MatVecExpOp.apply(H_weights[:, t], matrix_list, vec, input_num, T)
            vec_list.append(vec)
        inter_packed = torch.stack(vec_list, dim=1)
        final_state = inter_packed[:, steps, :]
        loss = 1 - _inner_product_2D(final_state, target_vecs, n, m)
        unitary_scale = _inner_product_2D(final_state, final_state, n, m)        # :335

    # ---- get_reg_loss (regularization_functions.py:7-97) ------------------------------------------------------
    rc = sysp.reg_coeffs
    reg_loss = loss
    if 'amplitude' in rc:
        reg_loss = reg_loss + (rc['amplitude'] / float(steps)) * _l2(ops_weight)
    if 'envelope' in rc:
        reg_loss = reg_loss + (rc['envelope'] / float(steps)) * _l2(tt(sysp.one_minus_gauss) * ops_weight)
    if 'dwdt' in rc:
        z2 = torch.zeros(k, 2, dtype=dtype)
        nw = torch.cat([z2, torch.cat([ops_weight, z2], 1)], 1)
        reg_loss = reg_loss + (rc['dwdt'] / float(steps)) * _l2((nw[:, 1:] - nw[:, :steps + 3]) / sysp.dt)
    if 'd2wdt2' in rc:
        reg_loss = reg_loss + (rc['d2wdt2'] / float(steps)) * _l2(
            (nw[:, 2:] - 2 * nw[:, 1:steps + 3] + nw[:, :steps + 2]) / (sysp.dt ** 2))
    if 'bandpass' in rc:
        a = rc['bandpass'] / float(steps)
        F = torch.abs(torch.fft.fft(ops_weight.to(torch.complex128 if dtype == torch.float64 else torch.complex64)))
        band_id = (np.array(rc['band']) * sysp.total_time).astype(int)
        half_id = int(steps / 2)
        reg_loss = reg_loss + a * (torch.sum(F[:, 0:band_id[0]]) + torch.sum(F[:, band_id[1]:half_id]))
    if 'forbidden_coeff_list' in rc:
        v_sorted = None
        if sysp.is_dressed:
            v_sorted = tt(c_to_r_mat(np.reshape(sort_ev(sysp.v_c, sysp.dressed_id), [n, n])))
        for j in range(m):
            inter_vec = inter_packed[:, :, j]                                    # (2n, steps+1)
            if sysp.is_dressed and rc.get('forbid_dressed', False):
                inter_vec = v_sorted.t() @ inter_vec
            for coeff, state in zip(rc['forbidden_coeff_list'], rc['states_forbidden_list']):
                pop = inter_vec[state, :] ** 2 + inter_vec[n + state, :] ** 2
                reg_loss = reg_loss + (coeff / float(steps)) * _l2(pop)
    if 'speed_up' in rc:
        tgt_all = target_vecs.reshape(2 * n, 1, m).repeat(1, steps + 1, 1)
        ip = _inner_product_3D(inter_packed, tgt_all, n, m)
        reg_loss = reg_loss + (rc['speed_up'] / float(steps)) * _l2(steps + 1 - ip)

    (g,) = torch.autograd.grad(reg_loss, base_t)                                 # compute_gradients :348
    return dict(grad=g.detach().numpy().astype(np.float64), loss=float(loss.detach()), reg_loss=float(reg_loss.detach()),
                unitary_scale=float(unitary_scale.detach()), grad_squared=float(_l2(g)),
                final_state=final_state.detach().numpy(), inter_vecs_packed=inter_packed.detach().numpy())
