// qoc_state_source.h -- source term of the state regularisers in the costate recursion, shared by every backward kernel.
// Reference: core/regularization_functions.py:69-97 (forbidden levels incl. the dressed rotation, speed_up), differentiated
// with respect to the propagated vectors.
#pragma once
#include "qoc_common.h"

// S[tau][a][j] = d(state regularisers)/dPsi_tau   (G = d/dRe + i d/dIm)
__device__ __forceinline__ cplx source_at(const QocDev& d, int b, int tau, int a, int j) {
    cplx s = cmake(0.0, 0.0);
    const int n = d.n, m = d.m;
    const cplx* p = d.inter + ((size_t)b * (d.steps + 1) + tau) * n * m;
    for (int f = 0; f < d.n_forb; ++f) {
        const int st = d.forb_state[f];
        if (d.forbid_dressed) {
            // 2 a_f |phi|^2 phi, phi = <dressed level f | Psi_tau[:, j]>: formed once per (tau, f, j) by k_loss of this evaluation (an
            // n-term dot product; recomputing it here for every row a made the dressed sources O(n) per entry: one C2 trajectory
            // 0.52 ms per iteration on the GEMM route against 0.29 with undressed levels)
            cfma(s, d.Vs[a * n + st], d.Fd[(((size_t)b * (d.steps + 1) + tau) * d.n_forb + f) * m + j]);
        } else if (a == st) {
            const cplx phi = p[st * m + j];
            const double pop = phi.x * phi.x + phi.y * phi.y;
            s = cadd(s, cscale(phi, 2.0 * d.forb_a[f] * pop));
        }
    }
    if (d.has_speed) {
        const double coef = -d.a_speed * d.su_resid[b] * 2.0 / ((double)m * (double)m);
        const cplx z = d.ztau[(size_t)b * (d.steps + 1) + tau];
        s = cadd(s, cscale(cmul(z, d.W[a * m + j]), coef));
    }
    return s;
}
