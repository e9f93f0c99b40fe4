// qoc_kernels_gemm.h -- "GEMM path" (QOC_PATH_GEMM): any n, m <= 32, unitary mode and state transfer.
//
// Matrices are zero-padded to N = 32*ceil(n/32) and live in HBM/L2 as plain row-major complex128.  Time is cut into NC
// chunks of S = 2^L ~ sqrt(steps) slices so that every chain has NC + S sequential steps instead of `steps`:
//   * exponentials for all (seed, slice) pairs: k_gemm_expm_fused (N <= 64, qoc_gemm_expm.h) or batched k_zgemm32 launches
//     (qoc_gemm_tiles.h) of the Paterson-Stockmeyer polynomial + squarings;
//   * a pairwise product tree gives the chunk products (and, in unitary mode, final_state at the root);
//   * forward / backward: chunk boundaries sequentially, then all chunks swept in parallel -- persistent VALU chain
//     kernels (qoc_gemm_chains.h) for N <= 64, m <= 8, one batched k_zgemm32 launch per step otherwise;
//   * control gradients: products H_k' [Psi_0 ... Psi_t ...] with a dot-product epilogue against conj(Lambda).
// State transfer runs either through the same propagators (anti-Hermitian generators) or "direct" (k_gemm_taylor_chain).
// This file: the small helper kernels and the host-side orchestration (QocGemm, qoc_gemm_setup/expm/forward/backward).
// Reference semantics: core/tensorflow_state.py:25-46, 49-65, 77-133, 204-261.
#pragma once
#include <string>
#include <vector>
#include "qoc_common.h"
#include "qoc_gemm_tiles.h"
#include "qoc_gemm_expm.h"
#include "qoc_gemm_chains.h"

// A_t = (H0' + sum_k u_k H_k') / 2^s for every (seed, slice), padded N x N           tensorflow_state.py:30-33
// Slices are padded to SP = NC*S per seed; a padded slice gets A = 0, i.e. K = I exactly.
// (item_first, item_count): the (seed, slice) items this launch assembles -- all of them, or the slices of one rank of a time-sharded
// engine
__global__ void __launch_bounds__(256) k_gemm_assemble(QocDev d, const cplx* __restrict__ HsP, cplx* __restrict__ Aout, int N, int SP,
    int sq,
                                                        size_t item_first, size_t item_count, int nn = 0) {
    const size_t NN = nn > 0 ? (size_t)nn : (size_t)N * N;         // (nn: entries per matrix of a packed stack, as in k_gemm_assemble_rows)
    const size_t total = item_count * NN;
    const double inv = 1.0 / (double)(1 << sq);
    for (size_t o0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o0 < total; o0 += (size_t)gridDim.x * blockDim.x) {
        const size_t o = o0 + item_first * NN;
        const size_t item = o / NN, e = o - item * NN;
        const int b = (int)(item / SP), t = (int)(item - (size_t)b * SP);
        cplx acc = cmake(0.0, 0.0);
        if (t < d.steps) {
            acc = cscale(HsP[e], inv);
            for (int kk = 0; kk < d.k; ++kk) {
                const double c = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv;
                const cplx h = HsP[(size_t)(kk + 1) * NN + e];
                acc.x = fma(c, h.x, acc.x); acc.y = fma(c, h.y, acc.y);
            }
        }
        Aout[o] = acc;
    }
}
// The same with the k + 1 Hamiltonian entries of a thread held in registers over a run of (seed, slice) items (k <= 8, N*N a multiple of
// 256): k_gemm_assemble re-reads them from L2 for every output entry -- (k + 1) x the written bytes through L2, 2.0 ms for the 4.2 GB of
// C3 x 64 -- this one is bound by the HBM writes alone.  blockIdx.x = 256-entry column of the matrix, blockIdx.y = run of items.
// (t0, tn): with tn > 0 the items are the slices t0 .. t0 + tn - 1 of EVERY seed (item = b * tn + t - t0), written to their usual place
// nn > 0: entries per matrix of the stack and of the output when that is not N * N (the packed anti-Hermitian image of
// qoc_gemm_chain_dpp.h: 2560)
__global__ void __launch_bounds__(256) k_gemm_assemble_rows(QocDev d, const cplx* __restrict__ HsP, cplx* __restrict__ Aout, int N, int SP,
    int sq, int per,
                                                             size_t item_first, size_t item_count, int t0 = 0, int tn = 0, int nn = 0) {
    const size_t NN = nn > 0 ? (size_t)nn : (size_t)N * N;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const double inv = 1.0 / (double)(1 << sq);
    cplx h[9];
#pragma unroll
    for (int kk = 0; kk < 9; ++kk) h[kk] = kk <= d.k ? cscale(HsP[(size_t)kk * NN + e], inv) : cmake(0.0, 0.0);
    const size_t items = item_first + item_count;
    const size_t i0 = item_first + (size_t)blockIdx.y * per, i1 = i0 + per < items ? i0 + per : items;
    for (size_t item = i0; item < i1; ++item) {
        int b, t;
        if (tn > 0) { b = (int)(item / tn); t = t0 + (int)(item - (size_t)b * tn); }
        else { b = (int)(item / SP); t = (int)(item - (size_t)b * SP); }
        cplx acc = cmake(0.0, 0.0);
        if (t < d.steps) {
            acc = h[0];
            const double* ub = d.u + (size_t)b * d.k * d.steps + t;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (kk < d.k) { const double c = ub[(size_t)kk * d.steps]; acc.x = fma(c, h[kk + 1].x, acc.x); acc.y = fma(c, h[kk + 1].y,
                    acc.y); }
        }
        Aout[((size_t)b * SP + t) * NN + e] = acc;
    }
}
// ---- squared-generator chain (qoc_gemm_chain_sq.h): B_t and B_t^2 of every (seed, slice), both in the packed anti-Hermitian / Hermitian
// image ---- coefficient row of item (b, t): [1, u_1 .. u_k, u_kk u_ll for kk <= ll (kk-major)] -- P = (k + 1)(k + 2) / 2 doubles, read as
// scalars by the assembly
__global__ void __launch_bounds__(256) k_gemm_sq_coefs(QocDev d, double* __restrict__ coef, int SP, int P) {
    const size_t total = (size_t)d.B * d.steps;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(o / d.steps), t = (int)(o - (size_t)b * d.steps);
        const double* ub = d.u + (size_t)b * d.k * d.steps + t;
        double* c = coef + ((size_t)b * SP + t) * P;
        double u[8];
        for (int kk = 0; kk < d.k; ++kk) u[kk] = ub[(size_t)kk * d.steps];
        c[0] = 1.0;
        int p = 1;
        for (int kk = 0; kk < d.k; ++kk) c[p++] = u[kk];
        for (int kk = 0; kk < d.k; ++kk)
            for (int ll = kk; ll < d.k; ++ll) c[p++] = u[kk] * u[ll];
    }
}
// B_t = h_0 + sum_k u_k h_k and B_t^2 = sum_p c_p q_p for the packed entry e of a thread (its k + 1 + P basis entries in registers over a
// run of items), written as [B | B^2] (2 x 2560 entries per item).  KK = number of controls.  (t0, tn) as in k_gemm_assemble_rows.
template <int KK>
__global__ void __launch_bounds__(256) k_gemm_assemble_sq(QocDev d, const cplx* __restrict__ HsPK, const cplx* __restrict__ HsSQ,
    const double* __restrict__ coef,
                                                           cplx* __restrict__ Aout, int SP, int per, size_t item_count, int t0, int tn) {
    constexpr int P = (KK + 1) * (KK + 2) / 2, GE = QOC_DPP_PK_ELEMS;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    cplx h[KK + 1], q[P];
#pragma unroll
    for (int kk = 0; kk <= KK; ++kk) h[kk] = HsPK[(size_t)kk * GE + e];
#pragma unroll
    for (int p = 0; p < P; ++p) q[p] = HsSQ[(size_t)p * GE + e];
    const size_t i0 = (size_t)blockIdx.y * per, i1 = i0 + per < item_count ? i0 + per : item_count;
    for (size_t item = i0; item < i1; ++item) {
        int b, t;
        if (tn > 0) { b = (int)(item / tn); t = t0 + (int)(item - (size_t)b * tn); }
        else { b = (int)(item / SP); t = (int)(item - (size_t)b * SP); }
        const double* c = coef + ((size_t)b * SP + t) * P;
        cplx accB = h[0], accS = q[0];
#pragma unroll
        for (int kk = 1; kk <= KK; ++kk) { const double u = c[kk]; accB.x = fma(u, h[kk].x, accB.x); accB.y = fma(u, h[kk].y, accB.y); }
#pragma unroll
        for (int p = 1; p < P; ++p) { const double u = c[p]; accS.x = fma(u, q[p].x, accS.x); accS.y = fma(u, q[p].y, accS.y); }
        cplx* out = Aout + ((size_t)b * SP + t) * (2 * GE);
        out[e] = accB;
        out[GE + e] = accS;
    }
}
// S = c0*I + c1*A (+ cT*A2): top block of the Paterson-Stockmeyer recursion
__global__ void __launch_bounds__(256) k_gemm_ps_init(const cplx* __restrict__ A, const cplx* __restrict__ A2, cplx* __restrict__ S,
                                                       size_t count, int N, double c0, double c1, double cT) {
    const size_t NN = (size_t)N * N;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < count; o += (size_t)gridDim.x * blockDim.x) {
        const size_t e = o % NN;
        const int row = (int)(e / N), col = (int)(e - (size_t)row * N);
        const cplx a = A[o];
        cplx v = cmake(c1 * a.x + (row == col ? c0 : 0.0), c1 * a.y);
        if (A2) { const cplx a2 = A2[o]; v.x = fma(cT, a2.x, v.x); v.y = fma(cT, a2.y, v.y); }
        S[o] = v;
    }
}
// Y[b] = [U0 | Psi0] padded (N x (xw+32), xw = N, or 0 in state transfer: no X chain); Psibnd[b][0] = Psi0 padded;
// inter[b][0] = V
__global__ void __launch_bounds__(256) k_gemm_chain_init(QocDev d, cplx* __restrict__ Y, cplx* __restrict__ Psibnd, int N, int NC, int xw) {
    const int ld = xw + QOC_TW;
    const size_t per = (size_t)N * ld;
    const size_t total = (size_t)d.B * per;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bb = o / per, e = o - bb * per;
        const int row = (int)(e / ld), col = (int)(e - (size_t)row * ld);
        cplx v = cmake(0.0, 0.0);
        if (row < d.n) {
            if (col < xw) { if (col < d.n) v = d.U0[row * d.n + col]; }
            else if (col - xw < d.m) v = d.Psi0[row * d.m + (col - xw)];
        }
        Y[o] = v;
        if (col >= xw) Psibnd[(bb * NC) * (size_t)N * QOC_TW + (size_t)row * QOC_TW + (col - xw)] = v;
    }
    const size_t nm = (size_t)d.n * d.m;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)d.B * nm; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bb = o / nm, e = o - bb * nm;
        d.inter[bb * (size_t)(d.steps + 1) * nm + e] = d.V[e];
    }
}
// chunk-start vectors Psibnd[b][c], c = 1 .. NC-1, from the thin blocks (columns N..N+31) of the per-step results: Ys holds one
// [B][N][ld] result per chunk step (slot c = the vectors at the START of chunk c), so the per-step products of N > 64 need no copy
// launch between them (31 launches of ~8 us with their gaps per iteration at n = 128)
__global__ void __launch_bounds__(256) k_gemm_take_bnd_all(QocDev d, const cplx* __restrict__ Ys, cplx* __restrict__ Psibnd, int N, int NC,
    int xw) {
    const int ld = xw + QOC_TW;
    const size_t per = (size_t)N * QOC_TW, slot = (size_t)d.B * N * ld;
    const size_t total = (size_t)d.B * (NC - 1) * per;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = o / per, e = o - bc * per;
        const size_t bb = bc / (NC - 1);
        const int c = 1 + (int)(bc - bb * (NC - 1));
        const int row = (int)(e / QOC_TW), col = (int)(e - (size_t)row * QOC_TW);
        Psibnd[(bb * NC + c) * per + e] = Ys[(size_t)c * slot + bb * (size_t)N * ld + (size_t)row * ld + xw + col];
    }
}
// inter[b][t+1] (API layout) from interP[b][t] (padded thin), t < steps
__global__ void __launch_bounds__(256) k_gemm_unpad_inter(QocDev d, const cplx* __restrict__ interP, int N, int SP) {
    const size_t nm = (size_t)d.n * d.m, per = (size_t)N * QOC_TW;
    const size_t total = (size_t)d.B * d.steps * nm;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = o / nm, e = o - bt * nm;
        const size_t bb = bt / d.steps, t = bt - bb * d.steps;
        const int row = (int)(e / d.m), col = (int)(e - (size_t)row * d.m);
        d.inter[(bb * (size_t)(d.steps + 1) + t + 1) * nm + e] = interP[(bb * SP + t) * per + (size_t)row * QOC_TW + col];
    }
}
// final_state, unitary_scale from the X block of Y                                     tensorflow_state.py:223-225
__global__ void __launch_bounds__(1024) k_gemm_take_final(QocDev d, const cplx* __restrict__ Y, int N) {
    __shared__ double red[32];
    const int b = blockIdx.x, ld = N + QOC_TW, n = d.n;
    const cplx* X = Y + (size_t)b * N * ld;
    double part = 0.0;
    // a wave per row, lanes along it (a thread per row walked the row alone, 64 rows apart from its neighbours: 0.46 ms at n = 512)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wv; c < n; c += nw) {
        double sr = 0.0, si = 0.0;
        for (int a = lane; a < n; a += 64) { const cplx v = X[(size_t)c * ld + a]; sr += v.x; si += v.y; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { sr += __shfl_xor(sr, off, 64); si += __shfl_xor(si, off, 64); }
        if (lane == 0) part += sr * sr + si * si;
    }
    for (int o = threadIdx.x; o < n * n; o += blockDim.x) d.Xfinal[(size_t)b * n * n + o] = X[(size_t)(o / n) * ld + (o % n)];
    const double tot = block_sum(part, red);
    if (threadIdx.x == 0) d.uscale[b] = tot / (double)n;
}
// sources SrcP[b][tau] (padded thin, tau = 0..SP-1; zero for tau = 0 and tau > steps) and the costate at the END of the
// last chunk Ebnd[b][NC-1]: -(2/m^2) z W, plus S_steps when there is no padded slice to add it through the recursion
// `cols` = columns written per row: QOC_TW, or the MV vector slots in the direct route, whose Taylor chains read nothing else of a thin
// panel (C3 x 64: 2.1 GB of zero columns, 0.34 ms per iteration, no longer written)
// `compact` (DPP chain, one vector): SrcP[b][tau][row] contiguous -- a thin panel puts the rows of ONE column 512 bytes apart, every
// 16-byte store its own memory transaction (C3 x 64: 0.11 ms for 4 M entries)
__global__ void __launch_bounds__(256) k_gemm_sources(QocDev d, cplx* __restrict__ SrcP, cplx* __restrict__ Ebnd, int N, int SP, int NC,
    int cols, int compact = 0) {
    const size_t per = (size_t)N * QOC_TW, perw = (size_t)N * cols;
    const bool need_src = d.n_forb > 0 || d.has_speed;
    const size_t total = (size_t)d.B * (need_src ? SP : 1) * perw;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = o / perw, ew = o - bt * perw;
        const int per_seed = need_src ? SP : 1;
        const int b = (int)(bt / per_seed), tau = (int)(bt - (size_t)b * per_seed);
        const int row = (int)(ew / cols), col = (int)(ew - (size_t)row * cols);
        const size_t e = (size_t)row * QOC_TW + col;
        const bool valid = row < d.n && col < d.m;
        if (need_src) {
            cplx s = cmake(0.0, 0.0);
            if (valid && tau >= 1 && tau <= d.steps) s = source_at(d, b, tau, row, col);
            SrcP[compact ? bt * (size_t)N + row : bt * per + e] = s;
        }
        if (tau == 0) {
            cplx v = cmake(0.0, 0.0);
            if (valid) {
                const double c0 = -2.0 / ((double)d.m * (double)d.m);
                v = cscale(cmul(d.zfin[b], d.W[row * d.m + col]), c0);
                if (need_src && SP == d.steps) v = cadd(v, source_at(d, b, d.steps, row, col));
            }
            Ebnd[((size_t)b * NC + (NC - 1)) * per + e] = v;
        }
    }
}
// z-free costate at the end of the pulse, Ebnd[b][NC-1] = -(2/m^2) W: start of a backward chain that does not wait for the overlap
__global__ void __launch_bounds__(256) k_gemm_zfree_end(QocDev d, cplx* __restrict__ Ebnd, int N, int NC) {
    const size_t per = (size_t)N * QOC_TW;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)d.B * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t b = o / per, e = o - b * per;
        const int row = (int)(e / QOC_TW), col = (int)(e - (size_t)row * QOC_TW);
        cplx v = cmake(0.0, 0.0);
        if (row < d.n && col < d.m) v = cscale(d.W[row * d.m + col], -2.0 / ((double)d.m * (double)d.m));
        Ebnd[(b * NC + (NC - 1)) * per + e] = v;
    }
}
// Lambda_t = z Lambda0_t for the time-major wide costates of a seed (LamP[b]: N rows x ldW), z = d.zfin[b]
__global__ void __launch_bounds__(256) k_gemm_scale_lam(QocDev d, cplx* __restrict__ LamP, int N, int ldW, int cols) {
    const size_t per = (size_t)N * cols;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)d.B * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t b = o / per, e = o - b * per;
        const size_t row = e / cols, col = e - row * cols;
        cplx* p = LamP + (b * N + row) * (size_t)ldW + col;
        *p = cmul(d.zfin[b], *p);
    }
}
// LamP[b][(c+1)S-1] = (Ebnd ? Ebnd[b][c] : 0): costate at the end of every chunk
__global__ void __launch_bounds__(256) k_gemm_set_chunk_ends(QocDev d, cplx* __restrict__ LamP, const cplx* __restrict__ Ebnd, int N, int S,
    int NC) {
    const size_t per = (size_t)N * QOC_TW;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)d.B * NC * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = o / per, e = o - bc * per;
        LamP[(bc * S + (S - 1)) * per + e] = Ebnd ? Ebnd[o] : cmake(0.0, 0.0);
    }
}
// dst[b] = src[b] for B matrices of NN elements (odd element of a product-tree level moves up unchanged)
__global__ void __launch_bounds__(256) k_gemm_copy_mats(cplx* __restrict__ dst, long long sD, const cplx* __restrict__ src, long long sS,
    int B, int NN) {
    const size_t total = (size_t)B * NN;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bb = o / NN, e = o - bb * NN;
        dst[bb * sD + e] = src[bb * sS + e];
    }
}
// inter[b][t+1] (API layout) from the time-major wide layout W[b][row][t*MV + col]
__global__ void __launch_bounds__(256) k_gemm_unpad_wide(QocDev d, const cplx* __restrict__ W, int N, int ldW, int MV) {
    const size_t nm = (size_t)d.n * d.m;
    const size_t total = (size_t)d.B * d.steps * nm;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = o / nm, e = o - bt * nm;
        const size_t bb = bt / d.steps, t = bt - bb * d.steps;
        const int row = (int)(e / d.m), col = (int)(e - (size_t)row * d.m);
        d.inter[(bb * (size_t)(d.steps + 1) + t + 1) * nm + e] = W[(bb * N + row) * (size_t)ldW + t * MV + col];
    }
}
// dLdu[b][k][t] = sum over row tiles and vector slots of the per-column dots (wide layout)
__global__ void __launch_bounds__(256) k_gemm_grad_reduce_wide(QocDev d, const double* __restrict__ partial, int tiles_m, int ldW, int MV) {
    const size_t total = (size_t)d.B * d.k * d.steps;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bk = o / d.steps;
        const int t = (int)(o - bk * d.steps);
        const double* p = partial + bk * tiles_m * (size_t)ldW + (size_t)t * MV;
        double s = 0.0;
        for (int i = 0; i < tiles_m; ++i)
            for (int jv = 0; jv < MV; ++jv) s += p[(size_t)i * ldW + jv];
        d.dLdu[o] = s;
    }
}
// dLdu[b][k][t] = sum over row tiles of the partial dots
__global__ void __launch_bounds__(256) k_gemm_grad_reduce(QocDev d, const double* __restrict__ partial, int tiles_m) {
    const size_t total = (size_t)d.B * d.steps * d.k;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = o / d.k;
        const int kk = (int)(o - bt * d.k);
        const int b = (int)(bt / d.steps), t = (int)(bt - (size_t)b * d.steps);
        const double* p = partial + (bt * d.k + kk) * tiles_m;
        double s = 0.0;
        for (int i = 0; i < tiles_m; ++i) s += p[i];
        d.dLdu[((size_t)b * d.k + kk) * d.steps + t] = s;
    }
}

// ---- gradients of large problems (N > 64, m <= 8) as ONE wide product per seed ----------------------------------------------------------
// The per-slice thin tiles [t][N][32] of Psi_t / Lambda_t carry m <= 8 useful columns of 32: k batched launches of 2000 padded thin
// products with a dot epilogue ran at ~21 TFLOP/s of mostly padding (C5: 12.7 ms of 215).  Re-packed time-major -- wide[row][t * 8 + col],
// the layout the persistent chains of N <= 64 write directly -- the products of ALL controls are one batched N x N x (8 steps) GEMM on
// k_zgemm_wg, and dL/du_{k,t} = Re sum conj(Lambda_t) (H_k' Psi_t) (tensorflow_state.py:61-63) is a column-block dot of its result.
#define QOC_WIDE_MV 8
__global__ void __launch_bounds__(256) k_gemm_to_wide(QocDev d, const cplx* __restrict__ thinP, const cplx* __restrict__ thinL,
                                                      cplx* __restrict__ wideP, cplx* __restrict__ wideL, int N, int W, int count) {
    const size_t total = (size_t)count * N * QOC_WIDE_MV;                                     // `count` slices from thinP / thinL on
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(o % QOC_WIDE_MV);
        const size_t tr = o / QOC_WIDE_MV;
        const int row = (int)(tr % N), t = (int)(tr / N);
        const size_t src = ((size_t)t * N + row) * QOC_TW + col, dst = (size_t)row * W + (size_t)t * QOC_WIDE_MV + col;
        wideP[dst] = thinP[src];
        wideL[dst] = thinL[src];
    }
}
// one wave per (control, slice): rows lane, lane + 64, ...; the 8 columns of a slice are one 128-byte line of a row
// (column block ti of the wide buffers is slice t_first + ti: the whole pulse, or the slices of one rank of a time-sharded engine)
__global__ void __launch_bounds__(256) k_gemm_dot_wide(QocDev d, int b, const cplx* __restrict__ wideC, const cplx* __restrict__ wideL,
    int N, int W,
                                                       int t_first, int count) {
    const int lane = threadIdx.x & 63;
    const size_t item = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (item >= (size_t)d.k * count) return;
    const int kk = (int)(item / count), t = (int)(item - (size_t)kk * count);
    const cplx* C = wideC + (size_t)kk * N * W + (size_t)t * QOC_WIDE_MV;
    const cplx* L = wideL + (size_t)t * QOC_WIDE_MV;
    double acc = 0.0;
    for (int row = lane; row < N; row += 64) {
#pragma unroll
        for (int col = 0; col < QOC_WIDE_MV; ++col) {
            const cplx c = C[(size_t)row * W + col], l = L[(size_t)row * W + col];
            acc = fma(l.x, c.x, acc); acc = fma(l.y, c.y, acc);                          // Re conj(l) c
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) d.dLdu[((size_t)b * d.k + kk) * d.steps + t_first + t] = acc;
}

// K[b][t] = I for the padded slices t = steps .. SP - 1 (set once: the launch-per-product route of ONE control set never computes them)
__global__ void __launch_bounds__(256) k_gemm_pad_identity(cplx* __restrict__ K, int B, int N, int steps, int SP) {
    const size_t NN = (size_t)N * N, per = (size_t)(SP - steps) * NN;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)B * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t b = o / per, r = o - b * per, t = steps + r / NN, e = r % NN;
        K[(b * SP + t) * NN + e] = cmake(e / N == e % N ? 1.0 : 0.0, 0.0);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Time is cut into NC chunks of S = 2^L slices (padded with identity slices to SP = NC*S).  A pairwise product tree over
// the K_t gives the chunk products at the batched-GEMM rate; the sequential part of each chain shrinks from `steps`
// launches to NC (chunk boundaries) + S (all chunks swept in parallel).
struct QocGemm {
    int N = 0, S = 1, L = 0, NC = 1, SP = 1;
    int MV = 0, ldW = 0;      // persistent mode: vector slots (1/2/4/8) and row stride of the time-major wide buffers
    // planned / local batch (QocDev::Bplan / B): split-K factors and kernel families are chosen for the planned batch
    double plan_scale = 1.0;
    bool direct = false;      // state transfer as Taylor mat-vec chains on the assembled generators (one chunk, no propagators)
    bool persistent = false;  // N <= 64, m <= 8: thin chains run as persistent VALU kernels instead of one launch per step
    bool reduce_in_tail = false;  // ... and the engine's split tail (k_finish_split_a) sums the per-tile gradient partials itself: no k_gemm_grad_reduce_wide launch
    cplx* HsP = nullptr;      // [k+1][N][N]
    cplx* HsPT = nullptr;     // dpp_chain: the same stack transposed -- k_gemm_assemble_rows then writes the generators column-major
    // dpp_chain with a state regulariser (forward chain alone in its launch): the generators of the slices from asm_split on are assembled
    // on a second stream BESIDE the forward chain over the first asm_split slices (64 of 256 CUs, 1.4 TB/s), which then continues from its
    // state
    hipStream_t aux = nullptr, chain_s = nullptr; hipEvent_t ev_ready = nullptr, ev_fwd = nullptr, ev_p1 = nullptr;
        int asm_split = 0, asm_tail_wgs = 512;
    // (round 5) the pulse is cut into asm_win.size() - 1 windows [asm_win[w], asm_win[w + 1]): window 0 is assembled in front of the chain,
    // window w >= 1 on the second stream while the chain walks window w - 1 (one chain launch per window, each continuing from the state
    // the previous one left in Aoff)
    std::vector<int> asm_win; std::vector<hipEvent_t> ev_win;
    // persistent state transfer: Psibnd[b][0] = Psi0 and inter[b][0] = V never change -- k_gemm_chain_init ran at set-up, not per iteration
    bool init_once = false;
    bool dpp_chain = false;   // direct route at N = 64, one state vector: k_gemm_taylor_chain_dpp (qoc_gemm_chain_dpp.h)
    bool antiherm = false;    // every generator anti-Hermitian (set by the engine before qoc_gemm_setup)
    // dpp_chain on anti-Hermitian generators: only the blocks on and below the block diagonal are assembled, stored and read
    bool dpp_packed = false;
    // dpp_packed, few enough control sets for the chains to be latency-bound: [B | B^2] per slice and k_gemm_taylor_chain_sq
    // (qoc_gemm_chain_sq.h)
    bool sq_chain = false;
    // qoc_config.variant of an explicit GEMM-path request: 1 = never the squared-generator chain, 2 = always where it applies
    int direct_variant = 0;
    cplx* HsSQ = nullptr;     // sq_chain: the (k + 1)(k + 2) / 2 packed basis matrices of B^2
    double* sqc = nullptr;    // sq_chain: [B][SP][P] coefficient rows (k_gemm_sq_coefs)
    // what qoc_taylor_chain_launch takes
    // dpp_chain on a padded problem (n <= 56 levels in N = 64) that is latency-bound (<= 128 control sets) or cannot be packed: columns per
    // wave 10 / 12 / 14 instead of 16 -- only the first 4 dpp_cw columns of the full image are assembled, stored, read and multiplied.  16:
    // off
    int dpp_cw = 16;
    int dpp_mode() const { return dpp_chain ? (sq_chain ? 3 : (dpp_packed ? 2 : (dpp_cw < 16 ? dpp_cw : 1))) : 0; }
    // entries of one slice
    size_t gen_elems() const {
        if (sq_chain) return (size_t)2 * QOC_DPP_PK_ELEMS;
        if (dpp_packed) return (size_t)QOC_DPP_PK_ELEMS;
        return dpp_chain && dpp_cw < 16 ? (size_t)256 * dpp_cw : (size_t)N * N;
    }
    cplx *A = nullptr, *P = nullptr, *K = nullptr, *A2 = nullptr;     // [B*SP][N][N]
    cplx* tree = nullptr;     // levels 1..L of the product tree: level l at tree_off[l], [B][SP >> l][N][N]
    size_t tree_off[8];
    cplx *Y0 = nullptr, *Y1 = nullptr;                               // [B][N][N+32]
    cplx* interP = nullptr;   // [B][SP][N][32]   Psi_t
    cplx* LamP = nullptr;     // [B][SP][N][32]   Lambda_t
    cplx* SrcP = nullptr;     // [B][SP][N][32]   S_tau (state regularisers only)
    cplx* KT = nullptr;       // persistent mode: K_t^T  [B*SP][N][N] (rows of K^H for the backward chains)
    cplx* PcT = nullptr;      // persistent mode: P_c^T  [B][NC][N][N] (== KT when S = 1)
    cplx* root = nullptr;     // persistent unitary mode: product tree above the chunk products, down to one matrix per seed
    ScanArgs scan;            // its levels (filled by qoc_gemm_forward each iteration; pointers are stable)
    cplx* zthin = nullptr;    // [N][32] zeros
    cplx *Psibnd = nullptr, *Ebnd = nullptr, *Aoff = nullptr;        // [B][NC][N][32] chunk-start Psi, chunk-end Lambda, affine offsets
    double* partial = nullptr; // [B*steps][k][N/32]
    // > 0: gradients of an N > 64 problem through ONE wide product per seed (k_gemm_to_wide, k_zgemm_wg, k_gemm_dot_wide)
    int wideW = 0;
    cplx *wideP = nullptr, *wideL = nullptr, *wideC = nullptr;   // [N][wideW], [N][wideW], [k][N][wideW]
    // time-axis sharding of one trajectory (qoc_gemm_ts.h): G ranks own runs of chunks; ts_rank < 0 emulates all of them in this engine
    int ts_G = 0, ts_rank = -1;
    std::vector<int> ts_cb;                                       // chunk boundaries: rank r owns [ts_cb[r], ts_cb[r + 1])
    cplx *ts_Rall = nullptr, *ts_Rtmp = nullptr;                  // [G][N][N] rank products (all-gathered in place), [2][N][N]
    // [G + 1][N][N + 32]: [X | Psi] at the rank boundaries; [G + 1][N][32]: costates there
    cplx *ts_Yr = nullptr, *ts_Er = nullptr;
    struct qoc_comm* ts_comm = nullptr;
};

// Unitary mode: any n.  State transfer: psi <- P(B_t) psi is the same chain with K_t = sum_{j<T} B_t^j/j! (no squaring);
// the reference's backward step lambda <- P(-B_t) lambda (tensorflow_state.py:118-131) equals K_t^dagger lambda exactly
// when every generator is anti-Hermitian (-i dt H with H Hermitian), which `antiherm` certifies at create time.
// Any state-transfer problem with n <= 64, m <= 8 can instead run "direct" (k_gemm_taylor_chain: the reference's own
// mat-vec recursion, forward and backward, on pre-assembled generators; no time parallelism, so it is the large-batch mode).
static inline bool qoc_gemm_direct_supported(const QocDev& d) { return d.state_transfer && d.n <= 64 && d.m <= 8 && d.T >= 1; }
// the polynomial coefficient tables (ExpmCoef, invf[]) hold 1/j! for j < QOC_GEMM_MAXT (the MFMA path stops at T = 22: this path takes
// over)
static inline bool qoc_gemm_supported(const QocDev& d, bool antiherm) {
    return d.m <= QOC_TW && d.T >= 1 && d.T <= QOC_GEMM_MAXT - 1 && (!d.state_transfer || antiherm || qoc_gemm_direct_supported(d));
}
static inline bool qoc_all_antihermitian(const cplx* Hs, int n, int count) {
    for (int q = 0; q < count; ++q) {
        const cplx* H = Hs + (size_t)q * n * n;
        for (int a = 0; a < n; ++a)
            for (int c = a; c < n; ++c)
                if (H[a * n + c].x != -H[c * n + a].x || H[a * n + c].y != H[c * n + a].y) return false;
    }
    return true;
}

static inline int qoc_gemm_setup(QocGemm& gm, const QocDev& d, const cplx* Hs_host, bool direct, std::vector<void*>& allocs,
    std::string& msg) {
    const int N = ((d.n + 31) / 32) * 32;
    gm.N = N;
    gm.plan_scale = (double)d.Bplan / (double)d.B;
    gm.persistent = N <= 64 && d.m <= 8;
    gm.MV = d.m <= 1 ? 1 : (d.m <= 2 ? 2 : (d.m <= 4 ? 4 : 8));
    gm.direct = direct && d.state_transfer && gm.persistent;
    {
        const char* e = qoc_exp_env("QOC_CHAIN_DPP");            // experimental switch: 0 = the butterfly kernel k_gemm_taylor_chain
        gm.dpp_chain = gm.direct && N == 64 && gm.MV == 1 && !(e && e[0] == '0');
        gm.dpp_packed = gm.dpp_chain && gm.antiherm;
        // padded problems: the FMAs of a mat-vec shrink with the columns a wave owns (48 -> 30 / 36 / 42 DPP FMAs), the bytes of a slice to
        // 256 cw entries (cw = 10: the packed size); where 256 chains are bound by the generator bytes (> 128 control sets) the packed
        // image stays ahead for cw > 10
        gm.dpp_cw = 16;
        if (gm.dpp_chain && d.n <= 56 && d.k <= 8 && gm.direct_variant != 2 && !qoc_exp_is("QOC_DPP_ACTIVE_COLUMNS", 0)) {
            const int cw = d.n <= 40 ? 10 : (d.n <= 48 ? 12 : 14);
            if (!gm.dpp_packed || d.Bplan <= 128 || cw == 10) { gm.dpp_cw = cw; gm.dpp_packed = false; }
        }
        // opt-in only (qoc_config.variant = 2 with path = GEMM): measured SLOWER than the plain chain at C3 x 64 (7.98 against 5.83 ms per
        // iteration) -- see the header of qoc_gemm_chain_sq.h and profiles/EXPERIMENTS.md
        gm.sq_chain = gm.dpp_packed && qoc_sq_chain_terms_ok(d.T) && d.k >= 1 && d.k <= 8 && gm.direct_variant == 2;
    }
    int L = 0;
    while (L < 6 && (1 << (2 * (L + 1))) <= d.steps) ++L;        // S = 2^L ~ sqrt(steps), at most 64
    // unitary chains get their chunk boundaries in log depth (k_gemm_scan_nodes), so a latency-bound launch (few (seed, chunk)
    // workgroups) prefers chunks half as long: C2 single trajectory 0.214 (S = 16) -> 0.198 ms (S = 8); 0.195 at S = 4
    if (gm.persistent && !d.state_transfer && !direct && L > 1 && (size_t)d.Bplan * ((d.steps + (1 << L) - 1) >> L) <= 64) --L;
    gm.L = L; gm.S = 1 << L;
    gm.NC = (d.steps + gm.S - 1) / gm.S;
    gm.SP = gm.NC * gm.S;
    if (gm.direct) { gm.L = L = 0; gm.S = d.steps; gm.NC = 1; gm.SP = d.steps; }   // one chunk, no padding, no tree
    gm.ldW = ((gm.SP * gm.MV + 31) / 32) * 32;
    const size_t NN = (size_t)N * N, BSP = (size_t)d.B * gm.SP, thin = (size_t)N * QOC_TW;
    std::vector<cplx> hp((size_t)(d.k + 1) * NN);
    for (auto& v : hp) { v.x = 0; v.y = 0; }
    for (int kk = 0; kk <= d.k; ++kk)
        for (int a = 0; a < d.n; ++a)
            for (int c = 0; c < d.n; ++c) hp[(size_t)kk * NN + (size_t)a * N + c] = Hs_host[(size_t)kk * d.n * d.n + (size_t)a * d.n + c];
    // every work buffer is carved out of ONE allocation: with one hipMalloc per buffer the placement after earlier engines of the
    // same process were freed decided the speed (n = 128 x 4: 8.6 or 17-20 ms per iteration for the same problem)
    std::vector<std::pair<void**, size_t>> wanted;
    auto al = [&](void** dst, size_t bytes) -> bool { wanted.emplace_back(dst,
        ((bytes ? bytes : 16) + 4095) & ~(size_t)4095); return true; };
    const bool need_src = d.n_forb > 0 || d.has_speed;
    size_t tree_elems = 0;
    for (int l = 1; l <= L; ++l) { gm.tree_off[l] = tree_elems; tree_elems += (size_t)d.B * (gm.SP >> l) * NN; }
    const bool fused = N <= 64 && !gm.direct;                    // k_gemm_expm_fused needs no A / A2 / ping-pong buffers
    size_t root_elems = 0;
    for (int cnt = gm.NC; cnt > 1; cnt = (cnt + 1) / 2) root_elems += (size_t)d.B * ((cnt + 1) / 2) * NN;
    const bool poly = !fused && !gm.direct;                      // launch-per-product route: A2 and ping-pong buffers
    bool ok = al((void**)&gm.HsP, hp.size() * sizeof(cplx)) && al((void**)&gm.HsPT, gm.dpp_chain ? hp.size() * sizeof(cplx) : 16) && (fused
        || al((void**)&gm.A, BSP * (gm.direct ? gm.gen_elems() : NN) * sizeof(cplx))) &&
              (!poly || al((void**)&gm.P, BSP * NN * sizeof(cplx))) && (!poly || al((void**)&gm.A2, BSP * NN * sizeof(cplx))) &&
              al((void**)&gm.root, (gm.persistent && !d.state_transfer) ? root_elems * sizeof(cplx) : 16) &&
              al((void**)&gm.K, gm.direct ? 16 : BSP * NN * sizeof(cplx)) && al((void**)&gm.tree, tree_elems * sizeof(cplx)) &&
              al((void**)&gm.KT, (gm.persistent && !gm.direct) ? BSP * NN * sizeof(cplx) : 16) &&
              al((void**)&gm.PcT, (gm.persistent && !gm.direct && L > 0) ? (size_t)d.B * gm.NC * NN * sizeof(cplx) : 16) &&
              // (per-step boundary products, N > 64 or m > 8: one result slot per chunk step, read back by ONE k_gemm_take_bnd_all)
              al((void**)&gm.Y0, (size_t)(gm.persistent ? 1 : gm.NC + 1) * d.B * N * (N + QOC_TW) * sizeof(cplx)) &&
              al((void**)&gm.Y1, gm.persistent ? (size_t)d.B * N * (N + QOC_TW) * sizeof(cplx) : 16) &&
              al((void**)&gm.interP, BSP * thin * sizeof(cplx)) && al((void**)&gm.LamP, BSP * thin * sizeof(cplx)) &&
              al((void**)&gm.Psibnd, (size_t)d.B * gm.NC * thin * sizeof(cplx)) &&
              al((void**)&gm.Ebnd, (size_t)d.B * gm.NC * thin * sizeof(cplx)) &&
              al((void**)&gm.Aoff, (size_t)d.B * gm.NC * thin * sizeof(cplx)) &&
              al((void**)&gm.zthin, thin * sizeof(cplx)) &&
              al((void**)&gm.partial, (size_t)d.B * d.k * (N / 32) * (gm.persistent ? (size_t)gm.ldW : (size_t)d.steps) * sizeof(double));
    if (ok && need_src) ok = al((void**)&gm.SrcP, BSP * thin * sizeof(cplx));
    const int sqP = (d.k + 1) * (d.k + 2) / 2;
    if (ok && gm.sq_chain) ok = al((void**)&gm.HsSQ, (size_t)sqP * QOC_DPP_PK_ELEMS * sizeof(cplx)) && al((void**)&gm.sqc,
        BSP * sqP * sizeof(double));
    // wide gradient products: large matrices with few vectors (row tiles in pairs and column tiles in fours: what k_zgemm_wg takes)
    gm.wideW = (!gm.persistent && N >= 128 && (N / 32) % 2 == 0
        && d.m <= QOC_WIDE_MV) ? (int)((((size_t)d.steps * QOC_WIDE_MV + 127) / 128) * 128) : 0;
    if (ok && gm.wideW > 0)
        ok = al((void**)&gm.wideP, (size_t)N * gm.wideW * sizeof(cplx)) && al((void**)&gm.wideL, (size_t)N * gm.wideW * sizeof(cplx)) &&
             al((void**)&gm.wideC, (size_t)d.k * N * gm.wideW * sizeof(cplx));
    if (ok && gm.ts_G > 0)
        ok = al((void**)&gm.ts_Rall, (size_t)gm.ts_G * NN * sizeof(cplx)) && al((void**)&gm.ts_Rtmp, 2 * NN * sizeof(cplx)) &&
             al((void**)&gm.ts_Yr, (size_t)(gm.ts_G + 1) * N * (N + QOC_TW) * sizeof(cplx)) && al((void**)&gm.ts_Er,
                 (size_t)(gm.ts_G + 1) * thin * sizeof(cplx));
    {
        size_t total = 0;
        for (auto& w : wanted) total += w.second;
        char* arena = nullptr;
        ok = ok && hipMalloc((void**)&arena, qoc_arena_bytes(total)) == hipSuccess;
        if (ok) {
            allocs.push_back(arena);
            size_t off = 0;
            for (auto& w : wanted) { *w.first = arena + off; off += w.second; }
        }
    }
    if (!ok) { msg = "GEMM path: out of device memory"; return -3; }
    if (hipMemcpy(gm.HsP, hp.data(), hp.size() * sizeof(cplx),
        hipMemcpyHostToDevice) != hipSuccess) { msg = "GEMM path: upload failed"; return -2; }
    if (gm.dpp_chain) {
        std::vector<cplx> ht(hp.size());
        // entries per matrix of the stack: packed, the first 4 cw columns of the column-major image, or all of it
        const size_t ge = gm.dpp_packed ? (size_t)QOC_DPP_PK_ELEMS : (gm.dpp_cw < 16 ? (size_t)256 * gm.dpp_cw : NN);
        for (int kk = 0; kk <= d.k; ++kk)
            for (int a = 0; a < N; ++a)
                for (int c = 0; c < N; ++c) {
                    if (!gm.dpp_packed) {
                        if (c < 4 * gm.dpp_cw) ht[(size_t)kk * ge + (size_t)c * N + a] = hp[(size_t)kk * NN + (size_t)a * N + c];
                        continue;
                    }
                    // packed: blocks on and below the block diagonal, column-major inside a block
                    const int R = a >> 4, C = c >> 4;
                    if (R >= C) ht[(size_t)kk * ge + (size_t)(R * (R + 1) / 2 + C) * 256 + (size_t)(c & 15) * 16 + (a & 15)]
                        = hp[(size_t)kk * NN + (size_t)a * N + c];
                }
        if (hipMemcpy(gm.HsPT, ht.data(), (size_t)(d.k + 1) * ge * sizeof(cplx),
            hipMemcpyHostToDevice) != hipSuccess) { msg = "GEMM path: upload failed"; return -2; }
    }
    if (gm.sq_chain) {
        // M_0 = A_0^2, M_k = A_0 A_k + A_k A_0, M_kl = A_k A_l + A_l A_k (k < l), M_kk = A_k^2 -- Hermitian, packed like the generators
        std::vector<cplx> hq((size_t)sqP * QOC_DPP_PK_ELEMS);
        std::vector<cplx> prod(NN);
        auto accumulate = [&](int x, int y, bool clear) {                        // prod (+)= A_x A_y
            const cplx* X = &hp[(size_t)x * NN]; const cplx* Y = &hp[(size_t)y * NN];
            for (int a = 0; a < N; ++a)
                for (int c = 0; c < N; ++c) {
                    double re = 0.0, im = 0.0;
                    for (int j = 0; j < N; ++j) { const cplx u = X[(size_t)a * N + j],
                        v = Y[(size_t)j * N + c]; re += u.x * v.x - u.y * v.y; im += u.x * v.y + u.y * v.x; }
                    cplx& o = prod[(size_t)a * N + c];
                    if (clear) { o.x = re; o.y = im; } else { o.x += re; o.y += im; }
                }
        };
        auto pack = [&](int p) {
            for (int a = 0; a < N; ++a)
                for (int c = 0; c < N; ++c) {
                    const int R = a >> 4, C = c >> 4;
                    if (R >= C) hq[(size_t)p * QOC_DPP_PK_ELEMS + (size_t)(R * (R + 1) / 2 + C) * 256 + (size_t)(c & 15) * 16 + (a & 15)]
                        = prod[(size_t)a * N + c];
                }
        };
        int p = 0;
        accumulate(0, 0, true); pack(p++);
        for (int kk = 1; kk <= d.k; ++kk) { accumulate(0, kk, true); accumulate(kk, 0, false); pack(p++); }
        for (int kk = 1; kk <= d.k; ++kk)
            for (int ll = kk; ll <= d.k; ++ll) { accumulate(kk, ll, true); if (ll != kk) accumulate(ll, kk, false); pack(p++); }
        if (hipMemcpy(gm.HsSQ, hq.data(), hq.size() * sizeof(cplx),
            hipMemcpyHostToDevice) != hipSuccess) { msg = "GEMM path: upload failed"; return -2; }
    }
    // the persistent chain kernels write only the first m (<= 8) of the 32 thin columns; the rest must read as zero
    bool zeroed = hipMemset(gm.zthin, 0, thin * sizeof(cplx)) == hipSuccess &&
                  hipMemset(gm.interP, 0, BSP * thin * sizeof(cplx)) == hipSuccess &&
                  hipMemset(gm.LamP, 0, BSP * thin * sizeof(cplx)) == hipSuccess &&
                  hipMemset(gm.Psibnd, 0, (size_t)d.B * gm.NC * thin * sizeof(cplx)) == hipSuccess &&
                  hipMemset(gm.Ebnd, 0, (size_t)d.B * gm.NC * thin * sizeof(cplx)) == hipSuccess &&
                  hipMemset(gm.Aoff, 0, (size_t)d.B * gm.NC * thin * sizeof(cplx)) == hipSuccess;
    if (gm.wideW > 0) zeroed = zeroed && hipMemset(gm.wideP, 0, (size_t)N * gm.wideW * sizeof(cplx)) == hipSuccess &&
                                         // (the columns beyond 8 steps)
                                         hipMemset(gm.wideL, 0, (size_t)N * gm.wideW * sizeof(cplx)) == hipSuccess;
    if (!zeroed) { msg = "GEMM path: clearing the work buffers failed"; return -2; }
    if (gm.persistent && d.state_transfer && gm.ts_G <= 0) {   // the constant starts of the chains, once (one launch less per iteration)
        const size_t work = ((size_t)d.B * N * QOC_TW + 255) / 256;
        hipLaunchKernelGGL(k_gemm_chain_init, dim3((unsigned)(work > 65535 ? 65535 : work)), dim3(256), 0, 0, d, gm.Y0, gm.Psibnd, N, gm.NC,
            0);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(0) != hipSuccess) {
            msg = "GEMM path: the chain starts could not be set";
            return -2;
        }
        gm.init_once = true;
    }
    if (poly && gm.SP > d.steps) {
        hipLaunchKernelGGL(k_gemm_pad_identity, dim3(4096), dim3(256), 0, 0, gm.K, d.B, N, d.steps, gm.SP);
        if (hipGetLastError() != hipSuccess
            || hipStreamSynchronize(0) != hipSuccess) { msg = "GEMM path: the padded propagators could not be set"; return -2; }
    }
    {
        // generators of the last 11/16 of the pulse assembled beside the forward chain's first part, by 512 long-running workgroups: the
        // chain's prefetch shares the memory system with them (a slice costs it 4-5.6 us beside an unthrottled assembly against 2.9 alone);
        // sweep of
        // (workgroups, split) at C3 x 64, ms per iteration: (8192, 3/16) 6.81, (2048, 3/16) 6.79, (512, 5/16) 6.67, (512, 8/16) 6.77,
        // (384, 6/16) 6.68, (256, 5/16) 7.35; one launch in front of the chain 7.03
        const char* e = qoc_exp_env("QOC_ASM_OVERLAP");             // experimental switch: 0 = one assembly launch in front of the chain
        // (256 chains fill the chip: 14.6 against 14.0 ms)
        if (gm.dpp_chain && need_src && d.k <= 8 && d.steps >= 64 && d.B <= 128 && !(e && e[0] == '0')) {
            // disjoint CU sets for the two kernels that run beside each other: the assembly's workgroups otherwise land on the chains' CUs
            // as well and take issue slots from waves whose every instruction is on the critical path.  QOC_ASM_CUMASK=0: plain second
            // stream, chain on the engine's
            {
                const char* cm = qoc_exp_env("QOC_ASM_CUMASK");
                int ncu = 0, dv = 0;
                if (hipGetDevice(&dv) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount,
                    dv) != hipSuccess) ncu = 0;
                const int chain_cus = cm && atoi(cm) > 0 ? atoi(cm) : 112;
                if (!(cm && cm[0] == '0') && ncu >= 128 && ncu <= 1024 && d.B + 16 <= chain_cus) {
                    std::vector<uint32_t> mc((ncu + 31) / 32, 0u), ma((ncu + 31) / 32, 0u);
                    for (int c = 0; c < ncu; ++c) (c < chain_cus ? mc : ma)[c / 32] |= 1u << (c % 32);
                    if (hipExtStreamCreateWithCUMask(&gm.chain_s, (uint32_t)mc.size(),
                        mc.data()) != hipSuccess) { gm.chain_s = nullptr; (void)hipGetLastError(); }
                    else if (hipExtStreamCreateWithCUMask(&gm.aux, (uint32_t)ma.size(),
                        ma.data()) != hipSuccess) { hipStreamDestroy(gm.chain_s); gm.chain_s = nullptr; gm.aux
                        = nullptr; (void)hipGetLastError(); }
                    if (gm.chain_s && (hipEventCreateWithFlags(&gm.ev_fwd, hipEventDisableTiming) != hipSuccess
                        || hipEventCreateWithFlags(&gm.ev_p1,
                        hipEventDisableTiming) != hipSuccess)) { msg = "GEMM path: events could not be created"; return -2; }
                }
            }
            if ((!gm.aux && hipStreamCreateWithFlags(&gm.aux, hipStreamNonBlocking) != hipSuccess) ||
                hipEventCreateWithFlags(&gm.ev_ready, hipEventDisableTiming) != hipSuccess) {
                msg = "GEMM path: second stream / events could not be created";
                return -2;
            }
            // on shared CUs: 512 long-running workgroups from 5/16 of the pulse on; on its own CUs the assembly runs unthrottled from 4/16
            // on
            // (C3 x 64, ms per iteration: masks of 80 / 96 / 112 / 128 CUs for the chains 6.31 / 6.31 / 6.25 / 6.34; shared CUs 6.40; 2048
            // workgroups 6.27 - 6.31; 3/16: 6.35 - 6.50)
            gm.asm_split = ((gm.chain_s ? 4 : 5) * d.steps) / 16;
            gm.asm_tail_wgs = gm.chain_s ? 8192 : 512;
            if (const char* t = qoc_exp_env("QOC_ASM_TAIL_WGS")) gm.asm_tail_wgs = atoi(t) > 0 ? atoi(t) : gm.asm_tail_wgs;
            if (const char* t = qoc_exp_env("QOC_ASM_SPLIT16")) gm.asm_split = (atoi(t) * d.steps) / 16;
            if (gm.asm_split < 1) gm.asm_split = 1;
            if (gm.asm_split > d.steps - 1) gm.asm_split = d.steps - 1;
            // windows: [0, asm_split) in front, the rest in nw - 1 equal windows beside the chain.  More than two windows buy nothing (C3 x
            // 64: 5.85 / 5.87 ms at nw = 2 / 4 with 4/16 in front, 5.83 with 2/16 and nw = 4: the chain part that runs beside an assembly
            // launch loses what the shorter head saves) and nine or more chain launches waiting on events of the second stream did not
            // finish at all on ROCm 7.2: profiles/r05_c3_windows.txt
            int nw = 2;
            if (const char* t = qoc_exp_env("QOC_ASM_WINDOWS")) nw = atoi(t) >= 2 ? (atoi(t) <= 4 ? atoi(t) : 4) : 2;
            if (nw - 1 > d.steps - gm.asm_split) nw = 1 + (d.steps - gm.asm_split);
            gm.asm_win.assign(1, 0);
            for (int w = 1; w <= nw; ++w) gm.asm_win.push_back(w == nw ? d.steps : gm.asm_split
                + (int)(((long long)(d.steps - gm.asm_split) * (w - 1)) / (nw - 1)));
            gm.ev_win.assign(nw, nullptr);
            for (int w = 1; w < nw; ++w)
                if (hipEventCreateWithFlags(&gm.ev_win[w],
                    hipEventDisableTiming) != hipSuccess) { msg = "GEMM path: events could not be created"; return -2; }
        }
    }
    return 0;
}
static inline void qoc_gemm_teardown(QocGemm& gm) {
    if (gm.aux) { hipStreamSynchronize(gm.aux); hipStreamDestroy(gm.aux); gm.aux = nullptr; }
    if (gm.chain_s) { hipStreamSynchronize(gm.chain_s); hipStreamDestroy(gm.chain_s); gm.chain_s = nullptr; }
    if (gm.ev_fwd) { hipEventDestroy(gm.ev_fwd); gm.ev_fwd = nullptr; }
    if (gm.ev_p1) { hipEventDestroy(gm.ev_p1); gm.ev_p1 = nullptr; }
    if (gm.ev_ready) { hipEventDestroy(gm.ev_ready); gm.ev_ready = nullptr; }
    for (auto& ev : gm.ev_win) if (ev) { hipEventDestroy(ev); ev = nullptr; }
}

template <bool CONJT, int EPI, int SK>
static inline void qoc_gemm_launch_sk(const GemmArgs& g, unsigned blocks, hipStream_t s) {
    const size_t lds = SK > 1 ? (size_t)(SK - 1) * 2048 * sizeof(double) : 0;
    hipLaunchKernelGGL((k_zgemm32<CONJT, EPI, SK>), dim3(blocks), dim3(64 * SK), lds, s, g);
}
// Kernels that use more than 64 KB of dynamic LDS must opt in, per device: called from qoc_gemm_setup (one engine = one device)
template <bool CONJT, int EPI>
static inline bool qoc_gemm_lds_opt_in_sk() {
    return hipFuncSetAttribute((const void*)k_zgemm32<CONJT, EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
        7 * 2048 * (int)sizeof(double)) == hipSuccess;
}
static inline bool qoc_gemm_lds_opt_in() {
    return qoc_gemm_lds_opt_in_sk<false, 0>() && qoc_gemm_lds_opt_in_sk<false, 1>() && qoc_gemm_lds_opt_in_sk<false, 2>()
        && qoc_gemm_lds_opt_in_sk<true, 0>() &&
           hipFuncSetAttribute((const void*)k_gemm_expm_fused<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
               2 * 64 * (64 + QOC_EXPM_LDPAD) * (int)sizeof(cplx)) == hipSuccess &&
           hipFuncSetAttribute((const void*)k_gemm_scan_nodes<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
               (int)qoc_scan_lds(64)) == hipSuccess &&
           qoc_zgemm_wg_opt_in();
}
// picks the split-K factor from the launch size: fill ~2 waves per SIMD (2048 waves) when the batch is small.  The split factor and the
// kernel family change the association of the sums, so they follow the PLANNED batch: QocGemm::plan_scale = planned / local batch
// (qoc_gemm_setup)
#ifndef QOC_SK_TARGET
#define QOC_SK_TARGET 2048     // waves a split-K launch aims at (~2 per SIMD)
#endif
// will this plain product run on k_zgemm_wg (the condition of the last-but-one branch below)?
static inline bool qoc_gemm_takes_wg(const QocGemm& gm, const GemmArgs& g) {
    const size_t real_tiles = (size_t)g.batch * g.tiles_m * g.tiles_n;
    const size_t tiles = (size_t)((double)real_tiles * gm.plan_scale + 0.5);
    const bool split = tiles * 2 <= QOC_SK_TARGET && (g.Kdim / 2) % 8 == 0;
    return !split && (g.tiles_m & 1) == 0 && (g.tiles_n & 3) == 0 && (g.Kdim % ZW_KC) == 0 && g.Kdim >= 128 && tiles >= 8 * 1024;
}
// sk_tiles: tile count the split-K factor is chosen for when the launch is one PART of a product (the parts must sum in the order of the
// whole)
static inline void qoc_gemm_launch(const QocGemm& gm, bool conjt, int epi, const GemmArgs& g, hipStream_t s, size_t sk_tiles = 0) {
    const size_t real_tiles = (size_t)g.batch * g.tiles_m * g.tiles_n;
    const unsigned blocks = (unsigned)real_tiles;
    const size_t tiles = (size_t)((double)(sk_tiles ? sk_tiles : real_tiles) * gm.plan_scale + 0.5);
    int sk = 1;
    if (tiles * 2 <= QOC_SK_TARGET && (g.Kdim / 2) % 8 == 0) sk = 2;
    if (tiles * 4 <= QOC_SK_TARGET && (g.Kdim / 4) % 8 == 0) sk = 4;
    if (tiles * 8 <= QOC_SK_TARGET && (g.Kdim / 8) % 8 == 0) sk = 8;
    if (epi == 2) {
        if (sk == 8) qoc_gemm_launch_sk<false, 2, 8>(g, blocks, s);
        else if (sk == 4) qoc_gemm_launch_sk<false, 2, 4>(g, blocks, s);
        else if (sk == 2) qoc_gemm_launch_sk<false, 2, 2>(g, blocks, s);
        else qoc_gemm_launch_sk<false, 2, 1>(g, blocks, s);
    } else if (epi == 1) {
        if (sk == 8) qoc_gemm_launch_sk<false, 1, 8>(g, blocks, s);
        else if (sk == 4) qoc_gemm_launch_sk<false, 1, 4>(g, blocks, s);
        else if (sk == 2) qoc_gemm_launch_sk<false, 1, 2>(g, blocks, s);
        else qoc_gemm_launch_sk<false, 1, 1>(g, blocks, s);
    } else if (conjt) {
        if (sk == 8) qoc_gemm_launch_sk<true, 0, 8>(g, blocks, s);
        else if (sk == 4) qoc_gemm_launch_sk<true, 0, 4>(g, blocks, s);
        else if (sk == 2) qoc_gemm_launch_sk<true, 0, 2>(g, blocks, s);
        else qoc_gemm_launch_sk<true, 0, 1>(g, blocks, s);
    } else if (sk == 1 && (g.tiles_m & 1) == 0 && (g.tiles_n & 3) == 0 && (g.Kdim % ZW_KC) == 0 && g.Kdim >= 128 && tiles >= 8 * 1024) {
        // large plain products: workgroup tiles of 64 x 128 on the 4x4x4 MFMA form
        qoc_zgemm_wg_launch(g, (unsigned)(real_tiles / 8), s);
    } else {
        if (sk == 8) qoc_gemm_launch_sk<false, 0, 8>(g, blocks, s);
        else if (sk == 4) qoc_gemm_launch_sk<false, 0, 4>(g, blocks, s);
        else if (sk == 2) qoc_gemm_launch_sk<false, 0, 2>(g, blocks, s);
        else qoc_gemm_launch_sk<false, 0, 1>(g, blocks, s);
    }
}

static inline int gemm_grid(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }
// the slices t0 .. t0 + tn - 1 of every seed (needs what k_gemm_assemble_rows needs: k <= 8, N*N a multiple of 256)
static inline void qoc_gemm_assemble_window(const QocDev& d, const cplx* HsP, cplx* Aout, int N, int SP, int t0, int tn, hipStream_t s,
    int target_wgs = 8192, int nn = 0) {
    const size_t NN = nn > 0 ? (size_t)nn : (size_t)N * N, items = (size_t)d.B * tn;
    const int gx = (int)(NN / 256);
    int per = (int)((items * gx + target_wgs - 1) / target_wgs);
    if (per < 4) per = 4;
    const int gy = (int)((items + per - 1) / per);
    hipLaunchKernelGGL(k_gemm_assemble_rows, dim3(gx, gy), dim3(256), 0, s, d, HsP, Aout, N, SP, 0, per, (size_t)0, items, t0, tn, nn);
}
// [B | B^2] of the slices t0 .. t0 + tn - 1 of every seed (tn = 0: all items) for the squared-generator chain
template <int KK>
static inline void qoc_gemm_assemble_sq_k(const QocDev& d, const cplx* HsPK, const cplx* HsSQ, const double* coef, cplx* Aout, int SP,
    int t0, int tn, hipStream_t s, int target_wgs) {
    const size_t items = (size_t)d.B * (tn > 0 ? tn : SP);
    const int gx = QOC_DPP_PK_ELEMS / 256;
    int per = (int)((items * gx + target_wgs - 1) / target_wgs);
    if (per < 4) per = 4;
    const int gy = (int)((items + per - 1) / per);
    hipLaunchKernelGGL(k_gemm_assemble_sq<KK>, dim3(gx, gy), dim3(256), 0, s, d, HsPK, HsSQ, coef, Aout, SP, per, items, t0, tn);
}
static inline void qoc_gemm_assemble_sq(const QocDev& d, const cplx* HsPK, const cplx* HsSQ, const double* coef, cplx* Aout, int SP, int t0,
    int tn, hipStream_t s, int target_wgs = 8192) {
    switch (d.k) {
        case 1: qoc_gemm_assemble_sq_k<1>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        case 2: qoc_gemm_assemble_sq_k<2>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        case 3: qoc_gemm_assemble_sq_k<3>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        case 4: qoc_gemm_assemble_sq_k<4>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        case 5: qoc_gemm_assemble_sq_k<5>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        case 6: qoc_gemm_assemble_sq_k<6>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        case 7: qoc_gemm_assemble_sq_k<7>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
        default: qoc_gemm_assemble_sq_k<8>(d, HsPK, HsSQ, coef, Aout, SP, t0, tn, s, target_wgs); break;
    }
}
static inline void qoc_gemm_assemble_launch(const QocDev& d, const cplx* HsP, cplx* Aout, int N, int SP, int sq, hipStream_t s,
                                            size_t item_first = 0, size_t item_count = 0, int nn = 0) {
    if (item_count == 0) item_count = (size_t)d.B * SP;
    const size_t NN = nn > 0 ? (size_t)nn : (size_t)N * N, items = item_count;
    if (d.k <= 8 && NN % 256 == 0 && items >= 64) {
        const int gx = (int)(NN / 256);
        int per = (int)((items * gx + 8191) / 8192);                     // ~8192 workgroups
        if (per < 4) per = 4;
        const int gy = (int)((items + per - 1) / per);
        if (gy <= 65535) { hipLaunchKernelGGL(k_gemm_assemble_rows, dim3(gx, gy), dim3(256), 0, s, d, HsP, Aout, N, SP, sq, per, item_first,
            item_count, 0, 0, nn); return; }
    }
    hipLaunchKernelGGL(k_gemm_assemble, dim3(gemm_grid(items * NN)), dim3(256), 0, s, d, HsP, Aout, N, SP, sq, item_first, item_count, nn);
}

// pairwise product tree: T_l[i] = T_{l-1}[2i+1] * T_{l-1}[2i]  (later slice on the left), T_0 = K
// (item_first, item_count): the chunk-aligned run of (seed, slice) items whose tree is built -- all of them by default
static inline void qoc_gemm_tree(QocGemm& gm, const QocDev& d, hipStream_t s, size_t item_first = 0, size_t item_count = 0) {
    const int N = gm.N;
    const size_t NN = (size_t)N * N;
    if (item_count == 0) item_count = (size_t)d.B * gm.SP;
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = g.ldb = g.ldc = N; g.Kdim = N; g.tiles_m = g.tiles_n = N / 32; g.alpha = 1.0;
    const cplx* prev = gm.K;
    for (int l = 1; l <= gm.L; ++l) {
        cplx* out = gm.tree + gm.tree_off[l];
        g.A = prev + ((item_first >> (l - 1)) + 1) * NN; g.sA = 2 * (long long)NN; g.Bm = prev + (item_first >> (l - 1)) * NN;
            g.sB = 2 * (long long)NN;
        g.C = out + (item_first >> l) * NN; g.sC = (long long)NN;
        g.batch = (int)(item_count >> l);
        // chunk products also transposed, for the backward boundary chain
        const bool want_t = gm.persistent && !gm.direct && l == gm.L;
        g.CT = want_t ? gm.PcT : nullptr; g.sCT = (long long)NN; g.ldct = N;
        qoc_gemm_launch(gm, false, 0, g, s);
        prev = out;
    }
}

static inline void qoc_gemm_expm_products(QocGemm& gm, const QocDev& d, hipStream_t s, size_t item_first, size_t item_count);
// K_t for all (seed, slice): the dominant part of the path (bracketed by the profiling events of the engine)
static inline void qoc_gemm_expm(QocGemm& gm, const QocDev& d, hipStream_t s) {
    const int N = gm.N;
    const size_t BS = (size_t)d.B * gm.SP;
    const int deg = d.state_transfer ? d.T - 1 : d.T;            // matvecexp sums j < T (tensorflow_state.py:88-96)
    const int nsq = d.state_transfer ? 0 : d.s;
    if (gm.direct) {                                             // the chains apply the Taylor series themselves
        const int nn = gm.dpp_packed ? QOC_DPP_PK_ELEMS : (gm.dpp_chain && gm.dpp_cw < 16 ? 256 * gm.dpp_cw : 0);
        if (gm.sq_chain) {
            const int P = (d.k + 1) * (d.k + 2) / 2;
            hipLaunchKernelGGL(k_gemm_sq_coefs, dim3(gemm_grid((size_t)d.B * d.steps)), dim3(256), 0, s, d, gm.sqc, gm.SP, P);
            if (gm.asm_split > 0) {
                const int nw = (int)gm.asm_win.size() - 1;
                qoc_gemm_assemble_sq(d, gm.HsPT, gm.HsSQ, gm.sqc, gm.A, gm.SP, 0, gm.asm_win[1], s);
                hipEventRecord(gm.ev_ready, s);
                hipStreamWaitEvent(gm.aux, gm.ev_ready, 0);
                for (int w = 1; w < nw; ++w) {
                    qoc_gemm_assemble_sq(d, gm.HsPT, gm.HsSQ, gm.sqc, gm.A, gm.SP, gm.asm_win[w], gm.asm_win[w + 1] - gm.asm_win[w], gm.aux,
                        gm.asm_tail_wgs);
                    hipEventRecord(gm.ev_win[w], gm.aux);
                }
            }
            else qoc_gemm_assemble_sq(d, gm.HsPT, gm.HsSQ, gm.sqc, gm.A, gm.SP, 0, 0, s);
            return;
        }
        // head on this stream, tail on the second one beside the forward chain's first part
        if (gm.asm_split > 0) {
            const int nw = (int)gm.asm_win.size() - 1;
            qoc_gemm_assemble_window(d, gm.HsPT, gm.A, N, gm.SP, 0, gm.asm_win[1], s, 8192, nn);
            // the head has the memory system to itself (started together, both took as long as the whole)
            hipEventRecord(gm.ev_ready, s);
            hipStreamWaitEvent(gm.aux, gm.ev_ready, 0);
            for (int w = 1; w < nw; ++w) {
                qoc_gemm_assemble_window(d, gm.HsPT, gm.A, N, gm.SP, gm.asm_win[w], gm.asm_win[w + 1] - gm.asm_win[w], gm.aux,
                    gm.asm_tail_wgs, nn);
                hipEventRecord(gm.ev_win[w], gm.aux);
            }
            return;
        }
        // dpp_chain: generators column-major
        qoc_gemm_assemble_launch(d, gm.dpp_chain ? gm.HsPT : gm.HsP, gm.A, N, gm.SP, 0, s, 0, 0, nn);
        return;
    }
    if (N <= 64) {
        ExpmCoef cf;
        { double f = 1.0; for (int j = 0; j < QOC_GEMM_MAXT; ++j) { if (j > 0) f *= (double)j; cf.c[j] = 1.0 / f; } }
        const size_t lds = 2 * (size_t)N * (N + QOC_EXPM_LDPAD) * sizeof(cplx);
        if (N == 32) hipLaunchKernelGGL(k_gemm_expm_fused<32>, dim3((unsigned)BS), dim3(128), lds, s, d, gm.HsP, gm.K,
            gm.persistent ? gm.KT : (cplx*)nullptr, gm.SP, deg, nsq, cf);
        else hipLaunchKernelGGL(k_gemm_expm_fused<64>, dim3((unsigned)BS), dim3(512), lds, s, d, gm.HsP, gm.K,
            gm.persistent ? gm.KT : (cplx*)nullptr, gm.SP, deg, nsq, cf);
        qoc_gemm_tree(gm, d, s);
        return;
    }
    // one control set: the padded slices (K = I exactly, written once by qoc_gemm_setup) are not computed -- C5: 16 of 2016 slices, 96
    // products
    qoc_gemm_expm_products(gm, d, s, 0, d.B == 1 ? (size_t)d.steps : BS);
    qoc_gemm_tree(gm, d, s);
}

// N > 64: K_t of the items [item_first, item_first + item_count) by batched launches (all items, or the slices of one rank of a
// time-sharded engine)
static inline void qoc_gemm_expm_products(QocGemm& gm, const QocDev& d, hipStream_t s, size_t item_first, size_t item_count) {
    const int N = gm.N;
    const size_t NN = (size_t)N * N, BS = item_count, off = item_first * NN;
    const int deg = d.state_transfer ? d.T - 1 : d.T;
    const int nsq = d.state_transfer ? 0 : d.s;
    qoc_gemm_assemble_launch(d, gm.HsP, gm.A, N, gm.SP, nsq, s, item_first, item_count);
    // Taylor polynomial sum_{j<=T} A^j/j! (tensorflow_state.py:37-41) in Paterson-Stockmeyer form over A2 = A*A:
    // S = B_m ; S = B_i + A2*S with B_i = c_{2i} I + c_{2i+1} A  (T = 5: 3 products instead of 4); then s squarings.
    cplx* const bufA = gm.A + off; cplx* const bufA2 = gm.A2 + off; cplx* const bufK = gm.K + off; cplx* const bufP = gm.P + off;
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = g.ldb = g.ldc = g.lde = N; g.sA = g.sB = g.sC = g.sE = (long long)NN; g.Kdim = N; g.tiles_m = g.tiles_n = N / 32;
        g.batch = (int)BS;
    double invf[QOC_GEMM_MAXT];
    { double f = 1.0; for (int j = 0; j < QOC_GEMM_MAXT; ++j) { if (j > 0) f *= (double)j; invf[j] = 1.0 / f; } }
    const int mm = deg >> 1;
    const bool even = (deg & 1) == 0;
    const int horner = deg >= 2 ? (even ? mm - 1 : mm) : 0;      // products after A2
    const int products = horner + nsq;                           // buffer flips until the result
    cplx* cur = (products % 2 == 0) ? bufK : bufP;               // buffers alternate cur -> other on every product
    cplx* oth = (products % 2 == 0) ? bufP : bufK;
    if (deg >= 2) {
        g.A = bufA; g.Bm = bufA; g.C = bufA2; g.E = nullptr; g.alpha = 1.0; g.beta = 0.0; g.gamma = 0.0;
        qoc_gemm_launch(gm, false, 0, g, s);                         // A2 = A*A
        // odd order on the workgroup-tiled kernel: the top block S = c_{2m} I + c_{2m+1} A is formed from A while the first Horner product
        // stages its right operand (GemmArgs::btrans) -- no k_gemm_ps_init pass (C5: 3.2 ms of reading and writing 8.4 GB each)
        const bool top_in_flight = !even && mm >= 1 && qoc_gemm_takes_wg(gm, g);
        if (even) hipLaunchKernelGGL(k_gemm_ps_init, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, bufA, bufA2, cur, BS * NN, N,
                                     invf[2 * mm - 2], invf[2 * mm - 1], invf[deg]);
        else if (!top_in_flight) hipLaunchKernelGGL(k_gemm_ps_init, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, bufA, (const cplx*)nullptr,
            cur, BS * NN, N,
                                                    invf[2 * mm], invf[2 * mm + 1], 0.0);
        for (int i = (even ? mm - 2 : mm - 1); i >= 0; --i) {    // S <- c_{2i} I + c_{2i+1} A + A2*S
            g.A = bufA2; g.Bm = cur; g.C = oth; g.E = bufA; g.alpha = 1.0; g.beta = invf[2 * i + 1]; g.gamma = invf[2 * i];
            g.btrans = 0;
            if (top_in_flight && i == mm - 1) { g.Bm = bufA; g.btrans = 1; g.bt_c0 = invf[2 * mm]; g.bt_c1 = invf[2 * mm + 1]; }
            qoc_gemm_launch(gm, false, 0, g, s);
            g.btrans = 0;
            cplx* t = cur; cur = oth; oth = t;
        }
    } else {
        hipLaunchKernelGGL(k_gemm_ps_init, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, bufA, (const cplx*)nullptr, cur, BS * NN, N, 1.0,
                           deg >= 1 ? 1.0 : 0.0, 0.0);
    }
    for (int sq = 0; sq < nsq; ++sq) {                       // M <- M M                    tensorflow_state.py:43-44
        g.A = cur; g.Bm = cur; g.C = oth; g.E = nullptr; g.alpha = 1.0; g.beta = 0.0; g.gamma = 0.0;
        qoc_gemm_launch(gm, false, 0, g, s);
        cplx* t = cur; cur = oth; oth = t;
    }
    (void)cur;                                               // == the K buffer by construction
}

static inline const cplx* qoc_gemm_chunk_products(const QocGemm& gm) { return gm.L > 0 ? gm.tree + gm.tree_off[gm.L] : gm.K; }

// lambda_{t-1} = P(-B_t) lambda_t + S_t   tensorflow_state.py:118-131 (direct route)
static inline ChainArgs qoc_gemm_direct_backward_args(const QocGemm& gm, const QocDev& d, bool need_src) {
    const int N = gm.N;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    ChainArgs a;
    memset(&a, 0, sizeof a);
    const size_t GE = gm.gen_elems();
    a.K = gm.A + (size_t)(d.steps - 1) * GE; a.sKb = (long long)GE * gm.SP; a.sKs = -(long long)GE;
    a.X0 = gm.Ebnd; a.sXb = (long long)thin;
    // compact sources
    if (need_src
        && gm.dpp_chain) { a.E = gm.SrcP + (size_t)(d.steps - 1) * N; a.sEb = (long long)N * gm.SP; a.sEs = -(long long)N; a.ldE = 1; }
    else if (need_src) { a.E = gm.SrcP + (size_t)(d.steps - 1) * thin; a.sEb = (long long)thin * gm.SP; a.sEs = -(long long)thin; }
    a.Out = gm.LamP + (long long)(d.steps - 2) * gm.MV; a.sOb = (long long)N * gm.ldW; a.sOs = -gm.MV; a.ldO = gm.ldW;
    a.store_initial = 1; a.CI = 1; a.len = d.steps - 1; a.m = d.m; a.nterms = d.T; a.sign = -1.0;
    return a;
}
// direct route without a state regulariser: backward chain beside the forward one (see qoc_gemm_forward)
static inline bool qoc_gemm_zfree_backward(const QocGemm& gm, const QocDev& d) { return gm.direct && !(d.n_forb > 0 || d.has_speed)
    && d.steps >= 2; }

// launch-per-step route in unitary mode: final_state / unitary_scale are formed when they are read back (qoc_gemm_final_state) -- inside
// the iterations the boundary chain carries the m vectors only, not the N columns of X beside them (C5: 63 products of 512 x 544 columns
// per iteration)
static inline bool qoc_gemm_lazy_final(const QocGemm& gm, const QocDev& d) { return !gm.persistent && !gm.direct && !d.state_transfer; }

static inline void qoc_gemm_forward(QocGemm& gm, const QocDev& d, hipStream_t s, bool with_final = false) {
    const int N = gm.N, xw = (d.state_transfer || (qoc_gemm_lazy_final(gm, d) && !with_final)) ? 0 : N, ld = xw + QOC_TW, S = gm.S,
        NC = gm.NC;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    const cplx* Pc = qoc_gemm_chunk_products(gm);                // [B][NC]
    if (!gm.init_once) hipLaunchKernelGGL(k_gemm_chain_init, dim3(gemm_grid((size_t)d.B * N * ld)), dim3(256), 0, s, d, gm.Y0, gm.Psibnd, N,
        NC, xw);
    if (gm.direct) {
        ChainArgs a;
        memset(&a, 0, sizeof a);
        const size_t GE = gm.gen_elems();
        a.K = gm.A; a.sKb = (long long)GE * gm.SP; a.sKs = (long long)GE;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin;
        a.Out = gm.interP; a.sOb = (long long)N * gm.ldW; a.sOs = gm.MV; a.ldO = gm.ldW;
        a.CI = 1; a.len = d.steps; a.m = d.m; a.nterms = d.T; a.sign = 1.0;
        // the chain writes inter[b][t + 1] itself (one vector: n contiguous entries per step)
        if (gm.dpp_chain) {
            a.Out2 = d.inter + d.n; a.sO2b = (long long)(d.steps + 1) * d.n; a.sO2s = d.n; a.n2 = d.n;
        }
        if (qoc_gemm_zfree_backward(gm, d)) {
            // no state regulariser: the costate is linear in the overlap z -- the backward chain starts from -(2/m^2) W and runs
            // beside the forward one; qoc_gemm_backward multiplies by z (C3 x 64: 13.2 -> 8 ms per iteration)
            hipLaunchKernelGGL(k_gemm_zfree_end, dim3(gemm_grid((size_t)d.B * thin)), dim3(256), 0, s, d, gm.Ebnd, N, NC);
            qoc_taylor_chain_launch2(N, a, qoc_gemm_direct_backward_args(gm, d, false), gm.zthin, d.B, s, gm.dpp_mode());
        }
        else if (gm.asm_split > 0) {
            // one chain launch per window, each from the state the previous one left in Aoff; with a CU mask the chains keep their own CUs
            // (the assembly of the later windows runs on the others) and the engine's stream joins after the last window
            const int nw = (int)gm.asm_win.size() - 1;
            hipStream_t cs = gm.chain_s ? gm.chain_s : s;
            if (gm.chain_s) { hipEventRecord(gm.ev_fwd, s); hipStreamWaitEvent(cs, gm.ev_fwd, 0); }
            for (int w = 0; w < nw; ++w) {
                ChainArgs p = a;
                const int t0 = gm.asm_win[w];
                p.len = gm.asm_win[w + 1] - t0;
                if (w > 0) { p.X0 = gm.Aoff; p.sXb = (long long)thin; hipStreamWaitEvent(cs, gm.ev_win[w], 0); }
                if (w + 1 < nw) { p.Fin = gm.Aoff; p.sFb = (long long)thin; }
                p.K = a.K + (long long)t0 * a.sKs; p.Out = a.Out + (long long)t0 * a.sOs; p.Out2 = a.Out2 + (long long)t0 * a.sO2s;
                qoc_taylor_chain_launch(N, p, gm.zthin, d.B, cs, gm.dpp_mode());
            }
            if (gm.chain_s) { hipEventRecord(gm.ev_p1, cs); hipStreamWaitEvent(s, gm.ev_p1, 0); }
        }
        else qoc_taylor_chain_launch(N, a, gm.zthin, d.B, s, gm.dpp_mode());
        if (!gm.dpp_chain) hipLaunchKernelGGL(k_gemm_unpad_wide, dim3(gemm_grid((size_t)d.B * d.steps * d.n * d.m)), dim3(256), 0, s, d,
            gm.interP, N, gm.ldW, gm.MV);
        return;
    }
    if (gm.persistent && d.state_transfer) {
        // chunk-start vectors Psibnd[c+1] = P_c Psibnd[c]: one persistent workgroup per seed.  State transfer has no use for
        // the upper product tree, and building it only for the scan costs more than the chain (C3: 0.58 vs 0.50 ms)
        ChainArgs a;
        memset(&a, 0, sizeof a);
        a.K = Pc; a.sKb = (long long)NN * NC; a.sKs = (long long)NN;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin * NC;
        a.Out = gm.Psibnd + thin; a.sOb = (long long)thin * NC; a.sOs = (long long)thin; a.ldO = QOC_TW;
        a.CI = 1; a.len = NC - 1; a.m = d.m;
        qoc_chain_launch(N, false, a, gm.zthin, d.B, s);
    }
    if (gm.persistent && !d.state_transfer) {
        // the product tree continues above the chunk products (log2(NC) launches): its root gives final_state =
        // (P_{NC-1} ... P_0) U0, its nodes give every chunk-boundary vector in log depth (k_gemm_scan_nodes)
        GemmArgs r;
        memset(&r, 0, sizeof r);
        r.lda = r.ldb = r.ldc = N; r.Kdim = N; r.tiles_m = r.tiles_n = N / 32; r.alpha = 1.0;
        const cplx* lvl = Pc;
        cplx* out = gm.root;
        ScanArgs& sc = gm.scan;
        memset(&sc, 0, sizeof sc);
        sc.lvl[0] = Pc; sc.sLb[0] = (long long)NC * NN; sc.cnt[0] = NC; sc.levels = 1;
        for (int cnt = NC; cnt > 1; cnt = (cnt + 1) / 2) {
            const int pairs = cnt / 2, nxt = (cnt + 1) / 2;
            r.A = lvl + NN; r.Bm = lvl; r.C = out; r.sA = r.sB = 2 * (long long)NN; r.sC = (long long)NN;
            r.inner = pairs; r.sA2 = r.sB2 = (long long)cnt * NN; r.sC2 = (long long)nxt * NN; r.batch = d.B * pairs;
            qoc_gemm_launch(gm, false, 0, r, s);
            if (cnt & 1)
                hipLaunchKernelGGL(k_gemm_copy_mats, dim3(gemm_grid((size_t)d.B * NN)), dim3(256), 0, s, out + (size_t)pairs * NN,
                                   (long long)nxt * NN, lvl + (size_t)(cnt - 1) * NN, (long long)cnt * NN, d.B, (int)NN);
            if (sc.levels < 10) { sc.lvl[sc.levels] = out; sc.sLb[sc.levels] = (long long)nxt * NN; sc.cnt[sc.levels] = nxt; ++sc.levels; }
            lvl = out;
            out += (size_t)d.B * nxt * NN;
        }
        sc.NC = NC;
        memset(&r, 0, sizeof r);
        r.A = lvl; r.sA = (long long)NN; r.lda = N; r.Bm = gm.Y0; r.C = gm.Y1; r.ldb = r.ldc = ld; r.sB = r.sC = (long long)N * ld;
        r.Kdim = N; r.tiles_m = N / 32; r.tiles_n = ld / 32; r.batch = d.B; r.alpha = 1.0;
        qoc_gemm_launch(gm, false, 0, r, s);
        hipLaunchKernelGGL(k_gemm_take_final, dim3(d.B), dim3(N > 64 ? 1024 : 256), 0, s, d, gm.Y1, N);
        // chunk-start vectors Psibnd[c] = P_{c-1} ... P_0 Psi0, c = 1 .. NC-1: one workgroup per (seed, chunk), <= log2(NC) nodes
        ScanArgs a = sc;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin * NC;
        a.Out = gm.Psibnd; a.sOb = (long long)thin * NC; a.sOc = (long long)thin;
        a.c0 = 1; a.nchains = NC - 1; a.suffix = 0;
        qoc_scan_launch(N, a, d.B, s);
    }
    // chunk boundaries: [X | Psi] <- P_c [X | Psi]   (X for final_state, Psi for the chunk starts)      :214-238
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.sA = (long long)NN * NC; g.ldb = g.ldc = ld; g.sB = g.sC = (long long)N * ld;
    g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = ld / 32; g.batch = d.B; g.alpha = 1.0;
    const size_t yslot = (size_t)d.B * N * ld;
    for (int c = 0; c < (gm.persistent ? 0 : NC); ++c) {
        g.A = Pc + (size_t)c * NN; g.Bm = gm.Y0 + (size_t)c * yslot; g.C = gm.Y0 + (size_t)(c + 1) * yslot;
        qoc_gemm_launch(gm, false, 0, g, s);
    }
    if (!gm.persistent && NC > 1)
        hipLaunchKernelGGL(k_gemm_take_bnd_all, dim3(gemm_grid((size_t)d.B * (NC - 1) * thin)), dim3(256), 0, s, d, gm.Y0, gm.Psibnd, N, NC,
            xw);
    if (xw > 0 && !gm.persistent) hipLaunchKernelGGL(k_gemm_take_final, dim3(d.B), dim3(N > 64 ? 1024 : 256), 0, s, d,
        gm.Y0 + (size_t)NC * yslot, N);
    // read-back of final_state: the boundary chain with X beside the vectors was all that was asked for
    if (with_final) return;
    if (gm.persistent) {
        // every chunk swept by its own persistent workgroup: Psi_{cS+j} = K_{cS+j} Psi_{cS+j-1}
        ChainArgs a;
        memset(&a, 0, sizeof a);
        a.K = gm.K; a.sKb = (long long)NN * gm.SP; a.sKc = (long long)NN * S; a.sKs = (long long)NN;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin * NC; a.sXc = (long long)thin;
        // time-major wide layout
        a.Out = gm.interP; a.sOb = (long long)N * gm.ldW; a.sOc = (long long)S * gm.MV; a.sOs = gm.MV; a.ldO = gm.ldW;
        a.CI = NC; a.len = S; a.m = d.m;
        qoc_chain_launch(N, false, a, gm.zthin, d.B * NC, s);
        hipLaunchKernelGGL(k_gemm_unpad_wide, dim3(gemm_grid((size_t)d.B * d.steps * d.n * d.m)), dim3(256), 0, s, d, gm.interP, N, gm.ldW,
            gm.MV);
        return;
    }
    // all chunks swept together: Psi_{cS+j} = K_{cS+j} Psi_{cS+j-1}, one launch per j, batch = B*NC
    GemmArgs h;
    memset(&h, 0, sizeof h);
    h.lda = N; h.sA = (long long)NN * S; h.ldb = h.ldc = QOC_TW; h.Kdim = N; h.tiles_m = N / 32; h.tiles_n = 1;
    h.batch = d.B * NC; h.alpha = 1.0; h.sC = (long long)thin * S;
    for (int j = 0; j < S; ++j) {
        h.A = gm.K + (size_t)j * NN;
        if (j == 0) { h.Bm = gm.Psibnd; h.sB = (long long)thin; }
        else { h.Bm = gm.interP + (size_t)(j - 1) * thin; h.sB = (long long)thin * S; }
        h.C = gm.interP + (size_t)j * thin;
        qoc_gemm_launch(gm, false, 0, h, s);
    }
    hipLaunchKernelGGL(k_gemm_unpad_inter, dim3(gemm_grid((size_t)d.B * d.steps * d.n * d.m)), dim3(256), 0, s, d, gm.interP, N, gm.SP);
}

// one backward pass over all chunks in parallel: Lambda_{cS+j-1} = K_{cS+j}^dagger Lambda_{cS+j} + S_{cS+j}, j = S-1 .. 1;
// the j = 0 product (result belongs to the previous chunk's end) goes to `first_out` [B][NC] when requested
static inline void qoc_gemm_bwd_sweep(QocGemm& gm, const QocDev& d, hipStream_t s, bool need_src, cplx* first_out) {
    const int N = gm.N, S = gm.S, NC = gm.NC;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.sA = (long long)NN * S; g.ldb = g.ldc = g.lde = QOC_TW; g.sB = (long long)thin * S; g.sE = (long long)thin * S;
    g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = 1; g.batch = d.B * NC; g.alpha = 1.0; g.beta = 1.0;
    for (int j = S - 1; j >= (first_out ? 0 : 1); --j) {
        g.A = gm.K + (size_t)j * NN; g.Bm = gm.LamP + (size_t)j * thin;
        g.E = need_src ? gm.SrcP + (size_t)j * thin : nullptr;
        if (j > 0) { g.C = gm.LamP + (size_t)(j - 1) * thin; g.sC = (long long)thin * S; }
        else { g.C = first_out; g.sC = (long long)thin; }
        qoc_gemm_launch(gm, true, 0, g, s);
    }
}

// gradients from the time-major wide layout: one product H_k' [Psi_0 ... Psi_{SP-1}] per control (batch = seeds), contracted column by
// column with conj(Lambda) (tensorflow_state.py:61-63) -- the columns [c_first, c_end) (multiples of 32) of it
static inline void qoc_gemm_wide_gradient(QocGemm& gm, const QocDev& d, hipStream_t s, int c_first, int c_end) {
    const int N = gm.N, tm = N / 32;
    const size_t NN = (size_t)N * N;
    GemmArgs h;
    memset(&h, 0, sizeof h);
    h.lda = N; h.ldb = h.ldl = gm.ldW; h.Kdim = N;
    h.tiles_m = tm; h.tiles_n = (c_end - c_first) / 32; h.Bm = gm.interP + c_first; h.L = gm.LamP + c_first;
    h.partial = gm.partial + c_first; h.ldp = gm.ldW; h.partial_stride = tm * gm.ldW;         // partial[b][k][tile_m][column]
    // one launch for all (seed, control) pairs: batch index bt = b*k + kk -> A = H'_{kk+1}, Bm / L = buffers of seed b
    h.A = gm.HsP + NN; h.inner = d.k; h.sA = (long long)NN; h.sA2 = 0;
    h.sB = h.sL = 0; h.sB2 = h.sL2 = (long long)N * gm.ldW;
    h.batch = d.B * d.k;
    qoc_gemm_launch(gm, false, 2, h, s, (size_t)h.batch * tm * (gm.ldW / 32));
}

static inline void qoc_gemm_backward(QocGemm& gm, const QocDev& d, hipStream_t s) {
    const int N = gm.N, S = gm.S, NC = gm.NC;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    const bool need_src = d.n_forb > 0 || d.has_speed;
    const cplx* Pc = qoc_gemm_chunk_products(gm);
    // the chain ran beside the forward one from -(2/m^2) W: Lambda_t = z Lambda0_t
    if (qoc_gemm_zfree_backward(gm, d)) {
        hipLaunchKernelGGL(k_gemm_scale_lam, dim3(gemm_grid((size_t)d.B * N * d.steps * gm.MV)), dim3(256), 0, s, d, gm.LamP, N, gm.ldW,
            d.steps * gm.MV);
    } else {
    {
        const int cols = gm.direct ? gm.MV : QOC_TW;
        hipLaunchKernelGGL(k_gemm_sources, dim3(gemm_grid((size_t)d.B * (need_src ? gm.SP : 1) * N * cols)), dim3(256), 0, s, d, gm.SrcP,
            gm.Ebnd, N, gm.SP, NC, cols, gm.dpp_chain ? 1 : 0);
    }
    }
    if (gm.direct) {
        // (the gradient products of the slices the chain has already left, on the second stream beside the rest of the chain: built and
        // measured in round 4, 6.19 against 6.17 ms at C3 x 64 -- the products slow the chain's prefetch as much as they save;
        // profiles/EXPERIMENTS.md)
        if (!qoc_gemm_zfree_backward(gm, d)) qoc_taylor_chain_launch(N, qoc_gemm_direct_backward_args(gm, d, need_src), gm.zthin, d.B, s,
            gm.dpp_mode());
    } else if (gm.persistent) {
        ChainArgs sw;                                        // one chunk, backwards: Lambda_{t-1} = K_t^dagger Lambda_t + S_t
        memset(&sw, 0, sizeof sw);
        // conj(K^T) = K^H
        sw.K = gm.KT + (size_t)(S - 1) * NN; sw.sKb = (long long)NN * gm.SP; sw.sKc = (long long)NN * S; sw.sKs = -(long long)NN;
        if (need_src) { sw.E = gm.SrcP + (size_t)(S - 1) * thin; sw.sEb = (long long)thin * gm.SP; sw.sEc = (long long)thin * S; sw.sEs
            = -(long long)thin; }
        sw.CI = NC; sw.m = d.m;
        if (need_src && NC > 1) {                            // affine offsets a_c: every chunk run from a zero costate
            ChainArgs a = sw;
            a.len = S; a.Fin = gm.Aoff; a.sFb = (long long)thin * NC; a.sFc = (long long)thin;
            qoc_chain_launch(N, true, a, gm.zthin, d.B * NC, s);
        }
        // chunk-end costates E_c = P_{c+1}^H ... P_{NC-1}^H E_{NC-1}, log depth (unitary mode: the tree exists)
        if (!need_src && !d.state_transfer) {
            ScanArgs a = gm.scan;
            a.X0 = gm.Ebnd + (size_t)(NC - 1) * thin; a.sXb = (long long)thin * NC;
            a.Out = gm.Ebnd; a.sOb = (long long)thin * NC; a.sOc = (long long)thin;
            a.c0 = 0; a.nchains = NC - 1; a.suffix = 1;
            qoc_scan_launch(N, a, d.B, s);
        // with sources the recursion is affine: E_{c-1} = P_c^dagger E_c + a_c, sequential
        } else {
            ChainArgs a;
            memset(&a, 0, sizeof a);
            const cplx* PcT = gm.L > 0 ? gm.PcT : gm.KT;
            a.K = PcT + (size_t)(NC - 1) * NN; a.sKb = (long long)NN * NC; a.sKs = -(long long)NN;
            a.X0 = gm.Ebnd + (size_t)(NC - 1) * thin; a.sXb = (long long)thin * NC;
            if (need_src) { a.E = gm.Aoff + (size_t)(NC - 1) * thin; a.sEb = (long long)thin * NC; a.sEs = -(long long)thin; }
            a.Out = gm.Ebnd + (long long)(NC - 2) * (long long)thin; a.sOb = (long long)thin * NC; a.sOs = -(long long)thin; a.ldO = QOC_TW;
            a.CI = 1; a.len = NC - 1; a.m = d.m;
            qoc_chain_launch(N, true, a, gm.zthin, d.B, s);
        }
        {
            ChainArgs a = sw;
            a.X0 = gm.Ebnd; a.sXb = (long long)thin * NC; a.sXc = (long long)thin;
            a.Out = gm.LamP + (long long)(S - 2) * gm.MV; a.sOb = (long long)N * gm.ldW; a.sOc = (long long)S * gm.MV; a.sOs = -gm.MV;
                a.ldO = gm.ldW;
            a.store_initial = 1; a.len = S - 1;
            qoc_chain_launch(N, true, a, gm.zthin, d.B * NC, s);
        }
    } else {
    if (need_src && NC > 1) {                                // affine offsets a_c: every chunk run from a zero costate
        hipLaunchKernelGGL(k_gemm_set_chunk_ends, dim3(gemm_grid((size_t)d.B * NC * thin)), dim3(256), 0, s, d, gm.LamP,
            (const cplx*)nullptr, N, S, NC);
        qoc_gemm_bwd_sweep(gm, d, s, true, gm.Aoff);
    }
    // chunk-end costates: E_{c-1} = P_c^dagger E_c + a_c
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.sA = (long long)NN * NC; g.ldb = g.ldc = g.lde = QOC_TW; g.sB = g.sC = g.sE = (long long)thin * NC;
    g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = 1; g.batch = d.B; g.alpha = 1.0; g.beta = 1.0;
    for (int c = NC - 1; c >= 1; --c) {
        g.A = Pc + (size_t)c * NN; g.Bm = gm.Ebnd + (size_t)c * thin; g.C = gm.Ebnd + (size_t)(c - 1) * thin;
        g.E = need_src ? gm.Aoff + (size_t)c * thin : nullptr;
        qoc_gemm_launch(gm, true, 0, g, s);
    }
    hipLaunchKernelGGL(k_gemm_set_chunk_ends, dim3(gemm_grid((size_t)d.B * NC * thin)), dim3(256), 0, s, d, gm.LamP, (const cplx*)gm.Ebnd,
        N, S, NC);
    qoc_gemm_bwd_sweep(gm, d, s, need_src, nullptr);
    }
    if (gm.persistent) {
        qoc_gemm_wide_gradient(gm, d, s, 0, gm.ldW);
        if (!gm.reduce_in_tail) hipLaunchKernelGGL(k_gemm_grad_reduce_wide, dim3(gemm_grid((size_t)d.B * d.steps * d.k)), dim3(256), 0, s, d, gm.partial, N / 32,
            gm.ldW, gm.MV);
        return;
    }
    if (gm.wideW > 0) {
        const int W = gm.wideW;
        GemmArgs h;
        memset(&h, 0, sizeof h);
        h.A = gm.HsP + NN; h.sA = (long long)NN; h.lda = N;                     // batch index = control
        h.Bm = gm.wideP; h.sB = 0; h.ldb = W;
        h.C = gm.wideC; h.sC = (long long)N * W; h.ldc = W;
        h.Kdim = N; h.tiles_m = N / 32; h.tiles_n = W / 32; h.batch = d.k; h.alpha = 1.0;
        for (int b = 0; b < d.B; ++b) {
            hipLaunchKernelGGL(k_gemm_to_wide, dim3(gemm_grid((size_t)d.steps * N * QOC_WIDE_MV)), dim3(256), 0, s, d,
                               (const cplx*)(gm.interP + (size_t)b * gm.SP * thin), (const cplx*)(gm.LamP + (size_t)b * gm.SP * thin),
                                   gm.wideP, gm.wideL, N, W, d.steps);
            qoc_gemm_launch(gm, false, 0, h, s);
            hipLaunchKernelGGL(k_gemm_dot_wide, dim3((unsigned)(((size_t)d.k * d.steps + 3) / 4)), dim3(256), 0, s, d, b,
                (const cplx*)gm.wideC, (const cplx*)gm.wideL, N, W, 0, d.steps);
        }
        return;
    }
    // gradients: for each control one batched product H_k' Psi_t contracted with conj(Lambda_t)   tensorflow_state.py:61-63
    GemmArgs h;
    memset(&h, 0, sizeof h);
    h.lda = N; h.sA = 0; h.ldb = QOC_TW; h.ldl = QOC_TW; h.Kdim = N; h.tiles_m = N / 32; h.tiles_n = 1;
    h.partial_stride = d.k * (N / 32);
    for (int b = 0; b < d.B; ++b) {
        h.batch = d.steps;
        h.Bm = gm.interP + (size_t)b * gm.SP * thin; h.sB = (long long)thin;
        h.L = gm.LamP + (size_t)b * gm.SP * thin; h.sL = (long long)thin;
        for (int kk = 0; kk < d.k; ++kk) {
            h.A = gm.HsP + (size_t)(kk + 1) * NN;
            h.partial = gm.partial + (size_t)b * d.steps * h.partial_stride;
            h.partial_offset = kk * (N / 32);
            qoc_gemm_launch(gm, false, 1, h, s);
        }
    }
    hipLaunchKernelGGL(k_gemm_grad_reduce, dim3(gemm_grid((size_t)d.B * d.steps * d.k)), dim3(256), 0, s, d, gm.partial, N / 32);
}
