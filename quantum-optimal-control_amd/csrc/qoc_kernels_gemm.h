// qoc_kernels_gemm.h -- "GEMM path": unitary mode for any n > 32 (and m <= 32), e.g. BASELINE config C5 (n = 512).
//
// Matrices are zero-padded to N = 32*ceil(n/32) and stay in HBM/L2 as plain row-major complex128; every product is a
// launch of k_zgemm32: ONE wavefront per 32x32 output tile, v_mfma_f64_16x16x4_f64 with the 3-multiplication complex
// form (12 MFMAs per 4-deep k-slice, 12 independent accumulator chains), operand fragments loaded straight from
// global memory in the MFMA A/B lane layouts (B rows are coalesced 256-byte segments; A is a 16-row x 64-byte gather
// that L1/L2 absorb), no LDS and no barriers.  The host sequences the launches:
//   * matrix exponentials: batched over ALL (seed, slice) pairs -- Horner form of the Taylor series + squarings;
//   * forward chain: one launch per slice on the concatenation [X | Psi] (N x (N+32)), i.e. X_t = K_t X_{t-1} and
//     Psi_t = K_t Psi_{t-1} together;
//   * backward chain: one launch per slice, Lambda_{t-1} = K_t^dagger Lambda_t + S_{t-1} (epilogue adds the sources);
//   * control gradients: for each control k ONE batched launch over all (seed, slice): the tile of H_k' Psi_t is
//     contracted with conj(Lambda_t) in the epilogue (deterministic per-tile partial sums, reduced by k_gemm_grad_reduce).
// Reference semantics: core/tensorflow_state.py:25-46, 49-65, 204-242.
#pragma once
#include <string>
#include <vector>
#include "qoc_common.h"

typedef double gd4 __attribute__((ext_vector_type(4)));
#define GMFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define QOC_TW 32          // thin (vector block) width

struct GemmArgs {
    const cplx* A; long long sA; int lda;      // left operand  (batch stride in elements; 0 = shared)
    const cplx* Bm; long long sB; int ldb;     // right operand
    cplx* C; long long sC; int ldc;            // output
    const cplx* E; long long sE; int lde;      // optional addend (nullptr = none)
    double alpha, beta, gamma;                 // C = alpha*op(A)*B + beta*E + gamma*I
    int Kdim;                                  // inner dimension (multiple of 4)
    int tiles_m, tiles_n;                      // output tiles per matrix
    int batch;
    // dot epilogue (EPI = 1): partial[batch][tile_m] = Re sum conj(L)*(A*B) over the tile
    const cplx* L; long long sL; int ldl;
    double* partial; int partial_stride;       // partial[(batch*partial_stride) + offset + tile_m]
    int partial_offset;
    int inner; long long sA2, sB2, sC2, sL2;   // inner > 0: batch index bt -> (bt / inner, bt % inner); A, Bm, C, L offsets = hi*s?2 + lo*s?
    int ldp;                                   // EPI = 2: per-COLUMN dots, partial[batch*stride + offset + tile_m*ldp + col]
};

// SK wavefronts of a workgroup split the inner dimension of ONE tile (small-batch chain launches are latency-bound when
// a single wave walks all of K); partial (re, im) tiles meet in LDS (16 KB per extra wave), wave 0 runs the epilogue.
template <bool CONJT, int EPI, int SK>
__global__ void __launch_bounds__(64 * SK) k_zgemm32(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sk_part[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = g.tiles_m * g.tiles_n;
    const int bt = blockIdx.x / tiles, tile = blockIdx.x - bt * tiles;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int r0 = tm * 32, c0 = tn * 32;
    const int bhi = g.inner > 0 ? bt / g.inner : 0, blo = g.inner > 0 ? bt - bhi * g.inner : bt;
    const cplx* __restrict__ A = g.A + (size_t)bhi * g.sA2 + (size_t)blo * g.sA;
    const cplx* __restrict__ Bm = g.Bm + (size_t)bhi * g.sB2 + (size_t)blo * g.sB;
    gd4 t1[2][2], t2[2][2], t3[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) { t1[I][J] = (gd4){0, 0, 0, 0}; t2[I][J] = (gd4){0, 0, 0, 0}; t3[I][J] = (gd4){0, 0, 0, 0}; }
    const int lr = lane & 15, lk = lane >> 4;
    const int kspan = g.Kdim / SK;
    for (int k0 = wv * kspan; k0 < (wv + 1) * kspan; k0 += 8) {     // two k-slices per trip: 8 loads in flight
        cplx a[2][2], b[2][2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int kk = k0 + 4 * qq + lk;
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                if (CONJT) a[qq][I] = A[(size_t)kk * g.lda + r0 + 16 * I + lr];
                else a[qq][I] = A[(size_t)(r0 + 16 * I + lr) * g.lda + kk];
            }
#pragma unroll
            for (int J = 0; J < 2; ++J) b[qq][J] = Bm[(size_t)kk * g.ldb + c0 + 16 * J + lr];
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const double ar = a[qq][I].x, ai = CONJT ? -a[qq][I].y : a[qq][I].y, as = ar + ai;
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    const double br = b[qq][J].x, bi = b[qq][J].y;
                    t1[I][J] = GMFMA(ar, br, t1[I][J]);
                    t2[I][J] = GMFMA(ai, bi, t2[I][J]);
                    t3[I][J] = GMFMA(as, br + bi, t3[I][J]);
                }
            }
    }
    // combine the 3-multiplication accumulators (linear, so partial K ranges simply add)
    gd4 re[2][2], im[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) { re[I][J] = t1[I][J] - t2[I][J]; im[I][J] = t3[I][J] - t1[I][J] - t2[I][J]; }
    if (SK > 1) {
        if (wv > 0) {
            double* dst = sk_part + (size_t)(wv - 1) * 2048 + lane;             // [32 values][64 lanes]
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[(((I * 2 + J) * 4 + r) * 2 + 0) * 64] = re[I][J][r];
                        dst[(((I * 2 + J) * 4 + r) * 2 + 1) * 64] = im[I][J][r];
                    }
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 1; w < SK; ++w) {
            const double* src = sk_part + (size_t)(w - 1) * 2048 + lane;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        re[I][J][r] += src[(((I * 2 + J) * 4 + r) * 2 + 0) * 64];
                        im[I][J][r] += src[(((I * 2 + J) * 4 + r) * 2 + 1) * 64];
                    }
        }
    }
    // D layout: register r of tile (I, J) <-> (row = r0 + 16I + (lane>>4) + 4r, col = c0 + 16J + (lane&15))
    double part = 0.0;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 16 * I + lk + 4 * r, col = c0 + 16 * J + lr;
                const double vre = re[I][J][r], vim = im[I][J][r];
                if (EPI == 0) {
                    cplx v = cmake(g.alpha * vre, g.alpha * vim);
                    if (g.E) {
                        const cplx e = g.E[(size_t)bt * g.sE + (size_t)row * g.lde + col];
                        v.x = fma(g.beta, e.x, v.x); v.y = fma(g.beta, e.y, v.y);
                    }
                    if (row == col) v.x += g.gamma;
                    g.C[(size_t)bhi * g.sC2 + (size_t)blo * g.sC + (size_t)row * g.ldc + col] = v;
                } else if (EPI == 1) {
                    const cplx l = g.L[(size_t)bt * g.sL + (size_t)row * g.ldl + col];
                    part = fma(l.x, vre, part); part = fma(l.y, vim, part);      // Re(conj(l) * y)
                }
            }
    if (EPI == 1) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) g.partial[(size_t)bt * g.partial_stride + g.partial_offset + tm] = part;
    }
    if (EPI == 2) {                                            // Re sum_rows conj(L[row][col]) * (A*B)[row][col] for every column
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            double pc = 0.0;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r0 + 16 * I + lk + 4 * r, col = c0 + 16 * J + lr;
                    const cplx l = g.L[(size_t)bhi * g.sL2 + (size_t)blo * g.sL + (size_t)row * g.ldl + col];
                    pc = fma(l.x, re[I][J][r], pc); pc = fma(l.y, im[I][J][r], pc);
                }
            pc += __shfl_xor(pc, 16, 64);
            pc += __shfl_xor(pc, 32, 64);
            if (lk == 0) g.partial[(size_t)bt * g.partial_stride + g.partial_offset + (size_t)tm * g.ldp + c0 + 16 * J + lr] = pc;
        }
    }
}

// A_t = (H0' + sum_k u_k H_k') / 2^s for every (seed, slice), padded N x N           tensorflow_state.py:30-33
// Slices are padded to SP = NC*S per seed; a padded slice gets A = 0, i.e. K = I exactly.
__global__ void __launch_bounds__(256) k_gemm_assemble(QocDev d, const cplx* __restrict__ HsP, cplx* __restrict__ Aout, int N, int SP, int sq) {
    const size_t NN = (size_t)N * N;
    const size_t total = (size_t)d.B * SP * NN;
    const double inv = 1.0 / (double)(1 << sq);
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
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
// copy the thin block of Y (columns N..N+31) into Psibnd[b][c]
__global__ void __launch_bounds__(256) k_gemm_take_bnd(QocDev d, const cplx* __restrict__ Y, cplx* __restrict__ Psibnd, int N, int NC, int c, int xw) {
    const int ld = xw + QOC_TW;
    const size_t per = (size_t)N * QOC_TW;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)d.B * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bb = o / per, e = o - bb * per;
        const int row = (int)(e / QOC_TW), col = (int)(e - (size_t)row * QOC_TW);
        Psibnd[(bb * NC + c) * per + e] = Y[bb * (size_t)N * ld + (size_t)row * ld + xw + col];
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
__global__ void __launch_bounds__(256) k_gemm_take_final(QocDev d, const cplx* __restrict__ Y, int N) {
    __shared__ double red[8];
    const int b = blockIdx.x, ld = N + QOC_TW, n = d.n;
    const cplx* X = Y + (size_t)b * N * ld;
    double part = 0.0;
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        cplx rs = cmake(0.0, 0.0);
        for (int a = 0; a < n; ++a) rs = cadd(rs, X[(size_t)c * ld + a]);
        part += rs.x * rs.x + rs.y * rs.y;
    }
    for (int o = threadIdx.x; o < n * n; o += blockDim.x) d.Xfinal[(size_t)b * n * n + o] = X[(size_t)(o / n) * ld + (o % n)];
    const double tot = block_sum(part, red);
    if (threadIdx.x == 0) d.uscale[b] = tot / (double)n;
}
// sources SrcP[b][tau] (padded thin, tau = 0..SP-1; zero for tau = 0 and tau > steps) and the costate at the END of the
// last chunk Ebnd[b][NC-1]: -(2/m^2) z W, plus S_steps when there is no padded slice to add it through the recursion
__global__ void __launch_bounds__(256) k_gemm_sources(QocDev d, cplx* __restrict__ SrcP, cplx* __restrict__ Ebnd, int N, int SP, int NC) {
    const size_t per = (size_t)N * QOC_TW;
    const bool need_src = d.n_forb > 0 || d.has_speed;
    const size_t total = (size_t)d.B * (need_src ? SP : 1) * per;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bt = o / per, e = o - bt * per;
        const int per_seed = need_src ? SP : 1;
        const int b = (int)(bt / per_seed), tau = (int)(bt - (size_t)b * per_seed);
        const int row = (int)(e / QOC_TW), col = (int)(e - (size_t)row * QOC_TW);
        const bool valid = row < d.n && col < d.m;
        if (need_src) {
            cplx s = cmake(0.0, 0.0);
            if (valid && tau >= 1 && tau <= d.steps) s = source_at(d, b, tau, row, col);
            SrcP[o] = s;
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
// LamP[b][(c+1)S-1] = (Ebnd ? Ebnd[b][c] : 0): costate at the end of every chunk
__global__ void __launch_bounds__(256) k_gemm_set_chunk_ends(QocDev d, cplx* __restrict__ LamP, const cplx* __restrict__ Ebnd, int N, int S, int NC) {
    const size_t per = (size_t)N * QOC_TW;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)d.B * NC * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t bc = o / per, e = o - bc * per;
        LamP[(bc * S + (S - 1)) * per + e] = Ebnd ? Ebnd[o] : cmake(0.0, 0.0);
    }
}
// dst[b] = src[b] for B matrices of NN elements (odd element of a product-tree level moves up unchanged)
__global__ void __launch_bounds__(256) k_gemm_copy_mats(cplx* __restrict__ dst, long long sD, const cplx* __restrict__ src, long long sS, int B, int NN) {
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

// ---- fused per-slice exponential for N <= 64 ---------------------------------------------------------------------------
// One workgroup per (seed, slice): A_t is assembled into LDS, the Paterson-Stockmeyer Taylor polynomial and the squarings
// run as MFMA products whose operands are read from two LDS-resident matrices (row stride N+1 elements: conflict-free for
// both the left-operand pattern, 16 rows x 1 column, and the right-operand pattern, 1 row x 16 columns), accumulators and
// the per-wave block of A stay in registers, and only K_t is written to HBM.  The launch-per-product route streams three
// B*SP*N*N buffers through HBM per product (7-11 products); this kernel writes one.
// Wave w owns tile row I = w / (N/32) and the tile-column pair Jp = w % (N/32) (2 tiles of 16x16, sharing the left operand).
struct ExpmCoef { double c[24]; };

template <int N>
__device__ __forceinline__ void lds_mm(const cplx* __restrict__ L, const cplx* __restrict__ R, int I, int Jp, int lane,
                                       gd4 (&re)[2], gd4 (&im)[2]) {
    constexpr int LD = N + 1;
    gd4 t1[2], t2[2], t3[2];
#pragma unroll
    for (int J = 0; J < 2; ++J) { t1[J] = (gd4){0, 0, 0, 0}; t2[J] = (gd4){0, 0, 0, 0}; t3[J] = (gd4){0, 0, 0, 0}; }
    const int lr = lane & 15, lk = lane >> 4;
    const cplx* lp = L + (16 * I + lr) * LD + lk;
    const cplx* rp = R + lk * LD + 32 * Jp + lr;
#pragma unroll 4
    for (int kk = 0; kk < N / 4; ++kk) {
        const cplx a = lp[4 * kk];
        const cplx b0 = rp[4 * kk * LD], b1 = rp[4 * kk * LD + 16];
        const double as = a.x + a.y;
        t1[0] = GMFMA(a.x, b0.x, t1[0]); t2[0] = GMFMA(a.y, b0.y, t2[0]); t3[0] = GMFMA(as, b0.x + b0.y, t3[0]);
        t1[1] = GMFMA(a.x, b1.x, t1[1]); t2[1] = GMFMA(a.y, b1.y, t2[1]); t3[1] = GMFMA(as, b1.x + b1.y, t3[1]);
    }
#pragma unroll
    for (int J = 0; J < 2; ++J) { re[J] = t1[J] - t2[J]; im[J] = t3[J] - t1[J] - t2[J]; }
}

template <int N>
__global__ void __launch_bounds__((N / 16) * (N / 16) * 32) k_gemm_expm_fused(QocDev d, const cplx* __restrict__ HsP, cplx* __restrict__ Kout,
                                                                               int SP, int deg, int nsq, ExpmCoef cf) {
    constexpr int LD = N + 1, NT = (N / 16) * (N / 16) * 32, NN = N * N;
    extern __shared__ __attribute__((aligned(16))) cplx ex_lds[];
    cplx* X = ex_lds;                   // A, then S / M
    cplx* Y = ex_lds + N * LD;          // A2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = wv / (N / 32), Jp = wv % (N / 32);
    const int lr = lane & 15, lk = lane >> 4;
    const int b = blockIdx.x / SP, t = blockIdx.x - b * SP;
    cplx* Kt = Kout + (size_t)blockIdx.x * NN;
    // D-layout coordinates of this lane's 2 x 4 accumulator elements
    auto drow = [&](int r) { return 16 * I + lk + 4 * r; };
    auto dcol = [&](int J) { return 32 * Jp + 16 * J + lr; };
    if (t >= d.steps) {                                           // padded slice: K = I exactly
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) Kt[(size_t)drow(r) * N + dcol(J)] = cmake(drow(r) == dcol(J) ? 1.0 : 0.0, 0.0);
        return;
    }
    // ---- A_t = (H0' + sum_k u_k H_k') / 2^s into X                                              tensorflow_state.py:30-33
    const double inv = 1.0 / (double)(1 << nsq);
    {
        constexpr int PER = NN / NT;
        cplx acc[PER];
#pragma unroll
        for (int x = 0; x < PER; ++x) acc[x] = cscale(HsP[tid + NT * x], inv);
        for (int kk = 0; kk < d.k; ++kk) {
            const double cu = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv;
            const cplx* H = HsP + (size_t)(kk + 1) * NN;
#pragma unroll
            for (int x = 0; x < PER; ++x) {
                const cplx hv = H[tid + NT * x];
                acc[x].x = fma(cu, hv.x, acc[x].x); acc[x].y = fma(cu, hv.y, acc[x].y);
            }
        }
#pragma unroll
        for (int x = 0; x < PER; ++x) { const int e = tid + NT * x; X[(e / N) * LD + (e % N)] = acc[x]; }
    }
    __syncthreads();
    cplx ablk[2][4];
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) ablk[J][r] = X[drow(r) * LD + dcol(J)];
    gd4 re[2], im[2];
    auto put = [&](cplx* dst) {
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[drow(r) * LD + dcol(J)] = cmake(re[J][r], im[J][r]);
    };
    // re/im <- c0*I + c1*A + (re/im already holding a product, scaled by 1) : the Horner addend in D layout
    auto add_b = [&](double c0, double c1) {
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                re[J][r] = fma(c1, ablk[J][r].x, re[J][r]) + (drow(r) == dcol(J) ? c0 : 0.0);
                im[J][r] = fma(c1, ablk[J][r].y, im[J][r]);
            }
    };
    const int mm = deg >> 1;
    const bool even = (deg & 1) == 0;
    if (deg >= 6) {
        // Paterson-Stockmeyer with cubes: P = B_0 + A3 (B_1 + A3 (B_2 + ...)), B_i = c_{3i} I + c_{3i+1} A + c_{3i+2} A2.
        // Degree 9 (state transfer, T = 10): A2, A3 + 2 Horner products = 4 instead of 5 with squares; never more.
        gd4 a2r[2], a2i[2];
        lds_mm<N>(X, X, I, Jp, lane, re, im);                     // A2 = A*A
        put(Y);
#pragma unroll
        for (int J = 0; J < 2; ++J) { a2r[J] = re[J]; a2i[J] = im[J]; }
        __syncthreads();
        lds_mm<N>(Y, X, I, Jp, lane, re, im);                     // A3 = A2*A
        __syncthreads();                                          // every wave is done reading A (X) and A2 (Y)
        put(Y);                                                   // Y = A3 from here on
        auto coef = [&](int j) { return j <= deg ? cf.c[j] : 0.0; };
        auto add_blk = [&](int i) {                               // re/im += B_i
            const double c0 = coef(3 * i), c1 = coef(3 * i + 1), c2 = coef(3 * i + 2);
#pragma unroll
            for (int J = 0; J < 2; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    re[J][r] += fma(c2, a2r[J][r], c1 * ablk[J][r].x) + (drow(r) == dcol(J) ? c0 : 0.0);
                    im[J][r] += fma(c2, a2i[J][r], c1 * ablk[J][r].y);
                }
        };
        const int nb = deg / 3;
        int first;
        if (deg % 3 == 0) {                                       // top block is the scalar c_deg: fold c_deg*A3 into B_{nb-1}
#pragma unroll
            for (int J = 0; J < 2; ++J) { re[J] = re[J] * cf.c[deg]; im[J] = im[J] * cf.c[deg]; }
            add_blk(nb - 1);
            first = nb - 2;
        } else {
#pragma unroll
            for (int J = 0; J < 2; ++J) { re[J] = (gd4){0, 0, 0, 0}; im[J] = (gd4){0, 0, 0, 0}; }
            add_blk(nb);
            first = nb - 1;
        }
        put(X);
        __syncthreads();
        for (int i = first; i >= 0; --i) {                        // S <- B_i + A3*S
            lds_mm<N>(Y, X, I, Jp, lane, re, im);
            add_blk(i);
            __syncthreads();
            if (i > 0 || nsq > 0) { put(X); __syncthreads(); }
        }
    } else if (deg >= 2) {
        lds_mm<N>(X, X, I, Jp, lane, re, im);                     // A2 = A*A
        __syncthreads();                                          // every wave is done reading A from X
        put(Y);
        if (even) {                                               // S = c_{2m-2} I + c_{2m-1} A + c_T A2
#pragma unroll
            for (int J = 0; J < 2; ++J) { re[J] = re[J] * cf.c[deg]; im[J] = im[J] * cf.c[deg]; }
            add_b(cf.c[2 * mm - 2], cf.c[2 * mm - 1]);
        } else {                                                  // S = c_{2m} I + c_{2m+1} A
#pragma unroll
            for (int J = 0; J < 2; ++J) { re[J] = (gd4){0, 0, 0, 0}; im[J] = (gd4){0, 0, 0, 0}; }
            add_b(cf.c[2 * mm], cf.c[2 * mm + 1]);
        }
        put(X);
        __syncthreads();
        for (int i = (even ? mm - 2 : mm - 1); i >= 0; --i) {     // S <- c_{2i} I + c_{2i+1} A + A2*S
            lds_mm<N>(Y, X, I, Jp, lane, re, im);
            add_b(cf.c[2 * i], cf.c[2 * i + 1]);
            __syncthreads();
            if (i > 0 || nsq > 0) { put(X); __syncthreads(); }
        }
    } else {
#pragma unroll
        for (int J = 0; J < 2; ++J) { re[J] = (gd4){0, 0, 0, 0}; im[J] = (gd4){0, 0, 0, 0}; }
        add_b(1.0, deg >= 1 ? 1.0 : 0.0);
        __syncthreads();
        if (nsq > 0) { put(X); __syncthreads(); }
    }
    for (int sq = 0; sq < nsq; ++sq) {                            // M <- M M                       tensorflow_state.py:43-44
        lds_mm<N>(X, X, I, Jp, lane, re, im);
        __syncthreads();
        if (sq + 1 < nsq) { put(X); __syncthreads(); }
    }
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) Kt[(size_t)drow(r) * N + dcol(J)] = cmake(re[J][r], im[J][r]);
}

// ---- persistent thin chains (N <= 64, m <= 8) ---------------------------------------------------------------------
// y <- op(K_j) y + E_j for `len` consecutive matrices, one workgroup per chain, y in LDS.  Launch-per-step chains cost
// ~5 us of launch latency per step; here a step costs N*N*m complex MACs on the VALU (fp64 FMA rate ~ MFMA rate on
// gfx950, and no padding of m to an MFMA tile) with K_j fetched two steps ahead into registers.
// The steady-state loop is one basic block (unconditional clamped prefetch, E always loaded -- from a zero buffer when
// there is no addend --, every finished value has exactly one owner lane): with conditional loads or stores in the loop hipcc's
// s_waitcnt placement has to assume the worst path and waits for the loads it has just issued.
struct ChainArgs {
    const cplx* K; long long sKb, sKc, sKs;     // matrix of step j: K + b*sKb + c*sKc + j*sKs  (elements; sKs may be negative)
    const cplx* X0; long long sXb, sXc;         // initial thin vectors (nullptr = zeros)
    const cplx* E; long long sEb, sEc, sEs;     // addend per step (a zero buffer with zero strides when there is none)
    cplx* Out; long long sOb, sOc, sOs; int ldO; // output per step (HAS_OUT): Out + b*sOb + c*sOc + j*sOs + row*ldO + jv
    cplx* Fin; long long sFb, sFc;              // optional final state
    int CI;                                     // chains per seed (blockIdx.x = b*CI + c)
    int len, m;
    int store_initial;                          // also store y0 at Out - sOs
    int nterms; double sign;                    // k_gemm_taylor_chain: y <- sum_{j<nterms} (sign*K)^j y / j!  (+ E)
};

// v from lane (l ^ OFF), OFF in {1, 2, 4, 8}, as DPP moves on the VALU (quad_perm; xor 4 = row_half_mirror then quad_perm
// [3,2,1,0]; xor 8 = row_ror:8) instead of ds_bpermute round trips through the LDS crossbar: the chain step is a dependent sequence, and
// three crossbar latencies per step were a tenth of it.
template <int OFF>
__device__ __forceinline__ double dpp_xor(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if constexpr (OFF == 1) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, true);
    } else if constexpr (OFF == 2) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, true);
    } else if constexpr (OFF == 4) {
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, true);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x1B, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x1B, 0xF, 0xF, true);
    } else {
        static_assert(OFF == 8, "dpp_xor: lane distance 1, 2, 4 or 8");
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xF, 0xF, true);   // row_ror:8
    }
    return __hiloint2double(hi, lo);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for the K/E prefetch of
// two steps ahead (and the output stores) at every step of a chain; the chains exchange data through LDS alone, and hipcc
// still places the vmcnt wait for each prefetched register stage before its first use.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Butterfly over the LPR lanes that share a result row (forward chain mapping).  While more than SPL values are alive the
// halves are exchanged (reduce-scatter: the lane with the bit set keeps the upper half), afterwards plain xor all-reduce.
// Compile-time recursion keeps every register index static.
template <int ALIVE, int OFF, int SPL, int MVT>
__device__ __forceinline__ void chain_butterfly(cplx (&acc)[MVT], int q) {
    if constexpr (OFF >= 1) {
        if constexpr (ALIVE > SPL) {
            constexpr int half = ALIVE / 2;
            const bool up = (q & OFF) != 0;
#pragma unroll
            for (int x = 0; x < half; ++x) {
                const cplx send = up ? acc[x] : acc[x + half];
                const cplx keep = up ? acc[x + half] : acc[x];
                acc[x].x = keep.x + dpp_xor<OFF>(send.x);
                acc[x].y = keep.y + dpp_xor<OFF>(send.y);
            }
            chain_butterfly<half, OFF / 2, SPL, MVT>(acc, q);
        } else {
#pragma unroll
            for (int x = 0; x < SPL; ++x) {
                acc[x].x += dpp_xor<OFF>(acc[x].x);
                acc[x].y += dpp_xor<OFF>(acc[x].y);
            }
            chain_butterfly<ALIVE, OFF / 2, SPL, MVT>(acc, q);
        }
    }
}

// y <- K_j^H y + E_j (backward chains).  Lane <-> column i of K, so that a wave reads whole rows (the 4 lanes of a quad must
// stay on 64 contiguous bytes: the texture-address unit serialises a quad that touches 4 cache lines, and a transposed read
// with a row-per-thread mapping was 2x slower per step); wave w owns rows (4e + w)*RPI + h, RPI = 64/N; x[r] is a broadcast
// LDS read; the 4*RPI partial rows meet in LDS (one extra barrier per step) and thread (w, h, i) finishes -- adds the
// source, writes LDS, stores -- the slots jv = sg + s*NSL, sg = (w*RPI + h) % NSL, NSL = min(MV, 4*RPI).
template <int N, int MV, bool HAS_OUT>
__global__ void __launch_bounds__(256) k_gemm_chain_adj(ChainArgs a) {
    constexpr int EL = N * N / 256, RPI = 64 / N;
    constexpr int NSL = MV < 4 * RPI ? MV : 4 * RPI;
    constexpr int SPL = MV / NSL;
    __shared__ __attribute__((aligned(16))) cplx y[2][N * MV];
    __shared__ __attribute__((aligned(16))) cplx part[4 * RPI * N * MV];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int i = lane % N, h = lane / N;
    const int sg = (wv * RPI + h) % NSL;
    auto slot = [&](int s) { return sg + s * NSL; };
    const int b = blockIdx.x / a.CI, c = blockIdx.x - b * a.CI;
    const cplx* Kp = a.K + b * a.sKb + c * a.sKc;
    const cplx* Ep = a.E + b * a.sEb + c * a.sEc + (size_t)i * QOC_TW;
    cplx* Op = HAS_OUT ? a.Out + b * a.sOb + c * a.sOc + (size_t)i * a.ldO : nullptr;
    cplx yfin[SPL];
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) yfin[sl] = cmake(0.0, 0.0);
    if (a.X0) {
        const cplx* x = a.X0 + b * a.sXb + c * a.sXc + (size_t)i * QOC_TW;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) yfin[sl] = x[slot(sl)];
    }
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) y[0][i * MV + slot(sl)] = yfin[sl];
    if (HAS_OUT && a.store_initial) {
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) (Op - a.sOs)[slot(sl)] = yfin[sl];
    }
    const int last = a.len - 1;
    auto load = [&](cplx (&kd)[EL], cplx (&ed)[SPL], int j) {
        const int jc = min(j, last);
        const cplx* Kj = Kp + (long long)jc * a.sKs;
#pragma unroll
        for (int e = 0; e < EL; ++e) kd[e] = Kj[(size_t)((4 * e + wv) * RPI + h) * N + i];
        const cplx* ej = Ep + (long long)jc * a.sEs;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) ed[sl] = ej[slot(sl)];
    };
    int cur = 0;
    auto step = [&](int j, const cplx (&ku)[EL], const cplx (&eu)[SPL]) {
        cplx acc[MV];
#pragma unroll
        for (int jv = 0; jv < MV; ++jv) acc[jv] = cmake(0.0, 0.0);
#pragma unroll
        for (int e = 0; e < EL; ++e) {
            const int r = (4 * e + wv) * RPI + h;
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) cfma_conj(acc[jv], ku[e], y[cur][r * MV + jv]);
        }
#pragma unroll
        for (int jv = 0; jv < MV; ++jv) part[((wv * RPI + h) * N + i) * MV + jv] = acc[jv];   // 4*RPI partial rows per column
        lds_barrier();
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) {
            const int jv = slot(sl);
            cplx t = part[i * MV + jv];
#pragma unroll
            for (int w = 1; w < 4 * RPI; ++w) t = cadd(t, part[(w * N + i) * MV + jv]);
            yfin[sl] = cadd(t, eu[sl]);
            y[cur ^ 1][i * MV + jv] = yfin[sl];
        }
        if (HAS_OUT) {
            cplx* oj = Op + (long long)j * a.sOs;
#pragma unroll
            for (int sl = 0; sl < SPL; ++sl) oj[slot(sl)] = yfin[sl];
        }
        lds_barrier();
        cur ^= 1;
    };
    if (a.len > 0) {
        // three register stages used round-robin by a 3x unrolled loop (rotating them with copies would make every
        // iteration wait for the newest load)
        cplx k0[EL], k1[EL], k2[EL], e0[SPL], e1[SPL], e2[SPL];
        load(k0, e0, 0);
        load(k1, e1, 1);
        lds_barrier();
        int j = 0;
        for (; j + 3 <= a.len; j += 3) {
            load(k2, e2, j + 2); step(j, k0, e0);
            load(k0, e0, j + 3); step(j + 1, k1, e1);
            load(k1, e1, j + 4); step(j + 2, k2, e2);
        }
        if (j < a.len) step(j, k0, e0);
        if (j + 1 < a.len) step(j + 1, k1, e1);
    } else {
        lds_barrier();
    }
    if (a.Fin) {
        cplx* f = a.Fin + b * a.sFb + c * a.sFc + (size_t)i * QOC_TW;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) f[slot(sl)] = yfin[sl];
    }
}

// ---- register-blocked forward mat-vec mapping (k_gemm_chain_fwd, k_gemm_taylor_chain) ---------------------------------
// Thread (g, c) = (tid / 16, tid % 16) owns the R x R block rows R*g + rr, columns c + 16*cc of K (R = N/16): a 16-lane DPP
// row reads 256 contiguous bytes per load, a thread reads only R entries of the vector per slot (the row-per-thread mapping
// re-read 16 at N = 64: 64 KB of LDS traffic per mat-vec, which bound the step) and the R*MV partial sums are combined by a
// reduce-scatter butterfly over the 16 lanes (DPP), after which lane c holds the finished values x = base + s, s < SPLB,
// of its row group in (row, slot)-major order; addend, LDS write and output store are done by the owner of each value.
template <int N, int MV>
struct BlockMap {
    static constexpr int R = N / 16, EL = R * R, V = R * MV;
    static constexpr int NSLB = V < 16 ? V : 16, SPLB = V / NSLB;
    int g, c, base;
    __device__ __forceinline__ BlockMap(int tid) : g(tid >> 4), c(tid & 15), base(((tid & 15) / (16 / NSLB)) * SPLB) {}
    __device__ __forceinline__ int row(int s) const { return R * g + (base + s) / MV; }
    __device__ __forceinline__ int slot(int s) const { return (base + s) % MV; }
    __device__ __forceinline__ void load(cplx (&kd)[EL], const cplx* Kj) const {
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int cc = 0; cc < R; ++cc) kd[rr * R + cc] = Kj[(size_t)(R * g + rr) * N + c + 16 * cc];
    }
    // acc[rr*MV + jv] = sum_cc K[rr][cc] * v[c + 16 cc][jv], then the 16-lane reduce-scatter: acc[0..SPLB) are this lane's values
    __device__ __forceinline__ void matvec(const cplx (&ku)[EL], const cplx* __restrict__ v, cplx (&acc)[V]) const {
#pragma unroll
        for (int x = 0; x < V; ++x) acc[x] = cmake(0.0, 0.0);
#pragma unroll
        for (int cc = 0; cc < R; ++cc) {
            cplx vv[MV];
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) vv[jv] = v[(c + 16 * cc) * MV + jv];
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int jv = 0; jv < MV; ++jv) cfma(acc[rr * MV + jv], ku[rr * R + cc], vv[jv]);
        }
        chain_butterfly<V, 8, SPLB, V>(acc, c);
    }
};

// Row-per-thread mapping with the same interface: thread (i, q) owns row i and the columns LPR*e + q.  Faster than the
// blocked mapping at N = 32 (4 vector reads per slot either way, a 3-level butterfly instead of 4); measured per 1000-slice
// direct iteration: N = 32: 4.65 (row) vs 4.81 ms (blocked); N = 64: 13.0 (row) vs 10.0 ms (blocked).
template <int N, int MV>
struct RowMap {
    static constexpr int LPR = 256 / N, EL = N / LPR, V = MV;
    static constexpr int NSLB = MV < LPR ? MV : LPR, SPLB = MV / NSLB;
    int i, q, base;
    __device__ __forceinline__ RowMap(int tid) : i(tid / LPR), q(tid % LPR), base(((tid % LPR) / (LPR / NSLB)) * SPLB) {}
    __device__ __forceinline__ int row(int) const { return i; }
    __device__ __forceinline__ int slot(int s) const { return base + s; }
    __device__ __forceinline__ void load(cplx (&kd)[EL], const cplx* Kj) const {
#pragma unroll
        for (int e = 0; e < EL; ++e) kd[e] = Kj[(size_t)i * N + LPR * e + q];
    }
    __device__ __forceinline__ void matvec(const cplx (&ku)[EL], const cplx* __restrict__ v, cplx (&acc)[V]) const {
#pragma unroll
        for (int jv = 0; jv < MV; ++jv) acc[jv] = cmake(0.0, 0.0);
#pragma unroll
        for (int e = 0; e < EL; ++e)
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) cfma(acc[jv], ku[e], v[(LPR * e + q) * MV + jv]);
        chain_butterfly<MV, LPR / 2, SPLB, MV>(acc, q);
    }
};
template <int N, int MV> struct FwdMap { using type = BlockMap<N, MV>; };
template <int MV> struct FwdMap<32, MV> { using type = RowMap<32, MV>; };

// y <- K_j y + E_j (forward chains) with the mapping FwdMap picks for N; same pipeline as k_gemm_chain_adj
template <int N, int MV, bool HAS_OUT>
__global__ void __launch_bounds__(256) k_gemm_chain_fwd(ChainArgs a) {
    using BM = typename FwdMap<N, MV>::type;
    constexpr int EL = BM::EL, SPL = BM::SPLB, V = BM::V;
    __shared__ __attribute__((aligned(16))) cplx y[2][N * MV];
    const BM bm(threadIdx.x);
    const int b = blockIdx.x / a.CI, c = blockIdx.x - b * a.CI;
    const cplx* Kp = a.K + b * a.sKb + c * a.sKc;
    const cplx* Ep = a.E + b * a.sEb + c * a.sEc;
    cplx* Op = HAS_OUT ? a.Out + b * a.sOb + c * a.sOc : nullptr;
    int thin_off[SPL], out_off[SPL], lds_off[SPL];
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) {
        thin_off[sl] = bm.row(sl) * QOC_TW + bm.slot(sl);
        out_off[sl] = bm.row(sl) * a.ldO + bm.slot(sl);
        lds_off[sl] = bm.row(sl) * MV + bm.slot(sl);
    }
    cplx yfin[SPL];
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) yfin[sl] = cmake(0.0, 0.0);
    if (a.X0) {
        const cplx* x = a.X0 + b * a.sXb + c * a.sXc;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) yfin[sl] = x[thin_off[sl]];
    }
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) y[0][lds_off[sl]] = yfin[sl];
    if (HAS_OUT && a.store_initial) {
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) (Op - a.sOs)[out_off[sl]] = yfin[sl];
    }
    const int last = a.len - 1;
    auto load = [&](cplx (&kd)[EL], cplx (&ed)[SPL], int j) {
        const int jc = min(j, last);
        bm.load(kd, Kp + (long long)jc * a.sKs);
        const cplx* ej = Ep + (long long)jc * a.sEs;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) ed[sl] = ej[thin_off[sl]];
    };
    int cur = 0;
    auto step = [&](int j, const cplx (&ku)[EL], const cplx (&eu)[SPL]) {
        cplx acc[V];
        bm.matvec(ku, y[cur], acc);
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) {
            yfin[sl] = cadd(acc[sl], eu[sl]);
            y[cur ^ 1][lds_off[sl]] = yfin[sl];
        }
        if (HAS_OUT) {
            cplx* oj = Op + (long long)j * a.sOs;
#pragma unroll
            for (int sl = 0; sl < SPL; ++sl) oj[out_off[sl]] = yfin[sl];
        }
        lds_barrier();
        cur ^= 1;
    };
    if (a.len > 0) {
        cplx k0[EL], k1[EL], k2[EL], e0[SPL], e1[SPL], e2[SPL];
        load(k0, e0, 0);
        load(k1, e1, 1);
        lds_barrier();
        int j = 0;
        for (; j + 3 <= a.len; j += 3) {
            load(k2, e2, j + 2); step(j, k0, e0);
            load(k0, e0, j + 3); step(j + 1, k1, e1);
            load(k1, e1, j + 4); step(j + 2, k2, e2);
        }
        if (j < a.len) step(j, k0, e0);
        if (j + 1 < a.len) step(j + 1, k1, e1);
    } else {
        lds_barrier();
    }
    if (a.Fin) {
        cplx* f = a.Fin + b * a.sFb + c * a.sFc;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) f[thin_off[sl]] = yfin[sl];
    }
}

// State transfer without propagators: psi <- sum_{j<T} (sign*B_t)^j psi / j! (+ E_t), one workgroup per seed walking all
// slices (tensorflow_state.py:88-96 forward, :118-131 backward with sign = -1 -- no anti-Hermiticity assumed).  Same mapping
// and prefetch structure as k_gemm_chain_fwd; a step is T-1 dependent mat-vecs on the register-resident B_t (the
// generator was assembled for all slices by k_gemm_assemble, so the chain streams 1 matrix per slice instead of k+1).
template <int N, int MV>
__global__ void __launch_bounds__(256) k_gemm_taylor_chain(ChainArgs a) {
    using BM = typename FwdMap<N, MV>::type;
    constexpr int EL = BM::EL, SPL = BM::SPLB, V = BM::V;
    __shared__ __attribute__((aligned(16))) cplx y[2][N * MV];
    const BM bm(threadIdx.x);
    const int b = blockIdx.x;
    const cplx* Kp = a.K + b * a.sKb;
    const cplx* Ep = a.E + b * a.sEb;
    cplx* Op = a.Out + b * a.sOb;
    int thin_off[SPL], out_off[SPL], lds_off[SPL];
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) {
        thin_off[sl] = bm.row(sl) * QOC_TW + bm.slot(sl);
        out_off[sl] = bm.row(sl) * a.ldO + bm.slot(sl);
        lds_off[sl] = bm.row(sl) * MV + bm.slot(sl);
    }
    cplx yfin[SPL];
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) yfin[sl] = cmake(0.0, 0.0);
    if (a.X0) {
        const cplx* x = a.X0 + b * a.sXb;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) yfin[sl] = x[thin_off[sl]];
    }
#pragma unroll
    for (int sl = 0; sl < SPL; ++sl) y[0][lds_off[sl]] = yfin[sl];
    if (a.store_initial) {
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) (Op - a.sOs)[out_off[sl]] = yfin[sl];
    }
    const int last = a.len - 1;
    auto load = [&](cplx (&kd)[EL], cplx (&ed)[SPL], int j) {
        const int jc = min(j, last);
        bm.load(kd, Kp + (long long)jc * a.sKs);
        const cplx* ej = Ep + (long long)jc * a.sEs;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) ed[sl] = ej[thin_off[sl]];
    };
    int cur = 0;
    auto step = [&](int j, const cplx (&ku)[EL], const cplx (&eu)[SPL]) {
        cplx out[SPL];
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) out[sl] = yfin[sl];
        double fact = 1.0;
        for (int ii = 1; ii < a.nterms; ++ii) {
            cplx acc[V];
            bm.matvec(ku, y[cur], acc);
            fact *= (double)ii;
            const double inv = 1.0 / fact;
            const bool lastterm = ii + 1 == a.nterms;
#pragma unroll
            for (int sl = 0; sl < SPL; ++sl) {
                const cplx w = cscale(acc[sl], a.sign);                       // psi_n = (sign*B) psi_n            :94 / :130
                out[sl].x = fma(w.x, inv, out[sl].x); out[sl].y = fma(w.y, inv, out[sl].y);   // += psi_n / factorial   :95 / :131
                // the last term is needed by nobody else: the buffer takes the new state (+ addend) instead
                y[cur ^ 1][lds_off[sl]] = lastterm ? cadd(out[sl], eu[sl]) : w;
            }
            lds_barrier();
            cur ^= 1;
        }
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) yfin[sl] = cadd(out[sl], eu[sl]);
        if (a.nterms <= 1) {                                                   // T = 1: psi unchanged (+ addend)
#pragma unroll
            for (int sl = 0; sl < SPL; ++sl) y[cur ^ 1][lds_off[sl]] = yfin[sl];
            lds_barrier();
            cur ^= 1;
        }
        cplx* oj = Op + (long long)j * a.sOs;
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) oj[out_off[sl]] = yfin[sl];
    };
    if (a.len > 0) {
        cplx k0[EL], k1[EL], k2[EL], e0[SPL], e1[SPL], e2[SPL];
        load(k0, e0, 0);
        load(k1, e1, 1);
        lds_barrier();
        int j = 0;
        for (; j + 3 <= a.len; j += 3) {
            load(k2, e2, j + 2); step(j, k0, e0);
            load(k0, e0, j + 3); step(j + 1, k1, e1);
            load(k1, e1, j + 4); step(j + 2, k2, e2);
        }
        if (j < a.len) step(j, k0, e0);
        if (j + 1 < a.len) step(j + 1, k1, e1);
    }
}

template <int N>
static inline void qoc_taylor_chain_launch_n(const ChainArgs& a, int blocks, hipStream_t s) {
    const int mv = a.m <= 1 ? 1 : (a.m <= 2 ? 2 : (a.m <= 4 ? 4 : 8));
    if (mv == 1) hipLaunchKernelGGL((k_gemm_taylor_chain<N, 1>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 2) hipLaunchKernelGGL((k_gemm_taylor_chain<N, 2>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 4) hipLaunchKernelGGL((k_gemm_taylor_chain<N, 4>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm_taylor_chain<N, 8>), dim3(blocks), dim3(256), 0, s, a);
}
static inline void qoc_taylor_chain_launch(int N, ChainArgs a, const cplx* zeros, int blocks, hipStream_t s) {
    if (!a.E) { a.E = zeros; a.sEb = a.sEc = a.sEs = 0; }
    if (N == 32) qoc_taylor_chain_launch_n<32>(a, blocks, s); else qoc_taylor_chain_launch_n<64>(a, blocks, s);
}

template <int N, bool HAS_OUT>
static inline void qoc_chain_adj_launch_n(const ChainArgs& a, int blocks, hipStream_t s) {
    const int mv = a.m <= 1 ? 1 : (a.m <= 2 ? 2 : (a.m <= 4 ? 4 : 8));
    if (mv == 1) hipLaunchKernelGGL((k_gemm_chain_adj<N, 1, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 2) hipLaunchKernelGGL((k_gemm_chain_adj<N, 2, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 4) hipLaunchKernelGGL((k_gemm_chain_adj<N, 4, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm_chain_adj<N, 8, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
}
template <int N, bool HAS_OUT>
static inline void qoc_chain_fwd_launch_n(const ChainArgs& a, int blocks, hipStream_t s) {
    const int mv = a.m <= 1 ? 1 : (a.m <= 2 ? 2 : (a.m <= 4 ? 4 : 8));
    if (mv == 1) hipLaunchKernelGGL((k_gemm_chain_fwd<N, 1, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 2) hipLaunchKernelGGL((k_gemm_chain_fwd<N, 2, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 4) hipLaunchKernelGGL((k_gemm_chain_fwd<N, 4, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm_chain_fwd<N, 8, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
}
template <int N>
static inline void qoc_chain_launch_c(bool conjt, const ChainArgs& a, int blocks, hipStream_t s) {
    if (conjt) { if (a.Out) qoc_chain_adj_launch_n<N, true>(a, blocks, s); else qoc_chain_adj_launch_n<N, false>(a, blocks, s); }
    else { if (a.Out) qoc_chain_fwd_launch_n<N, true>(a, blocks, s); else qoc_chain_fwd_launch_n<N, false>(a, blocks, s); }
}
// `zeros` = a zero thin buffer (N x 32) used as the addend when the chain has none
static inline void qoc_chain_launch(int N, bool conjt, ChainArgs a, const cplx* zeros, int blocks, hipStream_t s) {
    if (a.len <= 0 && !a.Fin && !a.store_initial) return;
    if (!a.E) { a.E = zeros; a.sEb = a.sEc = a.sEs = 0; }
    if (N == 32) qoc_chain_launch_c<32>(conjt, a, blocks, s); else qoc_chain_launch_c<64>(conjt, a, blocks, s);
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Time is cut into NC chunks of S = 2^L slices (padded with identity slices to SP = NC*S).  A pairwise product tree over
// the K_t gives the chunk products at the batched-GEMM rate; the sequential part of each chain shrinks from `steps`
// launches to NC (chunk boundaries) + S (all chunks swept in parallel).
struct QocGemm {
    int N = 0, S = 1, L = 0, NC = 1, SP = 1;
    int MV = 0, ldW = 0;      // persistent mode: vector slots (1/2/4/8) and row stride of the time-major wide buffers
    bool direct = false;      // state transfer as Taylor mat-vec chains on the assembled generators (one chunk, no propagators)
    bool persistent = false;  // N <= 64, m <= 8: thin chains run as persistent VALU kernels instead of one launch per step
    cplx* HsP = nullptr;      // [k+1][N][N]
    cplx *A = nullptr, *P = nullptr, *K = nullptr, *A2 = nullptr;     // [B*SP][N][N]
    cplx* tree = nullptr;     // levels 1..L of the product tree: level l at tree_off[l], [B][SP >> l][N][N]
    size_t tree_off[8];
    cplx *Y0 = nullptr, *Y1 = nullptr;                               // [B][N][N+32]
    cplx* interP = nullptr;   // [B][SP][N][32]   Psi_t
    cplx* LamP = nullptr;     // [B][SP][N][32]   Lambda_t
    cplx* SrcP = nullptr;     // [B][SP][N][32]   S_tau (state regularisers only)
    cplx* root = nullptr;     // persistent unitary mode: product tree above the chunk products, down to one matrix per seed
    cplx* zthin = nullptr;    // [N][32] zeros
    cplx *Psibnd = nullptr, *Ebnd = nullptr, *Aoff = nullptr;        // [B][NC][N][32] chunk-start Psi, chunk-end Lambda, affine offsets
    double* partial = nullptr; // [B*steps][k][N/32]
};

// Unitary mode: any n.  State transfer: psi <- P(B_t) psi is the same chain with K_t = sum_{j<T} B_t^j/j! (no squaring);
// the reference's backward step lambda <- P(-B_t) lambda (tensorflow_state.py:118-131) equals K_t^dagger lambda exactly
// when every generator is anti-Hermitian (-i dt H with H Hermitian), which `antiherm` certifies at create time.
// Any state-transfer problem with n <= 64, m <= 8 can instead run "direct" (k_gemm_taylor_chain: the reference's own
// mat-vec recursion, forward and backward, on pre-assembled generators; no time parallelism, so it is the large-batch mode).
static inline bool qoc_gemm_direct_supported(const QocDev& d) { return d.state_transfer && d.n <= 64 && d.m <= 8 && d.T >= 1; }
static inline bool qoc_gemm_supported(const QocDev& d, bool antiherm) {
    return d.m <= QOC_TW && d.T >= 1 && (!d.state_transfer || antiherm || qoc_gemm_direct_supported(d));
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

static inline int qoc_gemm_setup(QocGemm& gm, const QocDev& d, const cplx* Hs_host, bool direct, std::vector<void*>& allocs, std::string& msg) {
    const int N = ((d.n + 31) / 32) * 32;
    gm.N = N;
    gm.persistent = N <= 64 && d.m <= 8;
    gm.MV = d.m <= 1 ? 1 : (d.m <= 2 ? 2 : (d.m <= 4 ? 4 : 8));
    gm.direct = direct && d.state_transfer && gm.persistent;
    int L = 0;
    while (L < 6 && (1 << (2 * (L + 1))) <= d.steps) ++L;        // S = 2^L ~ sqrt(steps), at most 64
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
    auto al = [&](void** dst, size_t bytes) -> bool {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return false;
        allocs.push_back(p);
        *dst = p;
        return true;
    };
    const bool need_src = d.n_forb > 0 || d.has_speed;
    size_t tree_elems = 0;
    for (int l = 1; l <= L; ++l) { gm.tree_off[l] = tree_elems; tree_elems += (size_t)d.B * (gm.SP >> l) * NN; }
    const bool fused = N <= 64 && !gm.direct;                    // k_gemm_expm_fused needs no A / A2 / ping-pong buffers
    size_t root_elems = 0;
    for (int cnt = gm.NC; cnt > 1; cnt = (cnt + 1) / 2) root_elems += (size_t)d.B * ((cnt + 1) / 2) * NN;
    const bool poly = !fused && !gm.direct;                      // launch-per-product route: A2 and ping-pong buffers
    bool ok = al((void**)&gm.HsP, hp.size() * sizeof(cplx)) && (fused || al((void**)&gm.A, BSP * NN * sizeof(cplx))) &&
              (!poly || al((void**)&gm.P, BSP * NN * sizeof(cplx))) && (!poly || al((void**)&gm.A2, BSP * NN * sizeof(cplx))) &&
              al((void**)&gm.root, (gm.persistent && !d.state_transfer) ? root_elems * sizeof(cplx) : 16) &&
              al((void**)&gm.K, gm.direct ? 16 : BSP * NN * sizeof(cplx)) && al((void**)&gm.tree, tree_elems * sizeof(cplx)) &&
              al((void**)&gm.Y0, (size_t)d.B * N * (N + QOC_TW) * sizeof(cplx)) &&
              al((void**)&gm.Y1, (size_t)d.B * N * (N + QOC_TW) * sizeof(cplx)) &&
              al((void**)&gm.interP, BSP * thin * sizeof(cplx)) && al((void**)&gm.LamP, BSP * thin * sizeof(cplx)) &&
              al((void**)&gm.Psibnd, (size_t)d.B * gm.NC * thin * sizeof(cplx)) &&
              al((void**)&gm.Ebnd, (size_t)d.B * gm.NC * thin * sizeof(cplx)) &&
              al((void**)&gm.Aoff, (size_t)d.B * gm.NC * thin * sizeof(cplx)) &&
              al((void**)&gm.zthin, thin * sizeof(cplx)) &&
              al((void**)&gm.partial, (size_t)d.B * d.k * (N / 32) * (gm.persistent ? (size_t)gm.ldW : (size_t)d.steps) * sizeof(double));
    if (ok && need_src) ok = al((void**)&gm.SrcP, BSP * thin * sizeof(cplx));
    if (!ok) { msg = "GEMM path: out of device memory"; return -3; }
    if (hipMemcpy(gm.HsP, hp.data(), hp.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) { msg = "GEMM path: upload failed"; return -2; }
    // the persistent chain kernels write only the first m (<= 8) of the 32 thin columns; the rest must read as zero
    hipMemset(gm.zthin, 0, thin * sizeof(cplx));
    hipMemset(gm.interP, 0, BSP * thin * sizeof(cplx));
    hipMemset(gm.LamP, 0, BSP * thin * sizeof(cplx));
    hipMemset(gm.Psibnd, 0, (size_t)d.B * gm.NC * thin * sizeof(cplx));
    hipMemset(gm.Ebnd, 0, (size_t)d.B * gm.NC * thin * sizeof(cplx));
    hipMemset(gm.Aoff, 0, (size_t)d.B * gm.NC * thin * sizeof(cplx));
    return 0;
}

template <bool CONJT, int EPI, int SK>
static inline void qoc_gemm_launch_sk(const GemmArgs& g, unsigned blocks, hipStream_t s) {
    const size_t lds = SK > 1 ? (size_t)(SK - 1) * 2048 * sizeof(double) : 0;
    hipLaunchKernelGGL((k_zgemm32<CONJT, EPI, SK>), dim3(blocks), dim3(64 * SK), lds, s, g);
}
// Kernels that use more than 64 KB of dynamic LDS must opt in, per device: called from qoc_gemm_setup (one engine = one device)
template <bool CONJT, int EPI>
static inline void qoc_gemm_lds_opt_in_sk() {
    hipFuncSetAttribute((const void*)k_zgemm32<CONJT, EPI, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 2048 * (int)sizeof(double));
}
static inline void qoc_gemm_lds_opt_in() {
    qoc_gemm_lds_opt_in_sk<false, 0>(); qoc_gemm_lds_opt_in_sk<false, 1>(); qoc_gemm_lds_opt_in_sk<false, 2>(); qoc_gemm_lds_opt_in_sk<true, 0>();
    hipFuncSetAttribute((const void*)k_gemm_expm_fused<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 65 * (int)sizeof(cplx));
}
// picks the split-K factor from the launch size: fill ~2 waves per SIMD (2048 waves) when the batch is small
static inline void qoc_gemm_launch(bool conjt, int epi, const GemmArgs& g, hipStream_t s) {
    const size_t tiles = (size_t)g.batch * g.tiles_m * g.tiles_n;
    const unsigned blocks = (unsigned)tiles;
    int sk = 1;
    if (tiles * 2 <= 2048 && (g.Kdim / 2) % 8 == 0) sk = 2;
    if (tiles * 4 <= 2048 && (g.Kdim / 4) % 8 == 0) sk = 4;
    if (tiles * 8 <= 2048 && (g.Kdim / 8) % 8 == 0) sk = 8;
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
    } else {
        if (sk == 8) qoc_gemm_launch_sk<false, 0, 8>(g, blocks, s);
        else if (sk == 4) qoc_gemm_launch_sk<false, 0, 4>(g, blocks, s);
        else if (sk == 2) qoc_gemm_launch_sk<false, 0, 2>(g, blocks, s);
        else qoc_gemm_launch_sk<false, 0, 1>(g, blocks, s);
    }
}

static inline int gemm_grid(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }

// pairwise product tree: T_l[i] = T_{l-1}[2i+1] * T_{l-1}[2i]  (later slice on the left), T_0 = K
static inline void qoc_gemm_tree(QocGemm& gm, const QocDev& d, hipStream_t s) {
    const int N = gm.N;
    const size_t NN = (size_t)N * N;
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = g.ldb = g.ldc = N; g.Kdim = N; g.tiles_m = g.tiles_n = N / 32; g.alpha = 1.0;
    const cplx* prev = gm.K;
    for (int l = 1; l <= gm.L; ++l) {
        cplx* out = gm.tree + gm.tree_off[l];
        g.A = prev + NN; g.sA = 2 * (long long)NN; g.Bm = prev; g.sB = 2 * (long long)NN; g.C = out; g.sC = (long long)NN;
        g.batch = (int)((size_t)d.B * (gm.SP >> l));
        qoc_gemm_launch(false, 0, g, s);
        prev = out;
    }
}

// K_t for all (seed, slice): the dominant part of the path (bracketed by the profiling events of the engine)
static inline void qoc_gemm_expm(QocGemm& gm, const QocDev& d, hipStream_t s) {
    const int N = gm.N;
    const size_t NN = (size_t)N * N, BS = (size_t)d.B * gm.SP;
    const int deg = d.state_transfer ? d.T - 1 : d.T;            // matvecexp sums j < T (tensorflow_state.py:88-96)
    const int nsq = d.state_transfer ? 0 : d.s;
    if (gm.direct) {                                             // the chains apply the Taylor series themselves
        hipLaunchKernelGGL(k_gemm_assemble, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, d, gm.HsP, gm.A, N, gm.SP, 0);
        return;
    }
    if (N <= 64) {
        ExpmCoef cf;
        { double f = 1.0; for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; cf.c[j] = 1.0 / f; } }
        const size_t lds = 2 * (size_t)N * (N + 1) * sizeof(cplx);
        if (N == 32) hipLaunchKernelGGL(k_gemm_expm_fused<32>, dim3((unsigned)BS), dim3(128), lds, s, d, gm.HsP, gm.K, gm.SP, deg, nsq, cf);
        else hipLaunchKernelGGL(k_gemm_expm_fused<64>, dim3((unsigned)BS), dim3(512), lds, s, d, gm.HsP, gm.K, gm.SP, deg, nsq, cf);
        qoc_gemm_tree(gm, d, s);
        return;
    }
    hipLaunchKernelGGL(k_gemm_assemble, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, d, gm.HsP, gm.A, N, gm.SP, nsq);
    // Taylor polynomial sum_{j<=T} A^j/j! (tensorflow_state.py:37-41) in Paterson-Stockmeyer form over A2 = A*A:
    // S = B_m ; S = B_i + A2*S with B_i = c_{2i} I + c_{2i+1} A  (T = 5: 3 products instead of 4); then s squarings.
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = g.ldb = g.ldc = g.lde = N; g.sA = g.sB = g.sC = g.sE = (long long)NN; g.Kdim = N; g.tiles_m = g.tiles_n = N / 32; g.batch = (int)BS;
    double invf[24];
    { double f = 1.0; for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; invf[j] = 1.0 / f; } }
    const int mm = deg >> 1;
    const bool even = (deg & 1) == 0;
    const int horner = deg >= 2 ? (even ? mm - 1 : mm) : 0;      // products after A2
    const int products = horner + nsq;                           // buffer flips until the result
    cplx* cur = (products % 2 == 0) ? gm.K : gm.P;               // buffers alternate cur -> other on every product
    cplx* oth = (products % 2 == 0) ? gm.P : gm.K;
    if (deg >= 2) {
        g.A = gm.A; g.Bm = gm.A; g.C = gm.A2; g.E = nullptr; g.alpha = 1.0; g.beta = 0.0; g.gamma = 0.0;
        qoc_gemm_launch(false, 0, g, s);                         // A2 = A*A
        if (even) hipLaunchKernelGGL(k_gemm_ps_init, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, gm.A, gm.A2, cur, BS * NN, N,
                                     invf[2 * mm - 2], invf[2 * mm - 1], invf[deg]);
        else hipLaunchKernelGGL(k_gemm_ps_init, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, gm.A, (const cplx*)nullptr, cur, BS * NN, N,
                                invf[2 * mm], invf[2 * mm + 1], 0.0);
        for (int i = (even ? mm - 2 : mm - 1); i >= 0; --i) {    // S <- c_{2i} I + c_{2i+1} A + A2*S
            g.A = gm.A2; g.Bm = cur; g.C = oth; g.E = gm.A; g.alpha = 1.0; g.beta = invf[2 * i + 1]; g.gamma = invf[2 * i];
            qoc_gemm_launch(false, 0, g, s);
            cplx* t = cur; cur = oth; oth = t;
        }
    } else {
        hipLaunchKernelGGL(k_gemm_ps_init, dim3(gemm_grid(BS * NN)), dim3(256), 0, s, gm.A, (const cplx*)nullptr, cur, BS * NN, N, 1.0,
                           deg >= 1 ? 1.0 : 0.0, 0.0);
    }
    for (int sq = 0; sq < nsq; ++sq) {                       // M <- M M                    tensorflow_state.py:43-44
        g.A = cur; g.Bm = cur; g.C = oth; g.E = nullptr; g.alpha = 1.0; g.beta = 0.0; g.gamma = 0.0;
        qoc_gemm_launch(false, 0, g, s);
        cplx* t = cur; cur = oth; oth = t;
    }
    (void)cur;                                               // == gm.K by construction
    qoc_gemm_tree(gm, d, s);
}

static inline const cplx* qoc_gemm_chunk_products(const QocGemm& gm) { return gm.L > 0 ? gm.tree + gm.tree_off[gm.L] : gm.K; }

static inline void qoc_gemm_forward(QocGemm& gm, const QocDev& d, hipStream_t s) {
    const int N = gm.N, xw = d.state_transfer ? 0 : N, ld = xw + QOC_TW, S = gm.S, NC = gm.NC;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    const cplx* Pc = qoc_gemm_chunk_products(gm);                // [B][NC]
    hipLaunchKernelGGL(k_gemm_chain_init, dim3(gemm_grid((size_t)d.B * N * ld)), dim3(256), 0, s, d, gm.Y0, gm.Psibnd, N, NC, xw);
    if (gm.direct) {
        ChainArgs a;
        memset(&a, 0, sizeof a);
        a.K = gm.A; a.sKb = (long long)NN * gm.SP; a.sKs = (long long)NN;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin;
        a.Out = gm.interP; a.sOb = (long long)N * gm.ldW; a.sOs = gm.MV; a.ldO = gm.ldW;
        a.CI = 1; a.len = d.steps; a.m = d.m; a.nterms = d.T; a.sign = 1.0;
        qoc_taylor_chain_launch(N, a, gm.zthin, d.B, s);
        hipLaunchKernelGGL(k_gemm_unpad_wide, dim3(gemm_grid((size_t)d.B * d.steps * d.n * d.m)), dim3(256), 0, s, d, gm.interP, N, gm.ldW, gm.MV);
        return;
    }
    if (gm.persistent && !d.state_transfer) {
        // final_state = (P_{NC-1} ... P_0) U0: the product tree continues above the chunk products (log2(NC) launches)
        GemmArgs r;
        memset(&r, 0, sizeof r);
        r.lda = r.ldb = r.ldc = N; r.Kdim = N; r.tiles_m = r.tiles_n = N / 32; r.alpha = 1.0;
        const cplx* lvl = Pc;
        cplx* out = gm.root;
        for (int cnt = NC; cnt > 1; cnt = (cnt + 1) / 2) {
            const int pairs = cnt / 2, nxt = (cnt + 1) / 2;
            r.A = lvl + NN; r.Bm = lvl; r.C = out; r.sA = r.sB = 2 * (long long)NN; r.sC = (long long)NN;
            r.inner = pairs; r.sA2 = r.sB2 = (long long)cnt * NN; r.sC2 = (long long)nxt * NN; r.batch = d.B * pairs;
            qoc_gemm_launch(false, 0, r, s);
            if (cnt & 1)
                hipLaunchKernelGGL(k_gemm_copy_mats, dim3(gemm_grid((size_t)d.B * NN)), dim3(256), 0, s, out + (size_t)pairs * NN,
                                   (long long)nxt * NN, lvl + (size_t)(cnt - 1) * NN, (long long)cnt * NN, d.B, (int)NN);
            lvl = out;
            out += (size_t)d.B * nxt * NN;
        }
        memset(&r, 0, sizeof r);
        r.A = lvl; r.sA = (long long)NN; r.lda = N; r.Bm = gm.Y0; r.C = gm.Y1; r.ldb = r.ldc = ld; r.sB = r.sC = (long long)N * ld;
        r.Kdim = N; r.tiles_m = N / 32; r.tiles_n = ld / 32; r.batch = d.B; r.alpha = 1.0;
        qoc_gemm_launch(false, 0, r, s);
        hipLaunchKernelGGL(k_gemm_take_final, dim3(d.B), dim3(256), 0, s, d, gm.Y1, N);
    }
    if (gm.persistent) {
        // chunk-start vectors Psibnd[c+1] = P_c Psibnd[c]: one persistent workgroup per seed
        ChainArgs a;
        memset(&a, 0, sizeof a);
        a.K = Pc; a.sKb = (long long)NN * NC; a.sKs = (long long)NN;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin * NC;
        a.Out = gm.Psibnd + thin; a.sOb = (long long)thin * NC; a.sOs = (long long)thin; a.ldO = QOC_TW;
        a.CI = 1; a.len = NC - 1; a.m = d.m;
        qoc_chain_launch(N, false, a, gm.zthin, d.B, s);
    }
    // chunk boundaries: [X | Psi] <- P_c [X | Psi]   (X for final_state, Psi for the chunk starts)      :214-238
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.sA = (long long)NN * NC; g.ldb = g.ldc = ld; g.sB = g.sC = (long long)N * ld;
    g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = ld / 32; g.batch = d.B; g.alpha = 1.0;
    cplx *cur = gm.Y0, *oth = gm.Y1;
    for (int c = 0; c < (gm.persistent ? 0 : NC); ++c) {
        g.A = Pc + (size_t)c * NN; g.Bm = cur; g.C = oth;
        qoc_gemm_launch(false, 0, g, s);
        if (c + 1 < NC)
            hipLaunchKernelGGL(k_gemm_take_bnd, dim3(gemm_grid((size_t)d.B * thin)), dim3(256), 0, s, d, oth, gm.Psibnd, N, NC, c + 1, xw);
        cplx* x = cur; cur = oth; oth = x;
    }
    if (!d.state_transfer && !gm.persistent) hipLaunchKernelGGL(k_gemm_take_final, dim3(d.B), dim3(256), 0, s, d, cur, N);
    if (gm.persistent) {
        // every chunk swept by its own persistent workgroup: Psi_{cS+j} = K_{cS+j} Psi_{cS+j-1}
        ChainArgs a;
        memset(&a, 0, sizeof a);
        a.K = gm.K; a.sKb = (long long)NN * gm.SP; a.sKc = (long long)NN * S; a.sKs = (long long)NN;
        a.X0 = gm.Psibnd; a.sXb = (long long)thin * NC; a.sXc = (long long)thin;
        a.Out = gm.interP; a.sOb = (long long)N * gm.ldW; a.sOc = (long long)S * gm.MV; a.sOs = gm.MV; a.ldO = gm.ldW;   // time-major wide layout
        a.CI = NC; a.len = S; a.m = d.m;
        qoc_chain_launch(N, false, a, gm.zthin, d.B * NC, s);
        hipLaunchKernelGGL(k_gemm_unpad_wide, dim3(gemm_grid((size_t)d.B * d.steps * d.n * d.m)), dim3(256), 0, s, d, gm.interP, N, gm.ldW, gm.MV);
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
        qoc_gemm_launch(false, 0, h, s);
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
        qoc_gemm_launch(true, 0, g, s);
    }
}

static inline void qoc_gemm_backward(QocGemm& gm, const QocDev& d, hipStream_t s) {
    const int N = gm.N, S = gm.S, NC = gm.NC;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    const bool need_src = d.n_forb > 0 || d.has_speed;
    const cplx* Pc = qoc_gemm_chunk_products(gm);
    hipLaunchKernelGGL(k_gemm_sources, dim3(gemm_grid((size_t)d.B * (need_src ? gm.SP : 1) * thin)), dim3(256), 0, s, d, gm.SrcP, gm.Ebnd, N, gm.SP, NC);
    if (gm.direct) {                                         // lambda_{t-1} = P(-B_t) lambda_t + S_t   tensorflow_state.py:118-131
        ChainArgs a;
        memset(&a, 0, sizeof a);
        a.K = gm.A + (size_t)(d.steps - 1) * NN; a.sKb = (long long)NN * gm.SP; a.sKs = -(long long)NN;
        a.X0 = gm.Ebnd; a.sXb = (long long)thin;
        if (need_src) { a.E = gm.SrcP + (size_t)(d.steps - 1) * thin; a.sEb = (long long)thin * gm.SP; a.sEs = -(long long)thin; }
        a.Out = gm.LamP + (long long)(d.steps - 2) * gm.MV; a.sOb = (long long)N * gm.ldW; a.sOs = -gm.MV; a.ldO = gm.ldW;
        a.store_initial = 1; a.CI = 1; a.len = d.steps - 1; a.m = d.m; a.nterms = d.T; a.sign = -1.0;
        qoc_taylor_chain_launch(N, a, gm.zthin, d.B, s);
    } else if (gm.persistent) {
        ChainArgs sw;                                        // one chunk, backwards: Lambda_{t-1} = K_t^dagger Lambda_t + S_t
        memset(&sw, 0, sizeof sw);
        sw.K = gm.K + (size_t)(S - 1) * NN; sw.sKb = (long long)NN * gm.SP; sw.sKc = (long long)NN * S; sw.sKs = -(long long)NN;
        if (need_src) { sw.E = gm.SrcP + (size_t)(S - 1) * thin; sw.sEb = (long long)thin * gm.SP; sw.sEc = (long long)thin * S; sw.sEs = -(long long)thin; }
        sw.CI = NC; sw.m = d.m;
        if (need_src && NC > 1) {                            // affine offsets a_c: every chunk run from a zero costate
            ChainArgs a = sw;
            a.len = S; a.Fin = gm.Aoff; a.sFb = (long long)thin * NC; a.sFc = (long long)thin;
            qoc_chain_launch(N, true, a, gm.zthin, d.B * NC, s);
        }
        {                                                    // chunk-end costates E_{c-1} = P_c^dagger E_c + a_c
            ChainArgs a;
            memset(&a, 0, sizeof a);
            a.K = Pc + (size_t)(NC - 1) * NN; a.sKb = (long long)NN * NC; a.sKs = -(long long)NN;
            a.X0 = gm.Ebnd + (size_t)(NC - 1) * thin; a.sXb = (long long)thin * NC;
            if (need_src) { a.E = gm.Aoff + (size_t)(NC - 1) * thin; a.sEb = (long long)thin * NC; a.sEs = -(long long)thin; }
            a.Out = gm.Ebnd + (long long)(NC - 2) * (long long)thin; a.sOb = (long long)thin * NC; a.sOs = -(long long)thin; a.ldO = QOC_TW;
            a.CI = 1; a.len = NC - 1; a.m = d.m;
            qoc_chain_launch(N, true, a, gm.zthin, d.B, s);
        }
        {
            ChainArgs a = sw;
            a.X0 = gm.Ebnd; a.sXb = (long long)thin * NC; a.sXc = (long long)thin;
            a.Out = gm.LamP + (long long)(S - 2) * gm.MV; a.sOb = (long long)N * gm.ldW; a.sOc = (long long)S * gm.MV; a.sOs = -gm.MV; a.ldO = gm.ldW;
            a.store_initial = 1; a.len = S - 1;
            qoc_chain_launch(N, true, a, gm.zthin, d.B * NC, s);
        }
    } else {
    if (need_src && NC > 1) {                                // affine offsets a_c: every chunk run from a zero costate
        hipLaunchKernelGGL(k_gemm_set_chunk_ends, dim3(gemm_grid((size_t)d.B * NC * thin)), dim3(256), 0, s, d, gm.LamP, (const cplx*)nullptr, N, S, NC);
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
        qoc_gemm_launch(true, 0, g, s);
    }
    hipLaunchKernelGGL(k_gemm_set_chunk_ends, dim3(gemm_grid((size_t)d.B * NC * thin)), dim3(256), 0, s, d, gm.LamP, (const cplx*)gm.Ebnd, N, S, NC);
    qoc_gemm_bwd_sweep(gm, d, s, need_src, nullptr);
    }
    if (gm.persistent) {
        // gradients from the time-major wide layout: one product H_k' [Psi_0 ... Psi_{SP-1}] per control (batch = seeds),
        // contracted column by column with conj(Lambda)                                          tensorflow_state.py:61-63
        GemmArgs h;
        memset(&h, 0, sizeof h);
        const int tm = N / 32;
        h.lda = N; h.ldb = h.ldl = gm.ldW; h.Kdim = N;
        h.tiles_m = tm; h.tiles_n = gm.ldW / 32; h.Bm = gm.interP; h.L = gm.LamP;
        h.partial = gm.partial; h.ldp = gm.ldW; h.partial_stride = tm * gm.ldW;         // partial[b][k][tile_m][column]
        // one launch for all (seed, control) pairs: batch index bt = b*k + kk -> A = H'_{kk+1}, Bm / L = buffers of seed b
        h.A = gm.HsP + NN; h.inner = d.k; h.sA = (long long)NN; h.sA2 = 0;
        h.sB = h.sL = 0; h.sB2 = h.sL2 = (long long)N * gm.ldW;
        h.batch = d.B * d.k;
        qoc_gemm_launch(false, 2, h, s);
        hipLaunchKernelGGL(k_gemm_grad_reduce_wide, dim3(gemm_grid((size_t)d.B * d.steps * d.k)), dim3(256), 0, s, d, gm.partial, tm, gm.ldW, gm.MV);
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
            qoc_gemm_launch(false, 1, h, s);
        }
    }
    hipLaunchKernelGGL(k_gemm_grad_reduce, dim3(gemm_grid((size_t)d.B * d.steps * d.k)), dim3(256), 0, s, d, gm.partial, N / 32);
}
