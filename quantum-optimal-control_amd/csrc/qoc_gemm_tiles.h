// qoc_gemm_tiles.h -- k_zgemm32: batched complex-fp64 GEMM on 32x32 output tiles (GEMM path building block).
//
// ONE wavefront per 32x32 output tile, v_mfma_f64_16x16x4_f64 with the 3-multiplication complex form (12 MFMAs per 4-deep
// k-slice, 12 independent accumulator chains); operand fragments are loaded straight from global memory in the MFMA A/B
// lane layouts (B rows are coalesced 256-byte segments; A is a 16-row x 64-byte gather that L1/L2 absorb), no LDS and no
// barriers.  Epilogues: C = alpha*op(A)*B + beta*E + gamma*I (EPI 0), per-tile dot with conj(L) (EPI 1), per-column dots
// with conj(L) (EPI 2).  Small batches split the inner dimension over SK waves of a workgroup (partials meet in LDS).
#pragma once
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
    cplx* CT; long long sCT; int ldct;         // EPI = 0, optional: also store the transpose, CT[batch][col][row]
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
                    if (g.CT) g.CT[(size_t)bt * g.sCT + (size_t)col * g.ldct + row] = v;
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
