// qoc_gemm_expm.h -- k_gemm_expm_fused: LDS-resident per-slice matrix exponential for N <= 64 (GEMM path).
// Reference semantics: core/tensorflow_state.py:25-46 (get_matexp) and :83-96 (matvecexp polynomial, no squaring).
#pragma once
#include "qoc_gemm_tiles.h"

// ---- fused per-slice exponential for N <= 64 ---------------------------------------------------------------------------
// One workgroup per (seed, slice): A_t is assembled into LDS, the Paterson-Stockmeyer Taylor polynomial and the squarings
// run as MFMA products (4x4x4 form, lds_mm below) whose operands are read from two LDS-resident matrices (row stride N+1 elements), accumulators and
// the per-wave block of A stay in registers, and only K_t is written to HBM.  The launch-per-product route streams three
// B*SP*N*N buffers through HBM per product (7-11 products); this kernel writes one.
// Wave w owns tile row I = w / (N/32) and the tile-column pair Jp = w % (N/32) (2 tiles of 16x16, sharing the left operand).
#define QOC_GEMM_MAXT 48   // 1/j! tables of the GEMM path: Taylor orders up to 47
#ifndef QOC_EXPM_LDPAD
#define QOC_EXPM_LDPAD 1   // row stride of the LDS matrices = N + QOC_EXPM_LDPAD complex elements
#endif
#ifndef QOC_EXPM_MFMA4
#define QOC_EXPM_MFMA4 0   // 1: products on v_mfma_f64_4x4x4 (measured SLOWER here: 8 waves per CU fetch 6 KB of operands per 24 MFMAs from LDS)
#endif
struct ExpmCoef { double c[QOC_GEMM_MAXT]; };

// Products on v_mfma_f64_4x4x4_4b_f64 (round 4; the 16x16x4 instruction this kernel used before peaks at 48 TFLOP/s, the 4x4x4 form at 73):
// a D / B register of the 4x4x4 form is a 4-row x 16-column strip with lane = 16 row + column -- exactly ONE register r of the 16 x 16 D tile
// (row lk + 4 r, column lr) -- so the accumulators keep the D-tile layout of the rest of the kernel (component r of re[J] / im[J] = row strip r)
// and only the operand fetch changes: per 4-deep block kb of the inner index the wave reads its four 4 x 4 left blocks (rows 16 I + 4 ib ..,
// broadcast over the four lane groups of a block) and the two right strips of its tile-column pair, and issues 24 MFMAs (3-multiplication form).
template <int N>
__device__ __forceinline__ void lds_mm4(const cplx* __restrict__ L, const cplx* __restrict__ R, int I, int Jp, int lane,
                                       gd4 (&re)[2], gd4 (&im)[2]) {
    constexpr int LD = N + QOC_EXPM_LDPAD;
    double t1[2][4], t2[2][4], t3[2][4];
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) { t1[J][ib] = 0.0; t2[J][ib] = 0.0; t3[J][ib] = 0.0; }
    const int lr = lane & 15, lk = lane >> 4, li = lane & 3;
    const cplx* lp = L + (16 * I + li) * LD + lk;                 // block (ib, kb): lane 16 k + 4 b + i <-> L[16 I + 4 ib + i][4 kb + k]
    const cplx* rp = R + lk * LD + 32 * Jp + lr;                  // strip (kb, J):  lane 16 k + c       <-> R[4 kb + k][32 Jp + 16 J + c]
#pragma unroll 2
    for (int kb = 0; kb < N / 4; ++kb) {
        const cplx b0 = rp[4 * kb * LD], b1 = rp[4 * kb * LD + 16];
        cplx a[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) a[ib] = lp[4 * ib * LD + 4 * kb];
        const double bs0 = b0.x + b0.y, bs1 = b1.x + b1.y;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const double as = a[ib].x + a[ib].y;
            t1[0][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ib].x, b0.x, t1[0][ib], 0, 0, 0);
            t2[0][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ib].y, b0.y, t2[0][ib], 0, 0, 0);
            t3[0][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(as, bs0, t3[0][ib], 0, 0, 0);
            t1[1][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ib].x, b1.x, t1[1][ib], 0, 0, 0);
            t2[1][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[ib].y, b1.y, t2[1][ib], 0, 0, 0);
            t3[1][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(as, bs1, t3[1][ib], 0, 0, 0);
        }
    }
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) { re[J][ib] = t1[J][ib] - t2[J][ib]; im[J][ib] = t3[J][ib] - t1[J][ib] - t2[J][ib]; }
}

template <int N>
__device__ __forceinline__ void lds_mm16(const cplx* __restrict__ L, const cplx* __restrict__ R, int I, int Jp, int lane,
                                         gd4 (&re)[2], gd4 (&im)[2]) {
    constexpr int LD = N + QOC_EXPM_LDPAD;
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
__device__ __forceinline__ void lds_mm(const cplx* __restrict__ L, const cplx* __restrict__ R, int I, int Jp, int lane, gd4 (&re)[2], gd4 (&im)[2]) {
    if constexpr (QOC_EXPM_MFMA4) lds_mm4<N>(L, R, I, Jp, lane, re, im); else lds_mm16<N>(L, R, I, Jp, lane, re, im);
}

template <int N>
__global__ void __launch_bounds__((N / 16) * (N / 16) * 32) k_gemm_expm_fused(QocDev d, const cplx* __restrict__ HsP, cplx* __restrict__ Kout,
                                                                               cplx* __restrict__ KTout, int SP, int deg, int nsq, ExpmCoef cf) {
    constexpr int LD = N + QOC_EXPM_LDPAD, NT = (N / 16) * (N / 16) * 32, NN = N * N;
    extern __shared__ __attribute__((aligned(16))) cplx ex_lds[];
    cplx* X = ex_lds;                   // A, then S / M
    cplx* Y = ex_lds + N * LD;          // A2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int I = wv / (N / 32), Jp = wv % (N / 32);
    const int lr = lane & 15, lk = lane >> 4;
    const int b = blockIdx.x / SP, t = blockIdx.x - b * SP;
    cplx* Kt = Kout + (size_t)blockIdx.x * NN;
    cplx* KTt = KTout ? KTout + (size_t)blockIdx.x * NN : nullptr;     // K_t^T as well: the backward chains then read rows
    // D-layout coordinates of this lane's 2 x 4 accumulator elements
    auto drow = [&](int r) { return 16 * I + lk + 4 * r; };
    auto dcol = [&](int J) { return 32 * Jp + 16 * J + lr; };
    if (t >= d.steps) {                                           // padded slice: K = I exactly
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const cplx one = cmake(drow(r) == dcol(J) ? 1.0 : 0.0, 0.0);
                Kt[(size_t)drow(r) * N + dcol(J)] = one;
                if (KTt) KTt[(size_t)dcol(J) * N + drow(r)] = one;
            }
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
    if (KTt) {
        // the transpose leaves through LDS so that its rows are written coalesced (a direct store scatters 16-byte pieces
        // over 64 rows per instruction): X[row][col] <- K, then thread e reads X[e % N][e / N]
        __syncthreads();
        put(X);
        __syncthreads();
        constexpr int PER = NN / NT;
#pragma unroll
        for (int x = 0; x < PER; ++x) { const int e = tid + NT * x; KTt[e] = X[(e % N) * LD + (e / N)]; }
    }
}
