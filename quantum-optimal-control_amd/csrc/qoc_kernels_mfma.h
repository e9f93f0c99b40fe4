// qoc_kernels_mfma.h -- register-resident v_mfma_f64_16x16x4_f64 path for n <= 32, unitary mode (gfx950 / CDNA4).
//
// Design ("MFMA path"):
//  * Every matrix is zero-padded to NP = 32 and lives as 16x16 fp64 tiles in the MFMA C/D fragment layout
//      lane l, reg r  <->  element (row = (l>>4) + 4r, col = l&15)
//    which is *also* the B-operand layout of K-slice r.  A left-multiplication chain  P <- A * P  therefore never
//    moves P out of registers: the D output of one product is the B operand of the next.
//  * Left-multiplication acts on column blocks independently, so one wavefront owns ONE 16-column half (J) of a
//    32x32 matrix through the whole Taylor (Horner) recursion: 64 MFMAs per product per wave, 4 independent
//    accumulator chains, ~150 VGPRs -> >= 2 waves per SIMD.
//  * Only squaring (and handing K_t to the running chunk product) needs the matrix as a LEFT operand, i.e. in the
//    A-fragment layout (lane l <-> element (row = l&15, k = l>>4)) = the transposed D layout.  The two waves of a
//    matrix exchange their halves through a padded, transposed LDS image (one barrier per exchange, two buffers).
//  * Time is cut into C chunks per seed.  k_mfma_expm_chunk computes K_t for its chunk AND the chunk product
//    P_c = prod K_t; the thin sweeps (Psi_t = K_t Psi_{t-1}, Lambda_{t-1} = K_t^dagger Lambda_t) rebuild their chunk
//    boundary from the P_c's (<= C thin products) and then run their chunk -- every kernel has B*C-way parallelism.
//
// Reference semantics: core/tensorflow_state.py:25-46 (matexp), :204-242 (chain, inter vectors), :49-65 (gradient).
#pragma once
#include <string>
#include <vector>
#include "qoc_common.h"

#define QOC_NP 32                 // padded matrix dimension
#define QOC_LDR 33                // padded leading dimension of the transposed LDS image (complex elements)
#define QOC_MAXC 64               // max time chunks per seed

typedef double d4 __attribute__((ext_vector_type(4)));
#define QMFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

struct CTile { d4 re, im; };                      // 16x16 complex tile, D layout
struct AFrag { double re[2][8], im[2][8]; };      // 32x32 complex LEFT operand: [row block I][k-slice q], A layout

struct QocMfma {
    int C = 1;                // chunks per seed
    int L = 1;                // steps per chunk
    cplx* Hs_pad = nullptr;   // [k+1][32][32]  -i dt H, zero padded
    cplx* HsT_pad = nullptr;  // [k+1][32][32]  transposed
    cplx* U0_pad = nullptr;   // [32][32] zero padded
    cplx* K = nullptr;        // [B][steps][32][32]
    cplx* Pc = nullptr;       // [B][C][32][32] chunk products
    size_t bwd_lds = 0;
    bool h_in_lds = true;
};

// ---- fragment helpers ---------------------------------------------------------------------------------------------

__device__ __forceinline__ cplx ldg_c(const cplx* p) { return *p; }

// A-layout fragments of M from its TRANSPOSE stored plain row-major [32][32] (coalesced: 4 x 256 B per instruction).
__device__ __forceinline__ void afrag_from_transposed(const cplx* __restrict__ Mt, int lane, AFrag& A) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const cplx v = Mt[(4 * q + (lane >> 4)) * QOC_NP + 16 * I + (lane & 15)];
            A.re[I][q] = v.x; A.im[I][q] = v.y;
        }
}
// A-layout fragments of M^dagger from M stored plain row-major (same coalesced pattern, conjugated).
__device__ __forceinline__ void afrag_dagger(const cplx* __restrict__ M, int lane, AFrag& A) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const cplx v = M[(4 * q + (lane >> 4)) * QOC_NP + 16 * I + (lane & 15)];
            A.re[I][q] = v.x; A.im[I][q] = -v.y;
        }
}
// A-layout fragments of M from M stored plain row-major (gather: 16 rows x 64 B per instruction).
__device__ __forceinline__ void afrag_gather(const cplx* __restrict__ M, int lane, AFrag& A) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const cplx v = M[(16 * I + (lane & 15)) * QOC_NP + 4 * q + (lane >> 4)];
            A.re[I][q] = v.x; A.im[I][q] = v.y;
        }
}

// out[I] = sum_k A[I,k] * p[k]   for one 16-column block: 64 MFMAs, 4 independent accumulator chains.
__device__ __forceinline__ void mm_colblock(const AFrag& A, const CTile p[2], CTile out[2]) {
    d4 r0 = {0, 0, 0, 0}, i0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0}, i1 = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double br = p[q >> 2].re[q & 3], bi = p[q >> 2].im[q & 3], nbi = -bi;
        r0 = QMFMA(A.re[0][q], br, r0);
        i0 = QMFMA(A.re[0][q], bi, i0);
        r1 = QMFMA(A.re[1][q], br, r1);
        i1 = QMFMA(A.re[1][q], bi, i1);
        r0 = QMFMA(A.im[0][q], nbi, r0);
        i0 = QMFMA(A.im[0][q], br, i0);
        r1 = QMFMA(A.im[1][q], nbi, r1);
        i1 = QMFMA(A.im[1][q], br, i1);
    }
    out[0].re = r0; out[0].im = i0; out[1].re = r1; out[1].im = i1;
}

// D-layout column block J of a plain row-major [32][32] matrix.
__device__ __forceinline__ void colblock_load(const cplx* __restrict__ M, int J, int lane, CTile p[2]) {
#pragma unroll
    for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const cplx v = M[(16 * Ib + (lane >> 4) + 4 * r) * QOC_NP + 16 * J + (lane & 15)];
            p[Ib].re[r] = v.x; p[Ib].im[r] = v.y;
        }
}
__device__ __forceinline__ void colblock_store(cplx* __restrict__ M, int J, int lane, const CTile p[2]) {
#pragma unroll
    for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            M[(16 * Ib + (lane >> 4) + 4 * r) * QOC_NP + 16 * J + (lane & 15)] = cmake(p[Ib].re[r], p[Ib].im[r]);
}
__device__ __forceinline__ void colblock_identity(int J, int lane, CTile p[2]) {
#pragma unroll
    for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * Ib + (lane >> 4) + 4 * r, col = 16 * J + (lane & 15);
            p[Ib].re[r] = (row == col) ? 1.0 : 0.0; p[Ib].im[r] = 0.0;
        }
}

// Write a column block into the transposed LDS image img[col][row] (leading dimension QOC_LDR, complex).
__device__ __forceinline__ void lds_put_colblock(cplx* img, int Jcol0, int lane, const CTile p[2]) {
#pragma unroll
    for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            img[(Jcol0 + (lane & 15)) * QOC_LDR + 16 * Ib + (lane >> 4) + 4 * r] = cmake(p[Ib].re[r], p[Ib].im[r]);
}
// Read the A-layout fragments of the full 32x32 matrix held in the transposed image.
__device__ __forceinline__ void lds_get_afrag(const cplx* img, int lane, AFrag& A) {
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const cplx v = img[(4 * q + (lane >> 4)) * QOC_LDR + 16 * I + (lane & 15)];
            A.re[I][q] = v.x; A.im[I][q] = v.y;
        }
}

// ---- kernel E: K_t = matexp for every t of one chunk + chunk product P_c ---------------------------------------
// One workgroup = 2 waves = the two 16-column halves of the matrices of chunk (b, c).
__global__ void __launch_bounds__(128, 2) k_mfma_expm_chunk(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[2][QOC_NP * QOC_LDR];
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const int NN = QOC_NP * QOC_NP;
    const double inv_scale = 1.0 / (double)(1 << d.s);
    int flip = 0;
    CTile R[2];
    colblock_identity(J, lane, R);
    for (int t = t0; t < t1; ++t) {
        // ---- A_t = (H0' + sum_k u_k H_k') / 2^s : left-operand fragments + this wave's column block -----------
        AFrag A;
        CTile P[2];
        {
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const cplx h0 = mf.HsT_pad[(4 * q + (lane >> 4)) * QOC_NP + 16 * I + (lane & 15)];
                    A.re[I][q] = h0.x * inv_scale; A.im[I][q] = h0.y * inv_scale;
                }
#pragma unroll
            for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx h0 = mf.Hs_pad[(16 * Ib + (lane >> 4) + 4 * r) * QOC_NP + 16 * J + (lane & 15)];
                    P[Ib].re[r] = h0.x * inv_scale; P[Ib].im[r] = h0.y * inv_scale;
                }
#pragma unroll 1
            for (int kk = 0; kk < d.k; ++kk) {
                const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
                const cplx* __restrict__ HT = mf.HsT_pad + (size_t)(kk + 1) * NN;
                const cplx* __restrict__ HP = mf.Hs_pad + (size_t)(kk + 1) * NN;
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const cplx h = HT[(4 * q + (lane >> 4)) * QOC_NP + 16 * I + (lane & 15)];
                        A.re[I][q] = fma(ck, h.x, A.re[I][q]);
                        A.im[I][q] = fma(ck, h.y, A.im[I][q]);
                    }
#pragma unroll
                for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = HP[(16 * Ib + (lane >> 4) + 4 * r) * QOC_NP + 16 * J + (lane & 15)];
                        P[Ib].re[r] = fma(ck, h.x, P[Ib].re[r]);
                        P[Ib].im[r] = fma(ck, h.y, P[Ib].im[r]);
                    }
            }
        }
        // ---- Horner: P_{T-1} = I + A/T ; P_{j-1} = I + (A P_j)/j  -> sum_{j<=T} A^j/j!   (tensorflow_state.py:37-41)
        // identity on this wave's column block: tile Ib == J, register r, lanes with (l&15) - (l>>4) == 4r
        const int dlt = (lane & 15) - (lane >> 4);
        {
            const double invT = 1.0 / (double)d.T;
#pragma unroll
            for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double one = (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
                    P[Ib].re[r] = one + P[Ib].re[r] * invT; P[Ib].im[r] = P[Ib].im[r] * invT;
                }
        }
        for (int j = d.T - 1; j >= 1; --j) {
            CTile acc[2];
            mm_colblock(A, P, acc);
            const double invj = 1.0 / (double)j;
#pragma unroll
            for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double one = (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
                    P[Ib].re[r] = one + acc[Ib].re[r] * invj; P[Ib].im[r] = acc[Ib].im[r] * invj;
                }
        }
        // ---- squaring: M <- M*M, s times (:43-44); the left operand comes back through the LDS image -----------
        for (int sq = 0; sq < d.s; ++sq) {
            lds_put_colblock(img[flip], 16 * J, lane, P);
            __syncthreads();
            lds_get_afrag(img[flip], lane, A);
            flip ^= 1;
            CTile acc[2];
            mm_colblock(A, P, acc);
            P[0] = acc[0]; P[1] = acc[1];
        }
        // ---- K_t out; running chunk product R <- K_t R ------------------------------------------------------------
        colblock_store(mf.K + ((size_t)b * d.steps + t) * NN, J, lane, P);
        lds_put_colblock(img[flip], 16 * J, lane, P);
        __syncthreads();
        lds_get_afrag(img[flip], lane, A);
        flip ^= 1;
        CTile acc[2];
        mm_colblock(A, R, acc);
        R[0] = acc[0]; R[1] = acc[1];
    }
    colblock_store(mf.Pc + ((size_t)b * mf.C + c) * NN, J, lane, R);
}

// ---- kernel F: thin forward sweep  Psi_t = K_t Psi_{t-1}  (inter vectors) + final unitary ------------------------
// grid.x = B*C sweep waves + B*2 final-unitary waves, 4 waves per workgroup, no LDS, no barriers.
__global__ void __launch_bounds__(256) k_mfma_forward(QocDev d, QocMfma mf) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NN = QOC_NP * QOC_NP;
    const int n_sweep = d.B * mf.C;
    if (item < n_sweep) {
        const int b = item / mf.C, c = item - b * mf.C;
        const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
        CTile Psi[2];
#pragma unroll
        for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) v = d.Psi0[row * d.m + col];
                Psi[Ib].re[r] = v.x; Psi[Ib].im[r] = v.y;
            }
        cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
        if (c == 0) {                                                   // inter[0] = V  (tensorflow_state.py:232-233)
            for (int o = lane; o < d.n * d.m; o += 64) iv[o] = d.V[o];
        }
        AFrag A;
        for (int cc = 0; cc < c; ++cc) {                                // chunk boundary from the chunk products
            afrag_gather(mf.Pc + ((size_t)b * mf.C + cc) * NN, lane, A);
            CTile acc[2];
            mm_colblock(A, Psi, acc);
            Psi[0] = acc[0]; Psi[1] = acc[1];
        }
        for (int t = t0; t < t1; ++t) {
            afrag_gather(mf.K + ((size_t)b * d.steps + t) * NN, lane, A);
            CTile acc[2];
            mm_colblock(A, Psi, acc);
            Psi[0] = acc[0]; Psi[1] = acc[1];
            cplx* out = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
            for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                    if (row < d.n && col < d.m) out[row * d.m + col] = cmake(Psi[Ib].re[r], Psi[Ib].im[r]);
                }
        }
    } else if (item < n_sweep + d.B * 2) {
        // final_state = P_{C-1} ... P_0 U0 (tensorflow_state.py:223), one wave per 16-column half
        const int w = item - n_sweep, b = w >> 1, J = w & 1;
        CTile X[2];
        colblock_load(mf.U0_pad, J, lane, X);
        AFrag A;
        for (int cc = 0; cc < mf.C; ++cc) {
            afrag_gather(mf.Pc + ((size_t)b * mf.C + cc) * NN, lane, A);
            CTile acc[2];
            mm_colblock(A, X, acc);
            X[0] = acc[0]; X[1] = acc[1];
        }
        cplx* Xf = d.Xfinal + (size_t)b * d.n * d.n;
#pragma unroll
        for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = 16 * J + (lane & 15);
                if (row < d.n && col < d.n) Xf[row * d.n + col] = cmake(X[Ib].re[r], X[Ib].im[r]);
            }
    }
}

// unitary_scale = (1/n) sum_c |sum_a X[c][a]|^2                      tensorflow_state.py:225
__global__ void __launch_bounds__(64) k_mfma_uscale(QocDev d) {
    const int b = blockIdx.x, n = d.n, lane = threadIdx.x;
    const cplx* X = d.Xfinal + (size_t)b * n * n;
    double part = 0.0;
    for (int c = lane; c < n; c += 64) {
        cplx rs = cmake(0.0, 0.0);
        for (int a = 0; a < n; ++a) rs = cadd(rs, X[c * n + a]);
        part += rs.x * rs.x + rs.y * rs.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) d.uscale[b] = part / (double)n;
}

// ---- kernel B: thin backward sweep  Lambda_{t-1} = K_t^dagger Lambda_t  + control gradients ----------------------
// dL/du_{k,t} = Re sum_ab H_k'[a][b] Q_t[a][b],  Q_t = conj(Lambda_t) Psi_t^T  (rank-m outer product on the MFMA),
// which equals Re <Lambda_t, H_k' Psi_t> of the reference's matexp_op_grad (tensorflow_state.py:61-63).
// 4 waves per workgroup; LDS: D-layout image of the k control Hamiltonians (shared) + one transposition pad per wave.
template <bool H_IN_LDS>
__global__ void __launch_bounds__(256) k_mfma_backward(QocDev d, QocMfma mf, int single_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int NN = QOC_NP * QOC_NP;
    cplx* Hl = (cplx*)smem;                                                     // [k][4 tiles][4 regs][64 lanes]
    cplx* pad = (cplx*)(smem + (H_IN_LDS ? (size_t)d.k * NN * sizeof(cplx) : 0)) + (size_t)wv * 16 * QOC_LDR;
    if (H_IN_LDS) {
        for (int o = threadIdx.x; o < d.k * NN; o += blockDim.x) {
            // o = ((kk*4 + tile)*4 + r)*64 + l  ->  element (row, col) of H_{kk+1}'
            const int l = o & 63, r = (o >> 6) & 3, tile = (o >> 8) & 3, kk = o >> 10;
            const int row = 16 * (tile >> 1) + (l >> 4) + 4 * r, col = 16 * (tile & 1) + (l & 15);
            Hl[o] = mf.Hs_pad[(size_t)(kk + 1) * NN + row * QOC_NP + col];
        }
        __syncthreads();
    }
    const int CC = single_chunk ? 1 : mf.C;
    const int item = blockIdx.x * 4 + wv;
    if (item >= d.B * CC) return;
    const int b = item / CC, c = item - b * CC;
    const int t0 = single_chunk ? 0 : c * mf.L, t1 = single_chunk ? d.steps : min(t0 + mf.L, d.steps);
    const bool need_src = d.n_forb > 0 || d.has_speed;
    // terminal costate: -(2/m^2) z W (+ S_steps)
    CTile Lam[2];
    {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) {
                    v = cscale(cmul(z, d.W[row * d.m + col]), c0);
                    if (need_src) v = cadd(v, source_at(d, b, d.steps, row, col));
                }
                Lam[Ib].re[r] = v.x; Lam[Ib].im[r] = v.y;
            }
    }
    AFrag A;
    if (!single_chunk) {
        for (int cc = mf.C - 1; cc > c; --cc) {                          // Lambda at the end of this chunk
            afrag_dagger(mf.Pc + ((size_t)b * mf.C + cc) * NN, lane, A);
            CTile acc[2];
            mm_colblock(A, Lam, acc);
            Lam[0] = acc[0]; Lam[1] = acc[1];
        }
    }
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    for (int t = t1 - 1; t >= t0; --t) {
        // ---- Q = conj(Lambda_t) Psi_t^T ---------------------------------------------------------------------------
        lds_put_colblock(pad, 0, lane, Lam);                             // wave-private image: pad[j][row]
        double lr[2][4], li[2][4], pr[2][4], pi[2][4];
        const cplx* psi = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 16 * I + (lane & 15), j = 4 * q + (lane >> 4);
                const cplx lv = pad[j * QOC_LDR + row];
                lr[I][q] = lv.x; li[I][q] = lv.y;
                cplx pv = cmake(0.0, 0.0);
                if (row < d.n && j < d.m) pv = psi[row * d.m + j];
                pr[I][q] = pv.x; pi[I][q] = pv.y;
            }
        double g[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) g[kk] = 0.0;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int Jp = 0; Jp < 2; ++Jp) {
                d4 qr = {0, 0, 0, 0}, qi = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    qr = QMFMA(lr[I][q], pr[Jp][q], qr);
                    qi = QMFMA(lr[I][q], pi[Jp][q], qi);
                    qr = QMFMA(li[I][q], pi[Jp][q], qr);
                    qi = QMFMA(-li[I][q], pr[Jp][q], qi);
                }
                const int tile = I * 2 + Jp;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (kk >= d.k) continue;
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cplx h;
                        if (H_IN_LDS) {
                            h = Hl[((kk * 4 + tile) * 4 + r) * 64 + lane];
                        } else {
                            const int row = 16 * I + (lane >> 4) + 4 * r, col = 16 * Jp + (lane & 15);
                            h = mf.Hs_pad[(size_t)(kk + 1) * NN + row * QOC_NP + col];
                        }
                        acc = fma(h.x, qr[r], acc);
                        acc = fma(-h.y, qi[r], acc);
                    }
                    g[kk] += acc;
                }
            }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk >= d.k) continue;
            double v = g[kk];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) d.dLdu[((size_t)b * d.k + kk) * d.steps + t] = v;
        }
        if (t == 0) break;
        // ---- Lambda_{t-1} = K_t^dagger Lambda_t (+ S_{t-1}) ---------------------------------------------------------
        afrag_dagger(mf.K + ((size_t)b * d.steps + t) * NN, lane, A);
        CTile acc[2];
        mm_colblock(A, Lam, acc);
        Lam[0] = acc[0]; Lam[1] = acc[1];
        if (need_src) {
#pragma unroll
            for (int Ib = 0; Ib < 2; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                    if (row < d.n && col < d.m) {
                        const cplx sv = source_at(d, b, t, row, col);
                        Lam[Ib].re[r] += sv.x; Lam[Ib].im[r] += sv.y;
                    }
                }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------

static inline bool qoc_mfma_supported(const QocDev& d) {
    return !d.state_transfer && d.n <= QOC_NP && d.m <= 16 && d.k <= 8 && d.T >= 1;
}

static inline int qoc_mfma_setup(QocMfma& mf, const QocDev& d, int chunks_req, const cplx* Hs_host,
                                 std::vector<void*>& allocs, std::string& msg) {
    const int NP = QOC_NP, NN = NP * NP;
    int C = chunks_req;
    if (C <= 0) {
        C = (1024 + d.B - 1) / d.B;                  // ~2 waves per SIMD for the expm kernel (2 waves per chunk)
        if (C > 32) C = 32;
    }
    if (C > d.steps) C = d.steps;
    if (C > QOC_MAXC) C = QOC_MAXC;
    if (C < 1) C = 1;
    int L = (d.steps + C - 1) / C;
    C = (d.steps + L - 1) / L;                       // no empty chunks
    mf.C = C; mf.L = L;
    std::vector<cplx> hp((size_t)(d.k + 1) * NN), ht((size_t)(d.k + 1) * NN), u0((size_t)NN);
    for (auto& v : hp) { v.x = 0; v.y = 0; }
    ht = hp;
    for (auto& v : u0) { v.x = 0; v.y = 0; }
    for (int kk = 0; kk <= d.k; ++kk)
        for (int a = 0; a < d.n; ++a)
            for (int bcol = 0; bcol < d.n; ++bcol) {
                const cplx v = Hs_host[(size_t)kk * d.n * d.n + a * d.n + bcol];
                hp[(size_t)kk * NN + a * NP + bcol] = v;
                ht[(size_t)kk * NN + bcol * NP + a] = v;
            }
    std::vector<cplx> u0h((size_t)d.n * d.n);
    if (hipMemcpy(u0h.data(), d.U0, u0h.size() * sizeof(cplx), hipMemcpyDeviceToHost) != hipSuccess) { msg = "U0 readback failed"; return -2; }
    for (int a = 0; a < d.n; ++a)
        for (int bcol = 0; bcol < d.n; ++bcol) u0[a * NP + bcol] = u0h[a * d.n + bcol];
    auto up = [&](cplx** dst, const std::vector<cplx>& src) -> bool {
        void* p = nullptr;
        if (hipMalloc(&p, src.size() * sizeof(cplx)) != hipSuccess) return false;
        allocs.push_back(p);
        if (hipMemcpy(p, src.data(), src.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) return false;
        *dst = (cplx*)p;
        return true;
    };
    if (!up(&mf.Hs_pad, hp) || !up(&mf.HsT_pad, ht) || !up(&mf.U0_pad, u0)) { msg = "MFMA path: constant upload failed"; return -3; }
    auto al = [&](cplx** dst, size_t count) -> bool {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(cplx)) != hipSuccess) return false;
        allocs.push_back(p);
        *dst = (cplx*)p;
        return true;
    };
    if (!al(&mf.K, (size_t)d.B * d.steps * NN) || !al(&mf.Pc, (size_t)d.B * C * NN)) { msg = "MFMA path: out of device memory"; return -3; }
    const size_t pads = (size_t)4 * 16 * QOC_LDR * sizeof(cplx);
    const size_t hbytes = (size_t)d.k * NN * sizeof(cplx);
    mf.h_in_lds = (hbytes + pads) <= 160 * 1024;
    mf.bwd_lds = pads + (mf.h_in_lds ? hbytes : 0);
    if (mf.h_in_lds) {
        if (hipFuncSetAttribute((const void*)k_mfma_backward<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds) != hipSuccess) {
            msg = "MFMA path: cannot reserve LDS for the backward kernel";
            return -2;
        }
    }
    return 0;
}

static inline void qoc_mfma_launch_expm(QocMfma& mf, const QocDev& d, hipStream_t s) {
    hipLaunchKernelGGL(k_mfma_expm_chunk, dim3(d.B * mf.C), dim3(128), 0, s, d, mf);
}
static inline void qoc_mfma_launch_forward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    const int items = d.B * mf.C + d.B * 2;
    hipLaunchKernelGGL(k_mfma_forward, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    hipLaunchKernelGGL(k_mfma_uscale, dim3(d.B), dim3(64), 0, s, d);
}
static inline void qoc_mfma_launch_backward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    // state regularisers add a source at every step (affine recursion): run those sequentially per seed for now
    const int single = (d.n_forb > 0 || d.has_speed) ? 1 : 0;
    const int items = d.B * (single ? 1 : mf.C);
    if (mf.h_in_lds)
        hipLaunchKernelGGL(k_mfma_backward<true>, dim3((items + 3) / 4), dim3(256), mf.bwd_lds, s, d, mf, single);
    else
        hipLaunchKernelGGL(k_mfma_backward<false>, dim3((items + 3) / 4), dim3(256), mf.bwd_lds, s, d, mf, single);
}
