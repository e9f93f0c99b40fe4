// qoc_kernels_mfma.h -- register-resident v_mfma_f64_16x16x4_f64 path for n <= 32, unitary mode (gfx950 / CDNA4).
//
// Design ("MFMA path"):
//  * Every matrix is zero-padded to NP = 16*NT (NT = 1 for n <= 16, NT = 2 for n <= 32; the kernels are templated on
//    NT and the comments below describe NT = 2) and lives as 16x16 fp64 tiles in the MFMA C/D fragment layout
//      lane l, reg r  <->  element (row = (l>>4) + 4r, col = l&15)
//    which is *also* the B-operand layout of K-slice r.  A left-multiplication chain  P <- A * P  therefore never
//    moves P out of registers: the D output of one product is the B operand of the next.
//  * Left-multiplication acts on column blocks independently, so one wavefront owns ONE 16-column half (J) of a
//    32x32 matrix through the whole Taylor (Horner) recursion.
//  * Complex products use the 3-multiplication form  T1 = Ar Br, T2 = Ai Bi, T3 = (Ar+Ai)(Br+Bi),
//    Re = T1 - T2, Im = T3 - T1 - T2 :  48 MFMAs per 32x32x16 complex product per wave instead of 64,
//    6 independent accumulator chains.
//  * Only squaring (and handing K_t to the running chunk product) needs the matrix as a LEFT operand, i.e. in the
//    A-fragment layout (lane l <-> element (row = l&15, k = l>>4)) = the transposed D layout.  The two waves of a
//    matrix exchange their halves through a padded, transposed LDS image (one barrier per exchange, two buffers).
//  * HBM format "fragD(M)": 16 fragments f = cb*8 + q of 64 lanes x complex (1 KB each, perfectly coalesced),
//    lane l of fragment (cb, q) = M[4q + (l>>4)][16cb + (l&15)].  D-layout register (row block Ib, reg r) of column
//    block J is fragment J*8 + 4Ib + r; the A-operand (I, q) for a left product by M^dagger is conj(fragD(M)[I*8+q])
//    and for a left product by M it is fragD(M^T)[I*8+q].  K_t and the chunk products are stored as fragD(K) and
//    fragD(K^T), so every global access of these kernels is  uniform base + lane*16 + immediate.
//  * Time is cut into C chunks per seed.  k_mfma_expm_chunk computes K_t for its chunk AND the chunk product
//    P_c = prod K_t; the thin sweeps (Psi_t = K_t Psi_{t-1}, Lambda_{t-1} = K_t^dagger Lambda_t) rebuild their chunk
//    boundary from the P_c's (<= C thin products) and then run their chunk -- every kernel has B*C-way parallelism.
//
// Reference semantics: core/tensorflow_state.py:25-46 (matexp), :204-242 (chain, inter vectors), :49-65 (gradient).
#pragma once
#include <string>
#include <vector>
#include "qoc_common.h"

#define QOC_NP 32                 // largest padded matrix dimension (NT = 2)
#define QOC_MAXC 64               // max time chunks per seed
// per-NT constants: NP = 16 NT (padded size), QS = 4 NT (k-slices), LDR = NP + 1 (LDS image leading dimension),
// FR = 256 NT^2 (complex elements of one fragD matrix = NT*QS fragments x 64 lanes)
#define QNP (16 * NT)
#define QQS (4 * NT)
#define QLDR (16 * NT + 1)
#define QFR (256 * NT * NT)

typedef double d4 __attribute__((ext_vector_type(4)));
#define QMFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

struct CTile { d4 re, im; };                      // 16x16 complex tile, D layout
template <int NT> struct AFragT { double re[NT][4 * NT], im[NT][4 * NT]; };   // LEFT operand: [row block I][k-slice q], A layout

struct QocMfma {
    int C = 1;                // chunks per seed
    int L = 1;                // steps per chunk
    int mq = 4;               // ceil(m / 4): k-slices of the rank-m outer product
    int NT = 2;               // 16x16 tiles per matrix dimension (1: n <= 16, 2: n <= 32)
    int FR = 1024;            // complex elements per fragD matrix = 256 NT^2
    double invfact[24];       // 1/j!
    cplx* HfD = nullptr;      // [k+1] fragD(-i dt H), zero padded
    cplx* HfT = nullptr;      // [k+1] fragD((-i dt H)^T)
    cplx* U0fD = nullptr;     // fragD(U0), zero padded
    cplx* KfD = nullptr;      // [B][steps] fragD(K_t)
    cplx* KfT = nullptr;      // [B][steps] fragD(K_t^T); only when store_T (the 16x16x4 forward sweep reads it)
    bool store_T = true;      // false: NT = 2 sweeps on the 4x4x4 kernels, which gather K^T operands from KfD
    cplx* PfD = nullptr;      // [B][C] fragD(P_c)
    cplx* PfT = nullptr;      // [B][C] fragD(P_c^T)
    cplx* Aoff = nullptr;     // [B][C] affine offsets a_c of the backward recursion (D-layout column block, 512 cplx)
    cplx* LamD = nullptr;     // NT > 2: [B][steps][16 NT rows][16 columns] costates for the slice-parallel gradient kernel
    size_t grad_lds = 0;
    size_t bwd_lds = 0, bwd_lds2 = 0, bwd_lds3 = 0;
    bool h_in_lds = true, h_in_lds2 = true;
    int variant = 0;              // qoc_config.variant: 0 auto, 1 16x16x4, 2 4x4x4 two waves, 3 4x4x4 one wave
    int skew_c = 0, skew_b = 0;   // element skews per chunk / per seed that break the power-of-two strides of K storage
};

// element offset of K_t of seed b in KfD / KfT: consecutive slices are 16 KB apart; concurrent wavefronts differ in
// (seed, chunk), whose natural strides (L*16 KB, steps*16 KB) are powers of two for the usual sizes and alias HBM channels
__device__ __forceinline__ size_t kitem(const QocMfma& mf, int steps, int b, int t) {
    return (size_t)b * ((size_t)steps * mf.FR + (size_t)mf.C * mf.skew_c + mf.skew_b) + (size_t)t * mf.FR + (size_t)(t / mf.L) * mf.skew_c;
}

// ---- fragment helpers ---------------------------------------------------------------------------------------------

// A-operand fragments from a fragD matrix (pass fragD(M^T) to multiply by M, fragD(M) with CONJ to multiply by M^dagger)
template <int NT, bool CONJ>
__device__ __forceinline__ void afrag_load(const cplx* __restrict__ F, int lane, AFragT<NT>& A) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int q = 0; q < QQS; ++q) {
            const cplx v = F[(I * QQS + q) * 64 + lane];
            A.re[I][q] = v.x; A.im[I][q] = CONJ ? -v.y : v.y;
        }
}
// D-layout column block J from / to a fragD matrix
template <int NT>
__device__ __forceinline__ void colblock_load(const cplx* __restrict__ F, int J, int lane, CTile p[NT]) {
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const cplx v = F[(J * QQS + 4 * Ib + r) * 64 + lane];
            p[Ib].re[r] = v.x; p[Ib].im[r] = v.y;
        }
}
template <int NT>
__device__ __forceinline__ void colblock_store(cplx* __restrict__ F, int J, int lane, const CTile p[NT]) {
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) F[(J * QQS + 4 * Ib + r) * 64 + lane] = cmake(p[Ib].re[r], p[Ib].im[r]);
}
// the I = J half of an A-layout matrix is the J-th half of fragD(M^T)
template <int NT>
__device__ __forceinline__ void afrag_store_half(cplx* __restrict__ F, int J, int lane, const AFragT<NT>& A) {
#pragma unroll
    for (int q = 0; q < QQS; ++q) {
        double re = A.re[0][q], im = A.im[0][q];
#pragma unroll
        for (int Jc = 1; Jc < NT; ++Jc)
            if (Jc == J) { re = A.re[Jc][q]; im = A.im[Jc][q]; }
        F[(J * QQS + q) * 64 + lane] = cmake(re, im);
    }
}
template <int NT>
__device__ __forceinline__ void colblock_identity(int J, int lane, CTile p[NT]) {
    const int dlt = (lane & 15) - (lane >> 4);
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p[Ib].re[r] = (Ib == J && dlt == 4 * r) ? 1.0 : 0.0; p[Ib].im[r] = 0.0;
        }
}

// out[I] = sum_k A[I,k] * p[k] for one 16-column block, 3-multiplication complex arithmetic:
// 12 NT^2 MFMAs (48 for NT = 2), 3 NT independent accumulator chains.
template <int NT>
__device__ __forceinline__ void mm_colblock(const AFragT<NT>& A, const CTile p[NT], CTile out[NT]) {
    d4 a[NT], b[NT], c[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) { a[I] = (d4){0, 0, 0, 0}; b[I] = (d4){0, 0, 0, 0}; c[I] = (d4){0, 0, 0, 0}; }
#pragma unroll
    for (int q = 0; q < QQS; ++q) {
        const double br = p[q >> 2].re[q & 3], bi = p[q >> 2].im[q & 3], bs = br + bi;
#pragma unroll
        for (int I = 0; I < NT; ++I) a[I] = QMFMA(A.re[I][q], br, a[I]);
#pragma unroll
        for (int I = 0; I < NT; ++I) b[I] = QMFMA(A.im[I][q], bi, b[I]);
#pragma unroll
        for (int I = 0; I < NT; ++I) c[I] = QMFMA(A.re[I][q] + A.im[I][q], bs, c[I]);
    }
#pragma unroll
    for (int I = 0; I < NT; ++I) { out[I].re = a[I] - b[I]; out[I].im = c[I] - a[I] - b[I]; }
}

// Write a column block into the transposed LDS image img[col][row] (leading dimension QOC_LDR, complex).
template <int NT>
__device__ __forceinline__ void lds_put_colblock(cplx* img, int Jcol0, int lane, const CTile p[NT]) {
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            img[(Jcol0 + (lane & 15)) * QLDR + 16 * Ib + (lane >> 4) + 4 * r] = cmake(p[Ib].re[r], p[Ib].im[r]);
}
// Read the A-layout fragments of the full 32x32 matrix held in the transposed image.
template <int NT>
__device__ __forceinline__ void lds_get_afrag(const cplx* img, int lane, AFragT<NT>& A) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int q = 0; q < QQS; ++q) {
            const cplx v = img[(4 * q + (lane >> 4)) * QLDR + 16 * I + (lane & 15)];
            A.re[I][q] = v.x; A.im[I][q] = v.y;
        }
}

// The same product with v_mfma_f64_4x4x4_4b_f64 (17 cycles per 512 flops; the 16x16x4 shape issues every 103 cycles per 2048), left operand read block by block from the
// transposed LDS image (lane 16k+4b+i reads M[4ib+i][4kb+k], the 4 block lanes b share the address), right operand and
// result in the usual strip registers (a strip = 4 rows x 16 columns = one register of a CTile).
template <int NT>
__device__ __forceinline__ void mm_colblock4(const cplx* img, int lane, const CTile p[NT], CTile out[NT]) {
    double a[QQS], b[QQS], c[QQS];
#pragma unroll
    for (int s = 0; s < QQS; ++s) { a[s] = 0.0; b[s] = 0.0; c[s] = 0.0; }
    const cplx* base = img + (lane >> 4) * QLDR + (lane & 3);
    // blocks in (kb, ib) order through a 4-slot ring, three block steps (9 MFMAs) ahead -- see mm_full4
    constexpr int NS = QQS * QQS, RING = 4;
    cplx vb[RING];
    auto fetch = [&](int st, int slot) { vb[slot] = base[4 * (st / QQS) * QLDR + 4 * (st % QQS)]; };
#pragma unroll
    for (int st = 0; st < RING - 1; ++st) fetch(st, st);
    double br = 0.0, bi = 0.0, bs = 0.0;
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int kb = st / QQS, ib = st % QQS;
        if (st + RING - 1 < NS) fetch(st + RING - 1, (st + RING - 1) % RING);
        asm volatile("" ::: "memory");
        if (ib == 0) { br = p[kb >> 2].re[kb & 3]; bi = p[kb >> 2].im[kb & 3]; bs = br + bi; }
        const cplx v = vb[st % RING];
        a[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br, a[ib], 0, 0, 0);
        b[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi, b[ib], 0, 0, 0);
        c[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x + v.y, bs, c[ib], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < QQS; ++s) { out[s >> 2].re[s & 3] = a[s] - b[s]; out[s >> 2].im[s & 3] = c[s] - a[s] - b[s]; }
}
// one 16-column half (I = J) of the A-layout fragments, for the fragD(M^T) store
template <int NT>
__device__ __forceinline__ void lds_store_fragT_half(const cplx* img, cplx* __restrict__ F, int J, int lane) {
#pragma unroll
    for (int q = 0; q < QQS; ++q) F[(J * QQS + q) * 64 + lane] = img[(4 * q + (lane >> 4)) * QLDR + 16 * J + (lane & 15)];
}

// ---- kernel E: K_t = matexp for every t of one chunk + chunk product P_c ---------------------------------------
// One workgroup = 2 waves = the two 16-column halves of the matrices of chunk (b, c).
template <int NT>
__global__ void __launch_bounds__(64 * NT, (NT <= 2 ? 2 : 1)) k_mfma_expm_chunk(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[2][QNP * QLDR];
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;   // a finished seed keeps the results of its last evaluation (whole workgroup: no barrier yet)
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);     // identity: tile Ib == J, register r, lanes with dlt == 4r
    int flip = 0;
    CTile R[NT];
    colblock_identity<NT>(J, lane, R);
    for (int t = t0; t < t1; ++t) {
        // ---- A_t = (H0' + sum_k u_k H_k') / 2^s : left-operand fragments + this wave's column block -----------
        AFragT<NT> A;
        CTile P[NT];
        {
            afrag_load<NT, false>(mf.HfT, lane, A);
            colblock_load<NT>(mf.HfD, J, lane, P);
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int q = 0; q < QQS; ++q) { A.re[I][q] *= inv_scale; A.im[I][q] *= inv_scale; }
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib) { P[Ib].re *= inv_scale; P[Ib].im *= inv_scale; }
#pragma unroll 1
            for (int kk = 0; kk < d.k; ++kk) {
                const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
                const cplx* __restrict__ HT = mf.HfT + (size_t)(kk + 1) * QFR;
                const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
                for (int I = 0; I < NT; ++I)
#pragma unroll
                    for (int q = 0; q < QQS; ++q) {
                        const cplx h = HT[(I * QQS + q) * 64 + lane];
                        A.re[I][q] = fma(ck, h.x, A.re[I][q]);
                        A.im[I][q] = fma(ck, h.y, A.im[I][q]);
                    }
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                        P[Ib].re[r] = fma(ck, h.x, P[Ib].re[r]);
                        P[Ib].im[r] = fma(ck, h.y, P[Ib].im[r]);
                    }
            }
        }
        // ---- order-T Taylor polynomial sum_{j<=T} A^j/j! (tensorflow_state.py:37-41), Paterson-Stockmeyer form in
        //      A2 = A*A with blocks B_i = c_{2i} I + c_{2i+1} A:  S = B_m ; S = B_i + A2*S  -> 1 + ceil(T/2) - 1 products
        //      instead of T-1 (T=5: 3 instead of 4).  c_j = 1/j! from mf.invfact.
        if (d.T >= 2) {
            CTile AJ[NT];
            for (int Ib = 0; Ib < NT; ++Ib) AJ[Ib] = P[Ib];
            CTile A2J[NT];
            mm_colblock<NT>(A, AJ, A2J);
            lds_put_colblock<NT>(img[flip], 16 * J, lane, A2J);
            __syncthreads();
            lds_get_afrag<NT>(img[flip], lane, A);                      // A now holds the left-operand fragments of A2
            flip ^= 1;
            const int mm = d.T >> 1;
            int i;
            if ((d.T & 1) == 0) {                                   // top block is c_T I: S = B_{m-1} + c_T A2
                const double c0 = mf.invfact[2 * mm - 2], c1 = mf.invfact[2 * mm - 1], cT = mf.invfact[d.T];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + cT * A2J[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + cT * A2J[Ib].im[r];
                    }
                i = mm - 2;
            } else {                                                // S = B_m = c_{2m} I + c_{2m+1} A
                const double c0 = mf.invfact[2 * mm], c1 = mf.invfact[2 * mm + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r];
                    }
                i = mm - 1;
            }
            for (; i >= 0; --i) {
                CTile acc[NT];
                mm_colblock<NT>(A, P, acc);
                const double c0 = mf.invfact[2 * i], c1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + acc[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + acc[Ib].im[r];
                    }
            }
        } else {                                                    // T == 1: I + A
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[Ib].re[r] += (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
        }
        // ---- squaring: M <- M*M, s times (:43-44); the left operand comes back through the LDS image -----------
        for (int sq = 0; sq < d.s; ++sq) {
            lds_put_colblock<NT>(img[flip], 16 * J, lane, P);
            __syncthreads();
            lds_get_afrag<NT>(img[flip], lane, A);
            flip ^= 1;
            CTile acc[NT];
            mm_colblock<NT>(A, P, acc);
            for (int Ib = 0; Ib < NT; ++Ib) P[Ib] = acc[Ib];
        }
        // ---- K_t out (both operand forms); running chunk product R <- K_t R ----------------------------------------
        const size_t item = kitem(mf, d.steps, b, t);
        colblock_store<NT>(mf.KfD + item, J, lane, P);
        lds_put_colblock<NT>(img[flip], 16 * J, lane, P);
        __syncthreads();
        lds_get_afrag<NT>(img[flip], lane, A);
        flip ^= 1;
        if (mf.store_T) afrag_store_half<NT>(mf.KfT + item, J, lane, A);
        CTile acc[NT];
        mm_colblock<NT>(A, R, acc);
        for (int Ib = 0; Ib < NT; ++Ib) R[Ib] = acc[Ib];
    }
    const size_t pitem = (size_t)b * mf.C + c;
    colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R);
    lds_put_colblock<NT>(img[flip], 16 * J, lane, R);
    __syncthreads();
    AFragT<NT> A;
    lds_get_afrag<NT>(img[flip], lane, A);
    afrag_store_half<NT>(mf.PfT + pitem * QFR, J, lane, A);
}

// Two-wave variant on the 4x4x4 instruction (qoc_config.variant = 2): k_mfma_expm_chunk with every product done by mm_colblock4.  No A-operand fragments exist any
// more: A_t is assembled in strip layout only (half the Hamiltonian loads) and every left operand is an LDS image.
template <int NT>
__global__ void __launch_bounds__(64 * NT, (NT <= 2 ? 2 : 1)) k_mfma_expm_chunk4(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[2][QNP * QLDR];
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    int flip = 0;
    CTile R[NT];
    colblock_identity<NT>(J, lane, R);
    for (int t = t0; t < t1; ++t) {
        CTile P[NT];
        colblock_load<NT>(mf.HfD, J, lane, P);
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib) { P[Ib].re *= inv_scale; P[Ib].im *= inv_scale; }
#pragma unroll 1
        for (int kk = 0; kk < d.k; ++kk) {
            const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
            const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                    P[Ib].re[r] = fma(ck, h.x, P[Ib].re[r]);
                    P[Ib].im[r] = fma(ck, h.y, P[Ib].im[r]);
                }
        }
        if (d.T >= 2) {
            CTile AJ[NT];
            for (int Ib = 0; Ib < NT; ++Ib) AJ[Ib] = P[Ib];
            lds_put_colblock<NT>(img[flip], 16 * J, lane, AJ);
            __syncthreads();
            CTile A2J[NT];
            mm_colblock4<NT>(img[flip], lane, AJ, A2J);
            flip ^= 1;
            lds_put_colblock<NT>(img[flip], 16 * J, lane, A2J);
            __syncthreads();
            const cplx* a2img = img[flip];                          // stays valid through the Horner steps (no put until then)
            flip ^= 1;
            const int mm = d.T >> 1;
            int i;
            if ((d.T & 1) == 0) {
                const double c0 = mf.invfact[2 * mm - 2], c1 = mf.invfact[2 * mm - 1], cT = mf.invfact[d.T];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + cT * A2J[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + cT * A2J[Ib].im[r];
                    }
                i = mm - 2;
            } else {
                const double c0 = mf.invfact[2 * mm], c1 = mf.invfact[2 * mm + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r];
                    }
                i = mm - 1;
            }
            for (; i >= 0; --i) {
                CTile acc[NT];
                mm_colblock4<NT>(a2img, lane, P, acc);
                const double c0 = mf.invfact[2 * i], c1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + acc[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + acc[Ib].im[r];
                    }
            }
        } else {
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[Ib].re[r] += (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
        }
        // img[flip] is the buffer the A image lived in: every wave passed the barrier after reading it
        for (int sq = 0; sq < d.s; ++sq) {
            lds_put_colblock<NT>(img[flip], 16 * J, lane, P);
            __syncthreads();
            CTile acc[NT];
            mm_colblock4<NT>(img[flip], lane, P, acc);
            flip ^= 1;
            for (int Ib = 0; Ib < NT; ++Ib) P[Ib] = acc[Ib];
        }
        const size_t item = kitem(mf, d.steps, b, t);
        colblock_store<NT>(mf.KfD + item, J, lane, P);
        lds_put_colblock<NT>(img[flip], 16 * J, lane, P);
        __syncthreads();
        if (mf.store_T) lds_store_fragT_half<NT>(img[flip], mf.KfT + item, J, lane);
        CTile acc[NT];
        mm_colblock4<NT>(img[flip], lane, R, acc);
        flip ^= 1;
        for (int Ib = 0; Ib < NT; ++Ib) R[Ib] = acc[Ib];
    }
    const size_t pitem = (size_t)b * mf.C + c;
    colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R);
    lds_put_colblock<NT>(img[flip], 16 * J, lane, R);
    __syncthreads();
    lds_store_fragT_half<NT>(img[flip], mf.PfT + pitem * QFR, J, lane);
}

// DEFAULT for NT = 2 batches: one WAVE per (seed, chunk) owning all NT column blocks: every block load of the left operand
// feeds 3*NT MFMAs instead of 3, re+im comes pre-summed from a second image (no VALU in the product loop), and there is no
// workgroup barrier at all (a wave's LDS operations execute in order).  C2 x 64: 0.92 ms per launch = 72.6 TFLOP/s algorithmic.
template <int NT>
__device__ __forceinline__ void mm_full4(const cplx* img, const double* imgs, int lane, const CTile (&p)[NT][NT], CTile (&out)[NT][NT]) {
    double a[NT][QQS], b[NT][QQS], c[NT][QQS];
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int s = 0; s < QQS; ++s) { a[J][s] = 0.0; b[J][s] = 0.0; c[J][s] = 0.0; }
    const cplx* base = img + (lane >> 4) * QLDR + (lane & 3);
    const double* bases = imgs + (lane >> 4) * QLDR + (lane & 3);
    // 4x4 blocks of the left operand in (kb, ib) order through a 3-slot register ring, fetched TWO block steps (12 MFMAs,
    // ~200 cycles) ahead of their use: left to itself hipcc issues each ds_read one step ahead and the wave -- alone on
    // its SIMD -- stalls on LDS latency before every group of MFMAs.  The compiler fence after each fetch pins the order.
    constexpr int NS = QQS * QQS;
    cplx vb[3]; double sb[3];                                 // 5 slots (4 steps ahead) measured no better: 0.862 vs 0.855 ms
    auto fetch = [&](int st, int slot) {
        const int kb = st / QQS, ib = st % QQS;
        vb[slot] = base[4 * kb * QLDR + 4 * ib];
        sb[slot] = bases[4 * kb * QLDR + 4 * ib];             // re + im, summed once by the writer of the image
    };
    fetch(0, 0);
    fetch(1, 1);
    double br[NT], bi[NT], bs[NT];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int kb = st / QQS, ib = st % QQS;
        if (st + 2 < NS) fetch(st + 2, (st + 2) % 3);
        asm volatile("" ::: "memory");
        if (ib == 0) {
#pragma unroll
            for (int J = 0; J < NT; ++J) { br[J] = p[J][kb >> 2].re[kb & 3]; bi[J] = p[J][kb >> 2].im[kb & 3]; bs[J] = br[J] + bi[J]; }
        }
        const cplx v = vb[st % 3];
        const double vs = sb[st % 3];
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            a[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br[J], a[J][ib], 0, 0, 0);
            b[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi[J], b[J][ib], 0, 0, 0);
            c[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, bs[J], c[J][ib], 0, 0, 0);
        }
    }
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int s = 0; s < QQS; ++s) { out[J][s >> 2].re[s & 3] = a[J][s] - b[J][s]; out[J][s >> 2].im[s & 3] = c[J][s] - a[J][s] - b[J][s]; }
}
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NT>
__global__ void __launch_bounds__(64, 1) k_mfma_expm_chunk4w(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[QNP * QLDR];
    __shared__ __attribute__((aligned(16))) double imgs[QNP * QLDR];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    CTile R[NT][NT];
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_identity<NT>(J, lane, R[J]);
    auto put_all = [&](const CTile (&m)[NT][NT]) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            lds_put_colblock<NT>(img, 16 * J, lane, m[J]);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    imgs[(16 * J + (lane & 15)) * QLDR + 16 * Ib + (lane >> 4) + 4 * r] = m[J][Ib].re[r] + m[J][Ib].im[r];
        }
        wave_lds_fence();
    };
    for (int t = t0; t < t1; ++t) {
        CTile P[NT][NT];
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            colblock_load<NT>(mf.HfD, J, lane, P[J]);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib) { P[J][Ib].re *= inv_scale; P[J][Ib].im *= inv_scale; }
        }
#pragma unroll 1
        for (int kk = 0; kk < d.k; ++kk) {
            const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
            const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                        P[J][Ib].re[r] = fma(ck, h.x, P[J][Ib].re[r]);
                        P[J][Ib].im[r] = fma(ck, h.y, P[J][Ib].im[r]);
                    }
        }
        if (d.T >= 2) {
            CTile AJ[NT][NT], A2J[NT][NT];
#pragma unroll
            for (int J = 0; J < NT; ++J)
                for (int Ib = 0; Ib < NT; ++Ib) AJ[J][Ib] = P[J][Ib];
            put_all(AJ);
            mm_full4<NT>(img, imgs, lane, AJ, A2J);
            wave_lds_fence();
            put_all(A2J);
            const int mm = d.T >> 1;
            int i;
            double c0, c1, cT = 0.0;
            if ((d.T & 1) == 0) { c0 = mf.invfact[2 * mm - 2]; c1 = mf.invfact[2 * mm - 1]; cT = mf.invfact[d.T]; i = mm - 2; }
            else { c0 = mf.invfact[2 * mm]; c1 = mf.invfact[2 * mm + 1]; i = mm - 1; }
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[J][Ib].re[r] = one + c1 * AJ[J][Ib].re[r] + cT * A2J[J][Ib].re[r];
                        P[J][Ib].im[r] = c1 * AJ[J][Ib].im[r] + cT * A2J[J][Ib].im[r];
                    }
            for (; i >= 0; --i) {
                CTile acc[NT][NT];
                mm_full4<NT>(img, imgs, lane, P, acc);
                const double d0 = mf.invfact[2 * i], d1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double one = (Ib == J && dlt == 4 * r) ? d0 : 0.0;
                            P[J][Ib].re[r] = one + d1 * AJ[J][Ib].re[r] + acc[J][Ib].re[r];
                            P[J][Ib].im[r] = d1 * AJ[J][Ib].im[r] + acc[J][Ib].im[r];
                        }
            }
            wave_lds_fence();
        } else {
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) P[J][Ib].re[r] += (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
        }
        for (int sq = 0; sq < d.s; ++sq) {
            put_all(P);
            CTile acc[NT][NT];
            mm_full4<NT>(img, imgs, lane, P, acc);
            wave_lds_fence();
#pragma unroll
            for (int J = 0; J < NT; ++J)
                for (int Ib = 0; Ib < NT; ++Ib) P[J][Ib] = acc[J][Ib];
        }
        const size_t item = kitem(mf, d.steps, b, t);
#pragma unroll
        for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.KfD + item, J, lane, P[J]);
        put_all(P);
        if (mf.store_T) {
#pragma unroll
            for (int J = 0; J < NT; ++J) lds_store_fragT_half<NT>(img, mf.KfT + item, J, lane);
        }
        CTile acc[NT][NT];
        mm_full4<NT>(img, imgs, lane, R, acc);
        wave_lds_fence();
#pragma unroll
        for (int J = 0; J < NT; ++J)
            for (int Ib = 0; Ib < NT; ++Ib) R[J][Ib] = acc[J][Ib];
    }
    const size_t pitem = (size_t)b * mf.C + c;
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R[J]);
    put_all(R);
#pragma unroll
    for (int J = 0; J < NT; ++J) lds_store_fragT_half<NT>(img, mf.PfT + pitem * QFR, J, lane);
}

// ---- kernel F: thin forward sweep  Psi_t = K_t Psi_{t-1}  (inter vectors) + final unitary ------------------------
// grid.x = B*C sweep waves + B*2 final-unitary waves, 4 waves per workgroup, no LDS, no barriers.
template <int NT>
__global__ void __launch_bounds__(256) k_mfma_forward(QocDev d, QocMfma mf) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_sweep = d.B * mf.C;
    if (item < n_sweep) {
        const int b = item / mf.C, c = item - b * mf.C;
        if (d.skip_done && d.done[b]) return;
        const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
        CTile Psi[NT];
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
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
        AFragT<NT> A;
        for (int cc = 0; cc < c; ++cc) {                                // chunk boundary from the chunk products
            afrag_load<NT, false>(mf.PfT + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, Psi, acc);
            for (int Ib = 0; Ib < NT; ++Ib) Psi[Ib] = acc[Ib];
        }
        for (int t = t0; t < t1; ++t) {
            afrag_load<NT, false>(mf.KfT + kitem(mf, d.steps, b, t), lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, Psi, acc);
            for (int Ib = 0; Ib < NT; ++Ib) Psi[Ib] = acc[Ib];
            cplx* out = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                    if (row < d.n && col < d.m) out[row * d.m + col] = cmake(Psi[Ib].re[r], Psi[Ib].im[r]);
                }
        }
    } else if (item < n_sweep + d.B * NT) {
        // final_state = P_{C-1} ... P_0 U0 (tensorflow_state.py:223), one wave per 16-column half
        const int w = item - n_sweep, b = w / NT, J = w - b * NT;
        if (d.skip_done && d.done[b]) return;
        CTile X[NT];
        colblock_load<NT>(mf.U0fD, J, lane, X);
        AFragT<NT> A;
        for (int cc = 0; cc < mf.C; ++cc) {
            afrag_load<NT, false>(mf.PfT + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, X, acc);
            for (int Ib = 0; Ib < NT; ++Ib) X[Ib] = acc[Ib];
        }
        cplx* Xf = d.Xfinal + (size_t)b * d.n * d.n;
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = 16 * J + (lane & 15);
                if (row < d.n && col < d.n) Xf[row * d.n + col] = cmake(X[Ib].re[r], X[Ib].im[r]);
            }
    }
}

// ---- kernel F2: the thin forward sweep of NT = 2 on v_mfma_f64_4x4x4 ---------------------------------------------------
// Transposed recursion Psi_t^T = Psi_{t-1}^T K_t^T: the right operand (4 k-rows x 16 columns of K^T) is a fragD register of
// KfT as stored, the left operand a 4x4 block of Psi^T read from a wave-private LDS image (broadcast over the 4 blocks), the
// result register (I, jb) holds Psi[row 16 I + lane % 16][column 4 jb + lane / 16]: no output column is padding (a 16x16x4
// tile spends half of its columns on m = 8) -- 48 MQ MFMAs of 17 cycles per slice instead of 48 of ~100.  K_{t+1} is fetched
// while slice t multiplies.  Final-unitary waves as in k_mfma_forward.
#define F2_LDP 33
template <int NT, int MQ>
__global__ void __launch_bounds__(256) k_mfma_forward2(QocDev d, QocMfma mf) {
    constexpr int LDP = 16 * NT + 1;
    __shared__ __attribute__((aligned(16))) cplx f2_img[4][16 * LDP];             // per wave: image[column j][row]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wv;
    const int n_sweep = d.B * mf.C;
    if (item < n_sweep) {
        const int b = item / mf.C, c = item - b * mf.C;
        if (d.skip_done && d.done[b]) return;
        const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
        const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
        cplx* img = f2_img[wv];
        double pre[NT][MQ], pim[NT][MQ];
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const int row = 16 * I + lc, col = 4 * jb + lk;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) v = d.Psi0[row * d.m + col];
                pre[I][jb] = v.x; pim[I][jb] = v.y;
            }
        cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
        if (c == 0) {                                                   // inter[0] = V  (tensorflow_state.py:232-233)
            for (int o = lane; o < d.n * d.m; o += 64) iv[o] = d.V[o];
        }
        struct Frag { cplx f[NT][QQS]; };
        auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int q = 0; q < QQS; ++q)  // K^T[4q + lk][16 I + lc] = K[16 I + lc][4q + lk] gathered from fragD(K): quads of lanes (lk) read 64 contiguous bytes
                    fr.f[I][q] = F[((q >> 2) * QQS + 4 * I + (lc >> 2)) * 64 + 16 * (lc & 3) + 4 * (q & 3) + lk];
        };
        // Psi <- M Psi with M^T given by its fragD fragment
        auto product = [&](const Frag& fr) {
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) img[(4 * jb + lk) * LDP + 16 * I + lc] = cmake(pre[I][jb], pim[I][jb]);
            wave_lds_fence();
            double a[NT][MQ], bq[NT][MQ], cq[NT][MQ];
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) { a[I][jb] = 0.0; bq[I][jb] = 0.0; cq[I][jb] = 0.0; }
#pragma unroll
            for (int kb = 0; kb < QQS; ++kb) {
                cplx v[MQ];
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) v[jb] = img[(4 * jb + li4) * LDP + 4 * kb + lk];   // Psi[4 kb + lk][4 jb + li4]
#pragma unroll
                for (int I = 0; I < NT; ++I) {
                    const double br = fr.f[I][kb].x, bi = fr.f[I][kb].y, bs = br + bi;
#pragma unroll
                    for (int jb = 0; jb < MQ; ++jb) {
                        a[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x, br, a[I][jb], 0, 0, 0);
                        bq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].y, bi, bq[I][jb], 0, 0, 0);
                        cq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x + v[jb].y, bs, cq[I][jb], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) { pre[I][jb] = a[I][jb] - bq[I][jb]; pim[I][jb] = cq[I][jb] - a[I][jb] - bq[I][jb]; }
        };
        Frag A, A1;
        for (int cc = 0; cc < c; ++cc) {                                // chunk boundary from the chunk products
            load_frag(mf.PfD + ((size_t)b * mf.C + cc) * QFR, A);
            product(A);
        }
        auto step = [&](const Frag& fr, int t) {
            product(fr);
            cplx* out = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) {
                    const int row = 16 * I + lc, col = 4 * jb + lk;
                    if (row < d.n && col < d.m) out[row * d.m + col] = cmake(pre[I][jb], pim[I][jb]);
                }
        };
        const cplx* Kb = mf.KfD + kitem(mf, d.steps, b, t0);             // slices of one chunk are FR apart
        const int len = t1 - t0;
        load_frag(Kb, A);
        int t = 0;
        for (; t + 2 <= len; t += 2) {
            load_frag(Kb + (size_t)(t + 1) * mf.FR, A1); asm volatile("" ::: "memory"); step(A, t0 + t);
            load_frag(Kb + (size_t)min(t + 2, len - 1) * mf.FR, A); asm volatile("" ::: "memory"); step(A1, t0 + t + 1);
        }
        if (t < len) step(A, t0 + t);
    } else if (item < n_sweep + d.B * NT) {
        // final_state = P_{C-1} ... P_0 U0 (tensorflow_state.py:223), one wave per 16-column half
        const int w = item - n_sweep, b = w / NT, J = w - b * NT;
        if (d.skip_done && d.done[b]) return;
        CTile X[NT];
        colblock_load<NT>(mf.U0fD, J, lane, X);
        AFragT<NT> A;
        for (int cc = 0; cc < mf.C; ++cc) {
            afrag_load<NT, false>(mf.PfT + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, X, acc);
            for (int Ib = 0; Ib < NT; ++Ib) X[Ib] = acc[Ib];
        }
        cplx* Xf = d.Xfinal + (size_t)b * d.n * d.n;
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
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

// ---- kernel B0: affine offsets of the backward recursion when state regularisers add a source at every slice --------
// Lambda_{t-1} = K_t^dagger Lambda_t + S_{t-1} is affine; over chunk c it maps the chunk-end costate E to
// P_c^dagger E + a_c with a_c = result of running the chunk from a ZERO costate.  One wave per (seed, chunk >= 1).
template <int NT>
__global__ void __launch_bounds__(256) k_mfma_bwd_offsets(QocDev d, QocMfma mf) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (item >= d.B * mf.C) return;
    const int b = item / mf.C, c = item - b * mf.C;
    if (c == 0 || (d.skip_done && d.done[b])) return;                                  // a_0 is never used; finished seeds are frozen
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    CTile Z[NT];
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib) { Z[Ib].re = (d4){0, 0, 0, 0}; Z[Ib].im = (d4){0, 0, 0, 0}; }
    AFragT<NT> A;
    for (int t = t1 - 1; t >= t0; --t) {
        afrag_load<NT, true>(mf.KfD + kitem(mf, d.steps, b, t), lane, A);
        CTile acc[NT];
        mm_colblock<NT>(A, Z, acc);
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                cplx sv = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) sv = source_at(d, b, t, row, col);
                Z[Ib].re[r] = acc[Ib].re[r] + sv.x; Z[Ib].im[r] = acc[Ib].im[r] + sv.y;
            }
    }
    colblock_store<NT>(mf.Aoff + ((size_t)b * mf.C + c) * (QQS * 64), 0, lane, Z);
}

// ---- kernel B0': the affine offsets of NT = 2 on v_mfma_f64_4x4x4 ------------------------------------------------------
// Same recursion as k_mfma_bwd_offsets (Z <- K_t^dagger Z + S_t from a zero costate, one wave per (seed, chunk >= 1)) in the
// transposed form of k_mfma_forward2: Z^T <- Z^T conj(K_t), right operand = the fragD(K) registers as stored (contiguous loads),
// left operand = 4x4 blocks of Z^T from a wave-private LDS image.  The source term has conditional loads (waited for on the
// spot), so it is evaluated before the next K_t is fetched.  163 -> ~115 us per launch at the regularised C2 x 64.
template <int MQ, bool FULL>
__global__ void __launch_bounds__(256) k_mfma_bwd_offsets2(QocDev d, QocMfma mf) {
    constexpr int NT = 2;
    __shared__ __attribute__((aligned(16))) cplx o2_img[4][16 * F2_LDP];          // per wave: image[column j][row]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wv;
    if (item >= d.B * mf.C) return;
    const int b = item / mf.C, c = item - b * mf.C;
    if ((!FULL && c == 0) || (d.skip_done && d.done[b])) return;                       // a_0 is never used; finished seeds are frozen
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    cplx* img = o2_img[wv];
    double zre[2][MQ], zim[2][MQ];                                                      // (I, jb): Z[16 I + lc][4 jb + lk]
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) { zre[I][jb] = 0.0; zim[I][jb] = 0.0; }
    struct Frag { cplx f[2][8]; };
    auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int q = 0; q < 8; ++q) fr.f[I][q] = F[(I * QQS + q) * 64 + lane];        // K[4q + lk][16 I + lc]
    };
    double sre[2][MQ], sim[2][MQ];
    auto source = [&](int t) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const int row = 16 * I + lc, col = 4 * jb + lk;
                cplx sv = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) sv = source_at(d, b, t, row, col);
                sre[I][jb] = sv.x; sim[I][jb] = sv.y;
            }
    };
    const bool need_src = d.n_forb > 0 || d.has_speed;
    auto step = [&](const Frag& fr) {                                                  // Z <- K^dagger Z + S
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) img[(4 * jb + lk) * F2_LDP + 16 * I + lc] = cmake(zre[I][jb], zim[I][jb]);
        wave_lds_fence();
        double a[2][MQ], bq[2][MQ], cq[2][MQ];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { a[I][jb] = 0.0; bq[I][jb] = 0.0; cq[I][jb] = 0.0; }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            cplx v[MQ];
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) v[jb] = img[(4 * jb + li4) * F2_LDP + 4 * kb + lk];   // Z[4 kb + lk][4 jb + li4]
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const double br = fr.f[I][kb].x, bi = -fr.f[I][kb].y, bs = br + bi;
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) {
                    a[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x, br, a[I][jb], 0, 0, 0);
                    bq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].y, bi, bq[I][jb], 0, 0, 0);
                    cq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x + v[jb].y, bs, cq[I][jb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { zre[I][jb] = a[I][jb] - bq[I][jb] + sre[I][jb]; zim[I][jb] = cq[I][jb] - a[I][jb] - bq[I][jb] + sim[I][jb]; }
    };
    const cplx* Kb = mf.KfD + kitem(mf, d.steps, b, t0);                 // slices of one chunk are FR apart
    const int len = t1 - t0;
    Frag A, A1;
    auto zero_src = [&]() {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { sre[I][jb] = 0.0; sim[I][jb] = 0.0; }
    };
    if (FULL) {
        // FULL: the costate sweep itself (k > 4 controls: the gradients are formed by k_mfma_grad from the stored Lambda_t).
        // Terminal costate -(2/m^2) z W (+ S_steps), then E_{cc-1} = P_cc^dagger E_cc + a_cc down to the end of this chunk.
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const int row = 16 * I + lc, col = 4 * jb + lk;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) {
                    v = cscale(cmul(z, d.W[row * d.m + col]), c0);
                    if (need_src) v = cadd(v, source_at(d, b, d.steps, row, col));
                }
                zre[I][jb] = v.x; zim[I][jb] = v.y;
            }
        for (int cc = mf.C - 1; cc > c; --cc) {
            if (need_src) {
                const cplx* ao = mf.Aoff + ((size_t)b * mf.C + cc) * (QQS * 64) + 16 * (lc & 3) + lk;
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int jb = 0; jb < MQ; ++jb) { const cplx o = ao[(4 * I + (lc >> 2)) * 64 + 4 * jb]; sre[I][jb] = o.x; sim[I][jb] = o.y; }
            } else {
                zero_src();
            }
            load_frag(mf.PfD + ((size_t)b * mf.C + cc) * QFR, A);
            step(A);
        }
    }
    auto store_lam = [&](int t) {                                        // LamD[b][t][row][16 columns]
        if (!FULL) return;
        cplx* lo = mf.LamD + ((size_t)b * d.steps + t) * (16 * NT * 16);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) lo[(16 * I + lc) * 16 + 4 * jb + lk] = cmake(zre[I][jb], zim[I][jb]);
    };
    auto src_or_zero = [&](int t) { if (need_src && t > 0) source(max(t, 1)); else zero_src(); };
    load_frag(Kb + (size_t)(len - 1) * mf.FR, A);
    int i = 0;                                                           // step i handles slice t = t1 - 1 - i
    for (; i + 2 <= len; i += 2) {
        src_or_zero(t1 - 1 - i); load_frag(Kb + (size_t)(len - 2 - i) * mf.FR, A1); asm volatile("" ::: "memory"); store_lam(t1 - 1 - i); step(A);
        src_or_zero(t1 - 2 - i); load_frag(Kb + (size_t)max(len - 3 - i, 0) * mf.FR, A); asm volatile("" ::: "memory"); store_lam(t1 - 2 - i); step(A1);
    }
    if (i < len) { src_or_zero(t1 - 1 - i); store_lam(t1 - 1 - i); step(A); }
    if (FULL) return;
    cplx* out = mf.Aoff + ((size_t)b * mf.C + c) * (QQS * 64);             // D-layout 16x16x4 column block 0
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)                                     // all 16 columns: the 16x16x4 backward kernels read the whole block
            out[(4 * I + (lc >> 2)) * 64 + 16 * (lc & 3) + 4 * jb + lk] = jb < MQ ? cmake(zre[I][jb < MQ ? jb : 0], zim[I][jb < MQ ? jb : 0]) : cmake(0.0, 0.0);
}

// ---- kernel B: thin backward sweep  Lambda_{t-1} = K_t^dagger Lambda_t  + control gradients ----------------------
// dL/du_{k,t} = Re sum_ab H_k'[a][b] Q_t[a][b],  Q_t = conj(Lambda_t) Psi_t^T  (rank-m outer product on the MFMA),
// which equals Re <Lambda_t, H_k' Psi_t> of the reference's matexp_op_grad (tensorflow_state.py:61-63).
// 4 waves per workgroup; LDS: fragD image of the k control Hamiltonians (shared) + one transposition pad per wave.
template <int NT, bool H_IN_LDS, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_mfma_backward(QocDev d, QocMfma mf, int single_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cplx* Hl = (cplx*)smem;                                                     // [k] fragD(H_k')
    cplx* pad = (cplx*)(smem + (H_IN_LDS ? (size_t)d.k * QFR * sizeof(cplx) : 0)) + (size_t)wv * 16 * QLDR;
    if (H_IN_LDS) {
        for (int o = threadIdx.x; o < d.k * QFR; o += blockDim.x) Hl[o] = mf.HfD[QFR + o];
        __syncthreads();
    }
    const cplx* Hsrc = H_IN_LDS ? Hl : (mf.HfD + QFR);
    const int CC = single_chunk ? 1 : mf.C;
    const int item = blockIdx.x * 4 + wv;
    if (item >= d.B * CC) return;
    const int b = item / CC, c = item - b * CC;
    if (d.skip_done && d.done[b]) return;
    const int t0 = single_chunk ? 0 : c * mf.L, t1 = single_chunk ? d.steps : min(t0 + mf.L, d.steps);
    const bool need_src = d.n_forb > 0 || d.has_speed;
    // terminal costate: -(2/m^2) z W (+ S_steps)
    CTile Lam[NT];
    {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
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
    AFragT<NT> A;
    if (!single_chunk) {
        for (int cc = mf.C - 1; cc > c; --cc) {                          // Lambda at the end of this chunk
            afrag_load<NT, true>(mf.PfD + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, Lam, acc);
            for (int Ib = 0; Ib < NT; ++Ib) Lam[Ib] = acc[Ib];
            if (need_src) {                                              // E_{cc-1} = P_cc^dagger E_cc + a_cc
                CTile off[NT];
                colblock_load<NT>(mf.Aoff + ((size_t)b * mf.C + cc) * (QQS * 64), 0, lane, off);
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib) { Lam[Ib].re += off[Ib].re; Lam[Ib].im += off[Ib].im; }
            }
        }
    }
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    for (int t = t1 - 1; t >= t0; --t) {
        if constexpr (SPLIT) {
            // costates only: Lambda_t goes to LamD[b][t][row][16 columns] and k_mfma_grad forms the gradients slice-parallel
            cplx* lam_out = mf.LamD + ((size_t)b * d.steps + t) * (16 * NT * 16);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lam_out[(16 * Ib + (lane >> 4) + 4 * r) * 16 + (lane & 15)] = cmake(Lam[Ib].re[r], Lam[Ib].im[r]);
        } else {
        // ---- Q = conj(Lambda_t) Psi_t^T, 3-multiplication form:  Qr = T1 + T2, Qi = T3 - T1 + T2 with
        //      T1 = Lr Pr, T2 = Li Pi, T3 = (Lr - Li)(Pr + Pi) ------------------------------------------------------
        lds_put_colblock<NT>(pad, 0, lane, Lam);                             // wave-private image: pad[j][row]
        double lr[NT][4], li[NT][4], pr[NT][4], pi[NT][4];
        const cplx* psi = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lr[I][q] = 0.0; li[I][q] = 0.0; pr[I][q] = 0.0; pi[I][q] = 0.0;
                if (q < mf.mq) {
                    const int row = 16 * I + (lane & 15), j = 4 * q + (lane >> 4);
                    const cplx lv = pad[j * QLDR + row];
                    lr[I][q] = lv.x; li[I][q] = lv.y;
                    if (row < d.n && j < d.m) {
                        const cplx pv = psi[row * d.m + j];
                        pr[I][q] = pv.x; pi[I][q] = pv.y;
                    }
                }
            }
        double g[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) g[kk] = 0.0;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int Jp = 0; Jp < NT; ++Jp) {
                d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < mf.mq) {
                        t1v = QMFMA(lr[I][q], pr[Jp][q], t1v);
                        t2v = QMFMA(li[I][q], pi[Jp][q], t2v);
                        t3v = QMFMA(lr[I][q] - li[I][q], pr[Jp][q] + pi[Jp][q], t3v);
                    }
                }
                const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (kk >= d.k) continue;
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = Hsrc[(size_t)kk * QFR + (Jp * QQS + 4 * I + r) * 64 + lane];
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
        }
        if (t == 0) break;
        // ---- Lambda_{t-1} = K_t^dagger Lambda_t (+ S_{t-1}) ---------------------------------------------------------
        afrag_load<NT, true>(mf.KfD + kitem(mf, d.steps, b, t), lane, A);
        CTile acc[NT];
        mm_colblock<NT>(A, Lam, acc);
        for (int Ib = 0; Ib < NT; ++Ib) Lam[Ib] = acc[Ib];
        if (need_src) {
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
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

// ---- kernel B2 (NT = 2): the backward sweep with every (seed, chunk) item split over a PAIR of waves by row tile ----------
// The one-wave-per-item kernel above runs one wave per SIMD and its dependent 16x16x4 MFMA chains issue every ~143 cycles;
// here wave h of a pair owns the 16-row tile h of Lambda: it forms the Q tiles (h, 0..1) of the gradient contraction and row
// tile h of K_t^dagger Lambda_t (half the MFMAs, half the K fragments), so 2 waves per SIMD are resident (~103-cycle issue)
// and each chain is half as long.  The pair exchanges tiles through its transposed LDS images (the same image that feeds the
// A operand of Q), double-buffered, one workgroup barrier per slice; all trip counts are uniform over the workgroup
// (4 items = 8 waves share one LDS image of the control Hamiltonians): inactive steps only take part in the barriers.
// Measured at C2 x 64: 259 vs 301 us per launch (prefetching the K fragments one slice ahead made it 283: not the bound).
#define B2_LDP 17
template <bool H_IN_LDS>
__global__ void __launch_bounds__(512) k_mfma_backward2(QocDev d, QocMfma mf) {
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wv & 1, pair = wv >> 1;
    cplx* Hl = (cplx*)smem;                                                     // [k] fragD(H_k')
    cplx* pads = (cplx*)(smem + (H_IN_LDS ? (size_t)d.k * QFR * sizeof(cplx) : 0));   // [8 waves][2 buffers][16 * B2_LDP]
    double* gpart = (double*)(pads + 8 * 2 * 16 * B2_LDP);                     // [4 pairs][2 buffers][8]
    if (H_IN_LDS) {
        for (int o = threadIdx.x; o < d.k * QFR; o += blockDim.x) Hl[o] = mf.HfD[QFR + o];
    }
    const cplx* Hsrc = H_IN_LDS ? Hl : (mf.HfD + QFR);
    cplx* mypad = pads + (size_t)wv * 2 * 16 * B2_LDP;
    const cplx* otherpad = pads + (size_t)(wv ^ 1) * 2 * 16 * B2_LDP;
    const int item = blockIdx.x * 4 + pair;
    const bool item_ok = item < d.B * mf.C;
    const int b = item_ok ? item / mf.C : 0, c = item_ok ? item - b * mf.C : 0;
    const bool active = item_ok && !(d.skip_done && d.done[b]);
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const bool need_src = d.n_forb > 0 || d.has_speed;
    const int lk = lane >> 4, lc = lane & 15;
    // own tile of the costate, D layout: register r <-> (row 16h + lk + 4r, column lc)
    d4 ore = {0, 0, 0, 0}, oim = {0, 0, 0, 0};
    if (active) {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * h + lk + 4 * r;
            cplx v = cmake(0.0, 0.0);
            if (row < d.n && lc < d.m) {
                v = cscale(cmul(z, d.W[row * d.m + lc]), c0);
                if (need_src) v = cadd(v, source_at(d, b, d.steps, row, lc));
            }
            ore[r] = v.x; oim[r] = v.y;
        }
    }
    int buf = 0;
    auto put_own = [&](int bf) {                                                 // image[col][row16]
#pragma unroll
        for (int r = 0; r < 4; ++r) mypad[(bf * 16 + lc) * B2_LDP + lk + 4 * r] = cmake(ore[r], oim[r]);
    };
    auto get_other = [&](int bf, d4& xre, d4& xim) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const cplx v = otherpad[(bf * 16 + lc) * B2_LDP + lk + 4 * r]; xre[r] = v.x; xim[r] = v.y; }
    };
    // row tile h of M^dagger * Lambda with M given as fragD(M): 24 MFMAs
    auto dagger_product = [&](const cplx* __restrict__ F, const d4& xre, const d4& xim) {
        d4 a = {0, 0, 0, 0}, bq = {0, 0, 0, 0}, cq = {0, 0, 0, 0};
        cplx fr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fr[q] = F[(h * QQS + q) * 64 + lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool own = (q >> 2) == h;
            const double br = own ? ore[q & 3] : xre[q & 3], bi = own ? oim[q & 3] : xim[q & 3];
            const double ar = fr[q].x, ai = -fr[q].y;
            a = QMFMA(ar, br, a);
            bq = QMFMA(ai, bi, bq);
            cq = QMFMA(ar + ai, br + bi, cq);
        }
        ore = a - bq; oim = cq - a - bq;
    };
    put_own(0);
    __syncthreads();
    // ---- costate at the end of this chunk: E_{cc-1} = P_cc^dagger E_cc + a_cc, uniform trip count ---------------------
    for (int cc = mf.C - 1; cc >= 1; --cc) {
        if (active && cc > c) {
            d4 xre, xim;
            get_other(buf, xre, xim);
            dagger_product(mf.PfD + ((size_t)b * mf.C + cc) * QFR, xre, xim);
            if (need_src) {
                const cplx* off = mf.Aoff + ((size_t)b * mf.C + cc) * (QQS * 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const cplx v = off[(4 * h + r) * 64 + lane]; ore[r] += v.x; oim[r] += v.y; }
            }
        }
        put_own(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    for (int i = 0; i < mf.L; ++i) {
        const int t = t1 - 1 - i;
        const bool live = active && t >= t0;
        double g[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) g[kk] = 0.0;
        if (live) {
            // ---- Q tiles (h, 0..1) = conj(Lambda_t)[rows of tile h] Psi_t^T and the contraction with H_k' -----------------
            double lr[4], li[4], pr[2][4], pi[2][4];
            const cplx* psi = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lr[q] = 0.0; li[q] = 0.0;
#pragma unroll
                for (int Jp = 0; Jp < 2; ++Jp) { pr[Jp][q] = 0.0; pi[Jp][q] = 0.0; }
                if (q < mf.mq) {
                    const int j = 4 * q + lk;
                    const cplx lv = mypad[(buf * 16 + j) * B2_LDP + lc];          // Lambda[16h + lc][j]
                    lr[q] = lv.x; li[q] = lv.y;
#pragma unroll
                    for (int Jp = 0; Jp < 2; ++Jp) {
                        const int row = 16 * Jp + lc;
                        if (row < d.n && j < d.m) { const cplx pv = psi[row * d.m + j]; pr[Jp][q] = pv.x; pi[Jp][q] = pv.y; }
                    }
                }
            }
#pragma unroll
            for (int Jp = 0; Jp < 2; ++Jp) {
                d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < mf.mq) {
                        t1v = QMFMA(lr[q], pr[Jp][q], t1v);
                        t2v = QMFMA(li[q], pi[Jp][q], t2v);
                        t3v = QMFMA(lr[q] - li[q], pr[Jp][q] + pi[Jp][q], t3v);
                    }
                }
                const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (kk >= d.k) continue;
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx hv = Hsrc[(size_t)kk * QFR + (Jp * QQS + 4 * h + r) * 64 + lane];
                        acc = fma(hv.x, qr[r], acc);
                        acc = fma(-hv.y, qi[r], acc);
                    }
                    g[kk] += acc;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                if (kk >= d.k) continue;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) g[kk] += __shfl_down(g[kk], off, 64);
            }
            if (h == 1 && lane == 0) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) gpart[(pair * 2 + buf) * 8 + kk] = g[kk];
            }
            // ---- Lambda_{t-1} = K_t^dagger Lambda_t (+ S_{t-1}) -----------------------------------------------------------
            if (t > 0) {
                d4 xre, xim;
                get_other(buf, xre, xim);
                dagger_product(mf.KfD + kitem(mf, d.steps, b, t), xre, xim);
                if (need_src) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * h + lk + 4 * r;
                        if (row < d.n && lc < d.m) { const cplx sv = source_at(d, b, t, row, lc); ore[r] += sv.x; oim[r] += sv.y; }
                    }
                }
            }
        }
        put_own(buf ^ 1);
        __syncthreads();
        if (live && h == 0 && lane == 0) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
                if (kk < d.k) d.dLdu[((size_t)b * d.k + kk) * d.steps + t] = g[kk] + gpart[(pair * 2 + buf) * 8 + kk];
        }
        buf ^= 1;
    }
}

// ---- kernel G: control gradients of all slices in parallel (n > 32) ---------------------------------------------------
// For NT = 3/4 the fragD images of the control Hamiltonians (36 / 64 KB each) no longer fit in LDS next to the transposition
// pads of the sweep, and a sweep that reads them from L2 at every slice is bound by that stream (2.0 of 6.9 ms at n = 48 x 64).
// The sweep (k_mfma_backward<NT, false, true>) therefore only propagates the costates and stores Lambda_t; this kernel, with
// nothing but the images of up to 4 controls in LDS, forms Q = conj(Lambda_t) Psi_t^T and dL/du_{k,t} = Re sum_ab H_k'[a,b] Q[a,b]
// for every (seed, slice) independently: one wave per slice, the next slice's operands fetched while this one multiplies.
template <int NT, int MQ>
__global__ void __launch_bounds__(256) k_mfma_grad(QocDev d, QocMfma mf) {
    constexpr int KG = NT >= 4 ? 2 : 4;                                         // control images per pass: 2 x 64 KB or 4 x 36 KB of LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Hl = (cplx*)smem;                                                     // [<= KG] fragD(H_k')
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lk = lane >> 4, lc = lane & 15;
    const int total = d.B * d.steps, stride = gridDim.x * 4;
    struct Ops { double lr[NT][MQ], li[NT][MQ], pr[NT][MQ], pi[NT][MQ]; };
    auto fetch = [&](Ops& o, int s) {
        s = min(s, total - 1);
        const int b = s / d.steps, t = s - b * d.steps;
        const cplx* lam = mf.LamD + (size_t)s * (16 * NT * 16);
        const cplx* psi = d.inter + ((size_t)b * (d.steps + 1) + t + 1) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                const int j = 4 * q + lk;
                const cplx lv = lam[(16 * I + lc) * 16 + j];
                // rows >= n / columns >= m: clamped, finite, unmasked (they meet zero columns of Lambda / zero padding of H')
                const cplx pv = psi[min(16 * I + lc, d.n - 1) * d.m + min(j, d.m - 1)];
                o.lr[I][q] = lv.x; o.li[I][q] = lv.y; o.pr[I][q] = pv.x; o.pi[I][q] = pv.y;
            }
        asm volatile("" ::: "memory");
    };
    for (int k0 = 0; k0 < d.k; k0 += KG) {                                   // controls in groups of <= KG images
        const int kn = min(KG, d.k - k0);
        __syncthreads();
        for (int o = threadIdx.x; o < kn * QFR; o += blockDim.x) Hl[o] = mf.HfD[(size_t)(1 + k0) * QFR + o];
        __syncthreads();
        auto contract = [&](const Ops& o, int s) {
            if (s >= total) return;
            const int b = s / d.steps, t = s - b * d.steps;
            if (d.skip_done && d.done[b]) return;
            double g[KG];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) g[kk] = 0.0;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int Jp = 0; Jp < NT; ++Jp) {
                    d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < MQ; ++q) {
                        t1v = QMFMA(o.lr[I][q], o.pr[Jp][q], t1v);
                        t2v = QMFMA(o.li[I][q], o.pi[Jp][q], t2v);
                        t3v = QMFMA(o.lr[I][q] - o.li[I][q], o.pr[Jp][q] + o.pi[Jp][q], t3v);
                    }
                    const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk) {
                        if (kk >= kn) continue;
                        double acc = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const cplx h = Hl[(size_t)kk * QFR + (Jp * QQS + 4 * I + r) * 64 + lane];
                            acc = fma(h.x, qr[r], acc);
                            acc = fma(-h.y, qi[r], acc);
                        }
                        g[kk] += acc;
                    }
                }
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                if (kk >= kn) continue;
                double v = g[kk];
                v += dpp_xor<1>(v); v += dpp_xor<2>(v); v += dpp_xor<4>(v); v += dpp_xor<8>(v);
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (lane == 0) d.dLdu[((size_t)b * d.k + k0 + kk) * d.steps + t] = v;
            }
        };
        Ops o0, o1;
        int s = blockIdx.x * 4 + wv;
        fetch(o0, s);
        for (; s < total; s += 2 * stride) {
            fetch(o1, s + stride); contract(o0, s);
            fetch(o0, s + 2 * stride); contract(o1, s + stride);
        }
    }
}

// ---- kernel B3: k_mfma_backward2 with every per-slice latency taken off the dependent chain --------------------------
// Same split (pair of waves per (seed, chunk), tile h of the costate each), same LDS exchange.  What changes:
//  * the slice loop is branch-free (finished / out-of-range steps run on clamped addresses and only their store is
//    masked), so hipcc keeps counted vmcnt waits, and the K_t^dagger fragment and Psi_t of the NEXT slice are fetched at the
//    top of each step into a second register set (2x unrolled rotation): backward2 exposed two HBM round trips per slice
//    (Psi before the Q tiles, K before the costate product: ~5 of its 7.5 us per slice);
//  * the chunk-boundary recursion prefetches P_{cc-1} the same way and computes every step unconditionally (select);
//  * the workgroup barrier orders LDS only (lds_barrier), so the prefetch stays in flight across it;
//  * control gradients: 16-lane DPP butterflies, the 4 row partials of both waves go through LDS and lane kk of wave h = 0
//    adds the 8 partials of control kk (was: six ds_bpermute levels per control).
// Used for k <= 4 controls without state regularisers (no per-slice source term); anything else keeps backward2.
template <int MQ, bool SRC>
__global__ void __launch_bounds__(512) k_mfma_backward3(QocDev d, QocMfma mf) {
    constexpr int NT = 2, KC = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wv & 1, pair = wv >> 1;
    cplx* Hl = (cplx*)smem;                                                     // [KC] fragD(H_k'), zero beyond k
    cplx* pads = Hl + (size_t)KC * QFR;                                         // [8 waves][2 buffers][16 * B2_LDP]
    double* gpart = (double*)(pads + 8 * 2 * 16 * B2_LDP);                     // [4 pairs][2 buffers][2 waves][4 rows][KC]
    for (int o = threadIdx.x; o < KC * QFR; o += blockDim.x) Hl[o] = o < d.k * QFR ? mf.HfD[QFR + o] : cmake(0.0, 0.0);
    // costate images of the pair: image[buffer][column j][row % 16] (row stride B2_LDP), rows 0..15 in pad_lo, 16..31 in pad_hi
    cplx* mypad = pads + (size_t)wv * 2 * 16 * B2_LDP;
    const cplx* pad_lo = pads + (size_t)(2 * pair) * 2 * 16 * B2_LDP;
    const cplx* pad_hi = pads + (size_t)(2 * pair + 1) * 2 * 16 * B2_LDP;
    const int item = blockIdx.x * 4 + pair;
    const bool item_ok = item < d.B * mf.C;
    const int b = item_ok ? item / mf.C : 0, c = item_ok ? item - b * mf.C : 0;
    const bool active = item_ok && !(d.skip_done && d.done[b]);
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    // own part of the costate as the D operand of v_mfma_f64_4x4x4 on the TRANSPOSED recursion
    //   Lambda_{t-1}^T = Lambda_t^T conj(K_t):  register jb, lane 16 i + 4 blk + j  <->  Lambda[row 16h + 4 blk + j][column 4 jb + i],
    // so that the right operand (4 k-rows x 16 columns of conj(K)) is a fragD register exactly as the 16x16x4 kernels store it,
    // the left operand is a 4x4 block of Lambda^T read from the LDS image (broadcast over blk), and no output column is padding
    // (a 16x16x4 tile spends half of its columns on m = 8): 24 MQ MFMAs of 17 cycles instead of 24 of ~100.
    double ore[MQ], oim[MQ];
    {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) {
            const int row = 16 * h + lc, col = 4 * jb + lk;
            cplx v = cmake(0.0, 0.0);
            if (row < d.n && col < d.m) {
                v = cscale(cmul(z, d.W[row * d.m + col]), c0);
                if (SRC) v = cadd(v, source_at(d, b, d.steps, row, col));
            }
            ore[jb] = v.x; oim[jb] = v.y;
        }
    }
    struct Frag { cplx f[8]; };
    struct PsiReg { double pr[2][MQ], pi[2][MQ]; cplx own[MQ]; cplx zt; };   // own / zt: inputs of the source term (SRC only)
    auto put_own = [&](int bf) {
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) mypad[(bf * 16 + 4 * jb + lk) * B2_LDP + lc] = cmake(ore[jb], oim[jb]);
    };
    auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
        for (int q = 0; q < 8; ++q) fr.f[q] = F[(h * QQS + q) * 64 + lane];
    };
    // (ore, oim) <- rows of tile h of M^dagger Lambda, M given by its fragD fragment, Lambda by the image `bf` of the pair
    auto dagger_product = [&](const Frag& fr, int bf, double (&nre)[MQ], double (&nim)[MQ]) {
        double a[MQ], bq[MQ], cq[MQ];
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) { a[jb] = 0.0; bq[jb] = 0.0; cq[jb] = 0.0; }
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const cplx* src = (kb < 4 ? pad_lo : pad_hi) + (size_t)bf * 16 * B2_LDP + 4 * (kb & 3) + lk;
            const double br = fr.f[kb].x, bi = -fr.f[kb].y, bs = br + bi;
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const cplx v = src[(4 * jb + li4) * B2_LDP];                  // Lambda[4 kb + lk][4 jb + li4]
                a[jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br, a[jb], 0, 0, 0);
                bq[jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi, bq[jb], 0, 0, 0);
                cq[jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x + v.y, bs, cq[jb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) { nre[jb] = a[jb] - bq[jb]; nim[jb] = cq[jb] - a[jb] - bq[jb]; }
    };
    put_own(0);
    lds_barrier();
    int buf = 0;
    // ---- costate at the end of this chunk: E_{cc-1} = P_cc^dagger E_cc, uniform trip count, result kept only while cc > c ----
    {
        const cplx* Pb = mf.PfD + (size_t)b * mf.C * QFR;
        Frag f0, f1;
        auto bstep = [&](const Frag& fr, int cc) {
            double nre[MQ], nim[MQ];
            cplx off[MQ];
            if (SRC) {                                               // E_{cc-1} = P_cc^dagger E_cc + a_cc; a_cc is a D-layout 16x16x4 column block
                const cplx* ao = mf.Aoff + ((size_t)b * mf.C + cc) * (QQS * 64) + (4 * h + (lc >> 2)) * 64 + 16 * (lc & 3) + lk;
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) off[jb] = ao[4 * jb];
            }
            dagger_product(fr, buf, nre, nim);
            if (SRC) {
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) { nre[jb] += off[jb].x; nim[jb] += off[jb].y; }
            }
            const bool keep = cc > c;
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { ore[jb] = keep ? nre[jb] : ore[jb]; oim[jb] = keep ? nim[jb] : oim[jb]; }
            put_own(buf ^ 1);
            lds_barrier();
            buf ^= 1;
        };
        if (mf.C > 1) load_frag(Pb + (size_t)(mf.C - 1) * QFR, f0);
        int cc = mf.C - 1;
        for (; cc >= 2; cc -= 2) {
            load_frag(Pb + (size_t)(cc - 1) * QFR, f1); asm volatile("" ::: "memory"); bstep(f0, cc);
            load_frag(Pb + (size_t)max(cc - 2, 1) * QFR, f0); asm volatile("" ::: "memory"); bstep(f1, cc - 1);
        }
        if (cc == 1) bstep(f0, 1);
    }
    // ---- slices of the chunk, last to first ------------------------------------------------------------------------------
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    const int prow0 = min(lc, d.n - 1), prow1 = min(16 + lc, d.n - 1);
    auto fetch = [&](Frag& fr, PsiReg& ps, int i) {                          // operands of step i (slice t = t1 - 1 - i), clamped
        const int t = max(t1 - 1 - i, 0);
        load_frag(mf.KfD + kitem(mf, d.steps, b, t), fr);
        const cplx* psi = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const int j = 4 * q + lk, jc = min(j, d.m - 1);
            // out-of-range (row >= n, column >= m) entries read a clamped, finite element and need no mask: they only meet
            // the zero columns of Lambda (j >= m) or the zero padding of H_k' (row >= n); a masked load would be made
            // conditional by hipcc and waited for on the spot, draining the K prefetch with it
            const cplx p0 = psi[prow0 * d.m + jc], p1 = psi[prow1 * d.m + jc];
            ps.pr[0][q] = p0.x; ps.pi[0][q] = p0.y;
            ps.pr[1][q] = p1.x; ps.pi[1][q] = p1.y;
            if (SRC) ps.own[q] = (psi - (size_t)d.n * d.m)[(h ? prow1 : prow0) * d.m + jc];    // Psi_t at this lane's costate entries
        }
        if (SRC) ps.zt = *(d.has_speed ? d.ztau + (size_t)b * (d.steps + 1) + t : d.zfin + b);
        asm volatile("" ::: "memory");
    };
    // Source S_t of the state regularisers at this lane's costate entries (row 16h + lc, column 4 jb + lk).  Undressed forbidden
    // levels and speed_up need only Psi_t at those same entries and one scalar per slice, which fetch() brings in with the other
    // operands (unconditional loads, no wait on the spot).  A dressed forbidden level needs a whole column of Psi_t and falls
    // back to source_at(): its loads are conditional (hipcc waits for them on the spot, draining vmcnt), so that call sits BEFORE
    // the next operands are fetched -- what is in flight then are this step's operands, which are needed now anyway.
    const bool fast_src = !d.forbid_dressed;
    cplx wown[MQ];
    const double speed_coef = (SRC && d.has_speed) ? -d.a_speed * d.su_resid[b] * 2.0 / ((double)d.m * (double)d.m) : 0.0;
    if (SRC) {
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) wown[jb] = d.W[min(16 * h + lc, d.n - 1) * d.m + min(4 * jb + lk, d.m - 1)];
    }
    auto source_fast = [&](const PsiReg& ps, int i, double (&fre)[MQ], double (&fim)[MQ]) {
        const int t = t1 - 1 - i, row = 16 * h + lc;
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) {
            const cplx phi = ps.own[jb];
            const double pop = phi.x * phi.x + phi.y * phi.y;
            double w = 0.0;
            for (int f = 0; f < d.n_forb; ++f) w += (row == d.forb_state[f]) ? 2.0 * d.forb_a[f] * pop : 0.0;
            cplx sv = cscale(phi, w);
            const cplx zw = cscale(cmul(ps.zt, wown[jb]), speed_coef);
            sv.x += d.has_speed ? zw.x : 0.0; sv.y += d.has_speed ? zw.y : 0.0;
            const bool ok = t > 0 && row < d.n && 4 * jb + lk < d.m;
            fre[jb] = ok ? sv.x : 0.0; fim[jb] = ok ? sv.y : 0.0;
        }
    };
    double sre[MQ], sim[MQ];
    auto source = [&](int i) {
        if (!SRC || fast_src) return;
        const int t = t1 - 1 - i, tc = max(t, 1);
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) {
            const int row = 16 * h + lc, col = 4 * jb + lk;
            cplx sv = cmake(0.0, 0.0);
            if (t > 0 && row < d.n && col < d.m) sv = source_at(d, b, tc, row, col);
            sre[jb] = sv.x; sim[jb] = sv.y;
        }
    };
    auto step = [&](const Frag& fr, const PsiReg& ps, int i) {
        const int t = t1 - 1 - i;
        const bool live = active && t >= t0;
        // ---- Q tiles (h, 0..1) = conj(Lambda_t)[rows of tile h] Psi_t^T and the contraction with H_k' ---------------------
        double lr[MQ], li[MQ];
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const cplx lv = mypad[(buf * 16 + 4 * q + lk) * B2_LDP + lc];       // Lambda[16h + lc][4q + lk]
            lr[q] = lv.x; li[q] = lv.y;
        }
        double g[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) g[kk] = 0.0;
#pragma unroll
        for (int Jp = 0; Jp < 2; ++Jp) {
            d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                t1v = QMFMA(lr[q], ps.pr[Jp][q], t1v);
                t2v = QMFMA(li[q], ps.pi[Jp][q], t2v);
                t3v = QMFMA(lr[q] - li[q], ps.pr[Jp][q] + ps.pi[Jp][q], t3v);
            }
            const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx hv = Hl[(size_t)kk * QFR + (Jp * QQS + 4 * h + r) * 64 + lane];
                    acc = fma(hv.x, qr[r], acc);
                    acc = fma(-hv.y, qi[r], acc);
                }
                g[kk] += acc;
            }
        }
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {                                       // sum over the 16 lanes of a DPP row
            g[kk] += dpp_xor<1>(g[kk]); g[kk] += dpp_xor<2>(g[kk]); g[kk] += dpp_xor<4>(g[kk]); g[kk] += dpp_xor<8>(g[kk]);
        }
        if (lc == 0) {
            double* gp = gpart + ((((size_t)pair * 2 + buf) * 2 + h) * 4 + lk) * KC;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) gp[kk] = g[kk];
        }
        // ---- Lambda_{t-1} = K_t^dagger Lambda_t ------------------------------------------------------------------------------
        dagger_product(fr, buf, ore, oim);
        if (SRC) {
            if (fast_src) source_fast(ps, i, sre, sim);
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { ore[jb] += sre[jb]; oim[jb] += sim[jb]; }
        }
        put_own(buf ^ 1);
        lds_barrier();
        if (live && h == 0 && lane < d.k) {
            const double* gp = gpart + ((size_t)pair * 2 + buf) * 2 * 4 * KC + lane;
            double sum = 0.0;
#pragma unroll
            for (int x = 0; x < 8; ++x) sum += gp[x * KC];
            d.dLdu[((size_t)b * d.k + lane) * d.steps + t] = sum;
        }
        buf ^= 1;
    };
    Frag k0, k1;
    PsiReg p0, p1;
    fetch(k0, p0, 0);
    for (int i = 0; i < mf.L; i += 2) {
        source(i);     fetch(k1, p1, i + 1); step(k0, p0, i);
        source(i + 1); fetch(k0, p0, i + 2); step(k1, p1, i + 1);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------

static inline bool qoc_mfma_supported(const QocDev& d) {
    return !d.state_transfer && d.n <= 64 && d.m <= 16 && d.k <= 8 && d.T >= 1 && d.T <= 22;
}

// host: fragD image of a zero-padded n x n matrix (transpose optionally) for NT tiles per dimension
static inline void qoc_to_fragD(const cplx* M, int n, bool transpose, cplx* F, int NT) {
    const int QS = 4 * NT;
    for (int f = 0; f < NT * QS; ++f)
        for (int l = 0; l < 64; ++l) {
            const int cb = f / QS, q = f - cb * QS;
            int row = 4 * q + (l >> 4), col = 16 * cb + (l & 15);
            if (transpose) { const int tmp = row; row = col; col = tmp; }
            cplx v; v.x = 0.0; v.y = 0.0;
            if (row < n && col < n) v = M[(size_t)row * n + col];
            F[f * 64 + l] = v;
        }
}

static inline int qoc_mfma_setup(QocMfma& mf, const QocDev& d, int chunks_req, const cplx* Hs_host,
                                 std::vector<void*>& allocs, std::string& msg) {
    const int NT = d.n <= 16 ? 1 : (d.n <= 32 ? 2 : (d.n <= 48 ? 3 : 4));
    const int FR = 256 * NT * NT;
    mf.NT = NT; mf.FR = FR;
    int C = chunks_req;
    if (C <= 0) {
        // NT = 2: the default exponential kernel is ONE wave of 444 VGPRs per (seed, chunk), i.e. at most one resident wave per
        // SIMD: B*C must not exceed the 1024 SIMDs or a second, nearly empty round doubles the launch (48 seeds: C = 22 ->
        // 1056 items, 24.0k it/s; C = 21 -> 1008 items, 39.6k it/s).  Other NT: ~2 waves per SIMD.
        C = NT == 2 ? 1024 / d.B : (1024 + d.B - 1) / d.B;
        if (C > 32) C = 32;
    }
    if (C > d.steps) C = d.steps;
    if (C > QOC_MAXC) C = QOC_MAXC;
    if (C < 1) C = 1;
    int L = (d.steps + C - 1) / C;
    C = (d.steps + L - 1) / L;                       // no empty chunks
    mf.C = C; mf.L = L;
    mf.mq = (d.m + 3) / 4;
    {
        double f = 1.0;
        for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; mf.invfact[j] = 1.0 / f; }
    }
    std::vector<cplx> hd((size_t)(d.k + 1) * FR), ht((size_t)(d.k + 1) * FR), u0(FR);
    for (int kk = 0; kk <= d.k; ++kk) {
        qoc_to_fragD(Hs_host + (size_t)kk * d.n * d.n, d.n, false, hd.data() + (size_t)kk * FR, NT);
        qoc_to_fragD(Hs_host + (size_t)kk * d.n * d.n, d.n, true, ht.data() + (size_t)kk * FR, NT);
    }
    std::vector<cplx> u0h((size_t)d.n * d.n);
    if (hipMemcpy(u0h.data(), d.U0, u0h.size() * sizeof(cplx), hipMemcpyDeviceToHost) != hipSuccess) { msg = "U0 readback failed"; return -2; }
    qoc_to_fragD(u0h.data(), d.n, false, u0.data(), NT);
    auto up = [&](cplx** dst, const std::vector<cplx>& src) -> bool {
        void* p = nullptr;
        if (hipMalloc(&p, src.size() * sizeof(cplx)) != hipSuccess) return false;
        allocs.push_back(p);
        if (hipMemcpy(p, src.data(), src.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) return false;
        *dst = (cplx*)p;
        return true;
    };
    if (!up(&mf.HfD, hd) || !up(&mf.HfT, ht) || !up(&mf.U0fD, u0)) { msg = "MFMA path: constant upload failed"; return -3; }
    // the large buffers are carved out of ONE allocation (placement of separate hipMallocs after earlier engines of the process
    // were freed was worth a factor 2 on the GEMM path)
    std::vector<std::pair<cplx**, size_t>> wanted;
    auto al = [&](cplx** dst, size_t count) -> bool { wanted.emplace_back(dst, (count * sizeof(cplx) + 4095) & ~(size_t)4095); return true; };
    mf.skew_c = 5 * 16;                            // 1280 B per chunk
    mf.skew_b = 3 * 16;                            //  768 B per seed
    const size_t nk = (size_t)d.B * ((size_t)d.steps * FR + (size_t)C * mf.skew_c + mf.skew_b), np = (size_t)d.B * C * FR;
    mf.store_T = !((NT == 2 || NT == 3) && mf.variant != 1);     // the 4x4x4 forward sweep gathers K^T operands from fragD(K)
    const bool split_grad = (NT > 2 || (NT == 2 && d.k >= 6)) && mf.variant != 1;   // k = 5 still fits backward2's LDS (1.45 vs 1.49 ms)
    if (split_grad && !al(&mf.LamD, (size_t)d.B * d.steps * 16 * NT * 16)) { msg = "MFMA path: out of device memory"; return -3; }
    { const int kg = NT >= 4 ? 2 : 4; mf.grad_lds = (size_t)(d.k < kg ? d.k : kg) * FR * sizeof(cplx); }
    if (!al(&mf.KfD, nk) || (mf.store_T && !al(&mf.KfT, nk)) || !al(&mf.PfD, np) || !al(&mf.PfT, np) || !al(&mf.Aoff, (size_t)d.B * C * 4 * NT * 64)) { msg = "MFMA path: out of device memory"; return -3; }
    {
        size_t total = 0;
        for (auto& w : wanted) total += w.second;
        char* arena = nullptr;
        if (hipMalloc((void**)&arena, total) != hipSuccess) { msg = "MFMA path: out of device memory"; return -3; }
        allocs.push_back(arena);
        size_t off = 0;
        for (auto& w : wanted) { *w.first = (cplx*)(arena + off); off += w.second; }
    }
    const size_t pads = (size_t)4 * 16 * (16 * NT + 1) * sizeof(cplx);
    const size_t hbytes = (size_t)d.k * FR * sizeof(cplx);
    mf.h_in_lds = (hbytes + pads) <= 160 * 1024;
    mf.bwd_lds = pads + (mf.h_in_lds ? hbytes : 0);
    // row-split kernel (NT = 2): 8 waves x 2 image buffers x 16 x 17 cplx + the pair mailboxes of the gradient partials
    const size_t pads2 = (size_t)8 * 2 * 16 * B2_LDP * sizeof(cplx) + 4 * 2 * 8 * sizeof(double);
    mf.h_in_lds2 = (hbytes + pads2) <= 160 * 1024;
    mf.bwd_lds2 = pads2 + (mf.h_in_lds2 ? hbytes : 0);
    if (NT == 2 && mf.h_in_lds2 &&
        hipFuncSetAttribute((const void*)k_mfma_backward2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds2) != hipSuccess) {
        msg = "MFMA path: cannot reserve LDS for the row-split backward kernel";
        return -2;
    }
    // prefetching row-split kernel (NT = 2, k <= 4, no state regularisers): 4 control images + the pads + row partials
    mf.bwd_lds3 = (size_t)4 * FR * sizeof(cplx) + (size_t)8 * 2 * 16 * B2_LDP * sizeof(cplx) + 4 * 2 * 2 * 4 * 4 * sizeof(double);
    if (NT == 2 && (hipFuncSetAttribute((const void*)k_mfma_backward3<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess ||
                    hipFuncSetAttribute((const void*)k_mfma_backward3<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess ||
                    hipFuncSetAttribute((const void*)k_mfma_backward3<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess ||
                    hipFuncSetAttribute((const void*)k_mfma_backward3<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess)) {
        msg = "MFMA path: cannot reserve LDS for the prefetching backward kernel";
        return -2;
    }
    if (split_grad) {
        const void* gk = NT == 2 ? (mf.mq <= 2 ? (const void*)k_mfma_grad<2, 2> : (const void*)k_mfma_grad<2, 4>)
                       : NT == 3 ? (mf.mq <= 2 ? (const void*)k_mfma_grad<3, 2> : (const void*)k_mfma_grad<3, 4>)
                                 : (mf.mq <= 2 ? (const void*)k_mfma_grad<4, 2> : (const void*)k_mfma_grad<4, 4>);
        if (hipFuncSetAttribute(gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.grad_lds) != hipSuccess) { msg = "MFMA path: cannot reserve LDS for the gradient kernel"; return -2; }
    }
    if (mf.h_in_lds) {
        const hipError_t e1 = hipFuncSetAttribute((const void*)k_mfma_backward<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        const hipError_t e2 = hipFuncSetAttribute((const void*)k_mfma_backward<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        const hipError_t e3 = hipFuncSetAttribute((const void*)k_mfma_backward<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        const hipError_t e4 = hipFuncSetAttribute((const void*)k_mfma_backward<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        if ((NT == 1 ? e1 : (NT == 2 ? e2 : (NT == 3 ? e3 : e4))) != hipSuccess) {
            msg = "MFMA path: cannot reserve LDS for the backward kernel";
            return -2;
        }
    }
    return 0;
}

// which kernel computes the exponentials (1 = 16x16x4, 2 = 4x4x4 two waves, 3 = 4x4x4 one wave)
static inline int qoc_mfma_expm_variant(const QocMfma& mf, const QocDev& d) {
    if (mf.NT > 2) return mf.variant == 1 ? 1 : 2;      // n > 32: NT waves per item on 4x4x4 (n = 48 x 64: 4.3 vs 11.9 ms per launch)
    return mf.variant > 0 ? mf.variant : ((mf.NT == 2 && d.B * mf.C >= 512) ? 3 : 1);
}

template <int NT>
static inline void qoc_mfma_launch_all_expm(QocMfma& mf, const QocDev& d, hipStream_t s) {
    // AUTO: NT = 2 with at least half of the 1024 SIMDs busy -> one wave per (seed, chunk) on v_mfma_f64_4x4x4 (0.92 vs 1.21 ms
    // per launch at C2 x 64); NT = 1 and small launches keep the 16x16x4 kernel (C1: 0.072 vs 0.074 ms; one C2 trajectory:
    // 0.67 vs 0.75 ms).  qoc_config.variant forces one of the three kernels (parity tests, A/B runs).
    const int v = qoc_mfma_expm_variant(mf, d);
    if (v == 3 && NT <= 2) hipLaunchKernelGGL(k_mfma_expm_chunk4w<NT>, dim3(d.B * mf.C), dim3(64), 0, s, d, mf);
    else if (v == 2) hipLaunchKernelGGL(k_mfma_expm_chunk4<NT>, dim3(d.B * mf.C), dim3(64 * NT), 0, s, d, mf);
    else hipLaunchKernelGGL(k_mfma_expm_chunk<NT>, dim3(d.B * mf.C), dim3(64 * NT), 0, s, d, mf);
}
static inline void qoc_mfma_launch_expm(QocMfma& mf, const QocDev& d, hipStream_t s) {
    if (mf.NT == 1) qoc_mfma_launch_all_expm<1>(mf, d, s); else if (mf.NT == 2) qoc_mfma_launch_all_expm<2>(mf, d, s); else if (mf.NT == 3) qoc_mfma_launch_all_expm<3>(mf, d, s); else qoc_mfma_launch_all_expm<4>(mf, d, s);
}
static inline void qoc_mfma_launch_forward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    const int items = d.B * mf.C + d.B * mf.NT;
    if (mf.NT == 1) hipLaunchKernelGGL(k_mfma_forward<1>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    else if (mf.NT == 2 && mf.variant != 1) {
        // 4x4x4 sweep; like the backward choice this must not depend on the batch size (bit-identical seeds across shardings)
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_forward2<2, 2>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_forward2<2, 4>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    }
    else if (mf.NT == 3 && mf.variant != 1) {
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_forward2<3, 2>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_forward2<3, 4>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    }
    else if (mf.NT == 2) hipLaunchKernelGGL(k_mfma_forward<2>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    else if (mf.NT == 3) hipLaunchKernelGGL(k_mfma_forward<3>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    else hipLaunchKernelGGL(k_mfma_forward<4>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    if (!d.uscale_in_loss) hipLaunchKernelGGL(k_mfma_uscale, dim3(d.B), dim3(64), 0, s, d);
}
template <int NT>
static inline void qoc_mfma_launch_all_backward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    const int items = d.B * mf.C;
    if ((d.n_forb > 0 || d.has_speed) && mf.C > 1) {
        if (NT == 2 && mf.variant != 1) {
            if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_bwd_offsets2<2, false>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
            else hipLaunchKernelGGL((k_mfma_bwd_offsets2<4, false>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        } else {
            hipLaunchKernelGGL(k_mfma_bwd_offsets<NT>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        }
    }
    // NT = 2: each item split over a pair of waves (2 waves per SIMD for the batches AUTO sends here, 259 vs 301 us at C2 x 64).
    // The choice must not depend on the batch size: its gradient sums associate differently from the one-wave kernel, and a
    // seed has to evolve bit-identically whatever batch / GPU it is sharded into.  variant 1 keeps the one-wave kernel (A/B).
    if (NT == 2 && mf.variant != 1) {
        if (d.k <= 4) {
            const bool src = d.n_forb > 0 || d.has_speed;
            const dim3 g3((items + 3) / 4), b3(512);
            if (mf.mq <= 2) { if (src) hipLaunchKernelGGL((k_mfma_backward3<2, true>), g3, b3, mf.bwd_lds3, s, d, mf); else hipLaunchKernelGGL((k_mfma_backward3<2, false>), g3, b3, mf.bwd_lds3, s, d, mf); }
            else { if (src) hipLaunchKernelGGL((k_mfma_backward3<4, true>), g3, b3, mf.bwd_lds3, s, d, mf); else hipLaunchKernelGGL((k_mfma_backward3<4, false>), g3, b3, mf.bwd_lds3, s, d, mf); }
            return;
        }
        if (mf.LamD) {
            // k >= 6: the control images fit in LDS next to no sweep's pads; costate sweep + slice-parallel gradient kernel (4 images
            // per pass) instead of the row-split 16x16x4 sweep reading them from L2 (C2 x 64 with k = 8: 1.55 vs 1.71 ms per iteration)
            if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_bwd_offsets2<2, true>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
            else hipLaunchKernelGGL((k_mfma_bwd_offsets2<4, true>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
            const int slices = d.B * d.steps;
            int gg = (slices + 3) / 4; if (gg > 2048) gg = 2048;
            if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad<2, 2>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
            else hipLaunchKernelGGL((k_mfma_grad<2, 4>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
            return;
        }
        if (mf.h_in_lds2) hipLaunchKernelGGL((k_mfma_backward2<true>), dim3((items + 3) / 4), dim3(512), mf.bwd_lds2, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_backward2<false>), dim3((items + 3) / 4), dim3(512), mf.bwd_lds2, s, d, mf);
        return;
    }
    if (NT > 2 && mf.variant != 1) {
        // n > 32: the sweep only propagates the costates, the gradients are formed slice-parallel with the control images in LDS
        hipLaunchKernelGGL((k_mfma_backward<NT, false, true>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf, 0);   // no pad, no images
        const int slices = d.B * d.steps;
        int gg = (slices + 3) / 4; if (gg > 1024) gg = 1024;
        constexpr int GN = NT > 2 ? NT : 3;                                  // (never launched for NT <= 2)
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad<GN, 2>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_grad<GN, 4>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
        return;
    }
    if (mf.h_in_lds)
        hipLaunchKernelGGL((k_mfma_backward<NT, true>), dim3((items + 3) / 4), dim3(256), mf.bwd_lds, s, d, mf, 0);
    else
        hipLaunchKernelGGL((k_mfma_backward<NT, false>), dim3((items + 3) / 4), dim3(256), mf.bwd_lds, s, d, mf, 0);
}
static inline void qoc_mfma_launch_backward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    if (mf.NT == 1) qoc_mfma_launch_all_backward<1>(mf, d, s); else if (mf.NT == 2) qoc_mfma_launch_all_backward<2>(mf, d, s); else if (mf.NT == 3) qoc_mfma_launch_all_backward<3>(mf, d, s); else qoc_mfma_launch_all_backward<4>(mf, d, s);
}
