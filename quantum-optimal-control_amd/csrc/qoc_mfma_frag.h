// qoc_mfma_frag.h -- MFMA path: fragment layouts, the QocMfma descriptor and the register / LDS product helpers shared by the
// exponential kernels (qoc_mfma_expm.h) and the thin sweeps (qoc_mfma_forward.h, qoc_mfma_backward.h).  Design notes: header
// comment of qoc_kernels_mfma.h and DESIGN.md, section 4.1.
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
#define F2_LDP 33                 // row stride of the per-wave vector images of the 4x4x4 sweeps (forward2, bwd_offsets2)
// NT = 2 kernels templated on the ACTIVE 4-row strips of the padded 32 x 32 matrices (rows / columns >= n are zero): ceil(n / 4), at least 5
static inline int qoc_active_strips(int n) { return n > 28 ? 8 : (n > 24 ? 7 : (n > 20 ? 6 : 5)); }
#define QOC_QA_SWITCH(qa, F) do { switch (qa) { case 5: F(5); break; case 6: F(6); break; case 7: F(7); break; default: F(8); break; } } while (0)
// latency mode (a smaller problem is padded to 32 there as well): also 2 and 4 strips for n <= 8 / n <= 16
static inline int qoc_active_strips_lat(int n) { return n > 16 ? qoc_active_strips(n) : (n > 8 ? 4 : 2); }
#define QOC_QA_SWITCH_LAT(qa, F) do { switch (qa) { case 2: F(2); break; case 4: F(4); break; case 5: F(5); break; case 6: F(6); break; case 7: F(7); break; default: F(8); break; } } while (0)

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
    double pcoef[24];         // in-place exponential kernel: coefficients of the MONIC polynomial in S = sigma A, pcoef[j] = sigma^-j / j!, sigma = (1/T!)^(1/T)
    double psigma = 1.0;      //   sigma
    cplx* HsD = nullptr;      // [k+1] fragD(-i dt H) * sigma / 2^s: the same kernel assembles S_t = sigma A_t straight from these
    cplx* HfD = nullptr;      // [k+1] fragD(-i dt H), zero padded
    cplx* HfT = nullptr;      // [k+1] fragD((-i dt H)^T)
    cplx* U0fD = nullptr;     // fragD(U0), zero padded
    cplx* KfD = nullptr;      // [B][steps] fragD(K_t)
    cplx* KfT = nullptr;      // [B][steps] fragD(K_t^T); only when store_T (the 16x16x4 forward sweep reads it)
    bool store_T = true;      // false: NT = 2 sweeps on the 4x4x4 kernels, which gather K^T operands from KfD
    cplx* PfD = nullptr;      // [B][C] fragD(P_c)
    cplx* PfT = nullptr;      // [B][C] fragD(P_c^T)
    cplx* BndF = nullptr;     // [B][C][NT * MQ][64] chunk-start vectors Psi (sweep register layout), k_mfma_bnd_scan; null: the sweeps walk the chunk products
    cplx* BndA = nullptr;     // [B][C][NT * MQ][64] z-free costates at the chunk ends (no state regulariser); null: the backward sweep walks
    bool updown = false;      // both sweeps in one kernel, adjoint first (k_mfma_downup): NT = 2, k <= 5, no state regulariser; Psi_t is not stored (LamL holds the costates)
    size_t du_lds = 0;
    cplx* Aoff = nullptr;     // [B][C] affine offsets a_c of the backward recursion (D-layout column block, 512 cplx)
    cplx* Goff = nullptr;     // [B][NG] the same for whole groups of chunks (latency mode with a state regulariser)
    cplx* LamD = nullptr;     // NT > 2: [B][steps][16 NT rows][16 columns] costates for the slice-parallel gradient kernel
    size_t grad_lds = 0;
    double* gpart = nullptr;  // NT > 2: [NT row tiles][B][k][steps] partial control gradients of k_mfma_grad_rt
    int grad_rt = 0;          // NT > 2: row-tile gradient kernel (k <= 8)
    size_t bwd_lds = 0, bwd_lds3 = 0;
    bool h_in_lds = true;
    int variant = 0;              // qoc_config.variant: 0 auto, 1 16x16x4, 2 4x4x4 two waves, 3 4x4x4 one wave, 4 streamed image, 5 latency mode
    // latency mode (few seeds, variant 5): K_t by two waves per SLICE, chunk products and products of groups of G chunks by
    // k_mfma_chain_rows, two-level chunk boundaries in the sweeps, final_state / unitary_scale formed only when read back
    int G = 0, NG = 0;            // chunks per group (0: no groups), number of groups
    cplx* GfD = nullptr;          // [B][NG] fragD(product of the chunk products of a group)
    cplx* TfD = nullptr;          // [B] fragD(P_{C-1} ... P_0 U0), formed on demand (qoc_mfma_final_state)
    cplx* PsiL = nullptr;         // [B][steps][NT * MQ][64] Psi after slice t in the sweeps' own register layout (row 16 I + lane % 16, column
                                  // 4 jb + lane / 16): lane-contiguous 1 KB stores / loads; d.inter (API layout) is unpacked from it on read-back
    cplx* LamL = nullptr;         // [B][steps][NT * MQ][64] z-free costate BEFORE K_t^dagger is applied, same layout (qoc_mfma_latency.h)
    cplx* LamS = nullptr;         // [B][steps][NT * MQ][64] total costate c0 z Lambda0 + LambdaS when a state regulariser is present (k_mfma_sweep_src)
    cplx* AoffL = nullptr;        // [B][C][NT * MQ][64] / GoffL [B][NG][...]: chunk and group offsets of the source recursion, register layout
    cplx* GoffL = nullptr;
    bool lat_dressed = false;     // lat_src_fast with dressed forbidden levels (<= 4): amplitudes formed by k_mfma_loss_lat<NT, true>, sources from QocDev::Fd
    // experimental switches, read once in qoc_mfma_setup (qoc_exp_env): the padded problem in full in k_mfma_expm_rows / k_mfma_expm_slice2, the chunk
    // offsets of the source recursion by their own launches
    bool exp_rows_qa_full = false, exp_lat_qa8 = false, exp_lat_offsets_own = false;
    bool lat_src_fast = false;    // lat_sources on the thin affine sweeps (undressed forbidden levels / speed_up, NT = 2); else the batch kernels' recursion
    double* loss_part = nullptr;  // [B][steps + 1][2] per-time-point partials of k_mfma_loss_lat
    unsigned* lat_count = nullptr; // [B] workgroups of k_mfma_grad_lat that have finished (the last one runs the tail of the iteration)
    cplx* GfT = nullptr;          // [B][NG] fragD(G_g^T): with KfT / PfT the lane-contiguous operands of the forward sweep in latency mode
    bool latency = false;
    bool lat_sources = false;     // latency mode with a state regulariser: exponentials, chains and the forward sweep as above, then the
                                  // affine costate recursion of the batch kernels (k_mfma_bwd_offsets2 + k_mfma_backward3<SRC> / k_mfma_grad)
    int skew_c = 0, skew_b = 0;   // element skews per chunk / per seed that break the power-of-two strides of K storage
};

// element offset of K_t of seed b in KfD / KfT: consecutive slices are 16 KB apart; concurrent wavefronts differ in
// (seed, chunk), whose natural strides (L*16 KB, steps*16 KB) are powers of two for the usual sizes and alias HBM channels
__device__ __forceinline__ size_t kitem(const QocMfma& mf, int steps, int b, int t) {
    return (size_t)b * ((size_t)steps * mf.FR + (size_t)mf.C * mf.skew_c + mf.skew_b) + (size_t)t * mf.FR + (size_t)(t / mf.L) * mf.skew_c;
}

// ---- fragment helpers ---------------------------------------------------------------------------------------------

// A-operand fragments from a fragD matrix (pass fragD(M^T) to multiply by M, fragD(M) with CONJ to multiply by M^dagger)
// QA: the inner 4-index slices < QA only (the active ones of a padded problem, ceil(n / 4): the right operand's rows beyond are zero, so neither
// loaded here nor multiplied in mm_colblock<NT, QA>)
template <int NT, bool CONJ, int QA = 4 * NT>
__device__ __forceinline__ void afrag_load(const cplx* __restrict__ F, int lane, AFragT<NT>& A) {
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int q = 0; q < QA; ++q) {
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
template <int NT, int QA = 4 * NT>
__device__ __forceinline__ void mm_colblock(const AFragT<NT>& A, const CTile p[NT], CTile out[NT]) {
    d4 a[NT], b[NT], c[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) { a[I] = (d4){0, 0, 0, 0}; b[I] = (d4){0, 0, 0, 0}; c[I] = (d4){0, 0, 0, 0}; }
#pragma unroll
    for (int q = 0; q < QA; ++q) {
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
// the same image plus its re + im sums (left operand of the 3-multiplication products on v_mfma_f64_4x4x4)
template <int NT, bool SUMS>
__device__ __forceinline__ void lds_put_colblock_sum(cplx* img, double* imgs, int Jcol0, int lane, const CTile p[NT]) {
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = (Jcol0 + (lane & 15)) * QLDR + 16 * Ib + (lane >> 4) + 4 * r;
            img[o] = cmake(p[Ib].re[r], p[Ib].im[r]);
            if (SUMS) imgs[o] = p[Ib].re[r] + p[Ib].im[r];
        }
}
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
template <int NT, bool SUMS>
__device__ __forceinline__ void mm_colblock4(const cplx* img, const double* imgs, int lane, const CTile p[NT], CTile out[NT]) {
    double a[QQS], b[QQS], c[QQS];
#pragma unroll
    for (int s = 0; s < QQS; ++s) { a[s] = 0.0; b[s] = 0.0; c[s] = 0.0; }
    const cplx* base = img + (lane >> 4) * QLDR + (lane & 3);
    // blocks in (kb, ib) order through a 4-slot ring, three block steps (9 MFMAs) ahead -- see mm_full4
    constexpr int NS = QQS * QQS, RING = 4;
    const double* bases = imgs + (lane >> 4) * QLDR + (lane & 3);
    cplx vb[RING]; double sb[RING];
    auto fetch = [&](int st, int slot) { const int o = 4 * (st / QQS) * QLDR + 4 * (st % QQS); vb[slot] = base[o]; sb[slot] = SUMS ? bases[o] : 0.0; };
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
        c[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(SUMS ? sb[st % RING] : v.x + v.y, bs, c[ib], 0, 0, 0);
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

