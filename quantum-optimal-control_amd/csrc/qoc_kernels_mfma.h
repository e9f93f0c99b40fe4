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
//
// Files: qoc_mfma_frag.h (layouts, helpers); kernels in qoc_mfma_expm.h / qoc_mfma_forward.h / qoc_mfma_backward.h, each compiled in its
// own translation unit (qoc_mfma_expm.hip, qoc_mfma_forward.hip, qoc_mfma_backward.hip: the three compile in parallel) behind the
// host entry points declared below.
#pragma once
#include "qoc_mfma_frag.h"

// ---- host side ----------------------------------------------------------------------------------------------------

// State transfer (round 4): matvecexp (tensorflow_state.py:77-97) applies sum_{j < T} A^j / j! to the vectors, i.e. the propagator is the Taylor polynomial of
// degree T - 1 without squarings; with exactly anti-Hermitian generators its adjoint is the polynomial of -A that the reference's gradient (:99-133) applies,
// so the MFMA path runs state transfer as K_t of that degree + the same thin sweeps (the caller checks the generators: qoc_all_antihermitian).
// qoc_create lowers QocDev::T to the degree before qoc_mfma_setup; these predicates see the caller's T.
static inline int qoc_mfma_degree(const QocDev& d) { return d.state_transfer ? d.T - 1 : d.T; }
static inline bool qoc_mfma_supported(const QocDev& d) {
    return d.n <= 64 && d.m <= 16 && d.k <= 8 && qoc_mfma_degree(d) >= 1 && qoc_mfma_degree(d) <= 22;
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

// which kernel computes the exponentials (1 = 16x16x4, 2 = 4x4x4 two waves, 3 = 4x4x4 one wave, image written before each product,
// 4 = 4x4x4 one wave, image written strip by strip under the product's own MFMAs, 5 = latency mode: two waves per
// slice (k_mfma_expm_slice2) + k_mfma_chain_rows, 6 = 4x4x4, two waves per item and two waves per SIMD, no sums image (k_mfma_expm_pair), 7 = n > 32: four
// waves per item with a block of rows each (k_mfma_expm_rows), 8 = 4x4x4 one wave, row-strip-major products with the image rewritten in place
// (k_mfma_expm_inplace, qoc_mfma_expm_inplace.h; AUTO for NT = 2 batches since round 3))
static inline int qoc_mfma_expm_variant(const QocMfma& mf, const QocDev& d) {
    // n > 32: four waves per item, a block of rows each (7; AUTO: n = 48 x 64 seeds 3.3 ms per launch against 3.8 for 2 = one wave per
    // 16-column block and 11.9 for 1 = the same on 16x16x4)
    if (mf.latency) return 5;
    if (mf.NT > 2) return mf.variant == 1 ? 1 : (mf.variant == 2 ? 2 : 7);
    // AUTO, NT = 2: the in-place kernel whatever the batch (few (seed, chunk) items only occur with pinned chunk counts -- AUTO gives
    // small batches to the latency mode --, and a kernel that does not depend on the local batch keeps a restart bit-identical under any sharding)
    int v = mf.variant > 0 ? mf.variant : (mf.NT == 2 ? ((d.T >= 3 || d.Bplan * mf.C >= 512) ? 8 : 1) : 1);
    if (v == 7) v = 4;                                                  // the row-block kernel is an NT = 3 / 4 kernel
    if (v == 8 && (d.T < 3 || mf.NT != 2)) v = 4;                      // the in-place kernel needs at least one Horner product (T >= 3)
    if (v == 4 && (d.T < 2 || mf.NT != 2)) v = 3;                      // the streamed kernel starts from the product A * A
    if (v == 6 && mf.NT != 2) v = 3;                                    // the pair kernel is an NT = 2 kernel
    return v;
}
// latency mode: NT = 2 kernels (smaller problems are padded to 32).  With a state regulariser the costate is not linear in the
// overlap: the backward half then runs the batch kernels on the latency mode's chunks (QocMfma::lat_sources)
static inline bool qoc_mfma_latency_ok(const QocDev& d) {
    if (d.n > 32)    // NT = 3 / 4: at most 4 DRESSED forbidden levels (more need the batch kernels' affine recursion, NT = 2 only); 32 < n <= 48 with
                     // more than 4 controls runs the NT = 4 kernels on the padded problem (NT = 3 keeps four control images in LDS)
        return d.n <= 64 && d.m <= 16 && d.k <= 8 && qoc_mfma_degree(d) >= 2 && qoc_mfma_degree(d) <= 22 && !(d.n_forb > 4 && d.forbid_dressed);
    return d.m <= 16 && d.k <= 8 && qoc_mfma_degree(d) >= 2 && qoc_mfma_degree(d) <= 22;
}

// host entry points (defined next to their kernels)
int qoc_mfma_setup(QocMfma& mf, const QocDev& d, int chunks_req, const cplx* Hs_host, std::vector<void*>& allocs, std::string& msg);   // qoc_mfma_backward.hip
void qoc_mfma_launch_expm(QocMfma& mf, const QocDev& d, hipStream_t s);       // qoc_mfma_expm.hip
void qoc_mfma_launch_expm_inplace(QocMfma& mf, const QocDev& d, hipStream_t s);   // qoc_mfma_expm_inplace.hip (variant 8)
void qoc_mfma_launch_forward(QocMfma& mf, const QocDev& d, hipStream_t s);    // qoc_mfma_forward.hip
void qoc_mfma_launch_backward(QocMfma& mf, const QocDev& d, hipStream_t s);   // qoc_mfma_backward.hip
void qoc_mfma_final_state(QocMfma& mf, const QocDev& d, hipStream_t s);       // qoc_mfma_expm.hip: latency mode, on read-back
void qoc_mfma_uscale_state_transfer(const QocDev& d, hipStream_t s);            // qoc_mfma_forward.hip: state transfer on those routes, on read-back
void qoc_mfma_final_state_batch(QocMfma& mf, const QocDev& d, hipStream_t s);   // qoc_mfma_forward.hip: k_mfma_downup batches, on read-back
void qoc_mfma_unpack_inter(QocMfma& mf, const QocDev& d, hipStream_t s);      // qoc_mfma_forward.hip: latency mode, on read-back
int qoc_mfma_latency_setup(QocMfma& mf, const QocDev& d, std::string& msg);    // qoc_mfma_latency.hip
void qoc_mfma_latency_sweeps(QocMfma& mf, const QocDev& d, hipStream_t s);    //   forward + z-free adjoint sweep in one launch (lat_sources: forward only, then d.inter unpacked)
void qoc_mfma_latency_gradient(QocMfma& mf, const QocDev& d, const QocAdamDev* fused_tail, hipStream_t s);  //   slice-parallel gradient, overlap z, loss
