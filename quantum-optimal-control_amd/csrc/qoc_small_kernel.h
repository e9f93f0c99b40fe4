// qoc_small_kernel.h -- k_small_iter: the workgroup-resident GRAPE iteration for n <= 12 (see qoc_small.h for the mapping).
//
// Reference semantics, file:line under /root/reference/quantum_optimal_control/:
//   controls u = maxA sin(base)                         core/tensorflow_state.py:176-178
//   K_t = (sum_{j<=T} A^j / j!)^(2^s), A = H_t / 2^s    core/tensorflow_state.py:25-46    (state transfer: sum_{j<T} B^j / j!, :77-97)
//   chain, inter vectors, fidelity                      core/tensorflow_state.py:204-242, 282-340
//   first-order gradient Re<Lambda_{t+1}, H_k Psi_{t+1}> core/tensorflow_state.py:49-65 (:100-133 in state transfer)
//   regularisers                                        core/regularization_functions.py:15-95 (the bandpass, :47-67, by a direct DFT when the pulse fits ONE workgroup)
//   grad_squared, TF1 Adam                              core/tensorflow_state.py:342-356
//   stop rule, learning-rate schedule                   core/run_session.py:47-69
//
// Time points: Psi_tau = inter_vecs[tau], tau = 0 .. steps (inter_vecs[0] = V is a constant); slice t maps Psi_t to Psi_{t+1}.  The costate is
// kept as its adjoint Y_tau = Lambda_tau^dagger (m x n), so that the backward recursion Y_tau = Y_{tau+1} K_tau + S_tau^dagger is a RIGHT
// multiplication by the propagator in the very register layout the exponential leaves it in (column per lane); the control gradients are
// the contraction of the rank-m product Psi_{t+1} Y_{t+1} with the transposed control Hamiltonians.
#pragma once
#include "qoc_small.h"

namespace qsm {

// QOC_SMALL_TIMING (tools/small_phase_timing.sh builds such a library; never the product): workgroup 0 prints the shader-clock stamps of its phase
// boundaries in the last iteration of a launch
#ifdef QOC_SMALL_TIMING
#define QSM_STAMP(i) do { if (it == sd.iters - 1 && blockIdx.x == 0 && threadIdx.x == 0) stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QSM_STAMP(i) do { } while (0)
#endif

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// ---- the product primitive -----------------------------------------------------------------------------------------------------------
// o += bcast_C(a) * x:  (a.x, a.y) of lane C of this row of 16 lanes reaches every lane inside the FMA (row_newbcast, the one DPP control
// double-precision VALU has on gfx90a+); four VOP2 instructions per complex MAC, two accumulator chains interleaved.
//
// Hazard: a DPP read needs 2 wait states behind a VALU write of its source, and the compiler pads nothing for an asm statement -- under register pressure it places a
// copy (v_mov_b64 from a spill register) of the broadcast operand directly in front of a statement (seen in the n = 8 and the state-regulariser builds: wrong gradients).
// EVERY statement therefore opens with s_nop 1; statements are made of up to sixteen FMAs (four rows of a column step, or four columns of a row step) so that this costs 2 - 3 %.
// The translation units of the n <= 4 instances define QOC_SMALL_DPP_PAD 0: their builds have no such copies, and tools/dpp_hazard_scan.py (run by
// tests/test_abi.py on every built object) proves it on the machine code -- a build where that stops being true fails the test instead of the physics.
#ifndef QOC_SMALL_DPP_PAD
#define QOC_SMALL_DPP_PAD 1
#endif
#if QOC_SMALL_DPP_PAD
#define QPAD "s_nop 1\n\t"
#else
#define QPAD
#endif
#define QF(d, a, x, c) "v_fmac_f64_dpp " d ", " a ", " x " row_newbcast:" c " row_mask:0xf bank_mask:0xf\n\t"
template <int C>
__device__ __forceinline__ void cmac_dpp(cplx& o, const cplx& a, const cplx& x) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%6") QF("%1", "%2", "%5", "%6") QF("%0", "-%3", "%5", "%6") QF("%1", "%3", "%4", "%6")
                 : "+v"(o.x), "+v"(o.y) : "v"(a.x), "v"(a.y), "v"(x.x), "v"(x.y), "n"(C));
}
// two rows: o0 += bcast_C(a0) x, o1 += bcast_C(a1) x
template <int C>
__device__ __forceinline__ void cmac2_dpp(cplx& o0, cplx& o1, const cplx& a0, const cplx& a1, const cplx& x) {
    asm volatile(QPAD QF("%0", "%4", "%8", "%10") QF("%1", "%4", "%9", "%10") QF("%2", "%6", "%8", "%10") QF("%3", "%6", "%9", "%10")
                 QF("%0", "-%5", "%9", "%10") QF("%1", "%5", "%8", "%10") QF("%2", "-%7", "%9", "%10") QF("%3", "%7", "%8", "%10")
                 : "+v"(o0.x), "+v"(o0.y), "+v"(o1.x), "+v"(o1.y) : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(x.x), "v"(x.y), "n"(C));
}
// three and four rows
template <int C>
__device__ __forceinline__ void cmac3_dpp(cplx& o0, cplx& o1, cplx& o2, const cplx& a0, const cplx& a1, const cplx& a2, const cplx& x) {
    asm volatile(QPAD QF("%0", "%6", "%12", "%14") QF("%1", "%6", "%13", "%14") QF("%2", "%8", "%12", "%14") QF("%3", "%8", "%13", "%14") QF("%4", "%10", "%12", "%14") QF("%5", "%10", "%13", "%14")
                 QF("%0", "-%7", "%13", "%14") QF("%1", "%7", "%12", "%14") QF("%2", "-%9", "%13", "%14") QF("%3", "%9", "%12", "%14") QF("%4", "-%11", "%13", "%14") QF("%5", "%11", "%12", "%14")
                 : "+v"(o0.x), "+v"(o0.y), "+v"(o1.x), "+v"(o1.y), "+v"(o2.x), "+v"(o2.y)
                 : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(a2.x), "v"(a2.y), "v"(x.x), "v"(x.y), "n"(C));
}
template <int C>
__device__ __forceinline__ void cmac4_dpp(cplx& o0, cplx& o1, cplx& o2, cplx& o3, const cplx& a0, const cplx& a1, const cplx& a2, const cplx& a3, const cplx& x) {
    asm volatile(QPAD QF("%0", "%8", "%16", "%18") QF("%1", "%8", "%17", "%18") QF("%2", "%10", "%16", "%18") QF("%3", "%10", "%17", "%18") QF("%4", "%12", "%16", "%18") QF("%5", "%12", "%17", "%18") QF("%6", "%14", "%16", "%18") QF("%7", "%14", "%17", "%18")
                 QF("%0", "-%9", "%17", "%18") QF("%1", "%9", "%16", "%18") QF("%2", "-%11", "%17", "%18") QF("%3", "%11", "%16", "%18") QF("%4", "-%13", "%17", "%18") QF("%5", "%13", "%16", "%18") QF("%6", "-%15", "%17", "%18") QF("%7", "%15", "%16", "%18")
                 : "+v"(o0.x), "+v"(o0.y), "+v"(o1.x), "+v"(o1.y), "+v"(o2.x), "+v"(o2.y), "+v"(o3.x), "+v"(o3.y)
                 : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(a2.x), "v"(a2.y), "v"(a3.x), "v"(a3.y), "v"(x.x), "v"(x.y), "n"(C));
}
// two columns of one row: o += bcast_C0(a) x0 + bcast_{C0+1}(a) x1
template <int C0>
__device__ __forceinline__ void cmac_row2(cplx& o, const cplx& a, const cplx& x0, const cplx& x1) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%8") QF("%1", "%2", "%5", "%8") QF("%0", "-%3", "%5", "%8") QF("%1", "%3", "%4", "%8")
                 QF("%0", "%2", "%6", "%9") QF("%1", "%2", "%7", "%9") QF("%0", "-%3", "%7", "%9") QF("%1", "%3", "%6", "%9")
                 : "+v"(o.x), "+v"(o.y) : "v"(a.x), "v"(a.y), "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "n"(C0), "n"(C0 + 1));
}
// four columns of one row
template <int C0>
__device__ __forceinline__ void cmac_row4(cplx& o, const cplx& a, const cplx& x0, const cplx& x1, const cplx& x2, const cplx& x3) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%12") QF("%1", "%2", "%5", "%12") QF("%0", "-%3", "%5", "%12") QF("%1", "%3", "%4", "%12") QF("%0", "%2", "%6", "%13") QF("%1", "%2", "%7", "%13") QF("%0", "-%3", "%7", "%13") QF("%1", "%3", "%6", "%13")
                 QF("%0", "%2", "%8", "%14") QF("%1", "%2", "%9", "%14") QF("%0", "-%3", "%9", "%14") QF("%1", "%3", "%8", "%14") QF("%0", "%2", "%10", "%15") QF("%1", "%2", "%11", "%15") QF("%0", "-%3", "%11", "%15") QF("%1", "%3", "%10", "%15")
                 : "+v"(o.x), "+v"(o.y) : "v"(a.x), "v"(a.y), "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "v"(x2.x), "v"(x2.y), "v"(x3.x), "v"(x3.y),
                   "n"(C0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3));
}
// o += conj(bcast_C(a) * x)
template <int C>
__device__ __forceinline__ void cmac_dpp_conj(cplx& o, const cplx& a, const cplx& x) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%6") QF("%1", "-%2", "%5", "%6") QF("%0", "-%3", "%5", "%6") QF("%1", "-%3", "%4", "%6")
                 : "+v"(o.x), "+v"(o.y) : "v"(a.x), "v"(a.y), "v"(x.x), "v"(x.y), "n"(C));
}

// one column step of the product: o[r] += A[r][C] x[C] for every r
template <int N, int C, int R0 = 0>
__device__ __forceinline__ void mulb_col(const cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N]) {
    if constexpr (R0 + 4 <= N) { cmac4_dpp<C>(o[R0], o[R0 + 1], o[R0 + 2], o[R0 + 3], A[R0], A[R0 + 1], A[R0 + 2], A[R0 + 3], x[C]); mulb_col<N, C, R0 + 4>(A, x, o); }
    else if constexpr (R0 + 3 == N) cmac3_dpp<C>(o[R0], o[R0 + 1], o[R0 + 2], A[R0], A[R0 + 1], A[R0 + 2], x[C]);
    else if constexpr (R0 + 2 == N) cmac2_dpp<C>(o[R0], o[R0 + 1], A[R0], A[R0 + 1], x[C]);
    else if constexpr (R0 < N) cmac_dpp<C>(o[R0], A[R0], x[C]);
}
template <int N, int C = 0>
__device__ __forceinline__ void mulb_acc(const cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N]) {
    if constexpr (C < N) {
        mulb_col<N, C>(A, x, o);
        mulb_acc<N, C + 1>(A, x, o);
    }
}
// one row of the product: o[R_] += sum_c A[R_][c] x[c]
template <int N, int R_, int C = 0>
__device__ __forceinline__ void mulb_row(const cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N]) {
    if constexpr (C + 4 <= N) { cmac_row4<C>(o[R_], A[R_], x[C], x[C + 1], x[C + 2], x[C + 3]); mulb_row<N, R_, C + 4>(A, x, o); }
    else if constexpr (C + 2 <= N) { cmac_row2<C>(o[R_], A[R_], x[C], x[C + 1]); mulb_row<N, R_, C + 2>(A, x, o); }
    else if constexpr (C < N) cmac_dpp<C>(o[R_], A[R_], x[C]);
}
// every register a DPP read may touch is defined before the product (the compiler cannot see the DPP read inside the asm); then the 5 wait states a DPP read needs behind
// a write of EXEC (a product may open right behind a divergent branch)
template <int N>
__device__ __forceinline__ void dpp_guard(cplx (&A)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) asm volatile("" : "+v"(A[r].x), "+v"(A[r].y));
    asm volatile("s_nop 4");
}
// o[r] = sum_c A[r][c] x[c]: lane c holds A[.][c] (its column of the left operand), every lane its own column x of the right operand.
// The same statement is  Y' = Y M  (A = the lanes' columns of Y, x = the own column of M)  and the rank-m product  Psi Y.
template <int N>
__device__ __forceinline__ void mulb(cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) o[r] = cmake(0.0, 0.0);
    dpp_guard<N>(A);
    mulb_acc<N>(A, x, o);
}

// The costate side has m rows (Y = Lambda^dagger is m x n, m <= n state vectors: a two-qutrit gate has m = 4 of n = 9): products whose LEFT operand is such a matrix only
// form the rows r < m (a wave-uniform branch per row), products that SUM over its rows (Psi Y) only take the lanes c < m.
template <int N, int R_ = 0>
__device__ __forceinline__ void mulb_rows_acc(const cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N], int m) {
    if constexpr (R_ < N) {
        if (R_ < m) mulb_row<N, R_>(A, x, o);
        mulb_rows_acc<N, R_ + 1>(A, x, o, m);
    }
}
template <int N>
__device__ __forceinline__ void mulb_m(cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N], int m) {       // rows r >= m of A are zero: o[r] = 0 there
#pragma unroll
    for (int r = 0; r < N; ++r) o[r] = cmake(0.0, 0.0);
    dpp_guard<N>(A);
    mulb_rows_acc<N>(A, x, o, m);
}
template <int N, int C = 0>
__device__ __forceinline__ void mulb_cols_acc(const cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N], int m) {
    if constexpr (C < N) {
        if (C < m) mulb_col<N, C>(A, x, o);
        mulb_cols_acc<N, C + 1>(A, x, o, m);
    }
}
template <int N>
__device__ __forceinline__ void mulb_k(cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N], int m) {       // x[c] = 0 for c >= m (and the lanes c >= m of A hold nothing)
#pragma unroll
    for (int r = 0; r < N; ++r) o[r] = cmake(0.0, 0.0);
    dpp_guard<N>(A);
    mulb_cols_acc<N>(A, x, o, m);
}

// ---- split form, 5 <= n <= 8 (half of a row of 16 lanes would idle) ---------------------------------------------------------------------
// Lanes j and j + 8 of a row both hold column j: lane j (half h = 0) forms the REAL parts of the product's column, lane j + 8 (h = 1) the imaginary parts -- two DPP
// FMAs per (r, c) instead of four -- and the halves hand each other their part with one row_ror:8 move per double.  A matrix in registers is (mine, other): .x = the
// part this half forms (h = 0: re, h = 1: im), .y = the other one; the lanes c < 8 that row_newbcast reads are all of half 0, where (mine, other) = (re, im).
//   re: sum_c a_re x_re - a_im x_im = sum_c a_re mine_c - a_im other_c        im: sum_c a_re x_im + a_im x_re = sum_c a_re mine_c + a_im other_c
#ifndef QOC_SMALL_SPLIT
#define QOC_SMALL_SPLIT 1
#endif
// p += bcast_C(a.x) x.x, q += bcast_C(a.y) x.y for one, two or four rows of a column step
template <int C>
__device__ __forceinline__ void smac1(double& p, double& q, const cplx& a, const cplx& x) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%6") QF("%1", "%3", "%5", "%6") : "+v"(p), "+v"(q) : "v"(a.x), "v"(a.y), "v"(x.x), "v"(x.y), "n"(C));
}
template <int C>
__device__ __forceinline__ void smac2(double& p0, double& q0, double& p1, double& q1, const cplx& a0, const cplx& a1, const cplx& x) {
    asm volatile(QPAD QF("%0", "%4", "%8", "%10") QF("%1", "%5", "%9", "%10") QF("%2", "%6", "%8", "%10") QF("%3", "%7", "%9", "%10")
                 : "+v"(p0), "+v"(q0), "+v"(p1), "+v"(q1) : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(x.x), "v"(x.y), "n"(C));
}
template <int C>
__device__ __forceinline__ void smac4(double& p0, double& q0, double& p1, double& q1, double& p2, double& q2, double& p3, double& q3,
                                      const cplx& a0, const cplx& a1, const cplx& a2, const cplx& a3, const cplx& x) {
    asm volatile(QPAD QF("%0", "%8", "%16", "%18") QF("%1", "%9", "%17", "%18") QF("%2", "%10", "%16", "%18") QF("%3", "%11", "%17", "%18")
                 QF("%4", "%12", "%16", "%18") QF("%5", "%13", "%17", "%18") QF("%6", "%14", "%16", "%18") QF("%7", "%15", "%17", "%18")
                 : "+v"(p0), "+v"(q0), "+v"(p1), "+v"(q1), "+v"(p2), "+v"(q2), "+v"(p3), "+v"(q3)
                 : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(a2.x), "v"(a2.y), "v"(a3.x), "v"(a3.y), "v"(x.x), "v"(x.y), "n"(C));
}
template <int C>
__device__ __forceinline__ void smac3(double& p0, double& q0, double& p1, double& q1, double& p2, double& q2, const cplx& a0, const cplx& a1, const cplx& a2, const cplx& x) {
    asm volatile(QPAD QF("%0", "%6", "%12", "%14") QF("%1", "%7", "%13", "%14") QF("%2", "%8", "%12", "%14") QF("%3", "%9", "%13", "%14") QF("%4", "%10", "%12", "%14") QF("%5", "%11", "%13", "%14")
                 : "+v"(p0), "+v"(q0), "+v"(p1), "+v"(q1), "+v"(p2), "+v"(q2) : "v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y), "v"(a2.x), "v"(a2.y), "v"(x.x), "v"(x.y), "n"(C));
}
// ... for two or four columns of a row step
template <int C0>
__device__ __forceinline__ void srow2(double& p, double& q, const cplx& a, const cplx& x0, const cplx& x1) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%8") QF("%1", "%3", "%5", "%8") QF("%0", "%2", "%6", "%9") QF("%1", "%3", "%7", "%9")
                 : "+v"(p), "+v"(q) : "v"(a.x), "v"(a.y), "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "n"(C0), "n"(C0 + 1));
}
template <int C0>
__device__ __forceinline__ void srow4(double& p, double& q, const cplx& a, const cplx& x0, const cplx& x1, const cplx& x2, const cplx& x3) {
    asm volatile(QPAD QF("%0", "%2", "%4", "%12") QF("%1", "%3", "%5", "%12") QF("%0", "%2", "%6", "%13") QF("%1", "%3", "%7", "%13")
                 QF("%0", "%2", "%8", "%14") QF("%1", "%3", "%9", "%14") QF("%0", "%2", "%10", "%15") QF("%1", "%3", "%11", "%15")
                 : "+v"(p), "+v"(q) : "v"(a.x), "v"(a.y), "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "v"(x2.x), "v"(x2.y), "v"(x3.x), "v"(x3.y),
                   "n"(C0), "n"(C0 + 1), "n"(C0 + 2), "n"(C0 + 3));
}
template <int N, int C, int R0 = 0>
__device__ __forceinline__ void smulb_col(const cplx (&A)[N], const cplx (&x)[N], double (&o1)[N], double (&o2)[N]) {
    if constexpr (R0 + 4 <= N) {
        smac4<C>(o1[R0], o2[R0], o1[R0 + 1], o2[R0 + 1], o1[R0 + 2], o2[R0 + 2], o1[R0 + 3], o2[R0 + 3], A[R0], A[R0 + 1], A[R0 + 2], A[R0 + 3], x[C]);
        smulb_col<N, C, R0 + 4>(A, x, o1, o2);
    } else if constexpr (R0 + 3 == N) smac3<C>(o1[R0], o2[R0], o1[R0 + 1], o2[R0 + 1], o1[R0 + 2], o2[R0 + 2], A[R0], A[R0 + 1], A[R0 + 2], x[C]);
    else if constexpr (R0 + 2 == N) smac2<C>(o1[R0], o2[R0], o1[R0 + 1], o2[R0 + 1], A[R0], A[R0 + 1], x[C]);
    else if constexpr (R0 < N) smac1<C>(o1[R0], o2[R0], A[R0], x[C]);
}
template <int N, int R_, int C = 0>
__device__ __forceinline__ void smulb_row(const cplx (&A)[N], const cplx (&x)[N], double (&o1)[N], double (&o2)[N]) {
    if constexpr (C + 4 <= N) { srow4<C>(o1[R_], o2[R_], A[R_], x[C], x[C + 1], x[C + 2], x[C + 3]); smulb_row<N, R_, C + 4>(A, x, o1, o2); }
    else if constexpr (C + 2 <= N) { srow2<C>(o1[R_], o2[R_], A[R_], x[C], x[C + 1]); smulb_row<N, R_, C + 2>(A, x, o1, o2); }
    else if constexpr (C < N) smac1<C>(o1[R_], o2[R_], A[R_], x[C]);
}
template <int N, int C = 0>
__device__ __forceinline__ void smulb_acc(const cplx (&A)[N], const cplx (&x)[N], double (&o1)[N], double (&o2)[N]) {
    if constexpr (C < N) {
        smulb_col<N, C>(A, x, o1, o2);
        smulb_acc<N, C + 1>(A, x, o1, o2);
    }
}
template <int N, int R_ = 0>
__device__ __forceinline__ void smulb_rows_acc(const cplx (&A)[N], const cplx (&x)[N], double (&o1)[N], double (&o2)[N], int m) {
    if constexpr (R_ < N) {
        if (R_ < m) smulb_row<N, R_>(A, x, o1, o2);
        smulb_rows_acc<N, R_ + 1>(A, x, o1, o2, m);
    }
}
template <int N, int C = 0>
__device__ __forceinline__ void smulb_cols_acc(const cplx (&A)[N], const cplx (&x)[N], double (&o1)[N], double (&o2)[N], int m) {
    if constexpr (C < N) {
        if (C < m) smulb_col<N, C>(A, x, o1, o2);
        smulb_cols_acc<N, C + 1>(A, x, o1, o2, m);
    }
}
// mine = o1 + sgn o2 (sgn = -1 in half 0, +1 in half 1); other = the partner lane's mine
template <int N>
__device__ __forceinline__ void smulb_finish(const double (&o1)[N], const double (&o2)[N], cplx (&o)[N], double sgn) {
#pragma unroll
    for (int r = 0; r < N; ++r) o[r].x = fma(sgn, o2[r], o1[r]);
#pragma unroll
    for (int r = 0; r < N; ++r) o[r].y = dpp_xor<8>(o[r].x);
}
// WHICH 0: full product, 1: rows r < m of A, 2: summation index c < m
template <int N, int WHICH>
__device__ __forceinline__ void smulb(cplx (&A)[N], const cplx (&x)[N], cplx (&o)[N], int m, double sgn) {
    double o1[N], o2[N];
#pragma unroll
    for (int r = 0; r < N; ++r) { o1[r] = 0.0; o2[r] = 0.0; }
    dpp_guard<N>(A);
    if constexpr (WHICH == 0) smulb_acc<N>(A, x, o1, o2);
    else if constexpr (WHICH == 1) smulb_rows_acc<N>(A, x, o1, o2, m);
    else smulb_cols_acc<N>(A, x, o1, o2, m);
    smulb_finish<N>(o1, o2, o, sgn);
}
// an element of a matrix in LDS (natural re, im) as (mine, other)
template <bool SPL>
__device__ __forceinline__ cplx ldm(const cplx* p, int h) {
    if constexpr (SPL) { const double* q = (const double*)p; return cmake(q[h], q[1 - h]); }
    else return *p;
}
// (mine, other) <-> (re, im): the same exchange in half 1, nothing in half 0
__device__ __forceinline__ cplx swp(const cplx& v, int h) { return h ? cmake(v.y, v.x) : v; }

// sum over the 16 lanes of a row (result in every lane): DPP moves on the VALU (qoc_common.h: dpp_xor)
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_xor<1>(v); v += dpp_xor<2>(v); v += dpp_xor<4>(v); v += dpp_xor<8>(v);
    return v;
}

// sum over the first lanes of a row that can hold data of an N-column matrix (result in those lanes; the others must hold zero): 1 .. 4 DPP steps
template <int N>
__device__ __forceinline__ double row_sum_n(double v) {
    v += dpp_xor<1>(v);
    if constexpr (N > 2) v += dpp_xor<2>(v);
    if constexpr (N > 4) v += dpp_xor<4>(v);
    if constexpr (N > 8) v += dpp_xor<8>(v);
    return v;
}
// ... over the 8 lanes of a half row
__device__ __forceinline__ double row_sum8(double v) {
    v += dpp_xor<1>(v); v += dpp_xor<2>(v); v += dpp_xor<4>(v);
    return v;
}

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// ---- exchanges between the workgroups of one control set (Guideline 16: write-through payload, one flag word, relaxed polls) -----------
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store((gu64*)(unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((gu64*)(unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// every storing wave has drained its stores; then ONE lane raises the flag
__device__ __forceinline__ void publish_flag(unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store((gu32*)flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0: lane i polls the flag of workgroup i until it shows the epoch; bounded (a spin that never ends is a dead GPU box)
__device__ __forceinline__ void wait_flags(unsigned* flags /* stride 4 words */, int G, unsigned epoch, unsigned* err) {
    if (threadIdx.x < 64) {
        const int i = threadIdx.x;
        unsigned spins = __hip_atomic_load((gu32*)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? (1u << 22) : 0u;   // an earlier spin gave up: do not wait again
        while (true) {
            const unsigned v = i < G ? __hip_atomic_load((gu32*)(flags + 4 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
            if (__all((int)(v - epoch) >= 0)) break;
            if (++spins > (1u << 22)) { if (i == 0) __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
}

// count doubles from the exchange buffers into LDS, sixteen write-through loads in flight per thread (one at a time they cost a fabric round trip each);
// src(o) = the global address of element o, or nullptr for a pad element of value pad(o)
template <int THREADS, class SRC, class PAD, class DST>
__device__ __forceinline__ void fetch_sc1(double* dst, int count, SRC src, PAD pad, DST at) {
    for (int o0 = threadIdx.x; o0 < count; o0 += 16 * THREADS) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int o = o0 + u * THREADS;
            v[u] = 0.0;
            if (o < count) { const double* a = src(o); v[u] = a ? ld_sc1(a) : pad(o); }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int o = o0 + u * THREADS; if (o < count) dst[at(o)] = v[u]; }
    }
}
template <int THREADS, class SRC, class PAD>
__device__ __forceinline__ void fetch_sc1(double* dst, int count, SRC src, PAD pad) { fetch_sc1<THREADS>(dst, count, src, pad, [](int o) { return o; }); }

// The same for the big payloads (the subtree roots of all workgroups: 32 KB at n = 8) with 16-byte loads, eight in flight per thread: `pairs` complex numbers, src(p) = the
// 16-byte aligned global address of pair p or nullptr for a pad pair of value pad(p).  One asm statement holds the loads AND their s_waitcnt: the compiler does not track
// the arrival of a load it did not issue, so nothing may touch the destination registers in between.
typedef double qsm_d2 __attribute__((ext_vector_type(2)));
template <int THREADS, class SRC, class PAD>
__device__ __forceinline__ void fetch_sc1_pairs(cplx* dst, int pairs, SRC src, PAD pad, const cplx* any_valid) {
    for (int p0 = threadIdx.x; p0 < pairs; p0 += 8 * THREADS) {
        const cplx* a[8];
        bool real[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * THREADS;
            const cplx* q = p < pairs ? src(p) : nullptr;
            real[u] = q != nullptr;
            a[u] = q ? q : any_valid;
        }
        qsm_d2 v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                     "global_load_dwordx4 %1, %9, off sc1\n\t"
                     "global_load_dwordx4 %2, %10, off sc1\n\t"
                     "global_load_dwordx4 %3, %11, off sc1\n\t"
                     "global_load_dwordx4 %4, %12, off sc1\n\t"
                     "global_load_dwordx4 %5, %13, off sc1\n\t"
                     "global_load_dwordx4 %6, %14, off sc1\n\t"
                     "global_load_dwordx4 %7, %15, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
        const qsm_d2 v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + u * THREADS;
            if (p < pairs) dst[p] = real[u] ? cmake(v[u].x, v[u].y) : pad(p);
        }
    }
}

// The partial sums xs[4 gi], xs[4 gi + 1] of the G <= 32 workgroups of a control set, summed by every wave for itself: lane gi takes workgroup gi, a DPP tree over each
// row of 16 lanes, the two rows through readlane.  The same tree in every wave of every workgroup of the set: their copies of the sums are bit-identical (the stop rule
// must fall the same way everywhere), and a tree costs 0.3 k cycles where the loop over gi cost 2.7 k per iteration.
__device__ __forceinline__ double lane_value(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void sum_groups2(const double* xs, int G, double& a, double& b) {
    static_assert(QOC_SMALL_MAXG <= 32, "two rows of 16 lanes");
    const int l = threadIdx.x & 63;
    double va = l < G ? xs[4 * l] : 0.0, vb = l < G ? xs[4 * l + 1] : 0.0;
    va += dpp_xor<1>(va); vb += dpp_xor<1>(vb);
    va += dpp_xor<2>(va); vb += dpp_xor<2>(vb);
    va += dpp_xor<4>(va); vb += dpp_xor<4>(vb);
    va += dpp_xor<8>(va); vb += dpp_xor<8>(vb);
    a = lane_value(va, 0) + lane_value(va, 16);
    b = lane_value(vb, 0) + lane_value(vb, 16);
}

template <int THREADS>
__device__ __forceinline__ void wg_sum2(double& a, double& b, double* red /* 2 x waves doubles */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[2 * wid] = a; red[2 * wid + 1] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) { ta += red[2 * i]; tb += red[2 * i + 1]; }
    a = ta; b = tb;
}

// misc area (doubles)
enum { M_RED = 0, M_Z = 34, M_SUM = 36, M_ZN = 40, M_INVF = 48, M_MAXA = 80, M_FA = 88, M_DEC = 92 };

// ---- state regularisers at one time point (core/regularization_functions.py:69-95) ----------------------------------------------------
// P = the lane's column of Psi_tau.  sf[f] = 2 a_f |phi|^2 phi with phi = <level f | Psi_tau[:, j]> (bare or dressed: VfS holds the bra);
// returns the lane's share of sum_f a_f |phi|^4 / 2; ztau = <W, Psi_tau> (every lane of the row) when asked for.
template <int N>
__device__ __forceinline__ double state_terms(const cplx (&P)[N], const cplx* VfS, const double* faS, int nforb, const cplx* Wcol, int jj,
                                              bool lane_m, bool want_z, cplx (&sf)[QOC_SMALL_NF], cplx& ztau) {
    double val = 0.0;
#pragma unroll
    for (int f = 0; f < QOC_SMALL_NF; ++f) {
        sf[f] = cmake(0.0, 0.0);
        if (f < nforb) {
            cplx phi = cmake(0.0, 0.0);
#pragma unroll
            for (int c = 0; c < N; ++c) cfma_conj(phi, VfS[f * N + c], P[c]);
            const double pop = phi.x * phi.x + phi.y * phi.y, a = faS[f];
            if (lane_m) { sf[f] = cscale(phi, 2.0 * a * pop); val += a * 0.5 * pop * pop; }
        }
    }
    ztau = cmake(0.0, 0.0);
    if (want_z) {
        cplx zp = cmake(0.0, 0.0);
#pragma unroll
        for (int r = 0; r < N; ++r) cfma_conj(zp, Wcol[r * N + jj], P[r]);
        if (!lane_m) zp = cmake(0.0, 0.0);
        ztau = cmake(row_sum16(zp.x), row_sum16(zp.y));
    }
    return val;
}
// Y += S_tau^dagger: lane a holds column a of Y (entries j' < m).  Forbidden levels: S[a][j'] = Vf[a] sf[j'] (sf lives in lane j');
// speed_up: S[a][j'] = coef z_tau W[a][j'].
template <int N, int JP = 0>
__device__ __forceinline__ void add_forbidden(cplx (&Y)[N], cplx& s, const cplx& vfa) {
    if constexpr (JP < N) {
        cmac_dpp_conj<JP>(Y[JP], s, vfa);
        add_forbidden<N, JP + 1>(Y, s, vfa);
    }
}
template <int N>
__device__ __forceinline__ void add_sources(cplx (&Y)[N], cplx (&sf)[QOC_SMALL_NF], const cplx* VfS, int nforb, int jj, bool has_speed,
                                            double coef, const cplx& ztau, const cplx* Wd) {
#pragma unroll
    for (int f = 0; f < QOC_SMALL_NF; ++f) {
        if (f < nforb) {
            const cplx vfa = VfS[f * N + jj];
            asm volatile("" : "+v"(sf[f].x), "+v"(sf[f].y));
            asm volatile("s_nop 4");
            add_forbidden<N>(Y, sf[f], vfa);
        }
    }
    if (has_speed) {
        const cplx cz = cmake(coef * ztau.x, -coef * ztau.y);                // coef conj(z_tau)
#pragma unroll
        for (int jp = 0; jp < N; ++jp) cfma(Y[jp], cz, Wd[jp * N + jj]);
    }
}

// the same two on a (mine, other) matrix of the split form: the complex arithmetic per lane runs on (re, im) in both halves
template <int N, bool SPL>
__device__ __forceinline__ double state_terms_l(const cplx (&P)[N], const cplx* VfS, const double* faS, int nforb, const cplx* Wcol, int jj,
                                                bool lane_m, bool want_z, cplx (&sf)[QOC_SMALL_NF], cplx& ztau, int h) {
    if constexpr (SPL) {
        cplx Pn[N];
#pragma unroll
        for (int r = 0; r < N; ++r) Pn[r] = swp(P[r], h);
        return state_terms<N>(Pn, VfS, faS, nforb, Wcol, jj, lane_m, want_z, sf, ztau);
    } else return state_terms<N>(P, VfS, faS, nforb, Wcol, jj, lane_m, want_z, sf, ztau);
}
template <int N, bool SPL>
__device__ __forceinline__ void add_sources_l(cplx (&Y)[N], cplx (&sf)[QOC_SMALL_NF], const cplx* VfS, int nforb, int jj, bool has_speed,
                                              double coef, const cplx& ztau, const cplx* Wd, int h) {
    if constexpr (SPL) {
        cplx Sd[N];
#pragma unroll
        for (int r = 0; r < N; ++r) Sd[r] = cmake(0.0, 0.0);
        add_sources<N>(Sd, sf, VfS, nforb, jj, has_speed, coef, ztau, Wd);
#pragma unroll
        for (int r = 0; r < N; ++r) Y[r] = cadd(Y[r], swp(Sd[r], h));
    } else add_sources<N>(Y, sf, VfS, nforb, jj, has_speed, coef, ztau, Wd);
}

// =========================================================================================================================================
// MM = false: an instance for ONE workgroup per control set -- the exchange code, the deferred stop rule and their live state are compiled out (the 32-row instances of
// n <= 4 have 256 registers per lane: with the exchanges compiled in they spill in the loop; C1 takes one of these)
// the three products, in the form of the instance (SPL, sgn, m: locals of the kernel)
#define MULB(A, x, o) do { if constexpr (SPL) smulb<N, 0>(A, x, o, 0, sgn); else mulb<N>(A, x, o); } while (0)
#define MULB_M(A, x, o, m_) do { if constexpr (SPL) smulb<N, 1>(A, x, o, m_, sgn); else mulb_m<N>(A, x, o, m_); } while (0)
#define STATE_TERMS(P, wz) state_terms_l<N, SPL>(P, VfS, misc + M_FA, nforb, Wcol, jj, lane_m, wz, sf, ztau, h)
#define ADD_SOURCES(Ym, zt) add_sources_l<N, SPL>(Ym, sf, VfS, nforb, jj, has_speed, coef, zt, Wd, h)
#define MULB_K(A, x, o, m_) do { if constexpr (SPL) smulb<N, 2>(A, x, o, m_, sgn); else mulb_k<N>(A, x, o, m_); } while (0)
template <int N, int L, int R, bool SRC, bool MM = true>
__global__ void __launch_bounds__(R * 16) k_small_iter(QocDev d, QocAdamDev ap, QocSmallDev sd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int THREADS = R * 16, RL = R * L, LR = ilog2c(R), NN = N * N;
    constexpr int NP = qoc_small_node(N);                           // distance between the nodes of the product trees (qoc_small.h)
    constexpr int QE = (8 * RL + THREADS - 1) / THREADS;            // (k, t) elements per thread, k <= 8
    // split form (5 <= n <= 8): lanes j and j + 8 of the row share column j, half h forms the real (0) / imaginary (1) parts of the products
    constexpr bool SPL = QOC_SMALL_SPLIT && N >= 5 && N <= 8;
    const int tid = threadIdx.x, row = tid >> 4, lane16 = tid & 15, j = SPL ? (lane16 & 7) : lane16, h = SPL ? (lane16 >> 3) : 0;
    const double sgn = h ? 1.0 : -1.0;
    const int G = sd.G, g = blockIdx.x % G, b = blockIdx.x / G;
    const bool act = j < N, actw = act && h == 0;            // actw: the lane that writes column j to LDS / HBM and counts in sums over the columns
    const int jj = act ? j : N - 1;
    const int n = d.n, m = d.m, k = d.k, steps = d.steps, nforb = SRC ? d.n_forb : 0;
    const bool lane_m = j < m && h == 0;
    const int grow = g * R + row, t0 = grow * L, LTOT = LR + sd.LG;
    const bool multi = MM && G > 1;
    const bool has_speed = SRC && d.has_speed;

    const QocSmallLayout lo = qoc_small_layout(N, R, L, k, m, sd.Gp, SRC, d.has_band != 0);
    cplx* S = (cplx*)smem;
    cplx* HsC = S + lo.hsc; cplx* HsT = S + lo.hst; cplx* VfS = S + lo.vfs; cplx* Psi0c = S + lo.psi0; cplx* Wd = S + lo.wd;
    cplx* Wcol = S + lo.wcol; cplx* V0c = S + lo.v0; cplx* PsiN = S + lo.psin;
    cplx* treeM = S + lo.treeM; cplx* treeU = S + lo.treeU; cplx* treeO = S + lo.treeO; cplx* treeOU = S + lo.treeOU;
    cplx* twS = S + lo.twS; cplx* phS = S + lo.phS;
    cplx* qS = S + lo.qS; double* wS = (double*)(S + lo.wS); double* misc = (double*)(S + lo.misc); double* xsum = (double*)(S + lo.xsum);
    auto Wv = [&](int kk, int tl) -> double& { return wS[kk * (RL + 4) + 2 + tl]; };
    // node (level, index) of the two trees: levels below LR live in this workgroup (index relative to its first node of the level)
    auto lnode = [&](cplx* base, int l, int i) -> cplx* { return base + (size_t)((2 * R - (2 * R >> l)) + i) * NP; };
    auto unode = [&](cplx* base, int l, int i) -> cplx* { return base + (size_t)((2 * sd.Gp - (2 * sd.Gp >> l)) + i) * NP; };
    // the offsets of the affine costate recursion are m x N (rows j' < m of Y): their nodes are that small
    const int MN = m * N;
    auto lnodeO = [&](int l, int i) -> cplx* { return treeO + (size_t)((2 * R - (2 * R >> l)) + i) * MN; };
    auto unodeO = [&](int l, int i) -> cplx* { return treeOU + (size_t)((2 * sd.Gp - (2 * sd.Gp >> l)) + i) * MN; };

    // ---- prologue: constants to LDS -------------------------------------------------------------------------------------------------------
    const int it_start = d.iters[b], adam_t_start = d.adam_t[b];
    if (ap.mode == 1 && d.done[b]) return;                      // a finished control set keeps its last evaluation (uniform over its workgroups)
    if (g == 0 && tid == 0) sd.final_valid[b] = 0;              // final_state / unitary_scale of this launch's last evaluation: set in the epilogue when formed here
    bool tree_is_last = true;                                   // the product tree in LDS belongs to the evaluation the launch reports
    {
        const double sc = 1.0 / (double)(1 << d.s);
        for (int o = tid; o < (k + 1) * NN; o += THREADS) {
            const int kk = o / NN, rc = o - kk * NN, r = rc / N, c = rc - r * N;
            cplx v = cmake(0.0, 0.0);
            if (r < n && c < n) v = d.Hs[(size_t)kk * n * n + r * n + c];
            HsC[o] = cscale(v, sc);
            if (kk >= 1) HsT[(kk - 1) * NN + c * N + r] = v;
        }
        for (int o = tid; o < NN; o += THREADS) {
            const int r = o / N, c = o - r * N;                 // [r][lane c]
            const bool in = r < n && c < m;
            Psi0c[o] = in ? d.Psi0[r * m + c] : cmake(0.0, 0.0);
            Wd[c * N + r] = in ? cconj(d.W[r * m + c]) : cmake(0.0, 0.0);      // Wd[j'][a] = conj(W[a][j'])
            if (SRC) { Wcol[o] = in ? d.W[r * m + c] : cmake(0.0, 0.0); V0c[o] = in ? d.V[r * m + c] : cmake(0.0, 0.0); }
        }
        if (SRC) {
            for (int o = tid; o < QOC_SMALL_NF * N; o += THREADS) {
                const int f = o / N, c = o - f * N;
                cplx v = cmake(0.0, 0.0);
                if (f < nforb && c < n) {
                    const int st = d.forb_state[f];
                    v = d.forbid_dressed ? d.Vs[c * n + st] : cmake(c == st ? 1.0 : 0.0, 0.0);
                }
                VfS[o] = v;
            }
            if (tid < QOC_SMALL_NF) misc[M_FA + tid] = tid < nforb ? d.forb_a[tid] : 0.0;
        }
        if (tid < 32) { double f = 1.0; for (int i = 2; i <= tid; ++i) f *= (double)i; misc[M_INVF + tid] = 1.0 / f; }
        if (tid < 8) misc[M_MAXA + tid] = tid < k ? d.maxA[tid] : 0.0;
        for (int o = tid; o < k * (RL + 4); o += THREADS) wS[o] = 0.0;
        if (d.has_band) for (int r = tid; r < steps; r += THREADS) { double sn, cs; sincos(-2.0 * M_PI * (double)r / (double)steps, &sn, &cs); twS[r] = cmake(cs, sn); }
    }
    __syncthreads();
    // the (control, slice) elements of this thread: variable and Adam slots stay in registers for the whole launch
    double e_base[QE], e_m[QE], e_v[QE], e_w[QE], e_g[QE], e_c[QE];      // e_w = sin(base), e_c = cos(base) (the chain rule of the tail): ONE sincos per step
    int e_kk[QE], e_tl[QE];
    bool e_ok[QE];
#pragma unroll
    for (int e = 0; e < QE; ++e) {
        const int o = tid + e * THREADS, kk = o / RL, tl = o - kk * RL, t = g * RL + tl;
        e_kk[e] = kk; e_tl[e] = tl; e_ok[e] = kk < k && t < steps;
        e_base[e] = 0.0; e_m[e] = 0.0; e_v[e] = 0.0; e_w[e] = 0.0; e_g[e] = 0.0; e_c[e] = 1.0;
        if (e_ok[e]) {
            const size_t go = ((size_t)b * k + kk) * steps + t;
            e_base[e] = d.base[go];
            if (ap.mode != 0) { e_m[e] = d.adam_m[go]; e_v[e] = d.adam_v[go]; }
            sincos(e_base[e], &e_w[e], &e_c[e]);
            Wv(kk, tl) = e_w[e];
        }
    }
    __syncthreads();

    // n <= 4: a slice is a few hundred instructions between LDS round trips (profiles/r06_small_phase_timing.txt) -- the lane's columns of the drift and of the first two control
    // Hamiltonians (generator assembly) and their transposes (gradient contraction) stay in registers for the whole launch; further controls come from LDS as for larger n
    constexpr bool HOIST = N <= 4;
    cplx H0r[N], H1r[N], H2r[N], T0r[N], T1r[N];
    if constexpr (HOIST) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            H0r[r] = HsC[r * N + jj]; H1r[r] = HsC[NN + r * N + jj]; H2r[r] = k > 1 ? HsC[2 * NN + r * N + jj] : cmake(0.0, 0.0);
            T0r[r] = HsT[r * N + jj]; T1r[r] = k > 1 ? HsT[NN + r * N + jj] : cmake(0.0, 0.0);
        }
    }
    int it_count = it_start, adam_t = adam_t_start, done_now = 0;
    // beta^t of the Adam bias correction and the learning-rate schedule rate e^{-it / decay}: formed once per launch, advanced by one multiplication per
    // iteration (pow / exp are ~1000 instructions on every wave of the workgroup; the running products differ from them by <= iterations-per-launch ulps)
    double pow_b1 = ap.mode != 0 ? pow(0.9, (double)adam_t_start) : 1.0, pow_b2 = ap.mode != 0 ? pow(0.999, (double)adam_t_start) : 1.0;
    double lr_run = ap.mode == 1 ? ap.rate * exp(-(double)it_start / ap.decay) : 0.0;
    const double lr_step = ap.mode == 1 ? exp(-1.0 / ap.decay) : 1.0;
    double out_loss = 0.0, out_reg = 0.0, out_g2 = 0.0, out_regstate = 0.0;
    cplx out_z = cmake(0.0, 0.0);
    const double mm = (double)m * (double)m, c0 = -2.0 / mm;
    double* xA0 = sd.xA + ((size_t)b * G) * sd.xa_stride;
    double* xB = sd.xB + ((size_t)b * G) * sd.xb_stride;
    double* xS = sd.xS + ((size_t)b * G) * sd.xs_stride;
    unsigned* flags = sd.flags + ((size_t)b * G) * 4;

    // Several workgroups per control set: the stop rule needs grad_squared of the WHOLE pulse, i.e. an exchange.  Inside a burst the partial sums of
    // iteration i travel with exchange A of iteration i + 1 instead, the Adam step is taken at once, and if the sums then say "stop" the step is undone
    // (variable, Adam slots and counters of the evaluation that tripped the rule are kept beside the new ones): one exchange per iteration less.
    bool spec = false;
    double sp_reg = 0.0, sp_g2 = 0.0, sp_regstate = 0.0, sv_pow_b1 = 1.0, sv_pow_b2 = 1.0, sv_lr_run = 0.0;
    cplx sp_z = cmake(0.0, 0.0);
    int sv_it_count = 0, sv_adam_t = 0;
    double sv_base[QE], sv_m[QE], sv_v[QE], sv_w[QE], sv_g[QE];
#pragma unroll
    for (int e = 0; e < QE; ++e) { sv_base[e] = 0.0; sv_m[e] = 0.0; sv_v[e] = 0.0; sv_w[e] = 0.0; sv_g[e] = 0.0; }
#ifdef QOC_SMALL_TIMING
    unsigned long long stamp[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
    for (int it = 0; it < sd.iters; ++it) {
        const unsigned epoch = (unsigned)it + 1u;
        QSM_STAMP(0);
        // ---- P1: exponentials of the own slices, kept in registers; product of the row --------------------------------------------------
        cplx Kr[L][N];
        double uu0[L], uu1[L];                                                   // controls of the own slices (first two), read ahead of the products
        if constexpr (HOIST) {
#pragma unroll
            for (int i = 0; i < L; ++i) {
                uu0[i] = misc[M_MAXA] * Wv(0, row * L + i);
                uu1[i] = k > 1 ? misc[M_MAXA + 1] * Wv(1, row * L + i) : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int t = t0 + i;
#pragma unroll
            for (int r = 0; r < N; ++r) { const double dj = r == j ? 1.0 : 0.0; Kr[i][r] = cmake(h ? 0.0 : dj, h ? dj : 0.0); }
            if (t < steps && sd.Teff >= 1) {
                cplx A[N], Hn[N], acc[N];
                const int tl = row * L + i;
                if constexpr (HOIST) {
#pragma unroll
                    for (int r = 0; r < N; ++r) {
                        A[r].x = fma(uu1[i], H2r[r].x, fma(uu0[i], H1r[r].x, H0r[r].x));
                        A[r].y = fma(uu1[i], H2r[r].y, fma(uu0[i], H1r[r].y, H0r[r].y));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < N; ++r) A[r] = ldm<SPL>(HsC + r * N + jj, h);
                }
                for (int kk = HOIST ? 2 : 0; kk < k; ++kk) {
                    const double u = misc[M_MAXA + kk] * Wv(kk, tl);
                    const cplx* Hk = HsC + (kk + 1) * NN;
#pragma unroll
                    for (int r = 0; r < N; ++r) { const cplx hk = ldm<SPL>(Hk + r * N + jj, h); A[r].x = fma(u, hk.x, A[r].x); A[r].y = fma(u, hk.y, A[r].y); }
                }
#pragma unroll
                for (int r = 0; r < N; ++r) { Hn[r] = A[r]; Kr[i][r] = cadd(Kr[i][r], A[r]); }
                double fnext = misc[M_INVF + 2];                                  // (1 / ii! of the next term is read under the product of this one)
#pragma unroll 1
                for (int ii = 2; ii <= sd.Teff; ++ii) {                          // H_n = H H_n ; matexp += H_n / ii!      tensorflow_state.py:38-41
                    const double f = fnext;
                    fnext = misc[M_INVF + ii + 1];
                    MULB(A, Hn, acc);
#pragma unroll
                    for (int r = 0; r < N; ++r) { Hn[r] = acc[r]; Kr[i][r].x = fma(acc[r].x, f, Kr[i][r].x); Kr[i][r].y = fma(acc[r].y, f, Kr[i][r].y); }
                }
#pragma unroll 1
                for (int q = 0; q < d.s; ++q) {                                   // squarings                              tensorflow_state.py:43-44
                    MULB(Kr[i], Kr[i], acc);
#pragma unroll
                    for (int r = 0; r < N; ++r) Kr[i][r] = acc[r];
                }
            }
        }
        QSM_STAMP(1);
        cplx Mown[N];
#pragma unroll
        for (int r = 0; r < N; ++r) Mown[r] = Kr[0][r];
#pragma unroll
        for (int i = 1; i < L; ++i) {
            cplx acc[N];
            MULB(Kr[i], Mown, acc);
#pragma unroll
            for (int r = 0; r < N; ++r) Mown[r] = acc[r];
        }
        if (actw) {
            cplx* nd = lnode(treeM, 0, row);
#pragma unroll
            for (int r = 0; r < N; ++r) nd[r * N + j] = Mown[r];
        }
        // ---- P2: up-sweep of the product tree (later times on the left) ---------------------------------------------------------------------
#pragma unroll 1
        for (int l = 1; l <= LR; ++l) {
            // levels 1 and 2 combine rows of ONE wave (four rows per wave): the wave's own LDS traffic is ordered, no workgroup barrier
            if (l <= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
            if ((row & ((1 << l) - 1)) == 0) {
                const cplx* rn = lnode(treeM, l - 1, (row >> (l - 1)) + 1);
                cplx Ar[N], acc[N];
#pragma unroll
                for (int r = 0; r < N; ++r) Ar[r] = ldm<SPL>(rn + r * N + jj, h);
                MULB(Ar, Mown, acc);
                cplx* nd = lnode(treeM, l, row >> l);
#pragma unroll
                for (int r = 0; r < N; ++r) { Mown[r] = acc[r]; if (actw) nd[r * N + j] = acc[r]; }
            }
        }
        __syncthreads();
        QSM_STAMP(14);
        const cplx* rangeP = nullptr;                              // several workgroups, no state regulariser: start state and end costate of THIS workgroup (below)
        if (multi) {
            // exchange A: the subtree product of every workgroup of the control set + the halo controls of the neighbours.  Two buffers, by the parity of
            // the iteration: with the stop rule deferred nothing else separates a fast workgroup's next publication from a slow one's reads of this one
            double* xA = xA0 + (size_t)(it & 1) * sd.xa_parity;
            const double* root = (const double*)lnode(treeM, LR, 0);
            double* mine = xA + (size_t)g * sd.xa_stride;
            for (int o = tid; o < 2 * NN; o += THREADS) st_sc1(mine + o, root[o]);
            if (tid < 4 * k) { const int kk = tid >> 2, hh = tid & 3; st_sc1(mine + 2 * NN + tid, Wv(kk, hh < 2 ? hh : RL - 4 + hh)); }
            if (spec && tid == 0) { st_sc1(mine + 2 * NN + 32, sp_reg); st_sc1(mine + 2 * NN + 33, sp_g2); st_sc1(mine + 2 * NN + 34, sp_z.x); st_sc1(mine + 2 * NN + 35, sp_z.y); }
            publish_flag(flags + 4 * g + 0, epoch);
            wait_flags(flags + 0, G, epoch, sd.err);
            QSM_STAMP(15);
            if (spec) fetch_sc1<THREADS>(xsum, 4 * G, [&](int o) -> const double* { return xA + (size_t)(o >> 2) * sd.xa_stride + 2 * NN + 32 + (o & 3); }, [](int) { return 0.0; });
            static_assert(NP == NN, "the leaves are fetched as one dense array");
            fetch_sc1_pairs<THREADS>(unode(treeU, 0, 0), sd.Gp * NN,
                [&](int p) -> const cplx* { const int gi = p / NN; return gi < G ? (const cplx*)(xA + (size_t)gi * sd.xa_stride) + (p - gi * NN) : nullptr; },
                [&](int p) { const int e = p % NN; return cmake((e / N) == (e % N) ? 1.0 : 0.0, 0.0); },                               // identity leaves pad the tree
                (const cplx*)xA);
            if (tid < 4 * k) {                                    // halo: the neighbours' controls of this evaluation
                const int kk = tid >> 2, hh = tid & 3;
                const int src_g = hh < 2 ? g - 1 : g + 1, tl = hh < 2 ? hh - 2 : RL + (hh - 2), t = g * RL + tl;
                if (src_g >= 0 && src_g < G && t >= 0 && t < steps)
                    Wv(kk, tl) = ld_sc1(xA + (size_t)src_g * sd.xa_stride + 2 * NN + 4 * kk + (hh < 2 ? hh + 2 : hh - 2));
            }
            __syncthreads();
            QSM_STAMP(7);
            if (spec) {                                            // the stop rule of the previous iteration, one exchange late
                double reg = 0.0, g2 = 0.0;
                sum_groups2(xsum, G, reg, g2);
                g2 *= 0.5;
                const cplx z = cmake(xsum[2], xsum[3]);
                const double loss = 1.0 - (z.x * z.x + z.y * z.y) / mm;
                if ((loss < ap.conv_target) || (g2 < ap.min_grad)) {                    // run_session.py:56-60: undo the step taken past the stop
#pragma unroll
                    for (int e = 0; e < QE; ++e) { e_base[e] = sv_base[e]; e_m[e] = sv_m[e]; e_v[e] = sv_v[e]; e_w[e] = sv_w[e]; e_g[e] = sv_g[e]; }
                    it_count = sv_it_count; adam_t = sv_adam_t; pow_b1 = sv_pow_b1; pow_b2 = sv_pow_b2; lr_run = sv_lr_run;
                    out_loss = loss; out_regstate = sp_regstate; out_reg = loss + sp_regstate + reg; out_g2 = g2; out_z = z;
                    done_now = 1;
                    tree_is_last = false;                          // (the tree is the undone evaluation's: final_state is formed on read-back instead)
                    break;
                }
                spec = false;
                __syncthreads();                                   // (xsum is written again by the next exchange)
            }
            QSM_STAMP(16);
            if constexpr (SRC) {
#pragma unroll 1
                for (int l = 1; l <= sd.LG; ++l) {
                    for (int nd_i = row; nd_i < (sd.Gp >> l); nd_i += R) {
                        const cplx* rn = unode(treeU, l - 1, 2 * nd_i + 1);
                        const cplx* ln = unode(treeU, l - 1, 2 * nd_i);
                        cplx Ar[N], xl[N], acc[N];
#pragma unroll
                        for (int r = 0; r < N; ++r) { Ar[r] = ldm<SPL>(rn + r * N + jj, h); xl[r] = ldm<SPL>(ln + r * N + jj, h); }
                        MULB(Ar, xl, acc);
                        cplx* nd = unode(treeU, l, nd_i);
                        if (actw) {
#pragma unroll
                            for (int r = 0; r < N; ++r) nd[r * N + j] = acc[r];
                        }
                    }
                    __syncthreads();
                }
            } else {
                // z-free costate: this workgroup needs ONE start state and ONE end costate, i.e. the products of two RANGES of the subtree roots -- (Psi_0, M_0 .. M_{g-1}) and
                // (M_{g+1} .. M_{G-1}, W^dagger) in time order.  Both are reduced pairwise (later times on the left, an odd last element passes through), at most
                // Gp / 2 + 1 pairs per level over the R rows: ceil(log2(max(g + 1, G - g))) <= LG dependent products instead of the LG of a tree plus the LG of a walk down it.
                int ca = g + 1, cb = G - g;
                // (offsets into the LDS carve as 32-bit integers: the address arithmetic of a level is otherwise half as long as its product)
                const int offP = lo.treeU + sd.Gp * NP, offQ = offP + (sd.Gp / 2 + 2) * NP;
                int cur = -1, nxt = offP;
#pragma unroll 1
                while (ca > 1 || cb > 1) {
                    const int na = (ca + 1) >> 1, nb = (cb + 1) >> 1;
                    for (int w = row; w < na + nb; w += R) {
                        const bool pre = w < na;
                        const int i = pre ? w : w - na, c = pre ? ca : cb;
                        int o0, o1;
                        if (cur >= 0) { o0 = cur + (pre ? 2 * i : ca + 2 * i) * NP; o1 = o0 + NP; }
                        else if (pre) { o0 = i == 0 ? lo.psi0 : lo.treeU + (2 * i - 1) * NP; o1 = lo.treeU + 2 * i * NP; }
                        else { o0 = 2 * i == c - 1 ? lo.wd : lo.treeU + (g + 1 + 2 * i) * NP; o1 = 2 * i + 1 == c - 1 ? lo.wd : o0 + NP; }
                        const cplx* e0 = S + o0 + jj;
                        cplx xl[N], acc[N];
#pragma unroll
                        for (int r = 0; r < N; ++r) xl[r] = ldm<SPL>(e0 + r * N, h);
                        if (2 * i + 1 < c) {
                            const cplx* e1 = S + o1 + jj;
                            cplx Ar[N];
#pragma unroll
                            for (int r = 0; r < N; ++r) Ar[r] = ldm<SPL>(e1 + r * N, h);
                            MULB(Ar, xl, acc);
                        } else {
#pragma unroll
                            for (int r = 0; r < N; ++r) acc[r] = xl[r];
                        }
                        if (actw) {
                            cplx* nd = S + nxt + w * NP + j;
#pragma unroll
                            for (int r = 0; r < N; ++r) nd[r * N] = acc[r];
                        }
                    }
                    __syncthreads();
                    ca = na; cb = nb; cur = nxt; nxt = (nxt == offP) ? offQ : offP;
                }
                rangeP = S + cur;                                    // [0]: (M_{g-1} .. M_0) Psi_0, [1]: W^dagger (M_{G-1} .. M_{g+1})
            }
        }
        QSM_STAMP(2);
        // sibling of this row's ancestor at level l - 1 (global level numbering), and which side the row hangs on
        auto sibling = [&](cplx* loc, cplx* upp, int l, int& bit) -> const cplx* {
            bit = (grow >> (l - 1)) & 1;
            const int sib = ((grow >> l) << 1) + (1 - bit);
            return (l - 1 < LR) ? lnode(loc, l - 1, sib - ((g * R) >> (l - 1))) : unode(upp, l - 1 - LR, sib);
        };
        auto siblingO = [&](int l) -> const cplx* {
            const int bit = (grow >> (l - 1)) & 1, sib = ((grow >> l) << 1) + (1 - bit);
            return (l - 1 < LR) ? lnodeO(l - 1, sib - ((g * R) >> (l - 1))) : unodeO(l - 1 - LR, sib);
        };

        // ---- P3 / P4: start state and end costate of the row by a walk from the root; forward and backward sweep over the own slices --------
        cplx Phi[N], Y[N], Ps[L][N];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            Phi[r] = ldm<SPL>((rangeP ? rangeP : Psi0c) + r * N + jj, h);
            Y[r] = ldm<SPL>((rangeP ? rangeP + NP : Wd) + r * N + jj, h);
        }
        cplx zfin = cmake(0.0, 0.0);
        double reg_state = 0.0, coef = 0.0;
        if constexpr (!SRC) {
            // z-free costate: Lambda_tau = -(2 / m^2) z Lambda'_tau with Lambda'_N = W; one product per level -- the row is either the right
            // child (its start state passes the left sibling) or the left one (its end costate passes the right sibling)
            cplx Sib[(!MM && N <= 4) ? LR : 1][N];
            if constexpr (!MM && N <= 4) {                                        // one workgroup, tiny matrices: the log2(R) siblings of the walk in ONE batch of LDS reads
#pragma unroll
                for (int l = LR; l >= 1; --l) {
                    int bit;
                    const cplx* sn = sibling(treeM, treeU, l, bit);
#pragma unroll
                    for (int r = 0; r < N; ++r) Sib[l - 1][r] = sn[r * N + jj];
                }
#pragma unroll
                for (int l = LR; l >= 1; --l) {
                    const int bit = (grow >> (l - 1)) & 1;
                    cplx As[N], xs[N], acc[N];
#pragma unroll
                    for (int r = 0; r < N; ++r) { As[r] = bit ? Sib[l - 1][r] : Y[r]; xs[r] = bit ? Phi[r] : Sib[l - 1][r]; }
                    MULB(As, xs, acc);
#pragma unroll
                    for (int r = 0; r < N; ++r) { if (bit) Phi[r] = acc[r]; else Y[r] = acc[r]; }
                }
            } else
#pragma unroll 1
            for (int l = multi ? LR : LTOT; l >= 1; --l) {                      // (several workgroups: the levels above LR are in rangeP already)
                int bit;
                const cplx* sn = sibling(treeM, treeU, l, bit);
                cplx Ms[N], acc[N];
#pragma unroll
                for (int r = 0; r < N; ++r) Ms[r] = ldm<SPL>(sn + r * N + jj, h);
                if (l >= 3) {                                                     // the four rows of a wave share their ancestors from level 2 up: a uniform branch
                    if (bit) {
                        MULB(Ms, Phi, acc);
#pragma unroll
                        for (int r = 0; r < N; ++r) Phi[r] = acc[r];
                    } else {
                        MULB_M(Y, Ms, acc, m);
#pragma unroll
                        for (int r = 0; r < N; ++r) Y[r] = acc[r];
                    }
                } else {                                                          // rows of one wave on different sides: operands selected per row, one product
                    cplx As[N], xs[N];
#pragma unroll
                    for (int r = 0; r < N; ++r) { As[r] = bit ? Ms[r] : Y[r]; xs[r] = bit ? Phi[r] : Ms[r]; }
                    MULB(As, xs, acc);
#pragma unroll
                    for (int r = 0; r < N; ++r) { if (bit) Phi[r] = acc[r]; else Y[r] = acc[r]; }
                }
            }
#pragma unroll
            for (int i = 0; i < L; ++i) {
                if (i == 0) MULB(Kr[0], Phi, Ps[0]); else MULB(Kr[i], Ps[i - 1], Ps[i]);
            }
        } else {
            // ---- state regularisers: true costate.  Forward first (values, z_tau), then the affine offsets, then the walk for the costate ----
#pragma unroll 1
            for (int l = LTOT; l >= 1; --l) {
                int bit;
                const cplx* sn = sibling(treeM, treeU, l, bit);
                if (bit) {
                    cplx Ms[N], acc[N];
#pragma unroll
                    for (int r = 0; r < N; ++r) Ms[r] = ldm<SPL>(sn + r * N + jj, h);
                    MULB(Ms, Phi, acc);
#pragma unroll
                    for (int r = 0; r < N; ++r) Phi[r] = acc[r];
                }
            }
            QSM_STAMP(8);
            double fval = 0.0, zz2 = 0.0;
            cplx sf[QOC_SMALL_NF], ztau;
            if (grow == 0) {                                                      // tau = 0: inter_vecs[0] = V
                cplx P0[N];
#pragma unroll
                for (int r = 0; r < N; ++r) P0[r] = V0c[r * N + jj];
                fval += state_terms<N>(P0, VfS, misc + M_FA, nforb, Wcol, jj, lane_m, has_speed, sf, ztau);
                if (lane16 == 0) zz2 += ztau.x * ztau.x + ztau.y * ztau.y;
            }
#pragma unroll
            for (int i = 0; i < L; ++i) {
                if (i == 0) MULB(Kr[0], Phi, Ps[0]); else MULB(Kr[i], Ps[i - 1], Ps[i]);
                const int t = t0 + i;
                if (t < steps) {                                                  // tau = t + 1
                    const bool last = t == steps - 1;
                    fval += STATE_TERMS(Ps[i], has_speed || last);
                    if (lane16 == 0 && has_speed) zz2 += ztau.x * ztau.x + ztau.y * ztau.y;
                    if (last) {
                        if (lane16 == 0) { misc[M_ZN] = ztau.x; misc[M_ZN + 1] = ztau.y; }
                        if (actw) {
#pragma unroll
                            for (int r = 0; r < N; ++r) PsiN[r * N + j] = Ps[i][r];
                        }
                    }
                }
            }
            QSM_STAMP(9);
            wg_sum2<THREADS>(fval, zz2, misc + M_RED);
            auto fetch_a1 = [&]() {                                               // (after the flags of the exchange that carried the payload)
                const int glast = (steps - 1) / RL;
                fetch_sc1<THREADS>(xsum, 4 * G, [&](int o) -> const double* { return xS + (size_t)(o >> 2) * sd.xs_stride + (o & 3); }, [](int) { return 0.0; });
                if (g != glast) {
                    const double* from = xS + (size_t)glast * sd.xs_stride;
                    fetch_sc1<THREADS>((double*)PsiN, 2 * NN, [&](int o) -> const double* { return from + 4 + o; }, [](int) { return 0.0; });
                }
                __syncthreads();
                fval = 0.0; zz2 = 0.0;
                sum_groups2(xsum, G, fval, zz2);
                if (tid == 0) { misc[M_ZN] = xsum[4 * glast + 2]; misc[M_ZN + 1] = xsum[4 * glast + 3]; }
                __syncthreads();
            };
            if (multi) {
                // exchange A1: partial sums, and Psi_N with z_N from the workgroup that holds the last slice
                double* mine = xS + (size_t)g * sd.xs_stride;
                const int glast = (steps - 1) / RL;
                if (tid == 0) { st_sc1(mine + 0, fval); st_sc1(mine + 1, zz2); }
                if (g == glast) {
                    if (tid == 0) { st_sc1(mine + 2, misc[M_ZN]); st_sc1(mine + 3, misc[M_ZN + 1]); }
                    for (int o = tid; o < 2 * NN; o += THREADS) st_sc1(mine + 4 + o, ((const double*)PsiN)[o]);
                }
                // Without speed_up nothing below needs these sums before the offsets of the subtrees travel (the sources of forbidden levels are local in time):
                // the payload then rides on exchange A2 -- one exchange less per iteration
                if (has_speed) {
                    publish_flag(flags + 4 * g + 2, epoch);
                    wait_flags(flags + 2, G, epoch, sd.err);
                    fetch_a1();
                }
            }
            if (!multi || has_speed) zfin = cmake(misc[M_ZN], misc[M_ZN + 1]);
            const double resid = (double)(steps + 1) - zz2 / mm;
            reg_state = fval + (has_speed ? d.a_speed * 0.5 * resid * resid : 0.0);
            coef = has_speed ? -d.a_speed * resid * 2.0 / mm : 0.0;
            QSM_STAMP(10);
            // offsets of the row: O_tau = O_{tau+1} K_tau + S_tau^dagger for 1 <= tau <= steps - 1 (S_N belongs to the terminal costate)
            cplx Oown[N];
#pragma unroll
            for (int r = 0; r < N; ++r) Oown[r] = cmake(0.0, 0.0);
#pragma unroll
            for (int i = L - 1; i >= 0; --i) {
                const int t = t0 + i;
                cplx acc[N];
                MULB_M(Oown, Kr[i], acc, m);
#pragma unroll
                for (int r = 0; r < N; ++r) Oown[r] = acc[r];
                if (t >= 1 && t <= steps - 1) {
                    if (i == 0) STATE_TERMS(Phi, has_speed);
                    else STATE_TERMS(Ps[i > 0 ? i - 1 : 0], has_speed);
                    ADD_SOURCES(Oown, ztau);
                }
            }
            QSM_STAMP(11);
            if (actw) {
                cplx* nd = lnodeO(0, row);
#pragma unroll
                for (int r = 0; r < N; ++r) if (r < m) nd[r * N + j] = Oown[r];
            }
#pragma unroll 1
            for (int l = 1; l <= LR; ++l) {                                      // O = O_right M_left + O_left
                if (l <= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else __syncthreads();
                if ((row & ((1 << l) - 1)) == 0) {
                    const cplx* orn = lnodeO(l - 1, (row >> (l - 1)) + 1);
                    const cplx* mln = lnode(treeM, l - 1, row >> (l - 1));
                    cplx Ar[N], xl[N], acc[N];
#pragma unroll
                    for (int r = 0; r < N; ++r) { Ar[r] = r < m ? ldm<SPL>(orn + r * N + jj, h) : cmake(0.0, 0.0); xl[r] = ldm<SPL>(mln + r * N + jj, h); }
                    MULB_M(Ar, xl, acc, m);
                    cplx* nd = lnodeO(l, row >> l);
#pragma unroll
                    for (int r = 0; r < N; ++r) { Oown[r] = cadd(Oown[r], acc[r]); if (actw && r < m) nd[r * N + j] = Oown[r]; }
                }
            }
            __syncthreads();
            QSM_STAMP(12);
            if (multi) {
                // exchange A2: the offsets of the subtrees
                const double* root = (const double*)lnodeO(LR, 0);
                double* mine = xS + (size_t)g * sd.xs_stride + 4 + 2 * NN;
                for (int o = tid; o < 2 * MN; o += THREADS) st_sc1(mine + o, root[o]);
                publish_flag(flags + 4 * g + 3, epoch);
                wait_flags(flags + 3, G, epoch, sd.err);
                if (!has_speed) {                                                 // the payload of exchange A1 came with this one
                    fetch_a1();
                    zfin = cmake(misc[M_ZN], misc[M_ZN + 1]);
                    reg_state = fval;
                }
                fetch_sc1<THREADS>((double*)unodeO(0, 0), sd.Gp * 2 * MN,
                    [&](int o) -> const double* { const int gi = o / (2 * MN); return gi < G ? xS + (size_t)gi * sd.xs_stride + 4 + 2 * NN + (o - gi * 2 * MN) : nullptr; },
                    [](int) { return 0.0; });
                __syncthreads();
#pragma unroll 1
                for (int l = 1; l <= sd.LG; ++l) {
                    for (int nd_i = row; nd_i < (sd.Gp >> l); nd_i += R) {
                        const cplx* orn = unodeO(l - 1, 2 * nd_i + 1);
                        const cplx* oln = unodeO(l - 1, 2 * nd_i);
                        const cplx* mln = unode(treeU, l - 1, 2 * nd_i);
                        cplx Ar[N], xl[N], acc[N];
#pragma unroll
                        for (int r = 0; r < N; ++r) { Ar[r] = r < m ? ldm<SPL>(orn + r * N + jj, h) : cmake(0.0, 0.0); xl[r] = ldm<SPL>(mln + r * N + jj, h); }
                        MULB_M(Ar, xl, acc, m);
                        cplx* nd = unodeO(l, nd_i);
                        if (actw) {
#pragma unroll
                            for (int r = 0; r < N; ++r) if (r < m) nd[r * N + j] = cadd(acc[r], oln[r * N + j]);
                        }
                    }
                    __syncthreads();
                }
            }
            QSM_STAMP(13);
            // terminal costate Y_N = conj(c0 z) W^dagger + S_N^dagger
            {
                cplx PN[N];
#pragma unroll
                for (int r = 0; r < N; ++r) PN[r] = PsiN[r * N + jj];
                state_terms<N>(PN, VfS, misc + M_FA, nforb, Wcol, jj, lane_m, false, sf, ztau);
                const cplx cz = cmake(c0 * zfin.x, -c0 * zfin.y);
#pragma unroll
                for (int r = 0; r < N; ++r) Y[r] = cmul(cz, Wd[r * N + jj]);
                add_sources<N>(Y, sf, VfS, nforb, jj, has_speed, coef, zfin, Wd);
                if constexpr (SPL) {
#pragma unroll
                    for (int r = 0; r < N; ++r) Y[r] = swp(Y[r], h);
                }
            }
#pragma unroll 1
            for (int l = LTOT; l >= 1; --l) {
                int bit;
                const cplx* sn = sibling(treeM, treeU, l, bit);
                if (!bit) {                                                       // left child: the costate passes the right sibling
                    const cplx* on = siblingO(l);
                    cplx Ms[N], acc[N];
#pragma unroll
                    for (int r = 0; r < N; ++r) Ms[r] = ldm<SPL>(sn + r * N + jj, h);
                    MULB_M(Y, Ms, acc, m);
#pragma unroll
                    for (int r = 0; r < N; ++r) Y[r] = r < m ? cadd(acc[r], ldm<SPL>(on + r * N + jj, h)) : acc[r];
                }
            }
            // (the backward sweep below adds S_t^dagger after each slice)
        }

        QSM_STAMP(3);
        // ---- backward over the own slices: gradient inner products, costate ---------------------------------------------------------------
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            const int t = t0 + i, tl = row * L + i;
            cplx Rm[N];
            MULB_K(Ps[i], Y, Rm, m);                                          // Rm[c] (lane a) = (Psi_{t+1} Y_{t+1})[c][a]: a sum over the m state vectors
            if constexpr (SPL) {                                                   // the contractions below are complex arithmetic per lane: (re, im) in both halves
#pragma unroll
                for (int c = 0; c < N; ++c) Rm[c] = swp(Rm[c], h);
            }
            if constexpr (HOIST) {                                                 // the first two controls against the register copies, both reductions in flight together
                cplx q0 = cmake(0.0, 0.0), q1 = cmake(0.0, 0.0);
#pragma unroll
                for (int c = 0; c < N; ++c) { cfma(q0, T0r[c], Rm[c]); cfma(q1, T1r[c], Rm[c]); }
                if (!act) { q0 = cmake(0.0, 0.0); q1 = cmake(0.0, 0.0); }
                q0.x = row_sum_n<N>(q0.x); q0.y = row_sum_n<N>(q0.y);
                if (k > 1) { q1.x = row_sum_n<N>(q1.x); q1.y = row_sum_n<N>(q1.y); }
                if (lane16 == 0 && t < steps) { qS[tl] = q0; if (k > 1) qS[RL + tl] = q1; }
            }
            if constexpr (SPL) {                                                   // half h contracts the controls kk = h, h + 2, ...: sums over the 8 lanes of a half
                for (int k2 = 0; k2 < k; k2 += 2) {
                    const int kk = k2 + h;
                    const cplx* Hk = HsT + (kk < k ? kk : 0) * NN;
                    cplx q = cmake(0.0, 0.0);
#pragma unroll
                    for (int c = 0; c < N; ++c) cfma(q, Hk[c * N + jj], Rm[c]);
                    if (!act) q = cmake(0.0, 0.0);
                    q.x = row_sum8(q.x); q.y = row_sum8(q.y);
                    if (j == 0 && kk < k && t < steps) qS[kk * RL + tl] = q;
                }
            } else
            for (int kk = HOIST ? 2 : 0; kk < k; ++kk) {
                const cplx* Hk = HsT + kk * NN;
                cplx q = cmake(0.0, 0.0);
#pragma unroll
                for (int c = 0; c < N; ++c) cfma(q, Hk[c * N + jj], Rm[c]);
                if (!act) q = cmake(0.0, 0.0);
                q.x = row_sum_n<N>(q.x); q.y = row_sum_n<N>(q.y);
                if (j == 0 && t < steps) qS[kk * RL + tl] = q;
            }
            if (!SRC && i == L - 1 && row == 0) {                                // z = <W, Psi_N> = tr(Y_tau Psi_tau) at any tau (a workgroup's own copy)
                cplx dg = cmake(0.0, 0.0);
#pragma unroll
                for (int r = 0; r < N; ++r) if (r == j && h == 0) dg = Rm[r];
                dg.x = row_sum_n<N>(dg.x); dg.y = row_sum_n<N>(dg.y);
                if (lane16 == 0) { misc[M_Z] = dg.x; misc[M_Z + 1] = dg.y; }
            }
            if (i > 0) {
                cplx acc[N];
                MULB_M(Y, Kr[i], acc, m);
#pragma unroll
                for (int r = 0; r < N; ++r) Y[r] = acc[r];
            }
            if constexpr (SRC) {
                if (i > 0 && t >= 1 && t <= steps - 1) {                         // (i == 0: Y_{t0} belongs to the previous row's sweep)
                    cplx sf[QOC_SMALL_NF], ztau;
                    STATE_TERMS(Ps[i > 0 ? i - 1 : 0], has_speed);
                    ADD_SOURCES(Y, ztau);
                }
            }
        }
        __syncthreads();
        QSM_STAMP(4);

        // ---- P5: tail -- regularisers of the pulse, chain rule, grad_squared, stop rule, TF1 Adam (as finish_body, qoc_kernels_finish.h) ----
        {
#pragma clang fp contract(off)
            // the overlap: every workgroup of a control set holds its own copy (equal up to rounding) for its gradient elements; the loss and the
            // stop rule use workgroup 0's, which travels with the partial sums
            cplx z = SRC ? zfin : cmake(misc[M_Z], misc[M_Z + 1]);
            const double dt = d.dt;
            double reg = 0.0, g2 = 0.0;
            double dRb[QE];
#pragma unroll
            for (int e = 0; e < QE; ++e) dRb[e] = 0.0;
            if (d.has_band) {
                // bandpass regulariser (core/regularization_functions.py:47-67) by direct DFT, one workgroup per control set (the whole pulse is in LDS): the thread
                // that owns (control, slice t) forms the spectrum bin f = t, F_f = sum_t' w_t' e^{-2 pi i f t' / N}; value a sum_f cnt_f |F_f| with cnt_f = how often
                // the reference's two slices (f < lo; hi <= f < N / 2) contain f; then d/dw_t = a sum_f Re(cnt_f conj(F_f) / |F_f| e^{-2 pi i f t / N})
                const int half = steps / 2, lo_f = min(max(d.band_lo, 0), steps), hi_f = min(max(d.band_hi, 0), steps), fend = min(max(half, lo_f), steps);
#pragma unroll
                for (int e = 0; e < QE; ++e) {
                    if (e_ok[e]) {
                        const int kk = e_kk[e], f = e_tl[e];
                        const int cnt = (f < lo_f ? 1 : 0) + ((f >= hi_f && f < half) ? 1 : 0);
                        cplx ph = cmake(0.0, 0.0);
                        if (cnt > 0) {
                            double fr = 0.0, fi = 0.0;
                            int r = 0;
                            for (int t = 0; t < steps; ++t) {
                                const cplx tw = twS[r];
                                const double wv = Wv(kk, t);
                                fr = fma(wv, tw.x, fr); fi = fma(wv, tw.y, fi);
                                r += f; if (r >= steps) r -= steps;
                            }
                            const double mag = sqrt(fr * fr + fi * fi);
                            reg += d.a_band * (double)cnt * mag;
                            if (mag > 0.0) ph = cmake((double)cnt * fr / mag, -(double)cnt * fi / mag);
                        }
                        phS[kk * RL + f] = ph;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < QE; ++e) {
                    if (e_ok[e]) {
                        const int kk = e_kk[e], t = e_tl[e];
                        double acc = 0.0;
                        int r = 0;
                        for (int f = 0; f < fend; ++f) {
                            const cplx q = phS[kk * RL + f], tw = twS[r];
                            acc += q.x * tw.x - q.y * tw.y;
                            r += t; if (r >= steps) r -= steps;
                        }
                        dRb[e] = d.a_band * acc;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < QE; ++e) {
                if (e_ok[e]) {
                    const int kk = e_kk[e], tl = e_tl[e], t = g * RL + tl;
                    const double wv = Wv(kk, tl), wm1 = Wv(kk, tl - 1), wm2 = Wv(kk, tl - 2), wp1 = Wv(kk, tl + 1), wp2 = Wv(kk, tl + 2);
                    double dR = dRb[e];
                    if (d.has_amp) { reg += d.a_amp * 0.5 * wv * wv; dR += d.a_amp * wv; }                  // regularization_functions.py:15-18
                    if (d.has_env) {                                                                        // :21-25
                        const double ev = d.omg[(size_t)kk * steps + t];
                        reg += d.a_env * 0.5 * (ev * wv) * (ev * wv);
                        dR += d.a_env * ev * ev * wv;
                    }
                    if (d.has_dwdt) {                                                                       // :28-35 (padded differences: the last slice owns the closing one)
                        const double dm = (wv - wm1) / dt, dp = (wp1 - wv) / dt;
                        double acc = dm * dm;
                        if (t == steps - 1) acc += dp * dp;
                        reg += d.a_dwdt * 0.5 * acc;
                        dR += d.a_dwdt * (dm - dp) / dt;
                    }
                    if (d.has_d2wdt2) {                                                                     // :38-45
                        const double dt2 = dt * dt;
                        const double e0 = (wv - 2.0 * wm1 + wm2) / dt2, e1 = (wp1 - 2.0 * wv + wm1) / dt2, e2 = (wp2 - 2.0 * wp1 + wv) / dt2;
                        double acc = e0 * e0;
                        if (t == steps - 1) acc += e1 * e1 + e2 * e2;
                        reg += d.a_d2wdt2 * 0.5 * acc;
                        dR += d.a_d2wdt2 * (e0 - 2.0 * e1 + e2) / dt2;
                    }
                    const cplx q = qS[kk * RL + tl];
                    const double dLdu = SRC ? q.x : (c0 * z.x) * q.x + (c0 * z.y) * q.y;                    // Re(conj(c0 z) q')
                    const double gv = e_c[e] * (misc[M_MAXA + kk] * dLdu + dR);                     // tensorflow_state.py:176-178
                    e_g[e] = gv; g2 += gv * gv;
                }
            }
            wg_sum2<THREADS>(reg, g2, misc + M_RED);
            const bool defer = multi && ap.mode == 1 && it + 1 < sd.iters && it_count < ap.max_iterations;
            if (defer) {
                spec = true; sp_reg = reg; sp_g2 = g2; sp_z = z; sp_regstate = reg_state;
#pragma unroll
                for (int e = 0; e < QE; ++e) { sv_base[e] = e_base[e]; sv_m[e] = e_m[e]; sv_v[e] = e_v[e]; sv_w[e] = e_w[e]; sv_g[e] = e_g[e]; }
                sv_it_count = it_count; sv_adam_t = adam_t; sv_pow_b1 = pow_b1; sv_pow_b2 = pow_b2; sv_lr_run = lr_run;
            }
            if (multi && !defer) {
                // exchange B: partial sums of the regularisers and of grad_squared, summed in the same order by every workgroup
                double* mine = xB + (size_t)g * sd.xb_stride;
                if (tid == 0) { st_sc1(mine + 0, reg); st_sc1(mine + 1, g2); st_sc1(mine + 2, z.x); st_sc1(mine + 3, z.y); }
                publish_flag(flags + 4 * g + 1, epoch);
                wait_flags(flags + 1, G, epoch, sd.err);
                fetch_sc1<THREADS>(xsum, 4 * G, [&](int o) -> const double* { return xB + (size_t)(o >> 2) * sd.xb_stride + (o & 3); }, [](int) { return 0.0; });
                __syncthreads();
                reg = 0.0; g2 = 0.0;
                sum_groups2(xsum, G, reg, g2);
                z = cmake(xsum[2], xsum[3]);
            }
            QSM_STAMP(5);
            if (!defer) {
                g2 *= 0.5;
                const double loss = 1.0 - (z.x * z.x + z.y * z.y) / mm;
                out_loss = loss; out_regstate = reg_state; out_reg = loss + reg_state + reg; out_g2 = g2; out_z = z;
                if (ap.mode == 0) break;
                if (ap.mode == 1) {
                    const bool end = (loss < ap.conv_target) || (g2 < ap.min_grad) || (it_count >= ap.max_iterations);
                    if (end) { done_now = 1; break; }
                }
            }
            if (ap.mode == 1) it_count += 1;
            const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
            lr_run *= lr_step;                                     // rate e^{-it_count / decay}                     run_session.py:66
            const double lr = ap.mode == 1 ? lr_run : ap.lr[b];
            adam_t += 1;
            pow_b1 *= b1; pow_b2 *= b2;
            const double lr_t = lr * sqrt(1.0 - pow_b2) / (1.0 - pow_b1);
            __syncthreads();                                       // every thread has read its neighbours' controls
            const bool more = it + 1 < sd.iters;
#pragma unroll
            for (int e = 0; e < QE; ++e) {
                if (e_ok[e]) {
                    const double gv = e_g[e];
                    const double mv = b1 * e_m[e] + (1.0 - b1) * gv;
                    const double vv = b2 * e_v[e] + (1.0 - b2) * gv * gv;
                    e_m[e] = mv; e_v[e] = vv;
                    e_base[e] = e_base[e] - lr_t * mv / (sqrt(vv) + eps);
                    if (more) { sincos(e_base[e], &e_w[e], &e_c[e]); Wv(e_kk[e], e_tl[e]) = e_w[e]; }
                }
            }
            __syncthreads();
            QSM_STAMP(6);
        }
    }
#ifdef QOC_SMALL_TIMING
    if (blockIdx.x == 0 && tid == 0) {
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime(), ck1 = __builtin_amdgcn_s_memtime();
        printf("k_small_iter<%d,%d,%d,%d> G=%d iters=%d: launch %llu clk = %.2f us (100 MHz clock); last iteration: expm+row product %llu, up-sweep(+exchange A) %llu, "
               "walk+forward(+state terms, offsets) %llu, backward %llu, tail to sums(+exchange B) %llu, stop rule+Adam %llu clk\n", N, L, R, (int)SRC, G, sd.iters,
               ck1 - ck0, (double)(rt1 - rt0) / 100.0, stamp[1] - stamp[0], stamp[2] - stamp[1], stamp[3] - stamp[2], stamp[4] - stamp[3], stamp[5] - stamp[4],
               stamp[6] - stamp[5]);
        if (sd.G > 1) printf("   exchange A: local up-sweep %llu, payload stores + drain + flag + wait for all flags %llu, fetch of the payloads %llu, deferred stop rule %llu, upper sweep / range reduction %llu clk\n",
                             stamp[14] - stamp[1], stamp[15] - stamp[14], stamp[7] - stamp[15], stamp[16] - stamp[7], stamp[2] - stamp[16]);
        if (SRC) printf("   state-regulariser flow: walk for the start state %llu, forward + state terms %llu, sums (+ exchange A1) %llu, offsets of the row %llu, offset tree up-sweep %llu, "
                        "exchange A2 + upper offset tree %llu, terminal + walk for the costate %llu clk\n", stamp[8] - stamp[2], stamp[9] - stamp[8], stamp[10] - stamp[9], stamp[11] - stamp[10],
                        stamp[12] - stamp[11], stamp[13] - stamp[12], stamp[3] - stamp[13]);
    }
#endif

    // ---- epilogue: the state the C ABI reads back ----------------------------------------------------------------------------------------------
#pragma unroll
    for (int e = 0; e < QE; ++e) {
        if (e_ok[e]) {
            const size_t go = ((size_t)b * k + e_kk[e]) * steps + g * RL + e_tl[e];
            d.grad[go] = e_g[e];
            d.w[go] = e_w[e]; d.u[go] = misc[M_MAXA + e_kk[e]] * e_w[e];          // the controls the LAST evaluation ran on
            if (ap.mode != 0) { d.base[go] = e_base[e]; d.adam_m[go] = e_m[e]; d.adam_v[go] = e_v[e]; }
        }
    }
    // unitary mode: final_state = (product of all propagators) U0 is one product away from the root of the tree, unitary_scale = (1/n) sum_c |sum_a X[c][a]|^2 a row
    // reduction away from that (core/tensorflow_state.py:204-227): formed here, so that a poll of the progress line needs no second pass over the pulse
    const cplx* root = lnode(treeM, LR, 0);
    const bool want_final = !d.state_transfer && tree_is_last && g == 0;
    if (want_final && multi) {
        if constexpr (SRC) root = unode(treeU, sd.LG, 0);
        else {                                                      // (the iterations reduce ranges, not the tree: the product of ALL subtree roots once per launch, by every row of workgroup 0)
            cplx* bufs[2] = { treeU + (size_t)sd.Gp * NP, treeU + (size_t)(sd.Gp + sd.Gp / 2 + 2) * NP };
            const cplx* cur = unode(treeU, 0, 0);
            int which = 0;
            for (int c = sd.Gp; c > 1; c >>= 1) {
                cplx* nxt = bufs[which];
                for (int w = row; w < (c >> 1); w += R) {
                    const cplx* e0 = cur + (size_t)(2 * w) * NP;
                    cplx Ar[N], xl[N], acc[N];
#pragma unroll
                    for (int r = 0; r < N; ++r) { xl[r] = ldm<SPL>(e0 + r * N + jj, h); Ar[r] = ldm<SPL>(e0 + NP + r * N + jj, h); }
                    MULB(Ar, xl, acc);
                    if (actw) {
#pragma unroll
                        for (int r = 0; r < N; ++r) nxt[(size_t)w * NP + r * N + j] = acc[r];
                    }
                }
                __syncthreads();
                cur = nxt; which ^= 1;
            }
            root = cur;
        }
    }
    if (want_final && row == 0) {
        cplx Ar[N], Uc[N], X[N];
#pragma unroll
        for (int r = 0; r < N; ++r) {
            Ar[r] = ldm<SPL>(root + r * N + jj, h);
            Uc[r] = swp((r < n && j < n) ? d.U0[r * n + j] : cmake(0.0, 0.0), h);
        }
        MULB(Ar, Uc, X);
        double part = 0.0;
        const bool wr = j < n && h == 0;                                            // (half 0 holds (re, im))
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r < n && wr) d.Xfinal[(size_t)b * n * n + r * n + j] = X[r];
            const double sx = row_sum16(wr ? X[r].x : 0.0), sy = row_sum16(wr ? X[r].y : 0.0);
            if (r < n) part += sx * sx + sy * sy;
        }
        if (lane16 == 0) { d.uscale[b] = part / (double)n; sd.final_valid[b] = 1; }
    }
    if (g == 0 && tid == 0) {
        d.loss[b] = out_loss; d.reg_loss[b] = out_reg; d.g2[b] = out_g2; d.reg_state[b] = out_regstate; d.zfin[b] = out_z;
        if (ap.mode != 0) { d.iters[b] = it_count; d.adam_t[b] = adam_t; if (done_now) d.done[b] = 1; }
    }
}

#undef QF
#undef QPAD
#undef MULB
#undef MULB_M
#undef MULB_K
#undef STATE_TERMS
#undef ADD_SOURCES

}  // namespace qsm
