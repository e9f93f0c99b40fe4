// qoc_small.h -- the workgroup-resident GRAPE iteration for small Hilbert spaces (n <= 12): QOC_PATH_SMALL.
//
// The reference's own users run qubits, qutrits and two / three transmons, ONE control set per Grape() call
// (core/system_parameters.py:272-284; the n < 10 branch of Choose_exp_terms, :128-145, exists for these sizes; what runs is
// core/tensorflow_state.py:25-46, 204-242, 323-356 and the loop of core/run_session.py:47-69).  At these sizes an iteration is a few
// hundred kFLOP: what it costs is launches and round trips, not arithmetic.  Here the WHOLE iteration -- controls, per-slice exponentials,
// the chain, costates, control gradients, regularisers, TF1 Adam and the stop rule -- runs inside ONE launch, and qoc_iterate(iters) /
// qoc_run_adam loop inside that launch; nothing but the per-seed scalars and, at the end, the pulse goes back to HBM.
//
// Mapping (csrc/qoc_small_kernel.h): a ROW of 16 lanes owns L consecutive time slices; lane j of the row holds COLUMN j of every n x n matrix
// in registers (plain complex fp64, no padding to an MFMA tile).  A product C = A B is n^2 complex MACs per lane on v_fmac_f64_dpp
// row_newbcast:c -- A[r][c] reaches all lanes of the row from lane c's register r inside the FMA, so neither operand passes through LDS.  The
// chain over the rows is a product TREE in LDS (log-depth up-sweep, then a barrier-free walk from the root down to each row's start state and
// end costate); a pulse that does not fit one workgroup takes G workgroups per control set, which hand their subtree products and partial
// sums to each other through write-through (sc1) stores, one flag per workgroup and epoch (cdna_hip_programming.md, Guideline 16), and then
// reduce the two ranges of subtree products they need (start state, end costate) instead of a common upper tree.  5 <= n <= 8: lanes j and j + 8
// of a row share column j and form the real / imaginary parts of a product (2 n^2 FMAs per lane instead of 4 n^2).
#pragma once
#include <string>
#include <vector>

#include "qoc_common.h"

#define QOC_SMALL_NF 4          // forbidden levels the kernel keeps per time point (more: another path)
#define QOC_SMALL_MAXG 32       // workgroups per control set
#define QOC_SMALL_WG_BUDGET 128 // control sets x workgroups per control set that may spin on each other (all must be resident: 256 CUs)

// Distance between the nodes of the product trees, in complex numbers.  (n = 4, 8, 12 put the nodes on the 256-byte grid of the LDS banks; 64 bytes of padding per node
// were measured and changed nothing -- profiles/EXPERIMENTS.md, round 6 -- so the nodes are dense.)
__host__ __device__ constexpr int qoc_small_node(int N) { return N * N; }

// LDS carve of one workgroup, in units of one complex (16 bytes); the same function runs on the host (launch size) and in the kernel.
struct QocSmallLayout {
    int hsc, hst, vfs, psi0, wd, wcol, v0, psin, treeM, treeU, treeO, treeOU, qS, wS, misc, xsum, twS, phS, total;
};
__host__ __device__ inline QocSmallLayout qoc_small_layout(int N, int R, int L, int k, int m, int Gp, bool src, bool band = false) {
    QocSmallLayout lo;
    const int NN = N * N, RL = R * L, NP = qoc_small_node(N);
    int o = 0;
    lo.hsc = o; o += (k + 1) * NN;                 // generators / 2^s, row-major (lane j reads [r][j])
    lo.hst = o; o += k * NN;                       // control Hamiltonians transposed (lane a reads [c][a] = H_k[a][c])
    lo.vfs = o; o += src ? QOC_SMALL_NF * N : 0;   // bras of the forbidden levels
    lo.psi0 = o; o += NN;                          // U0 V (start of the chain), zero-padded to N columns
    lo.wd = o; o += NN;                            // W^dagger, [j'][a]
    lo.wcol = o; o += src ? NN : 0;                // W, [r][j]
    lo.v0 = o; o += src ? NN : 0;                  // V = inter_vecs[0]
    lo.psin = o; o += src ? NN : 0;                // Psi_N of this evaluation
    lo.treeM = o; o += (2 * R - 1) * NP;           // product tree of the rows of this workgroup
    // ... of the workgroups of the control set (leaves: their subtree roots).  Without a state regulariser the levels above the leaves are two ping-pong buffers of the
    // range reduction instead (Gp / 2 + 2 and Gp / 4 + 2 nodes)
    lo.treeU = o; o += (src || Gp == 1 ? 2 * Gp - 1 : Gp + Gp / 2 + Gp / 4 + 4) * NP;
    lo.treeO = o; o += src ? (2 * R - 1) * m * N : 0;   // offsets of the affine costate recursion, same shape; m x N nodes (rows j' < m of Y)
    lo.treeOU = o; o += src ? (2 * Gp - 1) * m * N : 0;
    lo.qS = o; o += k * RL;                        // <Lambda_{t+1}, H_k Psi_{t+1}> of the own slices
    lo.wS = o; o += (k * (RL + 4) + 1) / 2;        // sin(base) of the own slices + two halo slices either side (doubles)
    lo.misc = o; o += 64;                          // reductions, scalars, inverse factorials (128 doubles)
    lo.xsum = o; o += Gp > 1 ? 2 * Gp : 0;         // partial sums of the workgroups of the control set (4 doubles each)
    lo.twS = o; o += band ? RL : 0;                // bandpass regulariser (one workgroup per control set): e^{-2 pi i r / steps}, r < steps
    lo.phS = o; o += band ? k * RL : 0;            // ... and cnt_f conj(F_f) / |F_f| of the pulse spectrum
    lo.total = o;
    return lo;
}

struct QocSmallDev {            // kernel argument beside QocDev / QocAdamDev
    int iters;                  // loop iterations inside the launch
    int G, Gp, LG;              // workgroups per control set, padded to a power of two, log2 of that
    int Teff;                   // Taylor terms beyond the identity (unitary: T; state transfer: T - 1, no squarings)
    double* xA;                 // [2][B][G][XA] exchange A: subtree product + halo controls + the deferred partial sums of the previous iteration
    double* xB;                 // [B][G][XB] exchange B: partial sums of the tail (+ the overlap z from workgroup 0)
    double* xS;                 // [B][G][XS] exchanges of the state-regulariser flow: partial sums, Psi_N, z_N; offsets
    unsigned* flags;            // [B][G][4] one word per workgroup and exchange kind: the epoch it has published
    unsigned* err;              // [1] set when a spin timed out (the host turns it into QOC_ERR_HIP)
    int* final_valid;           // [B] 1: the launch left final_state / unitary_scale of its last evaluation in d.Xfinal / d.uscale (unitary mode)
    int xa_stride, xb_stride, xs_stride;
    long long xa_parity;        // doubles between the two copies of xA (exchange A alternates between them)
};

struct QocSmall {
    bool on = false;
    int N = 0, L = 0, R = 0, G = 1, inst = -1;
    bool src = false;
    size_t lds_bytes = 0;
    QocSmallDev sd{};
    size_t flag_bytes = 0;
    int B = 0;
};
// did the last launch leave final_state / unitary_scale of every control set behind (else the read-back re-forms them)?  Synchronises the stream.
bool qoc_small_final_valid(const QocSmall& sm, hipStream_t s);

// host entry points (csrc/qoc_small.hip)
bool qoc_small_supported(const QocDev& d, bool antiherm, int G_req, int R_req, std::string* why);
// picks (N, R, L, G) for the PLANNED batch (d.Bplan) and allocates the exchange buffers; G_req > 0 pins the workgroups per control set, R_req > 0 the rows
// of 16 lanes per workgroup (A/B runs)
int qoc_small_setup(QocSmall& sm, const QocDev& d, bool antiherm, int G_req, int R_req, std::vector<void*>& allocs, std::string& msg);
// one launch = `iters` loop iterations (mode 1) or one evaluation / explicit step (modes 0, 2)
int qoc_small_launch(QocSmall& sm, const QocDev& d, const QocAdamDev& ap, int iters, hipStream_t s, std::string& msg);
// worth taking over the MFMA latency mode / batch kernels for this problem and planned batch? (AUTO; tests/test_auto_plan.py restates it)
bool qoc_small_auto(const QocDev& d, bool antiherm);
