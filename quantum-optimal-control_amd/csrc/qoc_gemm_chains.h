// qoc_gemm_chains.h -- persistent thin-chain kernels of the GEMM path (N <= 64, m <= 8): k_gemm_chain_fwd (y <- K y + E),
// (with CONJ on the transposed copies: y <- K^H y + E) and k_gemm_taylor_chain (state-transfer Taylor recursion), with their
// lane mappings.
// Reference semantics: core/tensorflow_state.py:214-242 (chains), :77-133 (matvecexp forward / custom gradient).
#pragma once
#include "qoc_gemm_tiles.h"

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
    cplx* Out2; long long sO2b, sO2s; int n2;   // k_gemm_taylor_chain_dpp: second copy of the output, rows < n2 contiguous per step (the API's inter_vecs: no unpad pass)
    int ldE;                                    // k_gemm_taylor_chain_dpp: row stride of E (0 = QOC_TW, the thin panels; 1 = one vector per step, contiguous)
};

// dpp_xor<OFF> and lds_barrier() live in qoc_common.h (the MFMA-path backward sweep uses them too).

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

// ---- register-blocked forward mat-vec mapping (k_gemm_chain_fwd, k_gemm_taylor_chain) ---------------------------------
// Thread (g, c) = (tid / LG, tid % LG) owns the R x CC block rows R*g + rr, columns c + LG*cc of K (R = N*LG/256, CC = N/LG): an LG-lane
// group reads LG*16 contiguous bytes per load, a thread reads CC entries of the vector per slot and the R*MV partial sums are combined by
// a reduce-scatter butterfly over the LG lanes (DPP), after which lane c holds the finished values x = base + s, s < SPLB, of its row
// group in (row, slot)-major order; addend, LDS write and output store are done by the owner of each value.  LG trades LDS reads of the
// vector (CC per thread) against butterfly levels (log2 LG): see FwdMap below for the choice per N.
template <int N, int MV, int LG, int NTHR = 256>
struct BlockMap {
    // LG lanes share a group of R rows; a thread owns R x CC entries: rows R*g + rr, columns c + LG*cc
    static constexpr int R = N * LG / NTHR, CC = N / LG, EL = R * CC, V = R * MV;
    static constexpr int THREADS = NTHR;
    static constexpr int NSLB = V < LG ? V : LG, SPLB = V / NSLB;
    int g, c, base;
    __device__ __forceinline__ BlockMap(int tid) : g(tid / LG), c(tid % LG), base(((tid % LG) / (LG / NSLB)) * SPLB) {}
    __device__ __forceinline__ int row(int s) const { return R * g + (base + s) / MV; }
    __device__ __forceinline__ int slot(int s) const { return (base + s) % MV; }
    __device__ __forceinline__ void load(cplx (&kd)[EL], const cplx* Kj) const {
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) kd[rr * CC + cc] = Kj[(size_t)(R * g + rr) * N + c + LG * cc];
    }
    // acc[rr*MV + jv] = sum_cc K[rr][cc] * v[c + LG cc][jv], then the LG-lane reduce-scatter: acc[0..SPLB) are this lane's values
    template <bool CONJ>
    __device__ __forceinline__ void matvec(const cplx (&ku)[EL], const cplx* __restrict__ v, cplx (&acc)[V]) const {
#pragma unroll
        for (int x = 0; x < V; ++x) acc[x] = cmake(0.0, 0.0);
        if constexpr (CC * MV <= 16) {
            // all vector reads in flight before the first FMA: left to itself hipcc issues one ds_read per column block and waits for
            // it on the spot (four exposed LDS latencies per mat-vec, the step of a latency-bound chain)
            cplx vv[CC][MV];
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
#pragma unroll
                for (int jv = 0; jv < MV; ++jv) vv[cc][jv] = v[(c + LG * cc) * MV + jv];
            __builtin_amdgcn_sched_barrier(0);
#ifdef QOC_CHAIN_SPLIT_ACC
            // two partial sums per value (even / odd column blocks): twice as many independent FMA chains for a wave that is alone on its SIMD
            cplx acc1[V];
#pragma unroll
            for (int x = 0; x < V; ++x) acc1[x] = cmake(0.0, 0.0);
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int jv = 0; jv < MV; ++jv) {
                        cplx& t = (cc & 1) ? acc1[rr * MV + jv] : acc[rr * MV + jv];
                        if (CONJ) cfma_conj(t, ku[rr * CC + cc], vv[cc][jv]); else cfma(t, ku[rr * CC + cc], vv[cc][jv]);
                    }
#pragma unroll
            for (int x = 0; x < V; ++x) acc[x] = cadd(acc[x], acc1[x]);
#else
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int jv = 0; jv < MV; ++jv) { if (CONJ) cfma_conj(acc[rr * MV + jv], ku[rr * CC + cc], vv[cc][jv]); else cfma(acc[rr * MV + jv], ku[rr * CC + cc], vv[cc][jv]); }
#endif
        } else {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                cplx vv[MV];
#pragma unroll
                for (int jv = 0; jv < MV; ++jv) vv[jv] = v[(c + LG * cc) * MV + jv];
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int jv = 0; jv < MV; ++jv) { if (CONJ) cfma_conj(acc[rr * MV + jv], ku[rr * CC + cc], vv[jv]); else cfma(acc[rr * MV + jv], ku[rr * CC + cc], vv[jv]); }
            }
        }
        chain_butterfly<V, LG / 2, SPLB, V>(acc, c);
    }
};

// N = 32: a row per thread (LG = 8: 4 vector reads per slot either way, a 3-level butterfly instead of 4; per 1000-slice direct
// iteration 4.65 against 4.81 ms with LG = 16).  N = 64, one or two vectors: two rows per thread (LG = 8) -- 8 vector reads, all in
// flight together, and a 26-instruction butterfly instead of 64; with the reads issued one at a time (before round 2's end) the
// fewer-reads mapping LG = 16 was the faster one (10.0 against 13.0 ms with a row per thread, LG = 4).
#ifndef QOC_CHAIN_LG64
#define QOC_CHAIN_LG64 8
#endif
template <int N, int MV> struct FwdMap { using type = BlockMap<N, MV, 16>; };
template <int MV> struct FwdMap<32, MV> { using type = BlockMap<32, MV, 8>; };
template <> struct FwdMap<64, 1> { using type = BlockMap<64, 1, QOC_CHAIN_LG64>; };
template <> struct FwdMap<64, 2> { using type = BlockMap<64, 2, QOC_CHAIN_LG64>; };
// the Taylor chains of the direct state-transfer route (k_gemm_taylor_chain): thread count and lane-group width at N = 64, one or two vectors
#ifndef QOC_TAYLOR_THREADS64
#define QOC_TAYLOR_THREADS64 256
#endif
#ifndef QOC_TAYLOR_LG64
#define QOC_TAYLOR_LG64 QOC_CHAIN_LG64
#endif
template <int N, int MV> struct TaylorMap { using type = typename FwdMap<N, MV>::type; };
template <> struct TaylorMap<64, 1> { using type = BlockMap<64, 1, QOC_TAYLOR_LG64, QOC_TAYLOR_THREADS64>; };
template <> struct TaylorMap<64, 2> { using type = BlockMap<64, 2, QOC_TAYLOR_LG64, QOC_TAYLOR_THREADS64>; };

// y <- K_j y + E_j (CONJ: conj(K_j) y + E_j) with the mapping FwdMap picks for N.  Pipeline: K_j / E_j of the next two steps
// are in flight in three register stages used round-robin by a 3x unrolled branch-free loop; y lives in LDS (double buffer)
template <int N, int MV, bool CONJ, bool HAS_OUT>
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
        bm.template matvec<CONJ>(ku, y[cur], acc);
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
// Two argument sets in one launch (workgroups 0 .. nb0-1 run a0, the rest a1): without a state regulariser the costate is linear
// in the overlap z, so the backward chain starts from -(2/m^2) W and runs BESIDE the forward chain (k_gemm_scale_lam applies z
// afterwards) -- the batched direct route is two latency-bound chains of 1000 x (T-1) dependent mat-vecs, on 64 of the 256 CUs each.
template <int N, int MV>
__global__ void __launch_bounds__((TaylorMap<N, MV>::type::THREADS)) k_gemm_taylor_chain(ChainArgs a0, ChainArgs a1, int nb0) {
    using BM = typename TaylorMap<N, MV>::type;
    constexpr int EL = BM::EL, SPL = BM::SPLB, V = BM::V;
    __shared__ __attribute__((aligned(16))) cplx y[2][N * MV];
    __shared__ double tinv[64];                                            // 1 / ii!  (a division per term sat on the chain: ~14 fp64 instructions)
    const BM bm(threadIdx.x);
    const bool second = (int)blockIdx.x >= nb0;
    const ChainArgs a = second ? a1 : a0;
    const int b = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
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
    if (threadIdx.x < 64) {
        double fact = 1.0;
        for (int ii = 2; ii <= (int)threadIdx.x; ++ii) fact *= (double)ii;   // the running factorial of :92-95
        tinv[threadIdx.x] = 1.0 / fact;
    }
    int cur = 0;
    auto step = [&](int j, const cplx (&ku)[EL], const cplx (&eu)[SPL]) {
        cplx out[SPL];
#pragma unroll
        for (int sl = 0; sl < SPL; ++sl) out[sl] = yfin[sl];
        for (int ii = 1; ii < a.nterms; ++ii) {
            double inv = tinv[ii & 63];
            if (ii >= 64) { double fact = 1.0; for (int q = 2; q <= ii; ++q) fact *= (double)q; inv = 1.0 / fact; }
            cplx acc[V];
            bm.template matvec<false>(ku, y[cur], acc);
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

#include "qoc_gemm_chain_dpp.h"     // k_gemm_taylor_chain_dpp: N = 64, one vector, generators column-major
#include "qoc_gemm_chain_sq.h"      // k_gemm_taylor_chain_sq: the same chain on [B | B^2], 1 + ceil(T/2) - 1 dependent mat-vecs per slice

template <int N>
static inline void qoc_taylor_chain_launch_n(const ChainArgs& a0, const ChainArgs& a1, int nb0, int blocks, hipStream_t s) {
    const int mv = a0.m <= 1 ? 1 : (a0.m <= 2 ? 2 : (a0.m <= 4 ? 4 : 8));
    if (mv == 1) hipLaunchKernelGGL((k_gemm_taylor_chain<N, 1>), dim3(blocks), dim3(TaylorMap<N, 1>::type::THREADS), 0, s, a0, a1, nb0);
    else if (mv == 2) hipLaunchKernelGGL((k_gemm_taylor_chain<N, 2>), dim3(blocks), dim3(TaylorMap<N, 2>::type::THREADS), 0, s, a0, a1, nb0);
    else if (mv == 4) hipLaunchKernelGGL((k_gemm_taylor_chain<N, 4>), dim3(blocks), dim3(TaylorMap<N, 4>::type::THREADS), 0, s, a0, a1, nb0);
    else hipLaunchKernelGGL((k_gemm_taylor_chain<N, 8>), dim3(blocks), dim3(TaylorMap<N, 8>::type::THREADS), 0, s, a0, a1, nb0);
}
// dpp: 0 = k_gemm_taylor_chain, 1 = k_gemm_taylor_chain_dpp on full generators, 2 = on packed anti-Hermitian generators (qoc_gemm_chain_dpp.h),
//      3 = k_gemm_taylor_chain_sq on packed [B | B^2] (qoc_gemm_chain_sq.h), 10 / 12 / 14 = k_gemm_taylor_chain_dpp on the first 4 x 10 / 12 / 14
//      columns of full generators (padded problems of at most 40 / 48 / 56 levels)
static inline void qoc_taylor_chain_launch(int N, ChainArgs a, const cplx* zeros, int blocks, hipStream_t s, int dpp = 0) {
    if (!a.E) { a.E = zeros; a.sEb = a.sEc = a.sEs = 0; }
    if (dpp == 10) { hipLaunchKernelGGL((k_gemm_taylor_chain_dpp<false, 10>), dim3(blocks), dim3(256), 0, s, a, a, blocks); return; }
    if (dpp == 12) { hipLaunchKernelGGL((k_gemm_taylor_chain_dpp<false, 12>), dim3(blocks), dim3(256), 0, s, a, a, blocks); return; }
    if (dpp == 14) { hipLaunchKernelGGL((k_gemm_taylor_chain_dpp<false, 14>), dim3(blocks), dim3(256), 0, s, a, a, blocks); return; }
    if (dpp == 3) { hipLaunchKernelGGL(k_gemm_taylor_chain_sq, dim3(blocks), dim3(256), 0, s, a, a, blocks); return; }
    if (dpp == 2) { hipLaunchKernelGGL(k_gemm_taylor_chain_dpp<true>, dim3(blocks), dim3(256), 0, s, a, a, blocks); return; }
    if (dpp) { hipLaunchKernelGGL(k_gemm_taylor_chain_dpp<false>, dim3(blocks), dim3(256), 0, s, a, a, blocks); return; }
    if (N == 32) qoc_taylor_chain_launch_n<32>(a, a, blocks, blocks, s); else qoc_taylor_chain_launch_n<64>(a, a, blocks, blocks, s);
}
// two chains side by side: `blocks` workgroups each
static inline void qoc_taylor_chain_launch2(int N, ChainArgs a0, ChainArgs a1, const cplx* zeros, int blocks, hipStream_t s, int dpp = 0) {
    if (!a0.E) { a0.E = zeros; a0.sEb = a0.sEc = a0.sEs = 0; }
    if (!a1.E) { a1.E = zeros; a1.sEb = a1.sEc = a1.sEs = 0; }
    if (dpp == 10) { hipLaunchKernelGGL((k_gemm_taylor_chain_dpp<false, 10>), dim3(2 * blocks), dim3(256), 0, s, a0, a1, blocks); return; }
    if (dpp == 12) { hipLaunchKernelGGL((k_gemm_taylor_chain_dpp<false, 12>), dim3(2 * blocks), dim3(256), 0, s, a0, a1, blocks); return; }
    if (dpp == 14) { hipLaunchKernelGGL((k_gemm_taylor_chain_dpp<false, 14>), dim3(2 * blocks), dim3(256), 0, s, a0, a1, blocks); return; }
    if (dpp == 3) { hipLaunchKernelGGL(k_gemm_taylor_chain_sq, dim3(2 * blocks), dim3(256), 0, s, a0, a1, blocks); return; }
    if (dpp == 2) { hipLaunchKernelGGL(k_gemm_taylor_chain_dpp<true>, dim3(2 * blocks), dim3(256), 0, s, a0, a1, blocks); return; }
    if (dpp) { hipLaunchKernelGGL(k_gemm_taylor_chain_dpp<false>, dim3(2 * blocks), dim3(256), 0, s, a0, a1, blocks); return; }
    if (N == 32) qoc_taylor_chain_launch_n<32>(a0, a1, blocks, 2 * blocks, s); else qoc_taylor_chain_launch_n<64>(a0, a1, blocks, 2 * blocks, s);
}

template <int N, bool CONJ, bool HAS_OUT>
static inline void qoc_chain_fwd_launch_n(const ChainArgs& a, int blocks, hipStream_t s) {
    const int mv = a.m <= 1 ? 1 : (a.m <= 2 ? 2 : (a.m <= 4 ? 4 : 8));
    if (mv == 1) hipLaunchKernelGGL((k_gemm_chain_fwd<N, 1, CONJ, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 2) hipLaunchKernelGGL((k_gemm_chain_fwd<N, 2, CONJ, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else if (mv == 4) hipLaunchKernelGGL((k_gemm_chain_fwd<N, 4, CONJ, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gemm_chain_fwd<N, 8, CONJ, HAS_OUT>), dim3(blocks), dim3(256), 0, s, a);
}
// conj = true: y <- conj(M_j) y, used with M_j = K_j^T for the backward chains (K_j^H = conj(K_j^T): the rows of the transposed
// copy are read with the same coalesced pattern as the forward chains)
template <int N>
static inline void qoc_chain_launch_c(bool conj, const ChainArgs& a, int blocks, hipStream_t s) {
    if (conj) { if (a.Out) qoc_chain_fwd_launch_n<N, true, true>(a, blocks, s); else qoc_chain_fwd_launch_n<N, true, false>(a, blocks, s); }
    else { if (a.Out) qoc_chain_fwd_launch_n<N, false, true>(a, blocks, s); else qoc_chain_fwd_launch_n<N, false, false>(a, blocks, s); }
}
// `zeros` = a zero thin buffer (N x 32) used as the addend when the chain has none
static inline void qoc_chain_launch(int N, bool conjt, ChainArgs a, const cplx* zeros, int blocks, hipStream_t s) {
    if (a.len <= 0 && !a.Fin && !a.store_initial) return;
    if (!a.E) { a.E = zeros; a.sEb = a.sEc = a.sEs = 0; }
    if (N == 32) qoc_chain_launch_c<32>(conjt, a, blocks, s); else qoc_chain_launch_c<64>(conjt, a, blocks, s);
}

// ---- chunk-boundary vectors in log depth ----------------------------------------------------------------------------
// The boundary chains (Psibnd[c+1] = P_c Psibnd[c], Ebnd[c-1] = P_c^H Ebnd[c]) were NC - 1 dependent steps in one workgroup
// per seed.  The pairwise product tree above the chunk products already exists (unitary mode needs its root for
// final_state); level r, node j is the product of the chunks [j 2^r, min((j+1) 2^r, NC)).  So every boundary vector is
// <= 2 log2(NC) node applications away from the initial vector, independently of the others: one workgroup per (seed, chunk)
// walks the binary decomposition of its prefix [0, c) (forward) or suffix [c+1, NC) (backward, conjugate transposes, far end
// first).  Nodes are staged through a padded LDS image (both M x and M^H x read it conflict-free, so no transposed copies of
// the upper tree levels are needed) and the next node is fetched into registers while the current one multiplies.
struct ScanArgs {
    const cplx* lvl[10]; long long sLb[10]; int cnt[10];   // level r: node array of seed b at lvl[r] + b*sLb[r], cnt[r] nodes
    int levels;
    const cplx* X0; long long sXb;                          // [N][QOC_TW] start vector of seed b
    cplx* Out; long long sOb, sOc;                          // result of (b, c) at Out + b*sOb + c*sOc, [N][QOC_TW]
    int NC, c0, nchains, suffix;                            // chunks c0 .. c0 + nchains - 1 per seed
};

template <int N>
__global__ void __launch_bounds__(256) k_gemm_scan_nodes(ScanArgs a) {
    constexpr int LD = N + 1, PER = N * N / 256, NO = N / 32;        // NO outputs (columns tid/N... ) per thread
    extern __shared__ __attribute__((aligned(16))) cplx sc_lds[];
    cplx* M = sc_lds;                       // [N][LD]
    cplx* xv = sc_lds + N * LD;             // [2][N][8]
    __shared__ int s_list[24];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / a.nchains, c = a.c0 + blockIdx.x % a.nchains;
    if (tid == 0) {
        int n = 0;
        if (!a.suffix) {                                            // prefix [0, c): big blocks first = application order
            int pos = 0;
            for (int r = a.levels - 1; r >= 0; --r)
                if (pos + (1 << r) <= c) { s_list[n++] = (r << 16) | (pos >> r); pos += 1 << r; }
        } else {                                                    // suffix [c+1, NC): collected near end first, applied in reverse
            int lo = c + 1;
            while (lo < a.NC) {
                int r = 0;
                while (r + 1 < a.levels && (lo & ((1 << (r + 1)) - 1)) == 0 && (lo >> (r + 1)) < a.cnt[r + 1]) ++r;
                s_list[n++] = (r << 16) | (lo >> r);
                lo = min(lo + (1 << r), a.NC);
            }
        }
        s_n = n;
    }
    const int row = tid % N, cg = tid / N;                           // outputs (row, cg + (256/N) * o), o < NO
    {
        const cplx* x0 = a.X0 + (size_t)b * a.sXb;
        for (int e = tid; e < N * 8; e += 256) xv[e] = x0[(e >> 3) * QOC_TW + (e & 7)];
    }
    __syncthreads();
    const int n = s_n;
    auto node_ptr = [&](int i) {
        const int code = s_list[a.suffix ? n - 1 - i : i], r = code >> 16, j = code & 0xffff;
        return a.lvl[r] + (size_t)b * a.sLb[r] + (size_t)j * N * N;
    };
    cplx reg[PER];
    if (n > 0) {
        const cplx* p = node_ptr(0);
#pragma unroll
        for (int x = 0; x < PER; ++x) reg[x] = p[tid + 256 * x];
    }
    int cur = 0;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int x = 0; x < PER; ++x) { const int e = tid + 256 * x; M[(e / N) * LD + (e % N)] = reg[x]; }
        lds_barrier();
        {
            const cplx* p = node_ptr(min(i + 1, n - 1));             // clamped: the last step re-reads its own node
#pragma unroll
            for (int x = 0; x < PER; ++x) reg[x] = p[tid + 256 * x];
        }
        cplx acc[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[o] = cmake(0.0, 0.0);
        const cplx* xc = xv + cur * N * 8;
        if (!a.suffix) {
#pragma unroll 8
            for (int k = 0; k < N; ++k) {
                const cplx mv = M[row * LD + k];
#pragma unroll
                for (int o = 0; o < NO; ++o) cfma(acc[o], mv, xc[k * 8 + cg + (256 / N) * o]);
            }
        } else {
#pragma unroll 8
            for (int k = 0; k < N; ++k) {
                const cplx mv = M[k * LD + row];                      // (M^H x)[row] = sum_k conj(M[k][row]) x[k]
#pragma unroll
                for (int o = 0; o < NO; ++o) cfma_conj(acc[o], mv, xc[k * 8 + cg + (256 / N) * o]);
            }
        }
        cplx* xn = xv + (cur ^ 1) * N * 8;
#pragma unroll
        for (int o = 0; o < NO; ++o) xn[row * 8 + cg + (256 / N) * o] = acc[o];
        lds_barrier();
        cur ^= 1;
    }
    cplx* out = a.Out + (size_t)b * a.sOb + (size_t)c * a.sOc;
    const cplx* xc = xv + cur * N * 8;
    for (int e = tid; e < N * QOC_TW; e += 256) out[e] = (e & (QOC_TW - 1)) < 8 ? xc[(e / QOC_TW) * 8 + (e & (QOC_TW - 1))] : cmake(0.0, 0.0);
}
static inline size_t qoc_scan_lds(int N) { return ((size_t)N * (N + 1) + 2 * (size_t)N * 8) * sizeof(cplx); }
static inline void qoc_scan_launch(int N, const ScanArgs& a, int B, hipStream_t s) {
    if (a.nchains <= 0) return;
    if (N == 32) hipLaunchKernelGGL(k_gemm_scan_nodes<32>, dim3(B * a.nchains), dim3(256), qoc_scan_lds(32), s, a);
    else hipLaunchKernelGGL(k_gemm_scan_nodes<64>, dim3(B * a.nchains), dim3(256), qoc_scan_lds(64), s, a);
}
