// qoc_common.h -- device-side problem description and complex-fp64 helpers shared by all kernels.
// gfx950 only: no CUDA shims, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double2 cplx;   // (x = re, y = im), layout-compatible with numpy complex128

__device__ __forceinline__ cplx cmake(double r, double i) { cplx c; c.x = r; c.y = i; return c; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmake(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cmake(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cconj(cplx a) { return cmake(a.x, -a.y); }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return cmake(a.x * s, a.y * s); }
// acc += a*b
__device__ __forceinline__ void cfma(cplx& acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
__device__ __forceinline__ void cfma_conj(cplx& acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(-a.y, b.x, acc.y);
}

#define QOC_BLOCK 256

// Everything a kernel needs, passed by value (kernarg segment -> SGPRs).
struct QocDev {
    int n, k, steps, m, T, s, B;
    int Bplan;           // the batch size AUTO decisions are taken for (qoc_config.plan_seeds, else B): never an array extent
    int state_transfer;
    double dt;
    // regularisers (coefficients already divided by steps, regularization_functions.py:16,22,32,...)
    int has_amp, has_env, has_dwdt, has_d2wdt2, has_speed, has_band;
    double a_amp, a_env, a_dwdt, a_d2wdt2, a_speed, a_band;
    int band_lo, band_hi;
    int n_forb, forbid_dressed;
    const int* forb_state;    // [n_forb] device arrays: the reference accepts lists of any length (regularization_functions.py:81)
    const double* forb_a;     // [n_forb] coefficient / steps
    // constants in HBM
    const cplx* Hs;      // [k+1][n][n]
    const cplx* U0;      // [n][n]
    const cplx* V;       // [n][m]
    const cplx* W;       // [n][m]
    const cplx* Psi0;    // [n][m] = U0*V (unitary mode start vector); = V in state transfer
    const cplx* Vs;      // [n][n] or null
    const double* maxA;  // [k]
    const double* omg;   // [k][steps] or null
    // trainable + optimizer state
    double* base;        // [B][k][steps]
    double* adam_m;
    double* adam_v;
    int* adam_t;         // [B]
    int* iters;          // [B]
    int* done;           // [B]
    int skip_done;       // loop iterations only: kernels leave finished seeds untouched (their last evaluation stays readable)
    int uscale_in_loss;  // unitary mode: k_loss also forms unitary_scale from Xfinal (MFMA path: saves a launch per iteration)
    // per-evaluation intermediates
    double* w;           // [B][k][steps] sin(base)
    double* u;           // [B][k][steps] maxA*w
    double* w2;          // the same for the NEXT evaluation, written by the Adam tail of this one (null: k_controls forms them); the engine swaps
    double* u2;          //   the pairs between iterations, so that u / w stay "the controls of the last evaluation" until the next one starts
    double* dLdu;        // [B][k][steps]
    double* grad;        // [B][k][steps] d reg_loss / d base
    cplx* inter;         // [B][steps+1][n][m]
    cplx* Xfinal;        // [B][n][n]
    cplx* ztau;          // [B][steps+1] per-time-step overlap (speed_up)
    double* Fpop;        // [B][steps+1][n_forb][m] forbidden levels (bare or dressed): a_f |phi|^4 / 2, the entry's share of the regulariser (k_dress_amplitudes -> k_loss)
    cplx* Fd;            // [B][steps+1][n_forb][m] dressed forbidden levels: 2 a_f |phi|^2 phi with phi = <dressed level f | Psi_tau[:, j]> (k_dress_amplitudes -> source_at); null otherwise
    cplx* zfin;          // [B]
    double* su_resid;    // [B] (steps+1 - value) of speed_up
    double* loss;        // [B]
    double* reg_state;   // [B]
    double* reg_loss;    // [B]
    double* g2;          // [B]
    double* uscale;      // [B]
    cplx* band_ph;       // [B][k][steps] scratch for the bandpass gradient
    cplx* band_tw;       // [steps] e^{-2 pi i r / steps} (bandpass regulariser)
    double* band_mag;    // [B][k][steps] cnt_f |F_f| of the pulse spectrum (k_band_spectrum -> finish)
    double* band_dR;     // [B][k][steps] d(sum cnt |F|)/d w (k_band_gradient -> finish)
};

// Adam loop parameters handed to the finishing kernel.
struct QocAdamDev {
    int mode;            // 0: evaluation only; 1: loop iteration (decide + update on device); 2: explicit step with lr[]
    double rate, decay, conv_target, min_grad;
    int max_iterations;
    const double* lr;    // mode 2
};

// Deterministic workgroup reduction (sum) of one double per thread; result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double* red /* >= QOC_BLOCK/64 + 1 doubles of LDS */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

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

// A/B switches of the kernel families ("experimental switches"): a value of the environment only counts when QOC_EXPERIMENTAL=1 stands beside it,
// and every switch is read inside qoc_create -- never on a launch path -- into a field of the engine, so that two engines of a process (or the ranks of a
// launch) can only differ when somebody asked for it twice.  Production runs never set QOC_EXPERIMENTAL.
static inline const char* qoc_exp_env(const char* name) {
    const char* on = getenv("QOC_EXPERIMENTAL");
    return (on && on[0] == '1') ? getenv(name) : nullptr;
}
static inline bool qoc_exp_is(const char* name, int value) { const char* e = qoc_exp_env(name); return e && atoi(e) == value; }

// Size of an engine's work-buffer arena: a multiple of 64 MB.  With the exact size the allocator recycled blocks freed by earlier
// engines of the process, and where such a block landed decided the speed (n = 128 x 4 after a dozen other engines: 21 ms per
// iteration instead of 8.3; three of three long sequences back at 8.3-8.6 ms with the rounded size).
static inline size_t qoc_arena_bytes(size_t total) { const size_t g = (size_t)1 << 26; return (total + g - 1) & ~(g - 1); }
