// qoc_engine.hip -- host side of libqoc_hip.so: the C ABI of include/qoc.h over the HIP kernels.
//
// One engine handle == one GRAPE problem (shared Hamiltonians, n_seeds independent control sets) resident in the
// HBM of one MI355X, one HIP stream.  One iteration is a fixed sequence of kernel launches on that stream; the
// stop rule, learning-rate schedule and Adam update run on the device, so the host never has to synchronise
// inside the optimisation loop (it only polls the done flags every `poll_every` iterations).
#include "../../include/qoc.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "qoc_common.h"
#include "qoc_kernels_finish.h"
#include "qoc_kernels_generic.h"
#include "qoc_kernels_mfma.h"
#include "qoc_kernels_st.h"
#include "qoc_kernels_gemm.h"
#include "qoc_gemm_ts.h"
#include "qoc_small.h"

#include "qoc_plan_limits.h"            // the measured numbers of AUTO's table (QOC_PLAN_*), shared with tests/test_auto_plan.py
// (QOC_PLAN_LAT_WORK = 4608 seeds x time slices: since the batch sweeps take their chunk boundaries and final_state from k_mfma_bnd_scan
// the batch kernels are ahead from 10 seeds of 500 slices on -- 0.356 ms at 12 seeds against 0.444; 8 seeds: 0.349 against 0.305;
// profiles/r03_latency_sweep.txt)
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(QOC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                              \
    } while (0)

struct qoc_engine {
    qoc_config cfg;
    QocDev d;
    int path;
    int chunks;
    hipStream_t stream;
    std::vector<void*> allocs;
    char* arena_cur = nullptr;      // bump allocator over the chunks in `allocs` (dev_alloc)
    size_t arena_left = 0;
    // generic path
    cplx* K = nullptr;          // [B][steps][n][n]
    cplx* expm_scratch = nullptr;
    int expm_grid = 0;
    cplx* seed_scratch = nullptr;
    double* fin_part = nullptr;      // [B][fin_S + 2][2] partial sums of the split tail (k_finish_split_a / _b: control sets of 4097 .. 8192 (k, t) elements)
    int fin_S = 0;
    // mfma path
    QocMfma mf;
    QocGemm gm;
    QocSmall sm;                // workgroup-resident path (csrc/qoc_small.h)
    bool evaluated = false;
    // QOC_DEBUG_SKIP (timing experiments only): 1 controls, 2 exponentials, 4 forward, 8 loss, 16 backward, 32 finish
    int skip_mask = 0;
    // u2 / w2 hold maxA sin(base) of the CURRENT variable (written by the Adam tail or by qoc_get_uks)
    bool controls_ready = false;
    bool final_stale = false, inter_stale = false;   // MFMA latency mode: Xfinal / uscale not yet formed for the last evaluation
    double* step_lr = nullptr;  // [B] per-seed learning rates of qoc_adam_step
    // profiling of the dominant kernel
    bool profiling = false;
    std::vector<hipEvent_t> ev;  // pairs
    size_t ev_used = 0;
    double prof_ms = 0.0;
    int64_t prof_launches = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
};

template <typename T>
static int dev_alloc(qoc_engine* e, T** p, size_t count) {
    // bump allocation out of 64 MB-granular chunks (qoc_arena_bytes: where many small recycled blocks land decides the speed)
    if (count == 0) count = 1;
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (bytes > e->arena_left) {
        void* q = nullptr;
        const size_t chunk = qoc_arena_bytes(bytes);
        hipError_t err = hipMalloc(&q, chunk);
        if (err != hipSuccess) return fail(QOC_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", chunk, hipGetErrorString(err));
        e->allocs.push_back(q);
        e->arena_cur = (char*)q;
        e->arena_left = chunk;
    }
    *p = (T*)e->arena_cur;
    e->arena_cur += bytes;
    e->arena_left -= bytes;
    return QOC_OK;
}

template <typename T>
static int dev_upload(qoc_engine* e, const T** p, const T* host, size_t count) {
    T* q = nullptr;
    int rc = dev_alloc(e, &q, count);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(q, host, count * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return QOC_OK;
}

#define TRY(expr)          \
    do {                   \
        int rc_ = (expr);  \
        if (rc_) return rc_; \
    } while (0)

static int prof_begin(qoc_engine* e) {
    if (!e->profiling) return QOC_OK;
    if (e->ev_used + 2 > e->ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t x;
            HIP_TRY(hipEventCreate(&x));
            e->ev.push_back(x);
        }
    }
    HIP_TRY(hipEventRecord(e->ev[e->ev_used], e->stream));
    return QOC_OK;
}
static int prof_end(qoc_engine* e) {
    if (!e->profiling) return QOC_OK;
    HIP_TRY(hipEventRecord(e->ev[e->ev_used + 1], e->stream));
    e->ev_used += 2;
    return QOC_OK;
}
static int prof_collect(qoc_engine* e) {
    if (e->ev_used == 0) return QOC_OK;
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < e->ev_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
        e->prof_ms += ms;
        e->prof_launches += 1;
    }
    e->ev_used = 0;
    return QOC_OK;
}

// ---- one evaluation (+ optional on-device stop rule / Adam), enqueued on the engine stream ------------------------
// band_tw[r] = e^{-2 pi i r / N}, r < N: the direct DFT of the bandpass regulariser evaluated sincos for every (frequency, slice) pair --
// 2 k N^2 of them per seed and iteration, 1.7 ms for one C2 trajectory; the phase index f t mod N advances by additions instead
__global__ void __launch_bounds__(256) k_band_twiddles(cplx* tw, int N) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < N) { double c, s; unit_phase(r, 1, N, &c, &s); tw[r] = cmake(c, s); }
}

// Bandpass regulariser (regularization_functions.py:47-67) by direct DFT, outside the one-workgroup-per-seed finish kernel (where the
// 2 k N^2 terms of a seed took 0.4 ms of one C2 trajectory even with the phase table):
// k_band_spectrum: a wave per (seed, control, frequency): F_f = sum_t w_t e^{-2 pi i f t/N}; band_mag = cnt_f |F_f|, band_ph = cnt_f
// conj(F_f)/|F_f|
//   k_band_gradient: a thread per (seed, control, slice): band_dR = sum_f Re(band_ph_f e^{-2 pi i f t/N})
// cnt_f = how often the reference's two slices (f < lo; hi <= f < N/2) contain f.
__global__ void __launch_bounds__(256) k_band_spectrum(QocDev d) {
    const int steps = d.steps, lane = threadIdx.x & 63, half = steps / 2;
    const int lo = min(max(d.band_lo, 0), steps), hi = min(max(d.band_hi, 0), steps);
    const size_t total = (size_t)d.B * d.k * steps;
    for (size_t o = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); o < total; o += (size_t)gridDim.x * 4) {
        const int f = (int)(o % steps);
        const size_t bk = o / steps;
        if (d.skip_done && d.done[bk / d.k]) continue;
        const int cnt = (f < lo ? 1 : 0) + ((f >= hi && f < half) ? 1 : 0);
        cplx p = cmake(0.0, 0.0);
        double mg = 0.0;
        if (cnt > 0) {                                                       // (uniform over the wave)
            const double* wk = d.w + bk * steps;
            int r = (int)(((long long)f * lane) % steps);
            const int dr = (int)(((long long)f * 64) % steps);
            double fr = 0.0, fi = 0.0;
            for (int t = lane; t < steps; t += 64) {
                const cplx e = d.band_tw[r];
                fr = fma(wk[t], e.x, fr);
                fi = fma(wk[t], e.y, fi);
                r += dr; if (r >= steps) r -= steps;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { fr += __shfl_xor(fr, off, 64); fi += __shfl_xor(fi, off, 64); }
            const double mag = sqrt(fr * fr + fi * fi);
            mg = (double)cnt * mag;
            if (mag > 0.0) p = cmake((double)cnt * fr / mag, -(double)cnt * fi / mag);
        }
        if (lane == 0) { d.band_ph[o] = p; d.band_mag[o] = mg; }
    }
}
__global__ void __launch_bounds__(256) k_band_gradient(QocDev d) {
    const int steps = d.steps, half = steps / 2;
    const int lo = min(max(d.band_lo, 0), steps);
    const int fend = min(max(half, lo), steps);
    const size_t total = (size_t)d.B * d.k * steps;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(o % steps);
        const size_t bk = o / steps;
        if (d.skip_done && d.done[bk / d.k]) continue;
        const cplx* pk = d.band_ph + bk * steps;
        double acc = 0.0;
        int r = 0;                                                           // f t mod steps, advanced by t per frequency
        for (int f0 = 0; f0 < fend; f0 += 8) {
            cplx qv[8], e[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                qv[q] = pk[min(f0 + q, fend - 1)];
                e[q] = d.band_tw[r];
                r += t; if (r >= steps) r -= steps;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                // Re(ph_f e^{-2 pi i f t/N}); ph = 0 outside the counted bins
                if (f0 + q < fend) acc += qv[q].x * e[q].x - qv[q].y * e[q].y;
        }
        d.band_dR[o] = acc;
    }
}

// k_loss, preceded by what it needs per time point: the dressed-basis amplitudes of the forbidden levels, the overlaps of speed_up
static inline void launch_loss(const QocDev& d, hipStream_t s) {
    if (d.n_forb > 0) {
        const size_t total = (size_t)d.B * (d.steps + 1) * d.n_forb * d.m;
        size_t g = (total + 255) / 256;
        hipLaunchKernelGGL(k_dress_amplitudes, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, d);
    }
    if (d.has_speed) {
        const size_t g = ((size_t)d.B * (d.steps + 1) + 3) / 4;
        hipLaunchKernelGGL(k_time_overlaps, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, s, d);
    }
    hipLaunchKernelGGL(k_loss, dim3(d.B), dim3(QOC_BLOCK), 0, s, d);
}

// workgroup-resident path: ONE launch runs `iters` loop iterations (or one evaluation / explicit step)
static int enqueue_small(qoc_engine* e, const QocAdamDev& ap, int iters) {
    QocDev d = e->d;
    d.skip_done = ap.mode == 1 ? 1 : 0;
    std::string msg;
    TRY(prof_begin(e));
    const int rc = qoc_small_launch(e->sm, d, ap, iters, e->stream, msg);
    if (rc) return fail(rc == -1 ? QOC_ERR_INVALID : QOC_ERR_HIP, "workgroup-resident iteration: %s", msg.c_str());
    TRY(prof_end(e));
    HIP_TRY(hipGetLastError());
    e->evaluated = true;
    e->controls_ready = false;                 // the kernel forms its own controls from the variable; d.u / d.w hold those of the last evaluation
    e->final_stale = true;                     // final_state, unitary_scale and inter_vecs are formed on read-back (refresh_small)
    e->inter_stale = true;
    return QOC_OK;
}

static int enqueue_iteration(qoc_engine* e, const QocAdamDev& ap) {
    if (e->path == QOC_PATH_SMALL) return enqueue_small(e, ap, 1);
    // the slice kernel of the n <= 32 latency mode forms its own controls; everybody else reads u / w: from k_controls, or -- when the Adam
    // tail of the previous iteration (or qoc_get_uks) has left them in u2 / w2 -- by swapping the two pairs (one launch less per iteration)
    const bool own_controls = e->path == QOC_PATH_MFMA && e->mf.latency && e->mf.NT == 2;
    const bool swap_in = e->controls_ready && !own_controls && !(e->skip_mask & 1);
    if (swap_in) { std::swap(e->d.u, e->d.u2); std::swap(e->d.w, e->d.w2); }
    e->controls_ready = false;
    QocDev d = e->d;
    if (own_controls) { d.u2 = nullptr; d.w2 = nullptr; }
    d.skip_done = ap.mode == 1 ? 1 : 0;      // qoc_eval / explicit steps always evaluate every seed
    d.uscale_in_loss = (e->path == QOC_PATH_MFMA && !e->mf.latency && !e->mf.updown) ? 1 : 0;
    const int total = d.B * d.k * d.steps;
    int cgrid = (total + QOC_BLOCK - 1) / QOC_BLOCK;
    if (cgrid > 2048) cgrid = 2048;
    const int skip = e->skip_mask;
    const bool plain = !(d.has_amp || d.has_env || d.has_dwdt || d.has_d2wdt2 || d.has_band);
    // latency mode: the tail of the iteration runs in the last workgroup of the gradient kernel
    // (with the local pulse regularisers too -- amplitude, envelope, dwdt, d2wdt2; the bandpass DFT keeps its own launch)
    const bool fused_tail = e->path == QOC_PATH_MFMA && e->mf.latency && (!e->mf.lat_sources || e->mf.lat_src_fast) && !d.has_band
        && !(skip & (16 | 32));
    // (latency mode of the MFMA path: the slice kernel of the exponentials forms its own controls)
    if (!(skip & 1) && !own_controls && !swap_in) hipLaunchKernelGGL(k_controls, dim3(cgrid), dim3(QOC_BLOCK), 0, e->stream, d);
    if (e->path == QOC_PATH_MFMA) {
        TRY(prof_begin(e));
        if (!(skip & 2)) qoc_mfma_launch_expm(e->mf, d, e->stream);
        TRY(prof_end(e));
        if (!(skip & 4)) qoc_mfma_launch_forward(e->mf, d, e->stream);
        if (skip & 64) qoc_mfma_launch_forward(e->mf, d, e->stream);        // debug: the same launch again (cold-start vs steady cost)
        if (skip & 128) qoc_mfma_launch_backward(e->mf, d, e->stream);
        // latency mode / k_mfma_downup: inside the backward kernel
        if (!(skip & 8) && !e->mf.updown && (!e->mf.latency || (e->mf.lat_sources && !e->mf.lat_src_fast))) launch_loss(d, e->stream);
        if (!(skip & 16)) {
            if (fused_tail) qoc_mfma_latency_gradient(e->mf, d, &ap, e->stream);
            else qoc_mfma_launch_backward(e->mf, d, e->stream);
        }
    } else if (e->path == QOC_PATH_GEMM) {
        if (e->gm.ts_G > 0) {                                           // one trajectory sharded along the time axis (qoc_gemm_ts.h)
            const int rc = qoc_gemm_ts_evaluate(e->gm, d, e->stream, [&]() { launch_loss(d, e->stream); }, [&]() { return prof_begin(e); },
                [&]() { return prof_end(e); });
            if (rc == 1) return fail(QOC_ERR_HIP, "time-sharded iteration: clearing the gradient array failed");
            if (rc) return rc;                                          // (the message is the collective's / the profiler's)
        } else {
        // the hipEvent bracket of qoc_profile_read: the exponentials -- or, on the direct state-transfer route (no exponentials: the
        // assembly of the generators is all qoc_gemm_expm does there), the backward half of the iteration, which the backward Taylor chain
        // dominates
        const bool bracket_bwd = e->gm.direct;
        if (!bracket_bwd) TRY(prof_begin(e));
        qoc_gemm_expm(e->gm, d, e->stream);
        if (!bracket_bwd) TRY(prof_end(e));
        qoc_gemm_forward(e->gm, d, e->stream);
        launch_loss(d, e->stream);
        if (bracket_bwd) TRY(prof_begin(e));
        qoc_gemm_backward(e->gm, d, e->stream);
        if (bracket_bwd) TRY(prof_end(e));
        }
    } else if (!d.state_transfer) {
        TRY(prof_begin(e));
        hipLaunchKernelGGL(k_expm_generic, dim3(e->expm_grid), dim3(QOC_BLOCK), 0, e->stream, d, e->K, e->expm_scratch);
        TRY(prof_end(e));
        hipLaunchKernelGGL(k_fwd_generic, dim3(d.B), dim3(QOC_BLOCK), 0, e->stream, d, e->K, e->seed_scratch);
        launch_loss(d, e->stream);
        hipLaunchKernelGGL(k_bwd_generic, dim3(d.B), dim3(QOC_BLOCK), 0, e->stream, d, e->K, e->seed_scratch);
    } else if (e->path == QOC_PATH_ST_FUSED) {
        TRY(prof_begin(e));
        st_fused_launch(d, e->stream, true);
        TRY(prof_end(e));
        launch_loss(d, e->stream);
        st_fused_launch(d, e->stream, false);
    } else {
        TRY(prof_begin(e));
        hipLaunchKernelGGL(k_st_fwd_generic, dim3(d.B), dim3(QOC_BLOCK), 0, e->stream, d, e->seed_scratch);
        TRY(prof_end(e));
        launch_loss(d, e->stream);
        hipLaunchKernelGGL(k_st_bwd_generic, dim3(d.B), dim3(QOC_BLOCK), 0, e->stream, d, e->seed_scratch);
    }
    if (!(skip & 32) && !fused_tail) {
        const dim3 fb(d.k * d.steps >= 2048 ? 1024 : QOC_BLOCK);
        if (d.has_band) {
            const size_t items = (size_t)d.B * d.k * d.steps, g1 = (items + 3) / 4, g2 = (items + 255) / 256;
            hipLaunchKernelGGL(k_band_spectrum, dim3((unsigned)(g1 > 8192 ? 8192 : g1)), dim3(256), 0, e->stream, d);
            hipLaunchKernelGGL(k_band_gradient, dim3((unsigned)(g2 > 8192 ? 8192 : g2)), dim3(256), 0, e->stream, d);
        }
        // seeds of 4097 .. 8192 (k, t) elements (C3: 6 x 1000) keep their Adam slots in registers too: eight elements per thread
        // ... or, since round 6, spread over fin_S workgroups in two launches: one control set of 6000 elements is bound by the fp64 sin / cos / sqrt / divide of the
        // ONE compute unit k_finish_t runs it on (C3, one trajectory: 32.5 us; profiles/r06_kernel_stats_c3_single_trajectory.txt)
        const bool wide = d.k * d.steps > 4 * 1024 && d.k * d.steps <= 8 * 1024;
        const bool split = d.k * d.steps > 4 * 1024 && e->fin_part;
        if (split) {
            const dim3 sg((unsigned)e->fin_S, (unsigned)d.B);
            // (GEMM path, persistent chains: the wide gradient product left per-tile partial dots -- qoc_gemm_backward skipped its reduce launch, part A sums them)
            QocGradPartial gp{nullptr, 0, 0, 0};
            if (e->path == QOC_PATH_GEMM && e->gm.reduce_in_tail) gp = QocGradPartial{e->gm.partial, e->gm.N / 32, e->gm.ldW, e->gm.MV};
            if (plain) {
                hipLaunchKernelGGL(k_finish_split_a<true>, sg, dim3(256), 0, e->stream, d, ap, e->fin_part, gp);
                hipLaunchKernelGGL(k_finish_split_b<true>, sg, dim3(256), 0, e->stream, d, ap, e->fin_part);
            } else {
                hipLaunchKernelGGL(k_finish_split_a<false>, sg, dim3(256), 0, e->stream, d, ap, e->fin_part, gp);
                hipLaunchKernelGGL(k_finish_split_b<false>, sg, dim3(256), 0, e->stream, d, ap, e->fin_part);
            }
        }
        else if (plain && wide) hipLaunchKernelGGL((k_finish_t<true, 8>), dim3(d.B), fb, 0, e->stream, d, ap);
        else if (plain) hipLaunchKernelGGL(k_finish_t<true>, dim3(d.B), fb, 0, e->stream, d, ap);
        else if (wide) hipLaunchKernelGGL((k_finish_t<false, 8>), dim3(d.B), fb, 0, e->stream, d, ap);
        else hipLaunchKernelGGL(k_finish_t<false>, dim3(d.B), fb, 0, e->stream, d, ap);
    }
    HIP_TRY(hipGetLastError());
    e->evaluated = true;
    // the Adam tail ran: u2 / w2 belong to the moved variable
    e->controls_ready = ap.mode != 0 && !own_controls && !(skip & (16 | 32));
    // final_state / unitary_scale are formed when read back
    e->final_stale = (e->path == QOC_PATH_MFMA && (e->mf.latency || e->mf.updown)) ||
                     (e->path == QOC_PATH_GEMM && e->gm.ts_G <= 0 && qoc_gemm_lazy_final(e->gm, e->d));
    // inter_vecs too, unless the batch kernels' source recursion needed them anyway
    e->inter_stale = (e->path == QOC_PATH_MFMA && e->final_stale && (!e->mf.lat_sources || e->mf.lat_src_fast))
                     || (e->path == QOC_PATH_MFMA && e->mf.updown);                  // (k_mfma_downup stores no Psi_t either)
    return QOC_OK;
}

// latency mode of the MFMA path: final_state and unitary_scale of the last evaluation, formed on demand
// workgroup-resident path: inter_vecs, final_state and unitary_scale of the last evaluation, re-formed from its controls (d.u) by the any-size kernels
static int refresh_small(qoc_engine* e, bool need_inter) {
    if (!e->final_stale && !(need_inter && e->inter_stale)) return QOC_OK;
    // unitary mode: the launch itself leaves final_state and unitary_scale of its last evaluation (from the root of its product tree), unless a deferred stop rule
    // undid that evaluation's successor; inter_vecs are always re-formed here
    if (!need_inter && !e->d.state_transfer && qoc_small_final_valid(e->sm, e->stream)) { e->final_stale = false; return QOC_OK; }
    QocDev d = e->d;
    d.skip_done = 0;
    if (!d.state_transfer) {
        hipLaunchKernelGGL(k_expm_generic, dim3(e->expm_grid), dim3(QOC_BLOCK), 0, e->stream, d, e->K, e->expm_scratch);
        hipLaunchKernelGGL(k_fwd_generic, dim3(d.B), dim3(QOC_BLOCK), 0, e->stream, d, e->K, e->seed_scratch);
    } else {
        hipLaunchKernelGGL(k_st_fwd_generic, dim3(d.B), dim3(QOC_BLOCK), 0, e->stream, d, e->seed_scratch);
        qoc_mfma_uscale_state_transfer(d, e->stream);
    }
    HIP_TRY(hipGetLastError());
    e->final_stale = false;
    e->inter_stale = false;
    return QOC_OK;
}

// several workgroups per control set: a spin on another workgroup's flag that gave up (it never should) left garbage behind -- say so
static int small_check(qoc_engine* e) {
    if (e->path != QOC_PATH_SMALL || e->sm.G <= 1) return QOC_OK;
    unsigned err = 0;
    HIP_TRY(hipMemcpy(&err, e->sm.sd.err, sizeof err, hipMemcpyDeviceToHost));
    if (err) return fail(QOC_ERR_HIP, "workgroup-resident iteration: an exchange between the %d workgroups of a control set timed out (are all %d workgroups "
        "resident?)", e->sm.G, e->sm.G * e->d.B);
    return QOC_OK;
}

static int refresh_final(qoc_engine* e) {
    if (e->path == QOC_PATH_SMALL) return refresh_small(e, false);
    if (!e->final_stale) return QOC_OK;
    // the boundary chain once more, with X beside the vectors
    if (e->path == QOC_PATH_GEMM) qoc_gemm_forward(e->gm, e->d, e->stream, true);
    else if (e->d.state_transfer) {                                                    // no final_state; unitary_scale from Psi_N
        if (e->inter_stale) { qoc_mfma_unpack_inter(e->mf, e->d, e->stream); e->inter_stale = false; }
        qoc_mfma_uscale_state_transfer(e->d, e->stream);
    }
    else if (e->mf.latency) qoc_mfma_final_state(e->mf, e->d, e->stream);
    else qoc_mfma_final_state_batch(e->mf, e->d, e->stream);
    HIP_TRY(hipGetLastError());
    e->final_stale = false;
    return QOC_OK;
}

static QocAdamDev loop_params(const qoc_adam_params* p) {
    QocAdamDev ap;
    ap.mode = 1;
    ap.rate = p->rate;
    ap.decay = p->learning_rate_decay;
    ap.conv_target = p->conv_target;
    ap.min_grad = p->min_grad;
    ap.max_iterations = p->max_iterations;
    ap.lr = nullptr;
    return ap;
}

extern "C" {

const char* qoc_last_error(void) { return g_err.c_str(); }
const char* qoc_version(void) { return "qoc-hip 0.1 (gfx950)"; }

int qoc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int qoc_device_info(int32_t device, char* name, int32_t name_len, int32_t* compute_units, int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return QOC_OK;
}

int qoc_device_peer_access(int32_t device, int32_t peer, int32_t* can_access) {
    if (!can_access) return fail(QOC_ERR_INVALID, "qoc_device_peer_access: null argument");
    int can = 0;
    if (device == peer) can = 1;
    else HIP_TRY(hipDeviceCanAccessPeer(&can, device, peer));
    *can_access = can;
    return QOC_OK;
}

int qoc_create(const qoc_config* cfg, const double* Hs, const double* U0, const double* V, const double* W,
               const double* maxA, const double* one_minus_gauss, const int32_t* forbidden_states,
               const double* forbidden_coeffs, const double* Vs, qoc_handle* out) {
    if (!cfg || !Hs || !V || !W || !maxA || !out) return fail(QOC_ERR_INVALID, "qoc_create: null argument");
    if (cfg->plan_seeds < 0) return fail(QOC_ERR_INVALID,
        "qoc_create: plan_seeds = %d (0 = plan for n_seeds, > 0 = the batch AUTO plans for)", cfg->plan_seeds);
    if (cfg->n < 1 || cfg->k < 1 || cfg->steps < 1 || cfg->m < 1 || cfg->n_seeds < 1)
        return fail(QOC_ERR_INVALID, "qoc_create: n, k, steps, m, n_seeds must be >= 1");
    if (cfg->taylor_terms < 1 || cfg->scaling < 0 || cfg->scaling > 30)
        return fail(QOC_ERR_INVALID, "qoc_create: bad taylor_terms/scaling (%d, %d)", cfg->taylor_terms, cfg->scaling);
    if (!cfg->state_transfer && !U0) return fail(QOC_ERR_INVALID, "qoc_create: U0 required in unitary mode");
    if (cfg->n_forbidden < 0) return fail(QOC_ERR_INVALID, "qoc_create: n_forbidden must be >= 0");
    if (cfg->n_forbidden > 0 && (!forbidden_states || !forbidden_coeffs))
        return fail(QOC_ERR_INVALID, "qoc_create: forbidden lists missing");
    if (cfg->forbid_dressed && cfg->n_forbidden > 0 && !Vs)
        return fail(QOC_ERR_INVALID, "qoc_create: forbid_dressed needs Vs");
    if (cfg->has_envelope && !one_minus_gauss) return fail(QOC_ERR_INVALID, "qoc_create: envelope constant missing");
    if (cfg->has_d2wdt2 && !cfg->has_dwdt)
        return fail(QOC_ERR_INVALID, "qoc_create: d2wdt2 needs dwdt (reference: NameError new_weights)");
    for (int f = 0; f < cfg->n_forbidden; ++f)
        if (forbidden_states[f] < 0 || forbidden_states[f] >= cfg->n)
            return fail(QOC_ERR_INVALID, "qoc_create: forbidden state %d out of range", forbidden_states[f]);
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0)
        return fail(QOC_ERR_HIP, "qoc_create: no HIP device visible (%s) -- this engine has no CPU fallback",
                    de == hipSuccess ? "count = 0" : hipGetErrorString(de));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(QOC_ERR_INVALID, "qoc_create: device %d of %d", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    qoc_engine* e = new qoc_engine();
    e->cfg = *cfg;
    QocDev& d = e->d;
    memset(&d, 0, sizeof d);
    const int n = cfg->n, k = cfg->k, steps = cfg->steps, m = cfg->m, B = cfg->n_seeds;
    d.n = n; d.k = k; d.steps = steps; d.m = m; d.T = cfg->taylor_terms; d.s = cfg->state_transfer ? 0 : cfg->scaling;
    d.B = B; d.state_transfer = cfg->state_transfer; d.dt = cfg->dt;
    d.Bplan = cfg->plan_seeds > 0 ? cfg->plan_seeds : B;
    const double inv_steps = 1.0 / (double)steps;
    d.has_amp = cfg->has_amplitude; d.a_amp = cfg->c_amplitude * inv_steps;
    d.has_env = cfg->has_envelope; d.a_env = cfg->c_envelope * inv_steps;
    d.has_dwdt = cfg->has_dwdt; d.a_dwdt = cfg->c_dwdt * inv_steps;
    d.has_d2wdt2 = cfg->has_d2wdt2; d.a_d2wdt2 = cfg->c_d2wdt2 * inv_steps;
    d.has_speed = cfg->has_speed_up; d.a_speed = cfg->c_speed_up * inv_steps;
    d.has_band = cfg->has_bandpass; d.a_band = cfg->c_bandpass * inv_steps;
    d.band_lo = cfg->band_lo; d.band_hi = cfg->band_hi;
    d.n_forb = cfg->n_forbidden; d.forbid_dressed = cfg->forbid_dressed && cfg->n_forbidden > 0;
    int rc = QOC_OK;
    auto bail = [&](int code) { qoc_destroy(e); return code; };
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(QOC_ERR_HIP, "hipStreamCreate failed"));
    if (hipEventCreate(&e->t0) != hipSuccess || hipEventCreate(&e->t1) != hipSuccess) return bail(fail(QOC_ERR_HIP,
        "hipEventCreate failed"));

    const size_t nn = (size_t)n * n, nm = (size_t)n * m, ks = (size_t)k * steps;
    // U0*V on the host (tiny): start vector of the thin forward recursion
    std::vector<double> psi0(2 * nm);
    for (int a = 0; a < n; ++a)
        for (int j = 0; j < m; ++j) {
            double re = 0, im = 0;
            if (cfg->state_transfer) {
                re = V[2 * (a * m + j)]; im = V[2 * (a * m + j) + 1];
            } else {
                for (int c = 0; c < n; ++c) {
                    const double ur = U0[2 * (a * n + c)], ui = U0[2 * (a * n + c) + 1];
                    const double vr = V[2 * (c * m + j)], vi = V[2 * (c * m + j) + 1];
                    re += ur * vr - ui * vi;
                    im += ur * vi + ui * vr;
                }
            }
            psi0[2 * (a * m + j)] = re; psi0[2 * (a * m + j) + 1] = im;
        }
    std::vector<double> ident;
    if (!U0) {
        ident.assign(2 * nn, 0.0);
        for (int a = 0; a < n; ++a) ident[2 * (a * n + a)] = 1.0;
        U0 = ident.data();
    }
    if ((rc = dev_upload(e, &d.Hs, (const cplx*)Hs, (size_t)(k + 1) * nn))) return bail(rc);
    if ((rc = dev_upload(e, &d.U0, (const cplx*)U0, nn))) return bail(rc);
    if ((rc = dev_upload(e, &d.V, (const cplx*)V, nm))) return bail(rc);
    if ((rc = dev_upload(e, &d.W, (const cplx*)W, nm))) return bail(rc);
    if ((rc = dev_upload(e, &d.Psi0, (const cplx*)psi0.data(), nm))) return bail(rc);
    if (d.forbid_dressed && (rc = dev_upload(e, &d.Vs, (const cplx*)Vs, nn))) return bail(rc);
    if ((rc = dev_upload(e, &d.maxA, maxA, (size_t)k))) return bail(rc);
    if (one_minus_gauss && (rc = dev_upload(e, &d.omg, one_minus_gauss, ks))) return bail(rc);
    if (cfg->n_forbidden > 0) {
        std::vector<double> fa(cfg->n_forbidden);
        for (int f = 0; f < cfg->n_forbidden; ++f) fa[f] = forbidden_coeffs[f] * inv_steps;
        if ((rc = dev_upload(e, &d.forb_state, (const int*)forbidden_states, (size_t)cfg->n_forbidden))) return bail(rc);
        if ((rc = dev_upload(e, &d.forb_a, (const double*)fa.data(), (size_t)cfg->n_forbidden))) return bail(rc);
    }

#define ALLOC(ptr, count) if ((rc = dev_alloc(e, &(ptr), (count)))) return bail(rc)
    ALLOC(d.base, B * ks); ALLOC(d.adam_m, B * ks); ALLOC(d.adam_v, B * ks);
    ALLOC(d.adam_t, (size_t)B); ALLOC(d.iters, (size_t)B); ALLOC(d.done, (size_t)B);
    ALLOC(d.w, B * ks); ALLOC(d.u, B * ks); ALLOC(d.w2, B * ks); ALLOC(d.u2, B * ks); ALLOC(d.dLdu, B * ks); ALLOC(d.grad, B * ks);
    if (ks > 4 * 1024 && !qoc_exp_is("QOC_FINISH_SPLIT", 0)) {      // (the tail of such control sets runs over fin_S workgroups each; the switch: A/B runs)
        e->fin_S = (int)((ks + 255) / 256);
        if (e->fin_S > 64) e->fin_S = 64;                          // (longer pulses: several elements per thread)
        ALLOC(e->fin_part, (size_t)B * (e->fin_S + 2) * 2);
    }
    ALLOC(d.inter, (size_t)B * (steps + 1) * nm);
    ALLOC(d.Xfinal, (size_t)B * nn);
    ALLOC(d.ztau, (size_t)B * (steps + 1));
    if (d.n_forb > 0) ALLOC(d.Fpop, (size_t)B * (steps + 1) * d.n_forb * m);
    if (d.forbid_dressed && d.n_forb > 0) ALLOC(d.Fd, (size_t)B * (steps + 1) * d.n_forb * m);
    ALLOC(d.zfin, (size_t)B); ALLOC(d.su_resid, (size_t)B);
    ALLOC(d.loss, (size_t)B); ALLOC(d.reg_state, (size_t)B); ALLOC(d.reg_loss, (size_t)B);
    ALLOC(d.g2, (size_t)B); ALLOC(d.uscale, (size_t)B);
    if (d.has_band) { ALLOC(d.band_ph, B * ks); ALLOC(d.band_tw, (size_t)steps); ALLOC(d.band_mag, B * ks); ALLOC(d.band_dR, B * ks); }
    ALLOC(e->step_lr, (size_t)B);
    if (hipMemset(d.base, 0, B * ks * sizeof(double)) != hipSuccess ||
        hipMemset(d.adam_m, 0, B * ks * sizeof(double)) != hipSuccess ||
        hipMemset(d.adam_v, 0, B * ks * sizeof(double)) != hipSuccess ||
        hipMemset(d.adam_t, 0, B * sizeof(int)) != hipSuccess ||
        hipMemset(d.iters, 0, B * sizeof(int)) != hipSuccess ||
        hipMemset(d.done, 0, B * sizeof(int)) != hipSuccess ||
        hipMemset(d.su_resid, 0, B * sizeof(double)) != hipSuccess ||
        hipMemset(d.uscale, 0, B * sizeof(double)) != hipSuccess ||
        hipMemset(d.Xfinal, 0, (size_t)B * nn * sizeof(cplx)) != hipSuccess)
        return bail(fail(QOC_ERR_HIP, "qoc_create: clearing the state buffers failed"));
    if (d.has_band) {
        hipLaunchKernelGGL(k_band_twiddles, dim3((steps + 255) / 256), dim3(256), 0, 0, d.band_tw, steps);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(0) != hipSuccess)
            return bail(fail(QOC_ERR_HIP, "qoc_create: the phase table of the bandpass regulariser could not be formed"));
    }

    // ---- path selection -----------------------------------------------------------------------------------------
    int path = cfg->path;
    const bool antiherm = cfg->state_transfer ? qoc_all_antihermitian((const cplx*)Hs, n, k + 1) : true;
    // (state transfer: the propagator's adjoint is the reference's gradient only then)
    const bool mfma_ok = qoc_mfma_supported(d) && antiherm;
    const bool st_ok = st_fused_supported(d);
    const bool gemm_ok = qoc_gemm_supported(d, antiherm);
    // Measured with tools/path_sweep.py (profiles/r01_path_sweep.txt):
    //  * unitary, n <= 32: the register-resident MFMA chain kernels win on throughput (1.8 vs 2.3 ms per iteration of 64
    //    C2 seeds), the GEMM path (fused LDS-resident exponential + product tree + persistent thin chains) on latency
    //    (0.19 vs 0.47 ms for one C2 trajectory; level at 12 seeds, 0.56 vs 0.49 ms at 16 since the batch kernels split the pulse into
    //    up to 64 chunks: profiles/r02_latency_sweep.txt); 32 < n <= 48 (NT = 3: exponentials by
    //    three waves per item on v_mfma_f64_4x4x4, costate sweep + slice-parallel gradient kernel): the MFMA path wins from 8 seeds
    //    on (1.11 vs 1.50 ms at 8, 1.70 vs 2.72 at 16, 5.27 vs 9.88 ms at 64 seeds of n = 48; the GEMM path pads to N = 64), ties at
    //    4 (1.00 vs 0.95) and loses below (0.96 vs 0.62 ms at 2); 48 < n <= 64 (NT = 4; tools/n64_batch_sweep.py): with k <= 4 controls
    //    the MFMA path is ahead from 32 seeds on (4.96 vs 5.14 ms at 32, 9.16 vs 10.05 at 64, 17.4 vs 19.8 at 128 seeds of n = 64 x 500
    //    slices, since the row-tile gradient kernel); with more controls the GEMM path up to 64 seeds, the MFMA path beyond (round 3, after
    // k_mfma_expm_rows lost its scratch: k = 6 x 200 slices 4.33 vs 4.35 ms at 64 seeds, 7.86 vs 8.47 at 128; k = 8 x 1000 slices 20.05 vs
    //    20.54 at 64, 38.7 vs 40.7 at 128; at 32 seeds the GEMM path: 2.31 vs 2.59, 10.4 vs 10.7).
    // every batch-size-dependent choice below is taken for Bp = qoc_config.plan_seeds (else the local batch): a shard of a restart
    // batch then runs the same path, kernels and chunking as the whole batch would
    const bool direct_ok = qoc_gemm_direct_supported(d);
    if (cfg->state_transfer && cfg->path == QOC_PATH_GEMM && cfg->chunks > 1 && !antiherm)
        return bail(fail(QOC_ERR_INVALID,
            "qoc_create: the propagator route of the GEMM path (chunks > 1) needs exactly anti-Hermitian generators"));
    // a handful of control sets of an n <= 32 unitary problem (the reference's own use is ONE per Grape() call): the latency mode of
    // the MFMA path.  It spends a workgroup per time slice, so what decides is seeds x slices
    // (profiles/r02_latency_sweep.txt, r02_small_n_sweep.txt): C2 (500 slices) 0.083 ms against 0.189 (GEMM route) and 0.56 (batch
    // kernels) for one seed, still ahead at 12 seeds, level at 16; n <= 16 is padded to 32 and competes with the cheap NT = 1 batch
    // kernels: ahead up to 4 seeds (n = 16 x 500 slices: 0.081 against 0.203 ms for one seed, 0.163 against 0.213 for four)
    // (round 4, with k_mfma_expm_slice2 on the active strips: ahead up to 6 control sets with or without a state regulariser -- n = 16 x
    // 500 slices x 6: 0.180 against 0.213 ms, with a forbidden level 0.250 against 0.280; x 8: 0.229 / 0.213 and 0.308 / 0.280; n = 9 x 300
    // x 6: 0.106 / 0.140; profiles/r04_small_n_latency.txt) 32 < n <= 48 (NT = 3 kernels: k_mfma_expm_rows per slice, the same chains,
    // sweeps and gradient): one trajectory of n = 48 x 500 slices 0.165 ms against 0.454 (GEMM route) and 0.84 (batch kernels); ahead up to
    // 8 seeds (profiles/r02_mid_n_sweep.txt). With a state regulariser (forbidden levels, speed_up) the backward half is the affine
    // recursion of the batch kernels on the latency mode's chunks, with two-level boundaries (QocMfma::lat_sources): one C2 trajectory with
    // dwdt + forbidden levels 0.189 ms against 0.290 (GEMM route) and 0.72 (batch kernels); ahead up to ~4096 seed-slices
    // (tools/c2_forbidden_single.py).
    const bool lat_src = d.n_forb > 0 || d.has_speed;
    // n > 32 with ONE state vector: the direct route runs k_gemm_taylor_chain_dpp (round 4: 0.31 against 0.46 us per dependent mat-vec) and
    // wins earlier -- C3 (n = 64, k = 6, 1000 slices), propagator / direct route in ms: x 24 6.03 / 6.20, x 32 8.00 / 6.34; without
    // forbidden levels (both chains side by side) x 12 2.81 / 3.12, x 16 3.74 / 3.18 (profiles/r04_c3_route_sweep.txt); n = 40, 48 with k =
    // 4 x 500 slices, MFMA batch kernels / direct: x 24 1.48 / 1.64, x 32 1.84 / 1.70; with forbidden levels x 32 2.14 / 3.20, x 48 3.89 /
    // 3.32 (profiles/r04_st_direct_sweep.txt)
    const bool dpp_shape = n > 32 && m == 1;
    // (three-multiplication form of the DPP chain, profiles/r04_c3_route_sweep_gauss.txt: with forbidden levels x 20 5.05 / 5.43, x 22 5.56
    // / 5.45, x 24 6.02 / 5.49; without x 11 2.63 / 2.77, x 12 2.84 / 2.79, x 13 3.11 / 2.81 -- the limits moved from 28 / 14 to 22 / 12)
    const int ST_DIRECT_FROM = n <= 32 ? QOC_PLAN_ST_DIRECT_N32 : (dpp_shape ? (lat_src ? QOC_PLAN_ST_DIRECT_DPP_SRC
        : QOC_PLAN_ST_DIRECT_DPP) : QOC_PLAN_ST_DIRECT_N64);
    struct AutoPlan { int path; bool latency; bool gemm_direct; };
    // the batch-size-dependent part of AUTO as a function of the batch it plans for (tests/test_auto_plan.py restates this table row by
    // row) State transfer on the MFMA path (round 4, tools/st_path_sweep.py -> profiles/r04_state_transfer_paths.txt; m = 1, T = 10, 500
    // slices, ms per iteration, GEMM path / MFMA batch kernels / latency mode): n = 32 x 1: 0.125 / 0.300 / 0.073, x 4: 0.159 / 0.304 /
    // 0.143, x 16: 0.388 / 0.324, x 64: 1.30 / 0.99, x 256: 2.00 (direct Taylor chains) / 3.78; n = 16 x 1: 0.126 / 0.192 / 0.063, x 8:
    // 0.219 / 0.209 / 0.203, x 64: 1.28 / 0.34, x 256: 1.94 / 1.16; n = 48 x 1: 0.245 / 0.72 / 0.120, x 8: 1.02 / 0.77 / 0.58, x 16: 1.87 /
    // 1.11, x 64: 2.60 / 3.31 (with forbidden levels 4.85 / 4.00); n = 64 (C3: k = 6, 1000 slices) x 1: 0.417 / 3.0 / 0.449, x 64: 9.73 /
    // 18.1 -- so: the unitary table for n <= 32 and for 32 < n <= 48 with k <= 4 (NT = 3), the latency mode of n <= 16 up to 8 control sets
    // and from 25 levels on up to 4, the GEMM route for up to 8 control sets from 25 levels on, and the direct Taylor chains of the GEMM
    // path for the large batches they win (n <= 32: from 112 control sets of more than 20 levels, 28 with a state regulariser;
    // n > 32: from 48, 112).
    const bool st = cfg->state_transfer != 0;
    const bool mfma_auto = mfma_ok && (!st || n <= 32 || (n <= 48 && k <= 4));
    auto plan_for = [&](int Bp) -> AutoPlan {
        const bool nt4_batch = n > 48 && ((k <= 4 && Bp >= QOC_PLAN_NT4_MIN_SETS_K4) || Bp >= QOC_PLAN_NT4_MIN_SETS);
        // 16 < n <= 32 below the latency mode's reach (long pulses): the GEMM route up to a few control sets, fewer the smaller the active
        // part of the padded matrices is (500 slices, GEMM route / MFMA batch kernels in ms: n = 32 x 6 0.290 / 0.315, x 8 0.340 / 0.318; n
        // = 27 x 4 0.251 / 0.270, x 6 0.288 / 0.271; n = 20 x 2 0.185 / 0.196, x 4 0.249 / 0.196; with a forbidden level n = 32 x 8 0.436 /
        // 0.473, n = 27 x 8 level, n = 20 x 6 0.373 / 0.360)
        const int qa_g = (n + 3) / 4;
        const int gemm_small = (st && qa_g >= 7) ? QOC_PLAN_GEMM_SMALL_ST_WIDE
                               : lat_src ? (qa_g <= 5 ? QOC_PLAN_GEMM_SMALL_SRC_Q5 : qa_g == 6 ? QOC_PLAN_GEMM_SMALL_SRC_Q6
                                   : QOC_PLAN_GEMM_SMALL_SRC_Q78)
                                         : (qa_g <= 5 ? QOC_PLAN_GEMM_SMALL_Q5 : qa_g == 6 ? QOC_PLAN_GEMM_SMALL_Q6 : qa_g == 7
                                             ? QOC_PLAN_GEMM_SMALL_Q7 : QOC_PLAN_GEMM_SMALL_Q8);
        const bool st_big = st && direct_ok && cfg->chunks <= 1 &&
                            (n <= 32 ? (Bp >= QOC_PLAN_ST_BIG_N32
                                && n > (lat_src ? QOC_PLAN_ST_BIG_N32_MIN_LEVELS_SRC : QOC_PLAN_ST_BIG_N32_MIN_LEVELS))
                                     : Bp >= (dpp_shape ? (lat_src ? QOC_PLAN_ST_BIG_DPP_SRC : QOC_PLAN_ST_BIG_DPP) : (lat_src
                                         ? QOC_PLAN_ST_BIG_N64_SRC : QOC_PLAN_ST_BIG_N64)));
        const bool prefer_gemm = gemm_ok && ((n > 48 && !nt4_batch) || (n > 32 && Bp < QOC_PLAN_NT3_MIN_SETS) ||
                                             (n > 16 && n <= 32 && Bp <= gemm_small && m <= 8 && steps >= QOC_PLAN_GEMM_SMALL_MIN_SLICES)
                                                 || st_big);
        const long long lat_work = (long long)Bp * steps;
        // 16 < n <= 32: the batch kernels work on the ACTIVE 4-row strips qa = ceil(n / 4) of the padded matrices since round 4 and take
        // over earlier the smaller n is (tools/padded_latency_sweep.py, 500 slices: n = 20 / 24 / 27 / 32 level at ~5 / 6 / 7 / 8.5 control
        // sets; with a forbidden level the latency mode stays ahead up to 8, at n = 20 up to 7): seeds x slices <= 512 qa, with a state
        // regulariser min(4096, 768 qa)
        const int qa = (n + 3) / 4 < 5 ? 5 : (n + 3) / 4;
        const long long lat_limit = n <= 16 ? (lat_src ? QOC_PLAN_LAT_WORK_SRC : QOC_PLAN_LAT_WORK)
                                            : (lat_src ? std::min<long long>(QOC_PLAN_LAT_WORK_SRC,
                                                (long long)QOC_PLAN_LAT_WORK_PER_STRIP_SRC * qa)
                                                       : (long long)QOC_PLAN_LAT_WORK_PER_STRIP * qa);
        // NT = 4 (also 32 < n <= 48 with k > 4, padded): 0.268 against 0.458 ms (GEMM route) for one seed of 500 slices, level at 8; NT =
        // 3: the competitors are slower (tools/mid_n_sweep.py); state transfer from 25 levels on: 5 .. 8 control sets go to the GEMM route
        // -- n = 32 x 8: 0.220 against 0.261 ms, with forbidden levels 0.272 / 0.316
        const int lat_sets = n > 16 ? ((st && qa_g >= 7) ? QOC_PLAN_LAT_SETS_N32_ST_WIDE : QOC_PLAN_LAT_SETS_N32) : (st
            ? QOC_PLAN_LAT_SETS_N16_ST : QOC_PLAN_LAT_SETS_N16);
        const bool latency = cfg->path == QOC_PATH_AUTO && cfg->variant == 0 && mfma_auto && qoc_mfma_latency_ok(d)
            && steps >= QOC_PLAN_LAT_MIN_SLICES &&
                              (((n > 48 || (n > 32 && k > 4)) ? (lat_work <= QOC_PLAN_LAT_WORK_NT4 && Bp <= QOC_PLAN_LAT_SETS_NT4)
                                : n > 32 ? (lat_work <= QOC_PLAN_LAT_WORK_NT3 && Bp <= QOC_PLAN_LAT_SETS_NT3)
                                         : (lat_work <= lat_limit && Bp <= lat_sets)) ||
                               (Bp == 1 && steps <= QOC_PLAN_LAT_SINGLE_MAX_SLICES));
        AutoPlan p;
        p.latency = latency;
        p.gemm_direct = direct_ok && (!antiherm || cfg->chunks == 1 || (cfg->chunks == 0 && Bp >= ST_DIRECT_FROM));
        p.path = cfg->path != QOC_PATH_AUTO ? cfg->path
                 : latency ? QOC_PATH_MFMA : (mfma_auto
                     && !prefer_gemm) ? QOC_PATH_MFMA : (gemm_ok ? QOC_PATH_GEMM : (st_ok ? QOC_PATH_ST_FUSED : QOC_PATH_GENERIC));
        return p;
    };
    const AutoPlan plan = plan_for(d.Bplan);
    const bool latency_auto = plan.latency, gemm_direct = plan.gemm_direct;
    path = plan.path;
    // n <= 12, one or a few control sets (the reference's own use) and small batches: the workgroup-resident iteration (csrc/qoc_small.h) --
    // 5-20 us per iteration where the paths above pay 42-57 us of launches and dependent round trips whatever n (profiles/r06_small_n_latency.txt)
    // (QOC_EXPERIMENTAL=1 QOC_SMALL_AUTO=0: AUTO as it was before round 6, for A/B runs -- tools/small_n_latency.py)
    if (cfg->path == QOC_PATH_AUTO && cfg->variant == 0 && cfg->chunks == 0 && cfg->time_shards < 1 && !qoc_exp_is("QOC_SMALL_AUTO", 0) && qoc_small_auto(d, antiherm))
        path = QOC_PATH_SMALL;
    if (d.Bplan < B) {
        // a plan for FEWER control sets than the engine holds is legal (a rank that holds several shards of a planned batch keeps
        // bit-identity with them) but can cost a factor: say so once when it changes what AUTO would have picked for the resident batch
        const AutoPlan own = plan_for(B);
        if (own.path != plan.path || own.latency != plan.latency || own.gemm_direct != plan.gemm_direct)
            fprintf(stderr, "libqoc_hip: note: plan_seeds = %d < n_seeds = %d changes the AUTO plan (path %d%s instead of %d%s): kernels "
                "tuned for the smaller batch run on the larger one\n",
                    d.Bplan, B, plan.path, plan.latency ? " latency mode" : "", own.path, own.latency ? " latency mode" : "");
    }
    if (cfg->time_shards >= 1) {
        if (cfg->time_rank < -1 || cfg->time_rank >= cfg->time_shards) return bail(fail(QOC_ERR_INVALID,
            "qoc_create: time_rank %d of %d time shards", cfg->time_rank, cfg->time_shards));
        if (cfg->path != QOC_PATH_AUTO && cfg->path != QOC_PATH_GEMM) return bail(fail(QOC_ERR_INVALID,
            "qoc_create: time sharding runs on the GEMM path"));
        path = QOC_PATH_GEMM;
    }
    if (path == QOC_PATH_MFMA && !mfma_ok)
        return bail(fail(QOC_ERR_INVALID, "qoc_create: MFMA path needs n <= 64, m <= 16, k <= 8, a Taylor degree of 1 .. 22 and, in state "
            "transfer, exactly anti-Hermitian generators (n=%d m=%d k=%d T=%d)", n, m, k, d.T));
    if (path == QOC_PATH_ST_FUSED && !st_ok)
        return bail(fail(QOC_ERR_INVALID,
            "qoc_create: fused state-transfer path needs state_transfer, n <= 64, m <= 4, k <= 8 (n=%d m=%d k=%d)", n, m, k));
    if (path == QOC_PATH_GEMM && !gemm_ok)
        return bail(fail(QOC_ERR_INVALID,
            "qoc_create: GEMM path needs m <= 32 and, in state transfer, exactly anti-Hermitian generators or n <= 64, m <= 8 (m=%d)", m));
    if (path < QOC_PATH_GENERIC || path > QOC_PATH_SMALL) return bail(fail(QOC_ERR_INVALID, "qoc_create: unknown path %d", path));
    e->path = path;
    e->chunks = 1;
#ifdef QOC_DEBUG     // timing experiments only (tools/skip_timing.py builds its own library with -DQOC_DEBUG): never in the product library
    // wall-clock attribution of one kernel group (results are garbage)
    if (const char* sk = getenv("QOC_DEBUG_SKIP")) {
        e->skip_mask = atoi(sk);
        if (e->skip_mask) fprintf(stderr, "libqoc_hip: WARNING: QOC_DEBUG_SKIP=%d is set -- kernel groups are skipped or repeated, every "
            "result of this engine is garbage (timing experiments only)\n", e->skip_mask);
    }
#endif
    if (path == QOC_PATH_MFMA) {
        std::string msg;
        e->mf.variant = latency_auto ? 5 : cfg->variant;
        if (cfg->variant == 5 && !qoc_mfma_latency_ok(d))
            return bail(fail(QOC_ERR_INVALID, "qoc_create: the latency mode of the MFMA path (variant 5) needs n <= 32 with k <= 8 (or, "
                "with at most 4 dressed forbidden levels, n <= 64), "
                                              "a Taylor degree >= 2 (n=%d k=%d T=%d)", n, k, d.T));
        // state transfer: sum_{j < T} A^j / j! is the polynomial of degree T - 1 (no squarings: d.s = 0)
        d.T = qoc_mfma_degree(d);
        rc = qoc_mfma_setup(e->mf, d, cfg->chunks, (const cplx*)Hs, e->allocs, msg);
        if (rc) return bail(fail(rc, "qoc_create: %s", msg.c_str()));
        e->chunks = e->mf.C;
    } else if (path == QOC_PATH_GEMM) {
        std::string msg;
        if (cfg->time_shards >= 1) { e->gm.ts_G = cfg->time_shards; e->gm.ts_rank = cfg->time_rank; }
        e->gm.antiherm = antiherm;
        e->gm.direct_variant = cfg->path == QOC_PATH_GEMM ? cfg->variant : 0;
        rc = qoc_gemm_setup(e->gm, d, (const cplx*)Hs, gemm_direct, e->allocs, msg);
        if (rc) return bail(fail(rc, "qoc_create: %s", msg.c_str()));
        if (!qoc_gemm_lds_opt_in()) return bail(fail(QOC_ERR_HIP, "qoc_create: cannot reserve LDS for the GEMM-path kernels"));
        e->chunks = e->gm.NC;
        e->gm.reduce_in_tail = e->gm.persistent && e->fin_part != nullptr && e->skip_mask == 0;      // (the split tail sums the gradient partials: one launch less)
        if (e->gm.ts_G > 0) {
            std::string why;
            if (!qoc_gemm_ts_supported(e->gm, d, e->gm.ts_G, why))
                return bail(fail(QOC_ERR_INVALID, "qoc_create: time_shards = %d needs %s (n=%d m=%d chunks=%d)", e->gm.ts_G, why.c_str(), n,
                    m, e->gm.NC));
            qoc_gemm_ts_ranges(e->gm, e->gm.ts_G);
        }
    } else if (path == QOC_PATH_SMALL) {
        std::string msg;
        rc = qoc_small_setup(e->sm, d, antiherm, cfg->chunks, cfg->variant, e->allocs, msg);
        if (rc) return bail(fail(rc == -1 ? QOC_ERR_INVALID : (rc == -3 ? QOC_ERR_NOMEM : QOC_ERR_HIP), "qoc_create: %s (n=%d m=%d k=%d T=%d steps=%d seeds=%d)",
            msg.c_str(), n, m, k, d.T, steps, B));
        e->chunks = e->sm.G;
        // read-back (inter_vecs, final_state, unitary_scale) runs the any-size kernels on the controls of the last evaluation
        if (!cfg->state_transfer) {
            ALLOC(e->K, (size_t)B * steps * nn);
            int grid = B * steps;
            if (grid > 4096) grid = 4096;
            e->expm_grid = grid;
            ALLOC(e->expm_scratch, (size_t)grid * 3 * nn);
            ALLOC(e->seed_scratch, (size_t)B * (2 * nn + 2 * nm));
        } else {
            ALLOC(e->seed_scratch, (size_t)B * (nn + 3 * nm));
        }
    } else if (!cfg->state_transfer) {
        ALLOC(e->K, (size_t)B * steps * nn);
        int grid = B * steps;
        if (grid > 4096) grid = 4096;
        e->expm_grid = grid;
        ALLOC(e->expm_scratch, (size_t)grid * 3 * nn);
        ALLOC(e->seed_scratch, (size_t)B * (2 * nn + 2 * nm));
    } else if (path == QOC_PATH_GENERIC) {
        ALLOC(e->seed_scratch, (size_t)B * (nn + 3 * nm));
    }
#undef ALLOC
    {
        const hipError_t se = hipDeviceSynchronize();
        if (se != hipSuccess) return bail(fail(QOC_ERR_HIP, "qoc_create: %s", hipGetErrorString(se)));
    }
    *out = e;
    return QOC_OK;
}

int qoc_destroy(qoc_handle e) {
    if (!e) return QOC_OK;
    hipSetDevice(e->cfg.device);
    if (e->stream) hipStreamSynchronize(e->stream);
    qoc_comm_detach(e->gm.ts_comm);                    // (a time-sharded engine: its communicator may be destroyed from now on)
    qoc_gemm_teardown(e->gm);
    for (void* p : e->allocs) hipFree(p);
    for (hipEvent_t x : e->ev) hipEventDestroy(x);
    if (e->t0) hipEventDestroy(e->t0);
    if (e->t1) hipEventDestroy(e->t1);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return QOC_OK;
}

#define CHECK_H(h) if (!(h)) return fail(QOC_ERR_INVALID, "null handle"); HIP_TRY(hipSetDevice((h)->cfg.device))

int qoc_set_base(qoc_handle e, const double* base) {
    CHECK_H(e);
    if (!base) return fail(QOC_ERR_INVALID, "qoc_set_base: null base");
    const QocDev& d = e->d;
    const size_t cnt = (size_t)d.B * d.k * d.steps;
    // everything on the engine stream (it is non-blocking: the legacy null stream orders nothing against it), then one sync so
    // that the caller may reuse `base` and the next qoc_iterate sees the cleared optimiser state
    HIP_TRY(hipMemcpyAsync(d.base, base, cnt * sizeof(double), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(d.adam_m, 0, cnt * sizeof(double), e->stream));
    HIP_TRY(hipMemsetAsync(d.adam_v, 0, cnt * sizeof(double), e->stream));
    HIP_TRY(hipMemsetAsync(d.adam_t, 0, d.B * sizeof(int), e->stream));
    HIP_TRY(hipMemsetAsync(d.iters, 0, d.B * sizeof(int), e->stream));
    HIP_TRY(hipMemsetAsync(d.done, 0, d.B * sizeof(int), e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->evaluated = false;
    e->controls_ready = false;
    return QOC_OK;
}

int qoc_get_base(qoc_handle e, double* base) {
    CHECK_H(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(base, e->d.base, (size_t)e->d.B * e->d.k * e->d.steps * sizeof(double), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_get_scalars(qoc_handle e, double* loss, double* reg_loss, double* grad_squared, double* unitary_scale,
                    int32_t* iterations, int32_t* done) {
    CHECK_H(e);
    const QocDev& d = e->d;
    if (unitary_scale) TRY(refresh_final(e));
    HIP_TRY(hipStreamSynchronize(e->stream));
    TRY(small_check(e));
    const size_t sz = (size_t)d.B * sizeof(double);
    if (loss) HIP_TRY(hipMemcpy(loss, d.loss, sz, hipMemcpyDeviceToHost));
    if (reg_loss) HIP_TRY(hipMemcpy(reg_loss, d.reg_loss, sz, hipMemcpyDeviceToHost));
    if (grad_squared) HIP_TRY(hipMemcpy(grad_squared, d.g2, sz, hipMemcpyDeviceToHost));
    if (unitary_scale) HIP_TRY(hipMemcpy(unitary_scale, d.uscale, sz, hipMemcpyDeviceToHost));
    if (iterations) HIP_TRY(hipMemcpy(iterations, d.iters, d.B * sizeof(int), hipMemcpyDeviceToHost));
    if (done) HIP_TRY(hipMemcpy(done, d.done, d.B * sizeof(int), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_eval(qoc_handle e, double* loss, double* reg_loss, double* grad_squared, double* unitary_scale, double* grad) {
    CHECK_H(e);
    QocAdamDev ap;
    memset(&ap, 0, sizeof ap);
    ap.mode = 0;
    TRY(enqueue_iteration(e, ap));
    TRY(qoc_get_scalars(e, loss, reg_loss, grad_squared, unitary_scale, nullptr, nullptr));
    if (grad) HIP_TRY(hipMemcpy(grad, e->d.grad, (size_t)e->d.B * e->d.k * e->d.steps * sizeof(double), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_adam_step(qoc_handle e, const double* lr) {
    CHECK_H(e);
    if (!lr) return fail(QOC_ERR_INVALID, "qoc_adam_step: null lr");
    if (!e->evaluated) return fail(QOC_ERR_STATE, "qoc_adam_step: no evaluation since the last qoc_set_base");
    // per-seed learning rates live in the engine's arena (no allocation per step); the copy is ordered on the engine stream
    HIP_TRY(hipMemcpyAsync(e->step_lr, lr, e->d.B * sizeof(double), hipMemcpyHostToDevice, e->stream));
    QocAdamDev ap;
    memset(&ap, 0, sizeof ap);
    ap.mode = 2;
    ap.lr = e->step_lr;
    // the reference re-evaluates the gradient at the same parameters inside session.run([optimizer]) (run_session.py:69)
    TRY(enqueue_iteration(e, ap));
    HIP_TRY(hipStreamSynchronize(e->stream));                       // `lr` (pageable host memory) may be reused by the caller
    return QOC_OK;
}

int qoc_iterate(qoc_handle e, const qoc_adam_params* p, int32_t iters) {
    CHECK_H(e);
    if (!p) return fail(QOC_ERR_INVALID, "qoc_iterate: null params");
    const QocAdamDev ap = loop_params(p);
    if (e->path == QOC_PATH_SMALL) return iters > 0 ? enqueue_small(e, ap, iters) : QOC_OK;   // the loop runs inside the launch
    for (int i = 0; i < iters; ++i) TRY(enqueue_iteration(e, ap));
    return QOC_OK;
}

int qoc_sync(qoc_handle e) {
    CHECK_H(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    TRY(small_check(e));
    return QOC_OK;
}

int qoc_run_adam(qoc_handle e, const qoc_adam_params* p, int32_t* iterations_out) {
    CHECK_H(e);
    if (!p) return fail(QOC_ERR_INVALID, "qoc_run_adam: null params");
    const QocAdamDev ap = loop_params(p);
    const int poll = p->poll_every > 0 ? p->poll_every : 1;
    std::vector<int> done(e->d.B);
    // at most max_iterations updates + the evaluation that trips the stop rule
    const long long budget = (long long)p->max_iterations + 1;
    long long launched = 0;
    while (true) {
        int burst = poll;
        if (launched + burst > budget) burst = (int)(budget - launched);
        if (e->path == QOC_PATH_SMALL) TRY(enqueue_small(e, ap, burst));
        else for (int i = 0; i < burst; ++i) TRY(enqueue_iteration(e, ap));
        launched += burst;
        HIP_TRY(hipStreamSynchronize(e->stream));
        TRY(small_check(e));
        HIP_TRY(hipMemcpy(done.data(), e->d.done, e->d.B * sizeof(int), hipMemcpyDeviceToHost));
        bool all = true;
        for (int b = 0; b < e->d.B; ++b) all = all && done[b];
        if (all) break;
        if (launched >= budget) return fail(QOC_ERR_STATE, "qoc_run_adam: seeds not finished after %lld evaluations", launched);
    }
    if (iterations_out) HIP_TRY(hipMemcpy(iterations_out, e->d.iters, e->d.B * sizeof(int), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_get_uks(qoc_handle e, double* uks) {
    CHECK_H(e);
    if (!uks) return fail(QOC_ERR_INVALID, "qoc_get_uks: null output");
    const QocDev& d = e->d;
    // uks = maxA[k] * sin(base) of the CURRENT variable (run_session.py:112-117), evaluated on the device
    const int total = d.B * d.k * d.steps;
    int cgrid = (total + QOC_BLOCK - 1) / QOC_BLOCK;
    if (cgrid > 2048) cgrid = 2048;
    // (into u2 / w2: u / w keep the controls of the last evaluation for qoc_get_uks_evaluated and the regularisers' read-backs)
    if (!e->controls_ready) {
        QocDev dd = d;
        dd.u = d.u2; dd.w = d.w2;
        hipLaunchKernelGGL(k_controls, dim3(cgrid), dim3(QOC_BLOCK), 0, e->stream, dd);
        HIP_TRY(hipGetLastError());
        e->controls_ready = true;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(uks, d.u2, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_get_uks_evaluated(qoc_handle e, double* uks) {
    CHECK_H(e);
    if (!uks) return fail(QOC_ERR_INVALID, "qoc_get_uks_evaluated: null output");
    if (!e->evaluated) return fail(QOC_ERR_STATE, "qoc_get_uks_evaluated: nothing evaluated yet");
    // d.u still holds the controls the last evaluation ran on (an Adam step only moves `base`)
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(uks, e->d.u, (size_t)e->d.B * e->d.k * e->d.steps * sizeof(double), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_get_final_unitary(qoc_handle e, double* Uf) {
    CHECK_H(e);
    if (e->d.state_transfer) return fail(QOC_ERR_STATE, "qoc_get_final_unitary: state-transfer mode has no final unitary");
    if (!e->evaluated) return fail(QOC_ERR_STATE, "qoc_get_final_unitary: nothing evaluated yet");
    TRY(refresh_final(e));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(Uf, e->d.Xfinal, (size_t)e->d.B * e->d.n * e->d.n * sizeof(cplx), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_get_inter_vecs(qoc_handle e, double* inter) {
    CHECK_H(e);
    if (!e->evaluated) return fail(QOC_ERR_STATE, "qoc_get_inter_vecs: nothing evaluated yet");
    // one rank of a time-sharded run: its own slices, summed over the ranks (a collective)
    if (e->path == QOC_PATH_GEMM) TRY(qoc_gemm_ts_gather_inter(e->gm, e->d, e->stream));
    if (e->path == QOC_PATH_SMALL) TRY(refresh_small(e, true));
    else if (e->inter_stale) {                                       // latency mode: the sweeps keep Psi_t in their own layout
        qoc_mfma_unpack_inter(e->mf, e->d, e->stream);
        HIP_TRY(hipGetLastError());
        e->inter_stale = false;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(inter, e->d.inter, (size_t)e->d.B * (e->d.steps + 1) * e->d.n * e->d.m * sizeof(cplx), hipMemcpyDeviceToHost));
    return QOC_OK;
}

int qoc_profile_enable(qoc_handle e, int32_t on) {
    CHECK_H(e);
    TRY(prof_collect(e));
    e->profiling = on != 0;
    e->prof_ms = 0.0;
    e->prof_launches = 0;
    return QOC_OK;
}

int qoc_profile_read(qoc_handle e, const char** kernel_name, int64_t* launches, double* total_ms) {
    CHECK_H(e);
    TRY(prof_collect(e));
    if (kernel_name)
        *kernel_name = e->path == QOC_PATH_SMALL ? "k_small_iter (whole iterations)" : e->path == QOC_PATH_GEMM ? (e->gm.direct ? "k_gemm_taylor_chain (backward chain + sources + gradient products)"
                                                 : e->gm.N <= 64 ? "k_gemm_expm_fused (+ product tree)" : "k_zgemm_wg + k_zgemm32 (batched "
                                                     "matexp sequence)")
                       : e->path == QOC_PATH_MFMA ? (qoc_mfma_expm_variant(e->mf, e->d) == 7 ? "k_mfma_expm_rows"
                           : qoc_mfma_expm_variant(e->mf, e->d) == 6 ? "k_mfma_expm_pair" : qoc_mfma_expm_variant(e->mf, e->d) == 5
                           ? (e->mf.NT == 3 ? "k_mfma_expm_rows (per slice) + k_mfma_chain_rows" : "k_mfma_expm_slice2 + "
                           "k_mfma_chain_rows") : qoc_mfma_expm_variant(e->mf, e->d) == 8 ? "k_mfma_expm_inplace"
                           : qoc_mfma_expm_variant(e->mf, e->d) == 4 ? "k_mfma_expm_chunk4s" : qoc_mfma_expm_variant(e->mf, e->d) == 3
                           ? "k_mfma_expm_chunk4w" : qoc_mfma_expm_variant(e->mf, e->d) == 2 ? "k_mfma_expm_chunk4" : "k_mfma_expm_chunk")
                       : (e->path == QOC_PATH_ST_FUSED ? "k_st_fwd_fused" : (e->d.state_transfer ? "k_st_fwd_generic" : "k_expm_generic"));
    if (launches) *launches = e->prof_launches;
    if (total_ms) *total_ms = e->prof_ms;
    return QOC_OK;
}

int qoc_time_iterations(qoc_handle e, const qoc_adam_params* p, int32_t iters, double* elapsed_ms) {
    CHECK_H(e);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipEventRecord(e->t0, e->stream));
    TRY(qoc_iterate(e, p, iters));
    HIP_TRY(hipEventRecord(e->t1, e->stream));
    HIP_TRY(hipEventSynchronize(e->t1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->t0, e->t1));
    if (elapsed_ms) *elapsed_ms = ms;
    return QOC_OK;
}

int qoc_path_in_use(qoc_handle e) { return e ? e->path : QOC_ERR_INVALID; }
int qoc_chunks_in_use(qoc_handle e) { return e ? e->chunks : QOC_ERR_INVALID; }

// What AUTO resolved to, as one line of key=value pairs (tests/test_auto_plan.py pins the dispatch table of DESIGN.md section 4 with it):
// MFMA path:  path=mfma nt=<tiles> expm=<exponential kernel 1..8> chunks=<C>
// sweeps=<downup|split|row_tile_gradient|latency|latency_sources|one_wave>
//   GEMM path:  path=gemm route=<unitary|propagator|direct> chunks=<NC> slices_per_chunk=<S> chains=<persistent|launches>
//   others:     path=generic | path=st_fused
int qoc_plan_describe(qoc_handle e, char* buf, int32_t len) {
    if (!e || !buf || len < 1) return fail(QOC_ERR_INVALID, "qoc_plan_describe: null handle or buffer");
    char tmp[256];
    if (e->path == QOC_PATH_MFMA) {
        const QocMfma& mf = e->mf;
        const bool split = (mf.NT > 2 || (mf.NT == 2 && e->d.k >= 6)) && mf.variant != 1;
        const char* sweeps = mf.latency ? (mf.lat_sources ? "latency_sources" : "latency")
                             : mf.updown ? "downup" : (split ? (mf.grad_rt ? "row_tile_gradient" : "split") : (mf.variant == 1
                                 || mf.NT == 1 || mf.NT == 4 ? "one_wave" : "pair"));
        snprintf(tmp, sizeof tmp, "path=mfma nt=%d expm=%d chunks=%d sweeps=%s", mf.NT, qoc_mfma_expm_variant(mf, e->d), mf.C, sweeps);
    } else if (e->path == QOC_PATH_GEMM) {
        const QocGemm& g = e->gm;
        int w = snprintf(tmp, sizeof tmp, "path=gemm route=%s chunks=%d slices_per_chunk=%d chains=%s",
            g.direct ? "direct" : (e->d.state_transfer ? "propagator" : "unitary"),
                         g.NC, g.S, g.persistent ? "persistent" : "launches");
        if (g.ts_G > 0) w += snprintf(tmp + w, sizeof tmp - w, " time_shards=%d time_rank=%d", g.ts_G, g.ts_rank);
        // the kernel of the direct route's Taylor chains: squared (k_gemm_taylor_chain_sq on [B | B^2]), packed / full
        // (k_gemm_taylor_chain_dpp), butterfly (k_gemm_taylor_chain)
        if (g.direct) snprintf(tmp + w, sizeof tmp - w, " taylor_chain=%s",
            g.sq_chain ? "squared" : g.dpp_packed ? "packed"
            : g.dpp_chain ? (g.dpp_cw == 10 ? "columns40" : g.dpp_cw == 12 ? "columns48" : g.dpp_cw == 14 ? "columns56" : "full")
                : "butterfly");
    } else if (e->path == QOC_PATH_SMALL) {
        snprintf(tmp, sizeof tmp, "path=small n_pad=%d rows=%d slices_per_row=%d workgroups=%d state_sources=%d lds_kb=%d", e->sm.N, e->sm.R, e->sm.L, e->sm.G,
            e->sm.src ? 1 : 0, (int)((e->sm.lds_bytes + 1023) / 1024));
    } else {
        snprintf(tmp, sizeof tmp, "path=%s", e->path == QOC_PATH_ST_FUSED ? "st_fused" : "generic");
    }
    snprintf(buf, (size_t)len, "%s", tmp);
    return QOC_OK;
}

}  // extern "C"

#include "qoc_comm.h"
