// qoc_kernels_finish.h -- per-seed tail of one GRAPE iteration, fused into a single kernel:
//   pulse-shape regularisers and their gradient      core/regularization_functions.py:15-67
//   chain rule through u = maxA*sin(base)             core/tensorflow_state.py:176-178
//   grad_squared = sum g^2/2, reg_loss                core/tensorflow_state.py:348-353
//   stop rule + learning-rate schedule                core/run_session.py:56-66
//   TF1 Adam update                                   core/tensorflow_state.py:344-354 (tf.train.AdamOptimizer)
#pragma once
#include "qoc_common.h"

__device__ __forceinline__ double padded_w(const double* __restrict__ w, int steps, int i) {
    // new_weights = [0, 0, w_0 ... w_{steps-1}, 0, 0]          regularization_functions.py:29-31
    return (i >= 2 && i < steps + 2) ? w[i - 2] : 0.0;
}

__device__ __forceinline__ void unit_phase(int f, int t, int N, double* c, double* s) {
    // e^{-2 pi i f t / N} with the phase reduced in integers first
    const long long r = ((long long)f * (long long)t) % (long long)N;
    sincos(-2.0 * M_PI * (double)r / (double)N, s, c);
}

// two sums in one pass over the workgroup (one pair of barriers instead of two): results valid in every thread
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red /* >= 2 * (QOC waves) doubles */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { red[2 * wid] = a; red[2 * wid + 1] = b; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int i = 0; i < nw; ++i) { ta += red[2 * i]; tb += red[2 * i + 1]; }
    a = ta; b = tb;
}

// Launched with 1024 threads when a seed has >= 2048 (k, t) elements: the element loops are latency-bound.  The kernel is one
// dependent chain of global round trips for a single trajectory (17 us of a 133 us iteration): the Adam slots and everything else
// an element needs are fetched up front and kept in registers (up to QF_E elements per thread), and the two reductions share their
// barriers.
#define QF_E 4            // the default; QFE = 8 instances of the stand-alone kernel for seeds of 4097 .. 8192 elements (C3: k = 6 x 1000 slices)
// PLAIN: no pulse regulariser is configured -- the same kernel with those branches compiled out (a tenth of the code: for one
// trajectory this single-workgroup kernel is bound by its instruction fetch and its chain of global round trips, not by arithmetic)
// (a __device__ body: the latency mode of the MFMA path runs it in the last workgroup of its gradient kernel, qoc_mfma_latency.h)
// LEVEL: 0 = no pulse regulariser (PLAIN), 1 = the local ones (amplitude, envelope, dwdt, d2wdt2) but no bandpass DFT, 2 = all.
// SPLIT (one control set of many elements -- C3: 6 x 1000 -- is bound by the fp64 sin / cos / sqrt / divide of ONE compute unit: 34 us): 1 = part A, S workgroups
// per control set form the gradient elements and their partial sums (`part`: [B][S + 2][2] doubles, the last two rows a snapshot of the counters part B must not
// read while its first workgroup updates them); 2 = part B, every workgroup sums the partials in the same order, takes the stop rule, updates its own elements.
// gp (part A only): dL/du of an element still lies as per-tile partial dots of the wide gradient product (GEMM path, k_gemm_grad_reduce_wide's input) -- summed
// here, in that kernel's order, and stored to dLdu: one launch less in front of the tail
struct QocGradPartial { const double* partial; int tiles_m, ldW, MV; };
template <int LEVEL, int QFE = QF_E, int SPLIT = 0>
__device__ __forceinline__ void finish_body(const QocDev& d, const QocAdamDev& ap, const int b, double* red /* 34 doubles of LDS */, double* part = nullptr,
                                            const QocGradPartial* gp = nullptr) {
    // no implicit contraction into FMAs in here: the three flavours (and the fused / separate launches of the latency mode) must round
    // alike -- with a regulariser of weight zero they are bit-identical -- and which a*b + c the compiler contracts depends on the code
    // around it (an edit of the bandpass loops flipped one in the Adam update).  Every fma() below is written out.
#pragma clang fp contract(off)
    constexpr bool PLAIN = LEVEL == 0, BAND = LEVEL == 2;
    const int steps = d.steps, ks = d.k * steps;
    const double* w = d.w + (size_t)b * ks;
    const double* dLdu = d.dLdu + (size_t)b * ks;
    double* base = d.base + (size_t)b * ks;
    double* grad = d.grad + (size_t)b * ks;
    // the elements of this thread: tid, tid + tn, ... (one workgroup: its threads; split: the threads of the S workgroups of the control set)
    const int tid = SPLIT ? (int)(blockIdx.x * blockDim.x + threadIdx.x) : (int)threadIdx.x, tn = SPLIT ? (int)(gridDim.x * blockDim.x) : (int)blockDim.x;
    double* const pb = SPLIT ? part + (size_t)b * (gridDim.x + 2) * 2 : nullptr;
    const int it0 = SPLIT == 2 ? (int)pb[2 * gridDim.x] : d.iters[b];
    const int was_done = SPLIT == 2 ? (int)pb[2 * gridDim.x + 1] : d.done[b];
    const double dt = d.dt;
    const double loss = d.loss[b], reg_state_v = d.reg_state[b];
    const int adam_t0 = SPLIT == 2 ? (int)pb[2 * gridDim.x + 2] : d.adam_t[b];
    double* am = d.adam_m + (size_t)b * ks;
    double* av = d.adam_v + (size_t)b * ks;
    const bool in_regs = SPLIT == 0 && ks <= QFE * (int)blockDim.x;      // every element of this thread fits its register file
    double g_r[QFE], m_r[QFE], v_r[QFE], b_r[QFE];
    if (in_regs && ap.mode != 0) {
#pragma unroll
        for (int e = 0; e < QFE; ++e) {
            const int o = threadIdx.x + e * blockDim.x;
            m_r[e] = o < ks ? am[o] : 0.0; v_r[e] = o < ks ? av[o] : 0.0;
        }
    }

    // Learning rate and Adam bias correction of the step that follows IF the stop rule lets it happen: they depend on the iteration
    // counters only, so their exp / pow (a few hundred instructions on the dependent chain of a single-trajectory tail) are formed
    // here, under the latency of the loads above, not after the reduction that decides the stop rule.
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const double lr_next = ap.mode == 1 ? ap.rate * exp(-(double)(it0 + 1) / ap.decay) : (ap.mode == 2 ? ap.lr[b] : 0.0);   // run_session.py:66
    const double lr_t = lr_next * sqrt(1.0 - pow(b2, (double)(adam_t0 + 1))) / (1.0 - pow(b1, (double)(adam_t0 + 1)));

    double reg = 0.0;
    double g2 = 0.0;
    if constexpr (SPLIT != 2) {
    // ---- values that are not sums over (k,t) elements --------------------------------------------------------
    if (!PLAIN && d.has_dwdt) {                                                            // :28-35
        double acc = 0.0;
        for (int o = tid; o < d.k * (steps + 3); o += tn) {
            const int kk = o / (steps + 3), i = o - kk * (steps + 3);
            const double* wk = w + (size_t)kk * steps;
            const double dd = (padded_w(wk, steps, i + 1) - padded_w(wk, steps, i)) / dt;
            acc += dd * dd;
        }
        reg += d.a_dwdt * 0.5 * acc;
    }
    if (!PLAIN && d.has_d2wdt2) {                                                          // :38-45
        double acc = 0.0;
        for (int o = tid; o < d.k * (steps + 2); o += tn) {
            const int kk = o / (steps + 2), i = o - kk * (steps + 2);
            const double* wk = w + (size_t)kk * steps;
            const double e = (padded_w(wk, steps, i + 2) - 2.0 * padded_w(wk, steps, i + 1) + padded_w(wk, steps, i)) / (dt * dt);
            acc += e * e;
        }
        reg += d.a_d2wdt2 * 0.5 * acc;
    }
    if (BAND && d.has_band) {                                                              // :47-67: sum of cnt |F_f| (k_band_spectrum)
        double acc = 0.0;
        const double* bm = d.band_mag + (size_t)b * ks;
        for (int o = tid; o < ks; o += tn) acc += bm[o];
        reg += d.a_band * acc;
    }

    // ---- per-element: remaining values, d reg / d w, chain rule -------------------------------------------------
    for (int o = tid; o < ks; o += tn) {
        const int kk = o / steps, t = o - kk * steps;
        const double* wk = w + (size_t)kk * steps;
        const double wv = wk[t];
        double dR = 0.0;
        if (!PLAIN && d.has_amp) { reg += d.a_amp * 0.5 * wv * wv; dR += d.a_amp * wv; }                    // :15-18
        if (!PLAIN && d.has_env) {                                                                          // :21-25
            const double e = d.omg[o];
            reg += d.a_env * 0.5 * (e * wv) * (e * wv);
            dR += d.a_env * e * e * wv;
        }
        const int p = t + 2;
        if (!PLAIN && d.has_dwdt) {
            const double dm = (padded_w(wk, steps, p) - padded_w(wk, steps, p - 1)) / dt;          // d_{p-1}
            const double dp = (padded_w(wk, steps, p + 1) - padded_w(wk, steps, p)) / dt;          // d_p
            dR += d.a_dwdt * (dm - dp) / dt;
        }
        if (!PLAIN && d.has_d2wdt2) {
            const double dt2 = dt * dt;
            const double e0 = (padded_w(wk, steps, p) - 2.0 * padded_w(wk, steps, p - 1) + padded_w(wk, steps, p - 2)) / dt2;      // e_{p-2}
            const double e1 = (padded_w(wk, steps, p + 1) - 2.0 * padded_w(wk, steps, p) + padded_w(wk, steps, p - 1)) / dt2;      // e_{p-1}
            const double e2 = (padded_w(wk, steps, p + 2) - 2.0 * padded_w(wk, steps, p + 1) + padded_w(wk, steps, p)) / dt2;      // e_p
            dR += d.a_d2wdt2 * (e0 - 2.0 * e1 + e2) / dt2;
        }
        if (BAND && d.has_band) dR += d.a_band * d.band_dR[(size_t)b * ks + o];                            // (k_band_gradient)
        const double bv = base[o];
        double dl;
        if (SPLIT == 1 && gp && gp->partial) {
            const double* pp = gp->partial + ((size_t)b * d.k + kk) * gp->tiles_m * (size_t)gp->ldW + (size_t)t * gp->MV;
            dl = 0.0;
            for (int i = 0; i < gp->tiles_m; ++i)
                for (int jv = 0; jv < gp->MV; ++jv) dl += pp[(size_t)i * gp->ldW + jv];
            d.dLdu[(size_t)b * ks + o] = dl;
        } else dl = dLdu[o];
        const double g = cos(bv) * (d.maxA[kk] * dl + dR);
        grad[o] = g;
        g2 += g * g;
        if (in_regs) {
            const int e = (o - (int)threadIdx.x) / (int)blockDim.x;
#pragma unroll
            for (int q = 0; q < QFE; ++q) if (q == e) { g_r[q] = g; b_r[q] = bv; }
        }
    }
    block_sum2(reg, g2, red);
    }
    if constexpr (SPLIT == 1) {
        if (threadIdx.x == 0) {
            pb[2 * blockIdx.x] = reg; pb[2 * blockIdx.x + 1] = g2;
            if (blockIdx.x == 0) { pb[2 * gridDim.x] = (double)it0; pb[2 * gridDim.x + 1] = (double)was_done; pb[2 * gridDim.x + 2] = (double)adam_t0; }
        }
        return;
    }
    if constexpr (SPLIT == 2) {
        for (unsigned i = 0; i < gridDim.x; ++i) { reg += pb[2 * i]; g2 += pb[2 * i + 1]; }
    }
    const bool writer = threadIdx.x == 0 && (SPLIT == 0 || blockIdx.x == 0);      // the one thread of the control set that stores its scalars
    g2 *= 0.5;                                                                   // sum of tf.nn.l2_loss
    if (writer) {
        d.reg_loss[b] = loss + reg_state_v + reg;
        d.g2[b] = g2;
    }
    if (ap.mode == 0) return;

    // controls of the next evaluation (u2 / w2; the engine swaps them in instead of launching k_controls): a seed that does not move keeps its own
    auto keep_controls = [&]() {
        if (!d.u2) return;
        for (int o = tid; o < ks; o += tn) { d.w2[(size_t)b * ks + o] = w[o]; d.u2[(size_t)b * ks + o] = d.u[(size_t)b * ks + o]; }
    };
    if (ap.mode == 1) {                                                          // run_session.py:56-66
        if (was_done) { keep_controls(); return; }
        const bool end = (loss < ap.conv_target) || (g2 < ap.min_grad) || (it0 >= ap.max_iterations);
        if (end) {
            if (writer) d.done[b] = 1;
            keep_controls();
            return;
        }
        if (writer) d.iters[b] = it0 + 1;                                        // update_and_save :92
    }
    const int tstep = adam_t0 + 1;
    if (in_regs) {
#pragma unroll
        for (int e = 0; e < QFE; ++e) {
            const int o = threadIdx.x + e * blockDim.x;
            if (o < ks) {
                const double g = g_r[e];
                const double mm = b1 * m_r[e] + (1.0 - b1) * g;
                const double vv = b2 * v_r[e] + (1.0 - b2) * g * g;
                am[o] = mm; av[o] = vv;
                const double bn = b_r[e] - lr_t * mm / (sqrt(vv) + eps);
                base[o] = bn;
                if (d.u2) { const double wn = sin(bn); d.w2[(size_t)b * ks + o] = wn; d.u2[(size_t)b * ks + o] = d.maxA[o / steps] * wn; }   // = k_controls
            }
        }
    } else {
        for (int o = tid; o < ks; o += tn) {
            const double g = grad[o];
            const double mm = b1 * am[o] + (1.0 - b1) * g;
            const double vv = b2 * av[o] + (1.0 - b2) * g * g;
            am[o] = mm; av[o] = vv;
            const double bn = base[o] - lr_t * mm / (sqrt(vv) + eps);
            base[o] = bn;
            if (d.u2) { const double wn = sin(bn); d.w2[(size_t)b * ks + o] = wn; d.u2[(size_t)b * ks + o] = d.maxA[o / steps] * wn; }
        }
    }
    if (writer) d.adam_t[b] = tstep;
}

template <bool PLAIN, int QFE = QF_E>
__global__ void __launch_bounds__(1024) k_finish_t(QocDev d, QocAdamDev ap) {
    __shared__ double red[34];
    finish_body<PLAIN ? 0 : 2, QFE>(d, ap, blockIdx.x, red);
}

// control sets of more than 4096 elements (a single C3 trajectory: 6000): the two halves of finish_body over S workgroups per control set (grid: S x B)
template <bool PLAIN>
__global__ void __launch_bounds__(256) k_finish_split_a(QocDev d, QocAdamDev ap, double* part, QocGradPartial gp) {
    __shared__ double red[34];
    finish_body<PLAIN ? 0 : 2, QF_E, 1>(d, ap, blockIdx.y, red, part, &gp);
}
template <bool PLAIN>
__global__ void __launch_bounds__(256) k_finish_split_b(QocDev d, QocAdamDev ap, double* part) {
    __shared__ double red[34];
    finish_body<PLAIN ? 0 : 2, QF_E, 2>(d, ap, blockIdx.y, red, part);
}
