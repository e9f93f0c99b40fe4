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

// Launched with 1024 threads when a seed has >= 2048 (k, t) elements: the element loops are latency-bound
__global__ void __launch_bounds__(1024) k_finish(QocDev d, QocAdamDev ap) {
    __shared__ double red[17];
    const int b = blockIdx.x, steps = d.steps, ks = d.k * steps;
    const double* w = d.w + (size_t)b * ks;
    const double* dLdu = d.dLdu + (size_t)b * ks;
    double* base = d.base + (size_t)b * ks;
    double* grad = d.grad + (size_t)b * ks;
    const int it0 = d.iters[b];
    const int was_done = d.done[b];
    const double dt = d.dt;

    double reg = 0.0;
    // ---- values that are not sums over (k,t) elements --------------------------------------------------------
    if (d.has_dwdt) {                                                            // :28-35
        double acc = 0.0;
        for (int o = threadIdx.x; o < d.k * (steps + 3); o += blockDim.x) {
            const int kk = o / (steps + 3), i = o - kk * (steps + 3);
            const double* wk = w + (size_t)kk * steps;
            const double dd = (padded_w(wk, steps, i + 1) - padded_w(wk, steps, i)) / dt;
            acc += dd * dd;
        }
        reg += d.a_dwdt * 0.5 * acc;
    }
    if (d.has_d2wdt2) {                                                          // :38-45
        double acc = 0.0;
        for (int o = threadIdx.x; o < d.k * (steps + 2); o += blockDim.x) {
            const int kk = o / (steps + 2), i = o - kk * (steps + 2);
            const double* wk = w + (size_t)kk * steps;
            const double e = (padded_w(wk, steps, i + 2) - 2.0 * padded_w(wk, steps, i + 1) + padded_w(wk, steps, i)) / (dt * dt);
            acc += e * e;
        }
        reg += d.a_d2wdt2 * 0.5 * acc;
    }
    cplx* ph = d.band_ph ? d.band_ph + (size_t)b * ks : nullptr;
    const int half = steps / 2;
    const int lo = min(max(d.band_lo, 0), steps), hi = min(max(d.band_hi, 0), steps);
    if (d.has_band) {                                                            // :47-67
        double acc = 0.0;
        for (int o = threadIdx.x; o < d.k * steps; o += blockDim.x) {
            const int kk = o / steps, f = o - kk * steps;
            const int cnt = (f < lo ? 1 : 0) + ((f >= hi && f < half) ? 1 : 0);
            cplx p = cmake(0.0, 0.0);
            if (cnt > 0) {
                const double* wk = w + (size_t)kk * steps;
                double fr = 0.0, fi = 0.0;
                for (int t = 0; t < steps; ++t) {
                    double c, s;
                    unit_phase(f, t, steps, &c, &s);
                    fr = fma(wk[t], c, fr);
                    fi = fma(wk[t], s, fi);
                }
                const double mag = sqrt(fr * fr + fi * fi);
                acc += (double)cnt * mag;
                if (mag > 0.0) p = cmake((double)cnt * fr / mag, -(double)cnt * fi / mag);   // cnt * conj(F)/|F|
            }
            ph[o] = p;
        }
        reg += d.a_band * acc;
        __syncthreads();
    }

    // ---- per-element: remaining values, d reg / d w, chain rule -------------------------------------------------
    double g2 = 0.0;
    for (int o = threadIdx.x; o < ks; o += blockDim.x) {
        const int kk = o / steps, t = o - kk * steps;
        const double* wk = w + (size_t)kk * steps;
        const double wv = wk[t];
        double dR = 0.0;
        if (d.has_amp) { reg += d.a_amp * 0.5 * wv * wv; dR += d.a_amp * wv; }                    // :15-18
        if (d.has_env) {                                                                          // :21-25
            const double e = d.omg[o];
            reg += d.a_env * 0.5 * (e * wv) * (e * wv);
            dR += d.a_env * e * e * wv;
        }
        const int p = t + 2;
        if (d.has_dwdt) {
            const double dm = (padded_w(wk, steps, p) - padded_w(wk, steps, p - 1)) / dt;          // d_{p-1}
            const double dp = (padded_w(wk, steps, p + 1) - padded_w(wk, steps, p)) / dt;          // d_p
            dR += d.a_dwdt * (dm - dp) / dt;
        }
        if (d.has_d2wdt2) {
            const double dt2 = dt * dt;
            const double e0 = (padded_w(wk, steps, p) - 2.0 * padded_w(wk, steps, p - 1) + padded_w(wk, steps, p - 2)) / dt2;      // e_{p-2}
            const double e1 = (padded_w(wk, steps, p + 1) - 2.0 * padded_w(wk, steps, p) + padded_w(wk, steps, p - 1)) / dt2;      // e_{p-1}
            const double e2 = (padded_w(wk, steps, p + 2) - 2.0 * padded_w(wk, steps, p + 1) + padded_w(wk, steps, p)) / dt2;      // e_p
            dR += d.a_d2wdt2 * (e0 - 2.0 * e1 + e2) / dt2;
        }
        if (d.has_band) {
            double acc = 0.0;
            const cplx* pk = ph + (size_t)kk * steps;
            for (int f = 0; f < half || f < lo; ++f) {
                if (f >= steps) break;
                const cplx q = pk[f];
                if (q.x == 0.0 && q.y == 0.0) continue;
                double c, s;
                unit_phase(f, t, steps, &c, &s);
                acc += q.x * c - q.y * s;                              // Re(ph_f * e^{-2 pi i f t/N})
            }
            dR += d.a_band * acc;
        }
        const double g = cos(base[o]) * (d.maxA[kk] * dLdu[o] + dR);
        grad[o] = g;
        g2 += g * g;
    }
    reg = block_sum(reg, red);
    g2 = 0.5 * block_sum(g2, red);                                               // sum of tf.nn.l2_loss
    const double loss = d.loss[b];
    if (threadIdx.x == 0) {
        d.reg_loss[b] = loss + d.reg_state[b] + reg;
        d.g2[b] = g2;
    }
    if (ap.mode == 0) return;

    double lr;
    int tstep;
    if (ap.mode == 1) {                                                          // run_session.py:56-66
        if (was_done) return;
        const bool end = (loss < ap.conv_target) || (g2 < ap.min_grad) || (it0 >= ap.max_iterations);
        if (end) {
            if (threadIdx.x == 0) d.done[b] = 1;
            return;
        }
        const int it1 = it0 + 1;                                                 // update_and_save :92
        lr = ap.rate * exp(-(double)it1 / ap.decay);                             // :66
        if (threadIdx.x == 0) d.iters[b] = it1;
    } else {
        lr = ap.lr[b];
    }
    tstep = d.adam_t[b] + 1;
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    const double lr_t = lr * sqrt(1.0 - pow(b2, (double)tstep)) / (1.0 - pow(b1, (double)tstep));
    double* am = d.adam_m + (size_t)b * ks;
    double* av = d.adam_v + (size_t)b * ks;
    __syncthreads();                                                             // all reads of adam_t done
    for (int o = threadIdx.x; o < ks; o += blockDim.x) {
        const double g = grad[o];
        const double mm = b1 * am[o] + (1.0 - b1) * g;
        const double vv = b2 * av[o] + (1.0 - b2) * g * g;
        am[o] = mm; av[o] = vv;
        base[o] -= lr_t * mm / (sqrt(vv) + eps);
    }
    if (threadIdx.x == 0) d.adam_t[b] = tstep;
}
