// expm_phase_probe.hip -- where do the cycles of k_mfma_expm_chunk4w go?  A timed copy of the kernel (NT = 2, C2 shape: T = 5,
// s = 3, k = 4, 64 seeds x 16 chunks) that reads the shader clock (s_memtime) between its phases and sums the differences per
// phase over all waves.  Inputs are random (timing only).  Phases: 0 assemble A_t, 1 image writes (put_all), 2 MFMA product loop
// incl. operand fetches, 3 combine (a-b, c-a-b) + Horner terms, 4 K_t store, 5 everything else / timer overhead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "../quantum-optimal-control_amd/csrc/qoc_mfma_frag.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NT = 2;
#define NPH 8

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

// mm_full4 without the final combine (returns the three accumulator sets)
__device__ __forceinline__ void mm_acc(const cplx* img, const double* imgs, int lane, const CTile (&p)[NT][NT], double (&a)[NT][QQS], double (&b)[NT][QQS], double (&c)[NT][QQS]) {
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int s = 0; s < QQS; ++s) { a[J][s] = 0.0; b[J][s] = 0.0; c[J][s] = 0.0; }
    const cplx* base = img + (lane >> 4) * QLDR + (lane & 3);
    const double* bases = imgs + (lane >> 4) * QLDR + (lane & 3);
    constexpr int NS = QQS * QQS;
    cplx vb[3]; double sb[3];
    auto fetch = [&](int st, int slot) {
        const int kb = st / QQS, ib = st % QQS;
        vb[slot] = base[4 * kb * QLDR + 4 * ib];
        sb[slot] = bases[4 * kb * QLDR + 4 * ib];
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
}

template <bool TIMED>
__global__ void __launch_bounds__(64, 1) k_timed(QocDev d, QocMfma mf, unsigned long long* phase) {
    __shared__ __attribute__((aligned(16))) cplx img[QNP * QLDR];
    __shared__ __attribute__((aligned(16))) double imgs[QNP * QLDR];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    unsigned long long acc_t[NPH] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tl = TIMED ? now() : 0;
    auto lap = [&](int ph) { if (TIMED) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = now(); acc_t[ph] += t - tl; tl = t; } };
    auto lap_nodrain = [&](int ph) { if (TIMED) { const unsigned long long t = now(); acc_t[ph] += t - tl; tl = t; } };
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
        lap_nodrain(6);
        wave_lds_fence();
    };
    double a[NT][QQS], bb[NT][QQS], cc[NT][QQS];
    auto combine = [&](CTile (&out)[NT][NT]) {
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int s = 0; s < QQS; ++s) { out[J][s >> 2].re[s & 3] = a[J][s] - bb[J][s]; out[J][s >> 2].im[s & 3] = cc[J][s] - a[J][s] - bb[J][s]; }
    };
    lap(5);
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
        lap(0);
        CTile AJ[NT][NT], A2J[NT][NT];
#pragma unroll
        for (int J = 0; J < NT; ++J)
            for (int Ib = 0; Ib < NT; ++Ib) AJ[J][Ib] = P[J][Ib];
        put_all(AJ);
        lap(1);
        mm_acc(img, imgs, lane, AJ, a, bb, cc);
        lap(2);
        combine(A2J);
        lap(3);
        wave_lds_fence();
        put_all(A2J);
        lap(1);
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
        lap(3);
        for (; i >= 0; --i) {
            CTile acc[NT][NT];
            mm_acc(img, imgs, lane, P, a, bb, cc);
            lap(2);
            combine(acc);
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
            lap(3);
        }
        wave_lds_fence();
        for (int sq = 0; sq < d.s; ++sq) {
            put_all(P);
            lap(1);
            mm_acc(img, imgs, lane, P, a, bb, cc);
            lap(2);
            wave_lds_fence();
            combine(P);
            lap(3);
        }
        const size_t item = kitem(mf, d.steps, b, t);
#pragma unroll
        for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.KfD + item, J, lane, P[J]);
        lap(4);
        put_all(P);
        lap(1);
        mm_acc(img, imgs, lane, R, a, bb, cc);
        lap(2);
        wave_lds_fence();
        combine(R);
        lap(3);
    }
    const size_t pitem = (size_t)b * mf.C + c;
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R[J]);
    put_all(R);
#pragma unroll
    for (int J = 0; J < NT; ++J) lds_store_fragT_half<NT>(img, mf.PfT + pitem * QFR, J, lane);
    lap(5);
    if (TIMED && lane == 0) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) atomicAdd(&phase[ph], acc_t[ph]);
    }
}

int main() {
    const int B = 64, steps = 500, k = 4, C = 16, L = 32;
    QocDev d;
    memset(&d, 0, sizeof d);
    d.n = 32; d.k = k; d.steps = steps; d.m = 8; d.T = 5; d.s = 3; d.B = B;
    QocMfma mf;
    mf.C = C; mf.L = L; mf.NT = 2; mf.FR = 1024; mf.store_T = false;
    { double f = 1.0; for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; mf.invfact[j] = 1.0 / f; } }
    mf.skew_c = 80; mf.skew_b = 48;
    std::vector<cplx> h((size_t)(k + 1) * 1024);
    srand(1);
    for (auto& v : h) { v.x = 0.02 * (rand() / (double)RAND_MAX - 0.5); v.y = 0.02 * (rand() / (double)RAND_MAX - 0.5); }
    std::vector<double> u((size_t)B * k * steps);
    for (auto& v : u) v = rand() / (double)RAND_MAX - 0.5;
    double* du; unsigned long long* ph;
    CHECK(hipMalloc((void**)&mf.HfD, h.size() * sizeof(cplx)));
    CHECK(hipMemcpy(mf.HfD, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&du, u.size() * sizeof(double)));
    CHECK(hipMemcpy(du, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice));
    d.u = du;
    const size_t nk = (size_t)B * ((size_t)steps * 1024 + (size_t)C * mf.skew_c + mf.skew_b);
    CHECK(hipMalloc((void**)&mf.KfD, nk * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.PfD, (size_t)B * C * 1024 * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.PfT, (size_t)B * C * 1024 * sizeof(cplx)));
    CHECK(hipMalloc((void**)&ph, NPH * sizeof(unsigned long long)));
    CHECK(hipMemset(ph, 0, NPH * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int timed = 0; timed < 2; ++timed) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipMemset(ph, 0, NPH * sizeof(unsigned long long)));
            CHECK(hipEventRecord(e0));
            if (timed) hipLaunchKernelGGL(k_timed<true>, dim3(B * C), dim3(64), 0, 0, d, mf, ph);
            else hipLaunchKernelGGL(k_timed<false>, dim3(B * C), dim3(64), 0, 0, d, mf, ph);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%s kernel: %.3f ms per launch\n", timed ? "timed  " : "untimed", best);
    }
    unsigned long long hp[NPH];
    CHECK(hipMemcpy(hp, ph, sizeof hp, hipMemcpyDeviceToHost));
    const char* names[NPH] = {"assemble A_t", "image writes: fence wait", "MFMA product loop", "combine + Horner terms", "K_t store", "prologue / tail", "image writes: v_add + ds_write issue", "-"};
    double tot = 0;
    for (int i = 0; i < NPH; ++i) tot += (double)hp[i];
    const double waves = B * C, slices = (double)L;     // last chunk is shorter (500 = 15*32 + 20): averaged over waves anyway
    for (int i = 0; i < NPH; ++i)
        printf("%-34s %6.1f %%   %10.0f clock ticks per wave, %8.0f per slice\n", names[i], 100.0 * hp[i] / tot, hp[i] / waves, hp[i] / waves / slices);
    printf("clock ticks per wave total %.0f (s_memtime units)\n", tot / waves);
    return 0;
}
