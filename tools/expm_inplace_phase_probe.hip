// expm_inplace_phase_probe.hip -- phase timing of k_mfma_expm_inplace (the product kernel itself, compiled with its QOC_LAP hooks
// reading the shader clock): C2 shape (n = 32, T = 5, s = 3, k = 4, 64 seeds x 16 chunks), random inputs, timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define QOC_NPH 10
#ifdef QOC_PROBE_TIMED
__device__ unsigned long long g_phase[QOC_NPH];
#define QOC_LAP_INIT unsigned long long lap_acc[QOC_NPH] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long lap_last = __builtin_readcyclecounter();
#define QOC_LAP(ph) { __builtin_amdgcn_sched_barrier(0); const unsigned long long lap_now = __builtin_readcyclecounter(); lap_acc[ph] += lap_now - lap_last; lap_last = lap_now; __builtin_amdgcn_sched_barrier(0); }
#define QOC_LAP_DONE if (threadIdx.x == 0) { for (int ph_ = 0; ph_ < QOC_NPH; ++ph_) atomicAdd(&g_phase[ph_], lap_acc[ph_]); }
#endif
#include "../quantum-optimal-control_amd/csrc/qoc_mfma_expm_inplace.h"
#ifndef QOC_PROBE_TIMED
__device__ unsigned long long g_phase[QOC_NPH];
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

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
    double* du;
    CHECK(hipMalloc((void**)&mf.HfD, h.size() * sizeof(cplx)));
    CHECK(hipMemcpy(mf.HfD, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice));
    mf.HsD = mf.HfD;                                                   // timing only: the scaled copy of the kernel's assembly = the same random matrices
    for (int j = 0; j < 24; ++j) mf.pcoef[j] = mf.invfact[j];
    CHECK(hipMalloc((void**)&du, u.size() * sizeof(double)));
    CHECK(hipMemcpy(du, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice));
    d.u = du;
    const size_t nk = (size_t)B * ((size_t)steps * 1024 + (size_t)C * mf.skew_c + mf.skew_b);
    CHECK(hipMalloc((void**)&mf.KfD, nk * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.PfD, (size_t)B * C * 1024 * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.PfT, (size_t)B * C * 1024 * sizeof(cplx)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long zero[QOC_NPH] = {0};
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mfma_expm_inplace<4, false, false>), dim3(B * C), dim3(64), 0, 0, d, mf);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("k_mfma_expm_inplace: %.3f ms per launch\n", best);
#ifndef QOC_PROBE_TIMED
    return 0;
#endif
    unsigned long long hp[QOC_NPH];
    CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_phase), sizeof hp));
    const char* names[QOC_NPH] = {"first-slice assembly", "A*A product (+ X0)", "Horner products", "squaring products", "chunk product (+ next assembly)",
                                  "R copy, read-back of A", "tail", "-", "-", "-"};
    double tot = 0;
    for (int i = 0; i < QOC_NPH; ++i) tot += (double)hp[i];
    const double waves = B * C, slices = (double)L;
    for (int i = 0; i < QOC_NPH; ++i)
        printf("%-36s %6.1f %%   %10.0f ticks per wave, %8.0f per slice\n", names[i], 100.0 * hp[i] / tot, hp[i] / waves, hp[i] / waves / slices);
    printf("ticks per wave total %.0f; per slice: 2 Horner products, 3 squarings (T = 5, s = 3)\n", tot / waves);
    return 0;
}
