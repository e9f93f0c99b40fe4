// rows_phase_probe.hip -- phase timing of k_mfma_expm_rows<3, 4> (n = 48, T = 5, s = 3, k = 4; 64 seeds x 16 chunks), the kernel itself
// with its QOC_LAP hooks reading the shader clock (wave 0 of every workgroup).  Random inputs, timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define QOC_NPH 5
__device__ unsigned long long g_phase[QOC_NPH];
#define QOC_LAP_INIT unsigned long long lap_acc[QOC_NPH] = {0, 0, 0, 0, 0}; unsigned long long lap_last = __builtin_readcyclecounter();
#define QOC_LAP(ph) { __builtin_amdgcn_sched_barrier(0); const unsigned long long lap_now = __builtin_readcyclecounter(); lap_acc[ph] += lap_now - lap_last; lap_last = lap_now; __builtin_amdgcn_sched_barrier(0); }
#define QOC_LAP_DONE if (threadIdx.x == 0) { for (int ph_ = 0; ph_ < QOC_NPH; ++ph_) atomicAdd(&g_phase[ph_], lap_acc[ph_]); }
#include "../quantum-optimal-control_amd/csrc/qoc_mfma_expm_rows.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
#ifndef ROWS_NT
#define ROWS_NT 3
#endif
    constexpr int NT = ROWS_NT;
    const int B = 64, steps = 500, k = 4, C = 16, L = 32, FR = 256 * NT * NT;
    QocDev d;
    memset(&d, 0, sizeof d);
    d.n = 16 * NT; d.k = k; d.steps = steps; d.m = 8; d.T = 5; d.s = 3; d.B = B;
    QocMfma mf;
    mf.C = C; mf.L = L; mf.NT = NT; mf.FR = FR; mf.store_T = false;
    { double f = 1.0; for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; mf.invfact[j] = 1.0 / f; } }
    mf.skew_c = 80; mf.skew_b = 48;
    std::vector<cplx> h((size_t)(k + 1) * FR);
    srand(1);
    for (auto& v : h) { v.x = 0.02 * (rand() / (double)RAND_MAX - 0.5); v.y = 0.02 * (rand() / (double)RAND_MAX - 0.5); }
    std::vector<double> u((size_t)B * k * steps);
    for (auto& v : u) v = rand() / (double)RAND_MAX - 0.5;
    double* du;
    CHECK(hipMalloc((void**)&mf.HfD, h.size() * sizeof(cplx)));
    CHECK(hipMemcpy(mf.HfD, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&du, u.size() * sizeof(double)));
    CHECK(hipMemcpy(du, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice));
    d.u = du;
    const size_t nk = (size_t)B * ((size_t)steps * FR + (size_t)C * mf.skew_c + mf.skew_b);
    CHECK(hipMalloc((void**)&mf.KfD, nk * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.PfD, (size_t)B * C * FR * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.PfT, (size_t)B * C * FR * sizeof(cplx)));
    const size_t lds = qoc_expm_rows_lds<NT>();
    CHECK(hipFuncSetAttribute((const void*)k_mfma_expm_rows<NT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long zero[QOC_NPH] = {0};
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mfma_expm_rows<NT, 4>), dim3(B * C), dim3(256), lds, 0, d, mf);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("k_mfma_expm_rows<%d, 4>, n = %d, %zu B of LDS per workgroup: %.3f ms per launch (with the clock hooks)\n", NT, 16 * NT, lds, best);
    unsigned long long hp[QOC_NPH];
    CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_phase), sizeof hp));
    const char* names[QOC_NPH] = {"assembly of A_t", "publishes (LDS stores + barriers)", "products", "epilogues, K_t store", "P_c out"};
    double tot = 0;
    for (int i = 0; i < QOC_NPH; ++i) tot += (double)hp[i];
    const double wgs = B * C, slices = (double)L;
    for (int i = 0; i < QOC_NPH; ++i) printf("%-36s %6.1f %%   %9.0f ticks per slice (wave 0)\n", names[i], 100.0 * hp[i] / tot, hp[i] / wgs / slices);
    printf("ticks per slice total %.0f; 7 products of %d MFMAs x 17 cycles = %d pipe cycles per wave\n", tot / wgs / slices, 36 * NT * NT, 7 * 36 * NT * NT * 17);
    return 0;
}
