// slice_phase_probe.hip -- where do the 25 us of k_mfma_expm_slice2 go?  The kernel itself, compiled with its QOC_LAP hooks reading
// the shader clock (wave 0 of every workgroup), launched as the latency mode launches it: one C2 trajectory, 500 workgroups of two
// waves.  Random inputs, timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define QOC_NPH 5
#ifndef QOC_PROBE_UNTIMED
#define QOC_PROBE_TIMED
#endif
#ifdef QOC_PROBE_TIMED
__device__ unsigned long long g_phase[QOC_NPH];
__device__ unsigned long long g_first_start, g_last_end;
#define QOC_LAP_INIT unsigned long long lap_acc[QOC_NPH] = {0, 0, 0, 0, 0}; const unsigned long long lap_t0 = wall_clock64(); unsigned long long lap_last = __builtin_readcyclecounter();
#define QOC_LAP(ph) { __builtin_amdgcn_sched_barrier(0); const unsigned long long lap_now = __builtin_readcyclecounter(); lap_acc[ph] += lap_now - lap_last; lap_last = lap_now; __builtin_amdgcn_sched_barrier(0); }
#define QOC_LAP_DONE if (threadIdx.x == 0) { for (int ph_ = 0; ph_ < QOC_NPH; ++ph_) atomicAdd(&g_phase[ph_], lap_acc[ph_]); atomicMin(&g_first_start, lap_t0); atomicMax(&g_last_end, wall_clock64()); }
#endif
#include "../quantum-optimal-control_amd/csrc/qoc_mfma_expm_stream.h"
#ifndef QOC_PROBE_TIMED
__device__ unsigned long long g_phase[QOC_NPH];
__device__ unsigned long long g_first_start, g_last_end;
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spacer(double* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] += 1.0; }

int main() {
    const int B = 1, steps = 500, k = 4, C = 63, L = 8;
    QocDev d;
    memset(&d, 0, sizeof d);
    d.n = 32; d.k = k; d.steps = steps; d.m = 8; d.T = 5; d.s = 3; d.B = B;
    QocMfma mf;
    mf.C = C; mf.L = L; mf.NT = 2; mf.FR = 1024; mf.store_T = true; mf.latency = true;
    { double f = 1.0; for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; mf.invfact[j] = 1.0 / f; } }
    mf.skew_c = 80; mf.skew_b = 48;
    std::vector<cplx> h((size_t)(k + 1) * 1024);
    srand(1);
    for (auto& v : h) { v.x = 0.02 * (rand() / (double)RAND_MAX - 0.5); v.y = 0.02 * (rand() / (double)RAND_MAX - 0.5); }
    std::vector<double> base((size_t)B * k * steps), maxA(k, 1.0);
    for (auto& v : base) v = rand() / (double)RAND_MAX - 0.5;
    double *dbase, *dmaxA, *dw, *du, *spacer;
    CHECK(hipMalloc((void**)&mf.HfD, h.size() * sizeof(cplx)));
    CHECK(hipMemcpy(mf.HfD, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&dbase, base.size() * sizeof(double)));
    CHECK(hipMemcpy(dbase, base.data(), base.size() * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&dmaxA, k * sizeof(double)));
    CHECK(hipMemcpy(dmaxA, maxA.data(), k * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&dw, base.size() * sizeof(double)));
    CHECK(hipMalloc((void**)&du, base.size() * sizeof(double)));
    CHECK(hipMalloc((void**)&spacer, (1 << 20) * sizeof(double)));
    d.base = dbase; d.maxA = dmaxA; d.w = dw; d.u = du;
    const size_t nk = (size_t)B * ((size_t)steps * 1024 + (size_t)C * mf.skew_c + mf.skew_b);
    CHECK(hipMalloc((void**)&mf.KfD, nk * sizeof(cplx)));
    CHECK(hipMalloc((void**)&mf.KfT, nk * sizeof(cplx)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long zero[QOC_NPH] = {0}, big = ~0ull, nul = 0;
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero));
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_first_start), &big, sizeof big));
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_last_end), &nul, sizeof nul));
        hipLaunchKernelGGL(k_spacer, dim3(256), dim3(256), 0, 0, spacer, 1 << 20);        // another kernel in between, as in the engine
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mfma_expm_slice2<4>, dim3(B * steps), dim3(128), 0, 0, d, mf);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("k_mfma_expm_slice2<4>, %d workgroups: %.1f us per launch (hipEvents, best of 6)\n", B * steps, best * 1e3);
#ifndef QOC_PROBE_TIMED
    return 0;
#endif
    unsigned long long hp[QOC_NPH], fs, le;
    CHECK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_phase), sizeof hp));
    CHECK(hipMemcpyFromSymbol(&fs, HIP_SYMBOL(g_first_start), sizeof fs));
    CHECK(hipMemcpyFromSymbol(&le, HIP_SYMBOL(g_last_end), sizeof le));
    const char* names[QOC_NPH] = {"loads, controls, assembly of A_t", "publish (LDS stores + barrier) x 6", "products (192 MFMAs each) x 6", "epilogues (combine, Horner terms)", "K_t out (fragD + transposed copy)"};
    double tot = 0;
    for (int i = 0; i < QOC_NPH; ++i) tot += (double)hp[i];
    const double wgs = B * steps;
    for (int i = 0; i < QOC_NPH; ++i) printf("%-40s %6.1f %%   %9.0f shader-clock ticks per workgroup\n", names[i], 100.0 * hp[i] / tot, hp[i] / wgs);
    printf("ticks per workgroup total %.0f; first wave start .. last wave end %.2f us (100 MHz wall clock)\n", tot / wgs, (double)(le - fs) / 100.0);
    return 0;
}
