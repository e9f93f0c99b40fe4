// mfma_two_wave_probe.hip -- model of a 2-waves-per-SIMD exponential kernel: every wave alternates a burst of NM MFMAs (one
// ds_read_b128 per 3 MFMAs) with an epilogue of NV VALU instructions and NW ds_write_b128; the two waves of a SIMD are independent
// (no barrier).  Reported: aggregate ns per MFMA per SIMD, for 1 and 2 waves per SIMD.  Pure MFMA stream = 6.8-7.1 ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));

template <int NV, int NW>
__global__ void __launch_bounds__(512, 1) k_two(double* out, unsigned long long* ticks, int iters, int nwaves) {
    extern __shared__ double2 smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv >= nwaves) return;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    double x[4] = {1.0, 2.0, 3.0, 4.0};
    d2v lv = {1.0, 2.0}, rv;
    const unsigned lds = (unsigned)(wv * 1024 + lane * 2) * 16u + 64u;
    // desynchronise the two waves of a SIMD by half a period
    if (wv >= 4) {
#pragma unroll
        for (int r = 0; r < 192; ++r) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[r & 7]) : "v"(a), "v"(b));
    }
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 128; ++g) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(rv) : "v"(lds) : "memory");
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[(3 * g) & 7]) : "v"(a), "v"(b));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[(3 * g + 1) & 7]) : "v"(a), "v"(b));
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[(3 * g + 2) & 7]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int r = 0; r < NV; ++r) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[r & 3]) : "v"(a));
#pragma unroll
        for (int r = 0; r < NW; ++r) asm volatile("ds_write_b128 %0, %1" :: "v"(lds), "v"(lv) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = wall_clock64();
    double s = x[0] + x[1] + x[2] + x[3] + rv.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int NV, int NW>
static void run(double* out, unsigned long long* ticks) {
    const int iters = 3000;
    const size_t lds = 100 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_two<NV, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int nw = 4; nw <= 8; nw += 4) {
        CHECK(hipMemset(ticks, 0, 2048 * sizeof(unsigned long long)));
        hipLaunchKernelGGL((k_two<NV, NW>), dim3(256), dim3(512), lds, 0, out, ticks, 100, nw);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL((k_two<NV, NW>), dim3(256), dim3(512), lds, 0, out, ticks, iters, nw);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[2048];
        CHECK(hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost));
        double mx = 0;
        for (int i = 0; i < 2048; ++i) if ((double)h[i] > mx) mx = (double)h[i];
        const double mfma_per_simd = (double)iters * 384 * (nw / 4);
        printf("epilogue %3d v_add_f64 + %2d ds_write_b128: %d wave(s) per SIMD -> %6.2f ns per MFMA per SIMD\n", NV, NW, nw / 4, mx * 10.0 / mfma_per_simd);
    }
}

int main() {
    double* out; unsigned long long* ticks;
    CHECK(hipMalloc((void**)&out, 256 * 512 * sizeof(double)));
    CHECK(hipMalloc((void**)&ticks, 2048 * sizeof(unsigned long long)));
    run<0, 0>(out, ticks);
    run<200, 0>(out, ticks);
    run<200, 16>(out, ticks);
    run<400, 24>(out, ticks);
    return 0;
}
