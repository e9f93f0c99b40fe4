// mfma_partner_probe.hip -- two waves per SIMD: does a PARTNER wave's VALU / LDS-store work overlap with a wave's fp64 MFMA stream?
// 512-thread workgroups (8 waves, one workgroup per CU): waves 0-3 run back-to-back v_mfma_f64_4x4x4_4b_f64 and time themselves
// (wall_clock64, 100 MHz), waves 4-7 -- one per SIMD, beside an MFMA wave -- spin on one kind of instruction until the MFMA
// waves are done.  Decides whether a 2-waves-per-SIMD exponential kernel could hide its VALU epilogues (combine, Horner, sums).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2v __attribute__((ext_vector_type(2)));
enum { IDLE = 0, FMA64 = 1, MOV32 = 2, ACCRD = 3, DSW128 = 4, MFMA2 = 5, ADD64 = 6, BOTH = 7 };

template <int KIND>
__global__ void __launch_bounds__(512, 1) k_partner(double* out, unsigned long long* ticks, int iters) {
    extern __shared__ double2 smem[];
    __shared__ int done;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    if (wv < 4 || KIND == BOTH) {
        double acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0;
        const unsigned long long t0 = wall_clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        const unsigned long long t1 = wall_clock64();
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) { if (wv < 4) ticks[blockIdx.x * 4 + wv] = t1 - t0; else ticks[1024 + blockIdx.x * 4 + wv - 4] = t1 - t0; atomicAdd(&done, 1); }
    } else {
        double x[4] = {1.0, 2.0, 3.0, 4.0};
        float f = 1.0f;
        d2v lv = {1.0, 2.0};
        double acc2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned lds = (unsigned)((wv - 4) * 1024 + lane * 2) * 16u + 64u;
        volatile int* dn = &done;
        while (*dn < 4) {
            if (KIND == IDLE) { __builtin_amdgcn_s_sleep(8); continue; }
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if constexpr (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[r & 3]) : "v"(a), "v"(b));
                else if constexpr (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[r & 3]) : "v"(a));
                else if constexpr (KIND == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(f));
                else if constexpr (KIND == ACCRD) asm volatile("v_accvgpr_write_b32 a0, %1\n\tv_accvgpr_read_b32 %0, a0" : "=v"(f) : "v"(f) : "a0");
                else if constexpr (KIND == DSW128) { if ((r & 3) == 0) asm volatile("ds_write_b128 %0, %1" :: "v"(lds), "v"(lv) : "memory"); else asm volatile("s_nop 7"); }
                else if constexpr (KIND == MFMA2) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc2[r & 7]) : "v"(a), "v"(b));
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = x[0] + x[1] + x[2] + x[3] + f + lv.x + acc2[0] + acc2[1] + acc2[2] + acc2[3] + acc2[4] + acc2[5] + acc2[6] + acc2[7];
    }
}

template <int KIND>
static void run(const char* name, double* out, unsigned long long* ticks) {
    const int iters = 100000;
    const size_t lds = 100 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_partner<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_partner<KIND>), dim3(256), dim3(512), lds, 0, out, ticks, 2000);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k_partner<KIND>), dim3(256), dim3(512), lds, 0, out, ticks, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2048];
    CHECK(hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost));
    if (KIND == BOTH) { double s2 = 0; for (int i = 1024; i < 2048; ++i) s2 += (double)h[i]; printf("   second waves: %6.2f ns per MFMA\n", s2 / 1024 * 10.0 / ((double)iters * 32)); }
    double sum = 0;
    for (int i = 0; i < 1024; ++i) sum += (double)h[i];
    const double ns = sum / 1024 * 10.0 / ((double)iters * 32);
    printf("partner wave: %-34s MFMA wave: %6.2f ns per MFMA (%5.1f TFLOP/s per chip at this rate)\n", name, ns, 1024.0 * 512.0 / ns / 1e3);
}

int main() {
    double* out; unsigned long long* ticks;
    CHECK(hipMalloc((void**)&out, 256 * 512 * sizeof(double)));
    CHECK(hipMalloc((void**)&ticks, 2048 * sizeof(unsigned long long)));
    run<IDLE>("asleep", out, ticks);
    run<FMA64>("v_fma_f64 back to back", out, ticks);
    run<ADD64>("v_add_f64 back to back", out, ticks);
    run<MOV32>("v_mov_b32 back to back", out, ticks);
    run<ACCRD>("v_accvgpr_write/read back to back", out, ticks);
    run<DSW128>("ds_write_b128 every ~40 cycles", out, ticks);
    run<MFMA2>("the same MFMA stream", out, ticks);
    run<BOTH>("the same timed MFMA loop (both waves do equal work)", out, ticks);
    return 0;
}
