// mfma_cover_probe.hip -- what can one wave issue in the shadow of back-to-back v_mfma_f64_4x4x4_4b_f64?
//
// One wave per SIMD (1024 waves, 256 workgroups x 4 waves, LDS-limited to one workgroup per CU) runs a loop of 8 independent
// MFMA chains; after every MFMA it issues NS "side" instructions of one kind that depend on nothing in the MFMA stream.
// Reported: ns per MFMA (16 pipe cycles = 6.7 ns at 2.4 GHz when the side work is free).  Decides whether the per-product
// epilogue of k_mfma_expm_chunk4w (fp64 VALU combines, LDS image writes) can be software-pipelined under the next product's MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2v __attribute__((ext_vector_type(2)));
enum { NONE = 0, FMA64 = 1, ADD64 = 2, MOV32 = 3, DSW128 = 4, DSR128 = 5, FMA32 = 6, DSW64 = 7, GLD128 = 8, BPERM = 9 };

template <int KIND>
__device__ __forceinline__ void side(double& x, double y, double z, float& f, unsigned lds, d2v& lv, const double2* g, d2v& gv) {
    if constexpr (KIND == FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
    else if constexpr (KIND == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    else if constexpr (KIND == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(f) : "v"(f));
    else if constexpr (KIND == FMA32) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f) : "v"(f));
    else if constexpr (KIND == DSW128) asm volatile("ds_write_b128 %0, %1" :: "v"(lds), "v"(lv) : "memory");
    else if constexpr (KIND == DSW64) asm volatile("ds_write_b64 %0, %1" :: "v"(lds), "v"(x) : "memory");
    else if constexpr (KIND == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(lv) : "v"(lds) : "memory");
    else if constexpr (KIND == BPERM) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(f) : "v"(lds), "v"(lds) : "memory");
    else if constexpr (KIND == GLD128) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv) : "v"(g) : "memory");
}

template <int KIND, int NS, int PER>    // NS side instructions after every PER-th MFMA
__global__ void __launch_bounds__(256, 1) k_probe(double* out, const double2* gsrc, int iters, int data = 1) {
    extern __shared__ double2 smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned lds = (unsigned)(wv * 1024 + lane * 2) * 16u;   // byte offset in the dynamic LDS segment (no static LDS): conflict-free 16 B per lane
    if (iters < 0) smem[threadIdx.x] = make_double2(0.0, 0.0);
    const double2* g = gsrc + (size_t)(blockIdx.x * 256 + threadIdx.x);
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    if (data == 0) { a = 0.0; b = 0.0; }                                   // operand data: 0 = zeros, 1 = smooth, 2 = random mantissas
    if (data == 2) {
        unsigned long long x = 0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * 256 + threadIdx.x + 1);
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        a = __longlong_as_double((long long)((x & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)) - 1.5;
        x *= 0x94D049BB133111EBull; x ^= x >> 31;
        b = __longlong_as_double((long long)((x & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull)) - 1.5;
    }
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    double x[4] = {1.0, 2.0, 3.0, 4.0};
    float f = 1.0f;
    d2v lv = {1.0, 2.0}, gv = {0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                if ((rep * 8 + i) % PER == 0) {
#pragma unroll
                    for (int sdx = 0; sdx < NS; ++sdx) side<KIND>(x[sdx & 3], a, b, f, lds, lv, g, gv);
                }
            }
        if constexpr (KIND == DSW128 || KIND == DSR128 || KIND == DSW64 || KIND == BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == GLD128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    s += x[0] + x[1] + x[2] + x[3] + f + lv.x + lv.y + gv.x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NS, int PER>
static void run(const char* name, double* out, const double2* gsrc, int data = 1) {
    const int iters = 200000;
    const size_t lds = 100 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_probe<KIND, NS, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_probe<KIND, NS, PER>), dim3(256), dim3(256), lds, 0, out, gsrc, 200, data);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_probe<KIND, NS, PER>), dim3(256), dim3(256), lds, 0, out, gsrc, iters, data);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmas = (double)iters * 32;
    const double ns = ms * 1e6 / mfmas;
    printf("%-44s %7.2f ns per MFMA  (%5.1f TFLOP/s executed; side instr per MFMA %.2f)\n", name, ns,
           1024.0 * 512.0 / ns / 1e3, (double)NS / PER);
}

int main() {
    double* out;
    double2* gsrc;
    CHECK(hipMalloc((void**)&out, 256 * 256 * sizeof(double)));
    CHECK(hipMalloc((void**)&gsrc, 256 * 256 * sizeof(double2)));
    CHECK(hipMemset(gsrc, 0, 256 * 256 * sizeof(double2)));
    run<NONE, 0, 1>("MFMA only, operands all zero", out, gsrc, 0);
    run<NONE, 0, 1>("MFMA only, random mantissas", out, gsrc, 2);
    run<NONE, 0, 1>("MFMA only", out, gsrc);
    run<FMA64, 1, 1>("+1 v_fma_f64 per MFMA", out, gsrc);
    run<FMA64, 2, 1>("+2 v_fma_f64 per MFMA", out, gsrc);
    run<FMA64, 3, 1>("+3 v_fma_f64 per MFMA", out, gsrc);
    run<FMA64, 1, 2>("+1 v_fma_f64 per 2 MFMA", out, gsrc);
    run<ADD64, 1, 1>("+1 v_add_f64 per MFMA", out, gsrc);
    run<ADD64, 2, 1>("+2 v_add_f64 per MFMA", out, gsrc);
    run<MOV32, 1, 1>("+1 v_mov_b32 per MFMA", out, gsrc);
    run<MOV32, 3, 1>("+3 v_mov_b32 per MFMA", out, gsrc);
    run<FMA32, 1, 1>("+1 v_fma_f32 per MFMA", out, gsrc);
    run<FMA32, 3, 1>("+3 v_fma_f32 per MFMA", out, gsrc);
    run<DSW128, 1, 1>("+1 ds_write_b128 per MFMA", out, gsrc);
    run<DSW128, 1, 2>("+1 ds_write_b128 per 2 MFMA", out, gsrc);
    run<DSW128, 1, 4>("+1 ds_write_b128 per 4 MFMA", out, gsrc);
    run<DSW64, 1, 2>("+1 ds_write_b64 per 2 MFMA", out, gsrc);
    run<DSR128, 1, 1>("+1 ds_read_b128 per MFMA", out, gsrc);
    run<DSR128, 1, 2>("+1 ds_read_b128 per 2 MFMA", out, gsrc);
    run<DSR128, 1, 4>("+1 ds_read_b128 per 4 MFMA", out, gsrc);
    run<BPERM, 1, 1>("+1 ds_bpermute_b32 per MFMA", out, gsrc);
    run<BPERM, 2, 1>("+2 ds_bpermute_b32 per MFMA", out, gsrc);
    run<BPERM, 1, 2>("+1 ds_bpermute_b32 per 2 MFMA", out, gsrc);
    run<GLD128, 1, 2>("+1 global_load_dwordx4 per 2 MFMA", out, gsrc);
    run<GLD128, 1, 4>("+1 global_load_dwordx4 per 4 MFMA", out, gsrc);
    return 0;
}
