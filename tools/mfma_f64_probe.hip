// mfma_f64_probe.hip -- verifies the v_mfma_f64_16x16x4_f64 fragment layout assumed by qoc_kernels_mfma.h and
// measures its issue rate and the v_fma_f64 rate on the box it runs on.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D) {  // A 16x4 row-major, B 4x16 row-major
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) k_rate(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_cycles(double* out, long long* cyc, int iters) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_fma(double* out, int iters) {
    double x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// accumulators pinned to AGPRs (inline asm, "+a"): does the SrcC/D register file matter for the issue interval?
template <int NACC>
__global__ void __launch_bounds__(256) k_rate_agpr(double* out, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_mfma_f64_4x4x4_4b_f64: 4 blocks of 4x4x4, one double accumulator per lane (512 flops per instruction)
template <int NACC>
__global__ void __launch_bounds__(256) k_rate_4x4(double* out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MFMA and independent VALU FMA chains in the same wave: NF v_fma_f64 per MFMA.  If the FMAs are free next to the MFMAs the
// VALU can carry part of a product.
template <int NF>
__global__ void __launch_bounds__(256) k_rate_mixed(double* out, int iters) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double x[NF > 0 ? NF : 1];
    for (int i = 0; i < NF; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const double fa = 1.0000001, fb = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) x[f] = fma(x[f], fa, fb);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int f = 0; f < NF; ++f) s += x[f];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16x16x4 and 4x4x4 MFMAs interleaved in one wave: N4 small ones per big one
template <int N4>
__global__ void __launch_bounds__(256) k_rate_both(double* out, int iters) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
    double sm[4 * (N4 > 0 ? N4 : 1)];
    for (int i = 0; i < 4 * N4; ++i) sm[i] = 0.0;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < N4; ++f) sm[i * N4 + f] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, sm[i * N4 + f], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4 * N4; ++i) s += sm[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s %s CUs=%d clock=%d kHz mem=%.1f GB\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.totalGlobalMem / 1e9);
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = sin(1.0 + i); B[i] = cos(2.0 + 3 * i); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("LAYOUT max|D-ref| = %.3e  -> %s\n", err, err < 1e-14 ? "LAYOUT_OK (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15)" : "LAYOUT_MISMATCH");

    const int blocks = prop.multiProcessorCount, iters = 20000;
    double* dout; CK(hipMalloc(&dout, (size_t)blocks * 8 * 256 * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto bench = [&](const char* name, auto launch, double flop_per_thread_block_iter, int nblk, int nthr) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = flop_per_thread_block_iter * (double)iters * nblk;
        printf("%-34s %8.3f ms  %8.2f TFLOP/s\n", name, ms, flops / ms * 1e-9);
        return ms;
    };
    // one wave per SIMD (256 threads/block, 1 block/CU)
    float ms4 = bench("mfma_f64 4acc 1wave/SIMD", [&] { hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0, blocks, 256);
    bench("mfma_f64 8acc 1wave/SIMD", [&] { hipLaunchKernelGGL(k_rate<8>, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 8 * 4 * 2048.0, blocks, 256);
    bench("mfma_f64 1acc 1wave/SIMD (dependent)", [&] { hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 1 * 4 * 2048.0, blocks, 256);
    bench("mfma_f64 2acc 1wave/SIMD", [&] { hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 2 * 4 * 2048.0, blocks, 256);
    bench("mfma_f64 4acc 2waves/SIMD", [&] { hipLaunchKernelGGL(k_rate<4>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0, blocks * 2, 256);
    bench("mfma_f64 4acc 4waves/SIMD", [&] { hipLaunchKernelGGL(k_rate<4>, dim3(blocks * 4), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0, blocks * 4, 256);
    bench("mfma_f64 2acc 8waves/SIMD", [&] { hipLaunchKernelGGL(k_rate<2>, dim3(blocks * 8), dim3(256), 0, 0, dout, iters); }, 2 * 4 * 2048.0, blocks * 8, 256);
    bench("mfma_f64 AGPR acc 4acc 1wave/SIMD", [&] { hipLaunchKernelGGL(k_rate_agpr<4>, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0, blocks, 256);
    bench("mfma_f64 AGPR acc 4acc 2waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_agpr<4>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0, blocks * 2, 256);
    bench("mfma_f64 AGPR acc 4acc 4waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_agpr<4>, dim3(blocks * 4), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0, blocks * 4, 256);
    bench("mfma_f64_4x4x4 8acc 1wave/SIMD", [&] { hipLaunchKernelGGL(k_rate_4x4<8>, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 8 * 4 * 512.0, blocks, 256);
    bench("mfma_f64_4x4x4 8acc 4waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_4x4<8>, dim3(blocks * 4), dim3(256), 0, 0, dout, iters); }, 8 * 4 * 512.0, blocks * 4, 256);
    // mixed: flops counted = MFMA flops + FMA flops (2 per lane per FMA: 4 MFMA * NF FMAs * 256 threads * 2)
    bench("mixed 4 MFMA + 4x4 FMA 2waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_mixed<4>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0 + 4 * 4 * 256 * 2.0, blocks * 2, 256);
    bench("mixed 4 MFMA + 4x8 FMA 2waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_mixed<8>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0 + 4 * 8 * 256 * 2.0, blocks * 2, 256);
    bench("mixed 4 MFMA + 4x16 FMA 2waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_mixed<16>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0 + 4 * 16 * 256 * 2.0, blocks * 2, 256);
    bench("mixed 4 MFMA + 4x16 FMA 4waves/SIMD", [&] { hipLaunchKernelGGL(k_rate_mixed<16>, dim3(blocks * 4), dim3(256), 0, 0, dout, iters); }, 4 * 4 * 2048.0 + 4 * 16 * 256 * 2.0, blocks * 4, 256);
    bench("16x16x4 + 1x 4x4x4 interleaved 2w/SIMD", [&] { hipLaunchKernelGGL(k_rate_both<1>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * (2048.0 + 1 * 512.0), blocks * 2, 256);
    bench("16x16x4 + 2x 4x4x4 interleaved 2w/SIMD", [&] { hipLaunchKernelGGL(k_rate_both<2>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * (2048.0 + 2 * 512.0), blocks * 2, 256);
    bench("16x16x4 + 4x 4x4x4 interleaved 2w/SIMD", [&] { hipLaunchKernelGGL(k_rate_both<4>, dim3(blocks * 2), dim3(256), 0, 0, dout, iters); }, 4 * 4 * (2048.0 + 4 * 512.0), blocks * 2, 256);
    {   // in-kernel shader-clock cycles per MFMA (s_memtime ticks = shader cycles) and effective clock
        long long* dcyc; hipMalloc(&dcyc, sizeof(long long) * 2);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_cycles, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long hc[2]; hipMemcpy(hc, dcyc, sizeof hc, hipMemcpyDeviceToHost);
        printf("in-kernel: %.1f memtime ticks per MFMA (4 acc, 1 wave/SIMD); ticks/s = %.3f GHz over %.3f ms\n",
               (double)hc[0] / (iters * 4.0), (double)hc[0] / (ms * 1e-3) * 1e-9, ms);
    }
    bench("v_fma_f64 16chains 1wave/SIMD", [&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, dout, iters); }, 16 * 256 * 2.0, blocks, 256);
    bench("v_fma_f64 16chains 4waves/SIMD", [&] { hipLaunchKernelGGL(k_fma, dim3(blocks * 4), dim3(256), 0, 0, dout, iters); }, 16 * 256 * 2.0, blocks * 4, 256);
    // cycles per MFMA on one SIMD assuming the reported clock
    double cyc = (double)ms4 * 1e-3 * prop.clockRate * 1e3 / ((double)iters * 4);
    printf("approx cycles per v_mfma_f64_16x16x4 (4 acc, 1 wave/SIMD, at %d MHz) = %.1f\n", prop.clockRate / 1000, cyc);
    return err < 1e-14 ? 0 : 2;
}
