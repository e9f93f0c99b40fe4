// mfma_f64_4x4_product.hip -- de-risks a 4x4x4-based complex product: one wave computes D(32x16) = A(32x32)*B(32x16) with the
// 3-multiplication form, A blocks read from a column-major padded LDS image with 4-lane broadcast, B strips in registers.
// Reports executed-MFMA TFLOP/s for 2 and 4 waves per SIMD (128-thread workgroups = 2 column blocks, like k_mfma_expm_chunk).
#include <hip/hip_runtime.h>
#include <cstdio>

struct cplx { double x, y; };
#define LD 36

__global__ void __launch_bounds__(128) k_prod(double* out, int iters) {
    __shared__ __attribute__((aligned(16))) cplx img[32 * LD];      // img[col * LD + row] = A[row][col]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int o = tid; o < 32 * LD; o += 128) { img[o].x = 1e-3 * (o % 7); img[o].y = 1e-3 * (o % 5); }
    __syncthreads();
    double br[8], bi[8], bs[8];
    for (int s = 0; s < 8; ++s) { br[s] = 1.0 + lane * 1e-6 + s; bi[s] = 0.5 - lane * 1e-6; bs[s] = br[s] + bi[s]; }
    double t1[8], t2[8], t3[8];
    for (int s = 0; s < 8; ++s) t1[s] = t2[s] = t3[s] = 0.0;
    const int k = lane >> 4, i = lane & 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                const cplx a = img[(4 * kb + k) * LD + 4 * ib + i];
                t1[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, br[kb], t1[ib], 0, 0, 0);
                t2[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, bi[kb], t2[ib], 0, 0, 0);
                t3[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x + a.y, bs[kb], t3[ib], 0, 0, 0);
            }
        }
        // feed the result back as the next right operand (keeps the chain honest)
#pragma unroll
        for (int s = 0; s < 8; ++s) { br[s] = (t1[s] - t2[s]) * 1e-3; bi[s] = (t3[s] - t1[s] - t2[s]) * 1e-3; bs[s] = br[s] + bi[s]; }
    }
    double sum = 0;
    for (int s = 0; s < 8; ++s) sum += br[s] + bi[s];
    out[blockIdx.x * 128 + tid] = sum;
}

// variant of k_prod whose loads stay inside their k-block (no parking of 64 blocks in AGPRs)
__global__ void __launch_bounds__(128) k_prod_fenced(double* out, int iters) {
    __shared__ __attribute__((aligned(16))) cplx img[32 * LD];      // img[col * LD + row] = A[row][col]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int o = tid; o < 32 * LD; o += 128) { img[o].x = 1e-3 * (o % 7); img[o].y = 1e-3 * (o % 5); }
    __syncthreads();
    double br[8], bi[8], bs[8];
    for (int s = 0; s < 8; ++s) { br[s] = 1.0 + lane * 1e-6 + s; bi[s] = 0.5 - lane * 1e-6; bs[s] = br[s] + bi[s]; }
    double t1[8], t2[8], t3[8];
    for (int s = 0; s < 8; ++s) t1[s] = t2[s] = t3[s] = 0.0;
    const int k = lane >> 4, i = lane & 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                const cplx a = img[(4 * kb + k) * LD + 4 * ib + i];
                t1[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, br[kb], t1[ib], 0, 0, 0);
                t2[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, bi[kb], t2[ib], 0, 0, 0);
                t3[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x + a.y, bs[kb], t3[ib], 0, 0, 0);
            }
            asm volatile("" ::: "memory");                      // loads of the next k-block may not be hoisted above this point
        }
        // feed the result back as the next right operand (keeps the chain honest)
#pragma unroll
        for (int s = 0; s < 8; ++s) { br[s] = (t1[s] - t2[s]) * 1e-3; bi[s] = (t3[s] - t1[s] - t2[s]) * 1e-3; bs[s] = br[s] + bi[s]; }
    }
    double sum = 0;
    for (int s = 0; s < 8; ++s) sum += br[s] + bi[s];
    out[blockIdx.x * 128 + tid] = sum;
}

// variant: one wave owns all 32 columns (two strip sets): 6 MFMAs per A-block load
__global__ void __launch_bounds__(64) k_prod32(double* out, int iters) {
    __shared__ __attribute__((aligned(16))) cplx img[32 * LD];
    const int lane = threadIdx.x;
    for (int o = lane; o < 32 * LD; o += 64) { img[o].x = 1e-3 * (o % 7); img[o].y = 1e-3 * (o % 5); }
    __syncthreads();
    double br[2][8], bi[2][8], bs[2][8], t1[2][8], t2[2][8], t3[2][8];
    for (int c = 0; c < 2; ++c)
        for (int s = 0; s < 8; ++s) { br[c][s] = 1.0 + lane * 1e-6 + s + c; bi[c][s] = 0.5 - lane * 1e-6; bs[c][s] = br[c][s] + bi[c][s]; t1[c][s] = t2[c][s] = t3[c][s] = 0.0; }
    const int k = lane >> 4, i = lane & 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                const cplx a = img[(4 * kb + k) * LD + 4 * ib + i];
                const double as = a.x + a.y;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    t1[c][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, br[c][kb], t1[c][ib], 0, 0, 0);
                    t2[c][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, bi[c][kb], t2[c][ib], 0, 0, 0);
                    t3[c][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(as, bs[c][kb], t3[c][ib], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 8; ++s) { br[c][s] = (t1[c][s] - t2[c][s]) * 1e-3; bi[c][s] = (t3[c][s] - t1[c][s] - t2[c][s]) * 1e-3; bs[c][s] = br[c][s] + bi[c][s]; }
    }
    double sum = 0;
    for (int c = 0; c < 2; ++c) for (int s = 0; s < 8; ++s) sum += br[c][s] + bi[c][s];
    out[blockIdx.x * 64 + lane] = sum;
}

// variant: A kept in registers (no LDS in the loop): the pure issue limit of the 3-multiplication pattern
__global__ void __launch_bounds__(128) k_prod_reg(double* out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    double ar[8], ai[8], as[8];
    for (int s = 0; s < 8; ++s) { ar[s] = 1e-3 * (lane % 7 + s); ai[s] = 1e-3 * (lane % 5); as[s] = ar[s] + ai[s]; }
    double br[8], bi[8], bs[8], t1[8], t2[8], t3[8];
    for (int s = 0; s < 8; ++s) { br[s] = 1.0 + lane * 1e-6 + s; bi[s] = 0.5 - lane * 1e-6; bs[s] = br[s] + bi[s]; t1[s] = t2[s] = t3[s] = 0.0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                t1[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar[(ib + kb) & 7], br[kb], t1[ib], 0, 0, 0);
                t2[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai[(ib + kb) & 7], bi[kb], t2[ib], 0, 0, 0);
                t3[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(as[(ib + kb) & 7], bs[kb], t3[ib], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) { br[s] = (t1[s] - t2[s]) * 1e-3; bi[s] = (t3[s] - t1[s] - t2[s]) * 1e-3; bs[s] = br[s] + bi[s]; }
    }
    double sum = 0;
    for (int s = 0; s < 8; ++s) sum += br[s] + bi[s];
    out[blockIdx.x * 128 + tid] = sum;
}

// variant: A blocks from global memory (block-major "fragA4" image, 16 KB per matrix, L1/L2 resident), 16-column wave
__global__ void __launch_bounds__(128) k_prod_glb(double* out, const cplx* __restrict__ Ag, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    double br[8], bi[8], bs[8], t1[8], t2[8], t3[8];
    for (int s = 0; s < 8; ++s) { br[s] = 1.0 + lane * 1e-6 + s; bi[s] = 0.5 - lane * 1e-6; bs[s] = br[s] + bi[s]; t1[s] = t2[s] = t3[s] = 0.0; }
    const int k = lane >> 4, i = lane & 3;
    const cplx* Ab = Ag + (size_t)(blockIdx.x & 7) * 1024 + k * 4 + i;      // element (i, k) of a block at offset 16*block
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
            for (int ib = 0; ib < 8; ++ib) {
                const cplx a = Ab[(kb * 8 + ib) * 16];
                t1[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, br[kb], t1[ib], 0, 0, 0);
                t2[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, bi[kb], t2[ib], 0, 0, 0);
                t3[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x + a.y, bs[kb], t3[ib], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) { br[s] = (t1[s] - t2[s]) * 1e-3; bi[s] = (t3[s] - t1[s] - t2[s]) * 1e-3; bs[s] = br[s] + bi[s]; }
    }
    double sum = 0;
    for (int s = 0; s < 8; ++s) sum += br[s] + bi[s];
    out[blockIdx.x * 128 + tid] = sum;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2000;
    double* dout; hipMalloc(&dout, (size_t)cus * 64 * 128 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * 2 * wps;                       // 2 waves per block, 4 SIMDs per CU
        hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(128), 0, 0, dout, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_prod, dim3(blocks), dim3(128), 0, 0, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * 2 * iters * 192.0;
        printf("%d wave(s)/SIMD: %.3f ms, %.1f executed TFLOP/s (3M), %.1f algorithmic TFLOP/s, %.1f cycles per MFMA per SIMD at 2.37 GHz\n", wps, ms,
               mfma * 512 / ms * 1e-9, mfma * 512 / ms * 1e-9 * 8 / 6, ms * 1e-3 * 2.37e9 / (mfma / (cus * 4.0)));
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * 4 * wps;                       // 1 wave per block
        hipLaunchKernelGGL(k_prod32, dim3(blocks), dim3(64), 0, 0, dout, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_prod32, dim3(blocks), dim3(64), 0, 0, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * iters * 384.0;
        printf("32-col wave, %d wave(s)/SIMD: %.3f ms, %.1f executed TFLOP/s, %.1f cycles per MFMA\n", wps, ms, mfma * 512 / ms * 1e-9,
               ms * 1e-3 * 2.37e9 / (mfma / (cus * 4.0)));
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * 2 * wps;
        hipLaunchKernelGGL(k_prod_reg, dim3(blocks), dim3(128), 0, 0, dout, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_prod_reg, dim3(blocks), dim3(128), 0, 0, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * 2 * iters * 192.0;
        printf("register A, %d wave(s)/SIMD: %.3f ms, %.1f executed TFLOP/s, %.1f cycles per MFMA\n", wps, ms, mfma * 512 / ms * 1e-9,
               ms * 1e-3 * 2.37e9 / (mfma / (cus * 4.0)));
    }
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * 2 * wps;
        hipLaunchKernelGGL(k_prod_fenced, dim3(blocks), dim3(128), 0, 0, dout, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_prod_fenced, dim3(blocks), dim3(128), 0, 0, dout, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * 2 * iters * 192.0;
        printf("fenced LDS blocks, %d wave(s)/SIMD: %.3f ms, %.1f executed TFLOP/s, %.1f cycles per MFMA\n", wps, ms, mfma * 512 / ms * 1e-9,
               ms * 1e-3 * 2.37e9 / (mfma / (cus * 4.0)));
    }
    cplx* dA; hipMalloc(&dA, 8 * 1024 * sizeof(cplx)); hipMemset(dA, 0, 8 * 1024 * sizeof(cplx));
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * 2 * wps;
        hipLaunchKernelGGL(k_prod_glb, dim3(blocks), dim3(128), 0, 0, dout, dA, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k_prod_glb, dim3(blocks), dim3(128), 0, 0, dout, dA, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)blocks * 2 * iters * 192.0;
        printf("global A blocks, %d wave(s)/SIMD: %.3f ms, %.1f executed TFLOP/s, %.1f cycles per MFMA\n", wps, ms, mfma * 512 / ms * 1e-9,
               ms * 1e-3 * 2.37e9 / (mfma / (cus * 4.0)));
    }
    return 0;
}
