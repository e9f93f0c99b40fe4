// Micro-probe behind the round-5 re-cut of the direct state-transfer chain (csrc/qoc_gemm_chain_dpp.h): what do the pieces of one dependent
// 64 x 64 complex mat-vec cost on ONE workgroup of four waves (one per SIMD of a CU)?
//   * issue rate of v_fmac_f64_dpp (row_newbcast) against plain v_fma_f64 with VGPR / SGPR multiplicands
//   * the four-wave LDS exchange: ds_write_b128 + barrier + R x ds_read_b128 (R = 4, 2, 1), reads by all lanes or by 16 lanes
//   * v_permlane32_swap / v_permlane16_swap (gfx950) as the in-wave reduction over the four rows of 16 lanes
//   * s_barrier alone
// Prints core-clock cycles (s_memtime) per repetition, wave 0 of the workgroup.   chain_micro_probe [reps=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double2 cplx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define DPP3(J) \
    asm volatile("v_fmac_f64_dpp %0, %3, %6 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, %4, %7 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %2, %5, %8 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(k1), "+v"(k2), "+v"(k3) : "v"(xr), "v"(xd), "v"(xs), "v"(a[J]), "v"(b[J]), "v"(c[J]))
#define FMA3(J) \
    asm volatile("v_fma_f64 %0, %3, %6, %0\n\t" \
                 "v_fma_f64 %1, %4, %7, %1\n\t" \
                 "v_fma_f64 %2, %5, %8, %2" \
                 : "+v"(k1), "+v"(k2), "+v"(k3) : "v"(xr), "v"(xd), "v"(xs), "v"(a[J]), "v"(b[J]), "v"(c[J]))
#define FMA3S(J) \
    asm volatile("v_fma_f64 %0, %3, %6, %0\n\t" \
                 "v_fma_f64 %1, %4, %7, %1\n\t" \
                 "v_fma_f64 %2, %5, %8, %2" \
                 : "+v"(k1), "+v"(k2), "+v"(k3) : "s"(sxr), "s"(sxd), "s"(sxs), "v"(a[J]), "v"(b[J]), "v"(c[J]))
#define ALL16(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7); M(8); M(9); M(10); M(11); M(12); M(13); M(14); M(15)

// mode 0: 48 v_fmac_f64_dpp   1: 48 v_fma_f64 (VGPR)   2: 48 v_fma_f64 (SGPR multiplicand)
template <int MODE>
__global__ void __launch_bounds__(256) k_fma(double* out, unsigned long long* cyc, int reps) {
    const int l = threadIdx.x;
    double a[16], b[16], c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { a[j] = 1e-3 * (l + j); b[j] = 2e-3 * (l - j); c[j] = 1e-3 * j; }
    double xr = 1.0 + 1e-3 * l, xd = 0.5, xs = 0.25, acc = 0.0;
    const double sxr = __builtin_amdgcn_readfirstlane(l) * 1e-3 + 1.0, sxd = 0.5, sxs = 0.25;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        double k1 = 0.0, k2 = 0.0, k3 = 0.0;
        if (MODE == 0) { asm volatile("s_nop 1" : "+v"(xr), "+v"(xd), "+v"(xs)); ALL16(DPP3); }
        else if (MODE == 1) { ALL16(FMA3); }
        else { ALL16(FMA3S); }
        acc += k1 - k3 + k2;
        xr = acc * 1e-9 + 1.0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + l] = acc;
    if ((l & 63) == 0) cyc[l >> 6] = t1 - t0;
}

// The exchange: every lane writes its partial (b128), barrier, R reads + sum.  LANES16: only lanes < 16 of a wave read (and sum).
// SWAP: 1 read per lane (row q of 16 lanes reads the partial of wave q), then the in-wave sum over the four rows by v_permlane32_swap / 16_swap.
// SWAP2: 2 reads per lane (rows 0, 1 read waves 0, 1; rows 2, 3 read waves 2, 3), add, one v_permlane32_swap stage ... and a 16_swap stage is not needed
//        because rows {0,1} hold the same sum after the reads (both rows read both partials).
template <int R, bool LANES16, int SWAP>
__global__ void __launch_bounds__(256) k_exch(double* out, unsigned long long* cyc, int reps) {
    __shared__ __attribute__((aligned(16))) cplx part[2][4][64];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, idx = 16 * w + (l & 15), q = l >> 4;
    double pr = 1e-3 * tid, pi = -2e-3 * tid, sx = 0.0, sy = 0.0;
    int cur = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        part[cur][w][l] = make_double2(pr, pi);
        __syncthreads();
        double xr, xi;
        if (SWAP == 1) {
            const cplx s = part[cur][q][idx];
            double ar = s.x, ai = s.y, br = s.x, bi = s.y;
            // after permlane32_swap(a, b): a = [a.lo, b.lo], b = [a.hi, b.hi]  (b = copy of a: a = lo|lo, b = hi|hi)
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ar)[0]), "+v"(((int*)&br)[0]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ar)[1]), "+v"(((int*)&br)[1]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ai)[0]), "+v"(((int*)&bi)[0]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ai)[1]), "+v"(((int*)&bi)[1]));
            ar += br; ai += bi;
            br = ar; bi = ai;
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(((int*)&ar)[0]), "+v"(((int*)&br)[0]));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(((int*)&ar)[1]), "+v"(((int*)&br)[1]));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(((int*)&ai)[0]), "+v"(((int*)&bi)[0]));
            asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(((int*)&ai)[1]), "+v"(((int*)&bi)[1]));
            xr = ar + br; xi = ai + bi;
        } else if (SWAP == 2) {
            const int h = q >> 1;
            const cplx s0 = part[cur][2 * h][idx], s1 = part[cur][2 * h + 1][idx];
            double ar = s0.x + s1.x, ai = s0.y + s1.y, br = ar, bi = ai;
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ar)[0]), "+v"(((int*)&br)[0]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ar)[1]), "+v"(((int*)&br)[1]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ai)[0]), "+v"(((int*)&bi)[0]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(((int*)&ai)[1]), "+v"(((int*)&bi)[1]));
            xr = ar + br; xi = ai + bi;
        } else if (!LANES16 || l < 16) {
            cplx s[4];
#pragma unroll
            for (int k = 0; k < R; ++k) s[k] = part[cur][k][idx];
            if (R == 4) { xr = (s[0].x + s[1].x) + (s[2].x + s[3].x); xi = (s[0].y + s[1].y) + (s[2].y + s[3].y); }
            else if (R == 2) { xr = s[0].x + s[1].x; xi = s[0].y + s[1].y; }
            else { xr = s[0].x; xi = s[0].y; }
        } else { xr = 0.0; xi = 0.0; }
        sx += xr; sy += xi;
        pr = xr * 0.25 + 1e-3; pi = xi * 0.25 - 1e-3;
        cur ^= 1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + 2 * tid] = sx; out[blockIdx.x * 512 + 2 * tid + 1] = sy;
    if (l == 0) cyc[w] = t1 - t0;
}


// issue-rate sanity: 48 instructions per repetition on 12 independent accumulators.  KIND 0: v_fma_f32   1: v_fma_f64 (three VGPR pairs)   2: v_mul_f64
//   3: v_add_f64   4: v_fmac_f64 (VOP2 form, accumulator = destination)   5: v_fma_f64 with the multiplicand in an SGPR pair
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(double* out, unsigned long long* cyc, int reps) {
    const int l = threadIdx.x;
    double acc[12], a[4];
    float facc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) { acc[j] = 1e-3 * (l + j); facc[j] = 1e-3f * (l + j); }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = 1.0 + 1e-9 * (l + j);
    const double sa = 1.0 + 1e-9 * __builtin_amdgcn_readfirstlane(l);
    float fa = 1.0f + 1e-7f * l;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(facc[j]) : "v"(fa));
                if (KIND == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[u]), "v"(a[(u + 1) & 3]));
                if (KIND == 2) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[j]) : "v"(a[u]));
                if (KIND == 3) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[j]) : "v"(a[u]));
                if (KIND == 4) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[j]) : "v"(a[u]), "v"(a[(u + 1) & 3]));
                if (KIND == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "s"(sa), "v"(a[u]));
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 12; ++j) s += acc[j] + facc[j];
    out[blockIdx.x * 256 + l] = s;
    if ((l & 63) == 0) cyc[l >> 6] = t1 - t0;
}

__global__ void __launch_bounds__(256) k_barrier(unsigned long long* cyc, int reps) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) __builtin_amdgcn_s_barrier();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    double* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64 * 512 * sizeof(double)));
    CK(hipMalloc(&cyc, 4 * sizeof(unsigned long long)));
    unsigned long long h[4];
    std::vector<double> ho(512);
    auto report = [&](const char* name, bool show) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ho.data(), out, 512 * sizeof(double), hipMemcpyDeviceToHost));
        printf("%-64s %8.1f %8.1f %8.1f %8.1f cycles per repetition", name, h[0] / (double)reps, h[1] / (double)reps, h[2] / (double)reps, h[3] / (double)reps);
        if (show) printf("   (lane 0: %.6f %.6f  lane 17 of wave 1: %.6f)", ho[0], ho[1], ho[2 * (64 + 17)]);
        printf("\n");
    };
    for (int pass = 0; pass < 2; ++pass) {                 // pass 0 warms the clocks
        hipLaunchKernelGGL(k_fma<0>, dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("48 x v_fmac_f64_dpp row_newbcast (3 chains)", false);
        hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("48 x v_fma_f64, VGPR multiplicand", false);
        hipLaunchKernelGGL(k_fma<2>, dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("48 x v_fma_f64, SGPR multiplicand", false);
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(256), 0, 0, cyc, reps); if (pass) report("s_barrier alone, 4 waves", false);
        hipLaunchKernelGGL((k_exch<4, false, 0>), dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("exchange: write, barrier, 4 reads by all lanes, sum", true);
        hipLaunchKernelGGL((k_exch<2, false, 0>), dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("exchange: write, barrier, 2 reads by all lanes, sum", false);
        hipLaunchKernelGGL((k_exch<1, false, 0>), dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("exchange: write, barrier, 1 read by all lanes", false);
        hipLaunchKernelGGL((k_exch<4, true, 0>), dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("exchange: write, barrier, 4 reads by 16 lanes, sum", false);
        hipLaunchKernelGGL((k_exch<4, false, 1>), dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("exchange: 1 read + permlane32_swap + permlane16_swap sum", true);
        hipLaunchKernelGGL((k_exch<4, false, 2>), dim3(1), dim3(256), 0, 0, out, cyc, reps); if (pass) report("exchange: 2 reads + add + permlane32_swap sum", true);
    }
    // 64 workgroups at once (one per CU, as the chain kernel runs): the FMA sequences again
    hipLaunchKernelGGL(k_fma<0>, dim3(64), dim3(256), 0, 0, out, cyc, reps); report("64 workgroups: 48 x v_fmac_f64_dpp", false);
    hipLaunchKernelGGL(k_fma<1>, dim3(64), dim3(256), 0, 0, out, cyc, reps); report("64 workgroups: 48 x v_fma_f64 VGPR", false);

    hipLaunchKernelGGL(k_rate<0>, dim3(1), dim3(256), 0, 0, out, cyc, reps); report("rate: 48 x v_fma_f32 (12 accumulators)", false);
    hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(256), 0, 0, out, cyc, reps); report("rate: 48 x v_fma_f64 VGPR", false);
    hipLaunchKernelGGL(k_rate<2>, dim3(1), dim3(256), 0, 0, out, cyc, reps); report("rate: 48 x v_mul_f64", false);
    hipLaunchKernelGGL(k_rate<3>, dim3(1), dim3(256), 0, 0, out, cyc, reps); report("rate: 48 x v_add_f64", false);
    hipLaunchKernelGGL(k_rate<4>, dim3(1), dim3(256), 0, 0, out, cyc, reps); report("rate: 48 x v_fmac_f64", false);
    hipLaunchKernelGGL(k_rate<5>, dim3(1), dim3(256), 0, 0, out, cyc, reps); report("rate: 48 x v_fma_f64 SGPR multiplicand", false);
    hipLaunchKernelGGL(k_rate<1>, dim3(1), dim3(64), 0, 0, out, cyc, reps); report("rate: 48 x v_fma_f64 VGPR, ONE wave in the workgroup", false);
    return 0;
}
