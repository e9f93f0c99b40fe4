// Probe for the direct state-transfer chain (k_gemm_taylor_chain, C3 x 64): a 64 x 64 complex mat-vec of ONE workgroup without LDS reads of the
// vector and without a DPP butterfly.  Wave w owns the columns CW*w .. CW*w + CW - 1 (CW = 64 / W) of the generator, lane = row; the vector
// segment of the wave lives one entry per lane inside every row of 16 lanes and reaches the FMAs through the DP-DPP control row_newbcast:j
// (v_fmac_f64_dpp: gfx90a+), so a complex MAC is four VOP2 instructions and nothing else.  The W partial sums of a row meet in LDS: one
// ds_write_b128 per lane, one barrier, W ds_read_b128 per lane, W - 1 complex adds.
//   taylor_dpp_probe [seeds=64] [slices=200] [terms=10]
// prints us per dependent mat-vec for W = 4 / 8 and the max error against a host evaluation of the same recursion.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <complex>

typedef double2 cplx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define CMAC_(RE, IM, XR, XI, AR, AI, J) \
    asm volatile("v_fmac_f64_dpp %0, %2, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, %3, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %0, -%3, %5 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, %2, %5 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(RE), "+v"(IM) : "v"(XR), "v"(XI), "v"(AR), "v"(AI))
#define CMACN_(RE, IM, XR, XI, AR, AI, J) \
    asm volatile("v_fmac_f64_dpp %0, -%2, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, -%3, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, -%2, %5 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(RE), "+v"(IM) : "v"(XR), "v"(XI), "v"(AR), "v"(AI))
#define CMAC(S, J) do { if constexpr (NEG) CMACN_(re[S], im[S], xr, xi, a[J].x, a[J].y, J); else CMAC_(re[S], im[S], xr, xi, a[J].x, a[J].y, J); } while (0)

template <int CW, int NACC, bool NEG>
__device__ __forceinline__ void matvec_dpp(const cplx (&a)[CW], double xr, double xi, double& pr, double& pi) {
    double re[NACC], im[NACC];
#pragma unroll
    for (int s = 0; s < NACC; ++s) { re[s] = 0.0; im[s] = 0.0; }
    asm volatile("s_nop 1" ::: "memory");                  // VALU write of the vector entry -> DPP read: 2 wait states
    CMAC(0, 0); CMAC(1 % NACC, 1); CMAC(2 % NACC, 2); CMAC(3 % NACC, 3); CMAC(0, 4); CMAC(1 % NACC, 5); CMAC(2 % NACC, 6); CMAC(3 % NACC, 7);
    if constexpr (CW == 16) {
        CMAC(0, 8); CMAC(1 % NACC, 9); CMAC(2 % NACC, 10); CMAC(3 % NACC, 11); CMAC(0, 12); CMAC(1 % NACC, 13); CMAC(2 % NACC, 14); CMAC(3 % NACC, 15);
    }
    if constexpr (NACC == 4) { pr = (re[0] + re[1]) + (re[2] + re[3]); pi = (im[0] + im[1]) + (im[2] + im[3]); }
    else if constexpr (NACC == 2) { pr = re[0] + re[1]; pi = im[0] + im[1]; }
    else { pr = re[0]; pi = im[0]; }
}

// one workgroup per seed; B: seeds x slices x 64 x 64 (row-major), X0: seeds x 64, Out: seeds x slices x 64
template <int W, int NACC, bool LAPS>
__global__ void __launch_bounds__(64 * W) k_chain(const cplx* __restrict__ B, const cplx* __restrict__ X0, cplx* __restrict__ Out, int slices, int terms,
                                                    double sign, unsigned long long* Laps) {
    constexpr int CW = 64 / W;
    __shared__ __attribute__((aligned(16))) cplx part[2][W][64];
    __shared__ double tinv[64];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int xi_idx = CW * w + (l & (CW - 1));             // the vector entry this lane carries
    const cplx* Bp = B + (size_t)blockIdx.x * slices * 4096 + (size_t)CW * w * 64 + l;
    cplx* Op = Out + (size_t)blockIdx.x * slices * 64;
    if (tid < 64) { double f = 1.0; for (int i = 2; i <= tid; ++i) f *= (double)i; tinv[tid] = 1.0 / f; }
    cplx xv = X0[(size_t)blockIdx.x * 64 + xi_idx];
    __syncthreads();
    auto load = [&](cplx (&kd)[CW], int j) {
        const cplx* p = Bp + (size_t)min(j, slices - 1) * 4096;
#pragma unroll
        for (int c = 0; c < CW; ++c) kd[c] = p[c * 64];
    };
    int cur = 0;
    unsigned long long lap[3] = {0, 0, 0}, lap_slice = 0, t_slice = 0;
    const unsigned long long tstart = __builtin_readcyclecounter();
    auto step = [&](int j, const cplx (&a)[CW]) {
        double outr = xv.x, outi = xv.y;
        if (LAPS) { const unsigned long long tt = __builtin_readcyclecounter(); if (j > 0) lap_slice += tt - t_slice; }
        for (int ii = 1; ii < terms; ++ii) {
            double pr, pi;
            const unsigned long long t0 = LAPS ? __builtin_readcyclecounter() : 0;
            matvec_dpp<CW, NACC, true>(a, xv.x, xv.y, pr, pi);
            if (LAPS) asm volatile("" : "+v"(pr), "+v"(pi));
            const unsigned long long t1 = LAPS ? __builtin_readcyclecounter() : 0;
            part[cur][w][l] = make_double2(pr, pi);
            __syncthreads();
            const unsigned long long t2 = LAPS ? __builtin_readcyclecounter() : 0;
            cplx s[W];
#pragma unroll
            for (int q = 0; q < W; ++q) s[q] = part[cur][q][xi_idx];
            if constexpr (W == 4) { xv.x = ((s[0].x + s[1].x) + (s[2].x + s[3].x)); xv.y = ((s[0].y + s[1].y) + (s[2].y + s[3].y)); }
            else {
                xv.x = (((s[0].x + s[1].x) + (s[2].x + s[3].x)) + ((s[4].x + s[5].x) + (s[6].x + s[7].x)));
                xv.y = (((s[0].y + s[1].y) + (s[2].y + s[3].y)) + ((s[4].y + s[5].y) + (s[6].y + s[7].y)));
            }
            if (LAPS) asm volatile("" : "+v"(xv.x), "+v"(xv.y));
            const unsigned long long t3 = LAPS ? __builtin_readcyclecounter() : 0;
            lap[0] += t1 - t0; lap[1] += t2 - t1; lap[2] += t3 - t2;
            const double inv = tinv[ii];
            outr = fma(xv.x, inv, outr); outi = fma(xv.y, inv, outi);
            cur ^= 1;
        }
        if (LAPS) t_slice = __builtin_readcyclecounter();
        xv = make_double2(outr, outi);
        if (l < CW) Op[(size_t)j * 64 + xi_idx] = xv;
    };
    cplx k0[CW], k1[CW];
    load(k0, 0);
    int j = 0;
    for (; j + 2 <= slices; j += 2) {
        load(k1, j + 1); step(j, k0);
        load(k0, j + 2); step(j + 1, k1);
    }
    if (j < slices) step(j, k0);
    if (Laps && blockIdx.x == 0 && (tid & 63) == 0) { Laps[w * 4 + 0] = lap[0]; Laps[w * 4 + 1] = lap[1]; Laps[w * 4 + 2] = lap[2]; Laps[16 + w] = lap_slice; Laps[w * 4 + 3] = __builtin_readcyclecounter() - tstart; }
}

int main(int argc, char** argv) {
    const int seeds = argc > 1 ? atoi(argv[1]) : 64, slices = argc > 2 ? atoi(argv[2]) : 200, terms = argc > 3 ? atoi(argv[3]) : 10;
    const size_t nb = (size_t)seeds * slices * 4096;
    std::vector<cplx> hB(nb), hX((size_t)seeds * 64);
    srand(7);
    for (auto& v : hB) { v.x = 0.05 * (rand() / (double)RAND_MAX - 0.5); v.y = 0.05 * (rand() / (double)RAND_MAX - 0.5); }
    for (auto& v : hX) { v.x = rand() / (double)RAND_MAX - 0.5; v.y = rand() / (double)RAND_MAX - 0.5; }
    // host: seed 0 and the last seed
    auto host = [&](int sd, std::vector<std::complex<double>>& out) {
        std::vector<std::complex<double>> x(64), v(64), t(64), o(64);
        for (int i = 0; i < 64; ++i) x[i] = {hX[(size_t)sd * 64 + i].x, hX[(size_t)sd * 64 + i].y};
        out.resize((size_t)slices * 64);
        for (int j = 0; j < slices; ++j) {
            const cplx* b = &hB[((size_t)sd * slices + j) * 4096];
            v = x; o = x; double f = 1.0;
            for (int ii = 1; ii < terms; ++ii) {
                f *= ii;
                for (int r = 0; r < 64; ++r) { std::complex<double> s = 0; for (int c = 0; c < 64; ++c) s += std::complex<double>(b[c * 64 + r].x, b[c * 64 + r].y) * v[c]; t[r] = -s; }
                v = t;
                for (int r = 0; r < 64; ++r) o[r] += v[r] / f;
            }
            x = o;
            for (int r = 0; r < 64; ++r) out[(size_t)j * 64 + r] = x[r];
        }
    };
    std::vector<std::complex<double>> ref0, ref1;
    host(0, ref0); host(seeds - 1, ref1);
    cplx *dB, *dX, *dO;
    CK(hipMalloc(&dB, nb * sizeof(cplx))); CK(hipMalloc(&dX, hX.size() * sizeof(cplx))); CK(hipMalloc(&dO, (size_t)seeds * slices * 64 * sizeof(cplx)));
    CK(hipMemcpy(dB, hB.data(), nb * sizeof(cplx), hipMemcpyHostToDevice)); CK(hipMemcpy(dX, hX.data(), hX.size() * sizeof(cplx), hipMemcpyHostToDevice));
    std::vector<cplx> hO((size_t)seeds * slices * 64);
    unsigned long long* dL; CK(hipMalloc(&dL, 32 * 8)); unsigned long long hL[32];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        CK(hipMemset(dO, 0, hO.size() * sizeof(cplx)));
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        CK(hipMemcpy(hO.data(), dO, hO.size() * sizeof(cplx), hipMemcpyDeviceToHost));
        double err = 0;
        for (size_t i = 0; i < (size_t)slices * 64; ++i) {
            err = fmax(err, std::abs(std::complex<double>(hO[i].x, hO[i].y) - ref0[i]));
            const cplx z = hO[(size_t)(seeds - 1) * slices * 64 + i];
            err = fmax(err, std::abs(std::complex<double>(z.x, z.y) - ref1[i]));
        }
        printf("%-28s %8.3f ms per launch  %7.4f us per mat-vec  max err %.2e\n", name, ms, 1e3 * ms / ((double)slices * (terms - 1)), err);
        CK(hipMemcpy(hL, dL, sizeof(hL), hipMemcpyDeviceToHost));
        const double nmv = (double)slices * (terms - 1);
        printf("    between inner loops, ticks per slice: %.1f %.1f %.1f %.1f\n", hL[16] / (double)slices, hL[17] / (double)slices, hL[18] / (double)slices, hL[19] / (double)slices);
        for (int q = 0; q < 4; ++q) printf("    wave %d: counter ticks per mat-vec: FMAs %.1f  write+barrier %.1f  reads+sum %.1f  | whole kernel %.1f\n", q, hL[q * 4] / nmv, hL[q * 4 + 1] / nmv, hL[q * 4 + 2] / nmv, hL[q * 4 + 3] / nmv);
    };
    run("W=4 (256 thr) 1 acc", [&] { hipLaunchKernelGGL((k_chain<4, 1, false>), dim3(seeds), dim3(256), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    run("W=4 (256 thr) 1 acc laps", [&] { hipLaunchKernelGGL((k_chain<4, 1, true>), dim3(seeds), dim3(256), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    run("W=4 (256 thr) 2 acc", [&] { hipLaunchKernelGGL((k_chain<4, 2, false>), dim3(seeds), dim3(256), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    run("W=4 (256 thr) 4 acc", [&] { hipLaunchKernelGGL((k_chain<4, 4, false>), dim3(seeds), dim3(256), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    run("W=8 (512 thr) 1 acc", [&] { hipLaunchKernelGGL((k_chain<8, 1, false>), dim3(seeds), dim3(512), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    run("W=8 (512 thr) 1 acc laps", [&] { hipLaunchKernelGGL((k_chain<8, 1, true>), dim3(seeds), dim3(512), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    run("W=8 (512 thr) 2 acc", [&] { hipLaunchKernelGGL((k_chain<8, 2, false>), dim3(seeds), dim3(512), 0, 0, dB, dX, dO, slices, terms, -1.0, dL); });
    return 0;
}
