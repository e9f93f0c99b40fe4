// matrix_fetch_probe.hip -- how long does ONE wave need to bring a 16 KB (32x32 complex) matrix from global memory into registers?
// The thin sweeps of the MFMA path fetch one such matrix per time slice and wave; in-kernel clocks showed ~2 us per step against
// 0.4 us of MFMA work.  Variables: access pattern (the transposed GATHER of k_mfma_forward2 from fragD storage vs lane-contiguous
// 1 KB loads), matrices in flight (prefetch depth), number of waves (63 = one trajectory, 1024 = batch), and who wrote the data
// (a previous kernel on other CUs, as in the engine).  Reported: us per matrix per wave and aggregate GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double2 cplx;

__global__ void k_fill(cplx* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2(1e-3 * (double)(i & 1023), 1.0);
}

template <int GATHER, int PD>
__global__ void __launch_bounds__(256) k_fetch(const cplx* __restrict__ base, int per_wave, unsigned long long* ticks, double* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, w = blockIdx.x * 4 + wv;
    const int lk = lane >> 4, lc = lane & 15;
    const cplx* mine = base + (size_t)w * per_wave * 1024;
    cplx f[PD + 1][16];
    auto load = [&](int t, int slot) {
        const cplx* F = mine + (size_t)min(t, per_wave - 1) * 1024;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int I = e >> 3, q = e & 7;
            if (GATHER) f[slot][e] = F[((q >> 2) * 8 + 4 * I + (lc >> 2)) * 64 + 16 * (lc & 3) + 4 * (q & 3) + lk];
            else f[slot][e] = F[e * 64 + lane];
        }
    };
    double acc = 0.0;
    const unsigned long long t0 = wall_clock64();
#pragma unroll
    for (int q = 0; q < PD; ++q) load(q, q);
    for (int t = 0; t < per_wave; t += PD + 1) {
#pragma unroll
        for (int q = 0; q <= PD; ++q) {
            load(t + q + PD, (q + PD) % (PD + 1));
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 16; ++e) acc += f[q][e].x + f[q][e].y;       // consume (stands for the product)
            // ~0.4 us of dependent work per matrix, like the 48-96 MFMAs of a thin product
#pragma unroll
            for (int r = 0; r < 60; ++r) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %1, %0" : "+v"(acc) : "v"(acc));
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) ticks[w] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int GATHER, int PD>
static void run(const cplx* base, int waves, int per_wave, unsigned long long* ticks, double* sink, cplx* scratch, size_t scratch_n) {
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, scratch, scratch_n);       // another kernel in between, as in the engine
    hipLaunchKernelGGL((k_fetch<GATHER, PD>), dim3(waves / 4), dim3(256), 0, 0, base, per_wave, ticks, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(waves);
    CHECK(hipMemcpy(h.data(), ticks, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mx = 0, sum = 0;
    for (auto v : h) { sum += (double)v; if ((double)v > mx) mx = (double)v; }
    const double us = sum / waves / 100.0 / per_wave;
    printf("%-10s %d in flight, %4d waves x %3d matrices: %6.2f us per matrix per wave (slowest wave %6.1f us total), %7.1f GB/s aggregate\n",
           GATHER ? "gather" : "contiguous", PD, waves, per_wave, us, mx / 100.0, (double)waves * per_wave * 16384.0 / (mx / 100.0 * 1e-6) / 1e9);
}

int main() {
    const int per_wave = 24;
    const size_t n = (size_t)1024 * per_wave * 1024;
    cplx *data, *scratch;
    unsigned long long* ticks; double* sink;
    CHECK(hipMalloc((void**)&data, n * sizeof(cplx)));
    CHECK(hipMalloc((void**)&scratch, (size_t)64 << 20));
    CHECK(hipMalloc((void**)&ticks, 1024 * sizeof(unsigned long long)));
    CHECK(hipMalloc((void**)&sink, 256 * 256 * sizeof(double)));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, data, n);
    CHECK(hipDeviceSynchronize());
    for (int waves : {64, 1024}) {
        run<1, 1>(data, waves, per_wave, ticks, sink, scratch, ((size_t)64 << 20) / sizeof(cplx));
        run<1, 2>(data, waves, per_wave, ticks, sink, scratch, ((size_t)64 << 20) / sizeof(cplx));
        run<1, 3>(data, waves, per_wave, ticks, sink, scratch, ((size_t)64 << 20) / sizeof(cplx));
        run<0, 1>(data, waves, per_wave, ticks, sink, scratch, ((size_t)64 << 20) / sizeof(cplx));
        run<0, 2>(data, waves, per_wave, ticks, sink, scratch, ((size_t)64 << 20) / sizeof(cplx));
        run<0, 3>(data, waves, per_wave, ticks, sink, scratch, ((size_t)64 << 20) / sizeof(cplx));
    }
    return 0;
}
