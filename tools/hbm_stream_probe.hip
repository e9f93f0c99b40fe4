// hbm_stream_probe.hip -- what read bandwidth does the sweep access pattern (one wave walks its own run of 16 KB
// matrices, 1024 waves) reach on this MI355X, as a function of the number of 16 KB blocks a wave keeps in flight,
// next to a plain many-waves grid-stride read of the same buffer?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_gridstride(const double2* __restrict__ p, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 12345.678) out[0] = acc;
}

// one wave per item; the item's run = `steps` blocks of 1024 double2 (16 KB); DEPTH blocks in flight
template <int DEPTH, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_wave_runs(const double2* __restrict__ p, int steps, double* out) {
    const int lane = threadIdx.x & 63;
    const size_t item = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const double2* q = p + item * (size_t)steps * 1024 + lane;
    double2 buf[DEPTH][16];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) buf[d][j] = q[(size_t)d * 1024 + j * 64];
    double acc = 0.0;
    for (int t = 0; t < steps; t += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int tn = min(t + u + DEPTH - 1, steps - 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) buf[(u + DEPTH - 1) % DEPTH][j] = q[(size_t)tn * 1024 + j * 64];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += buf[u][j].x + buf[u][j].y;
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

// the sweep's mix: per 16 KB block read, 4 KB written (every lane with (lane & 15) < 8 stores 16 B, 8 times) to a second run
template <int DEPTH, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_wave_runs_rw(const double2* __restrict__ p, double2* __restrict__ w, int steps, double* out) {
    const int lane = threadIdx.x & 63;
    const size_t item = (size_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const double2* q = p + item * (size_t)steps * 1024 + lane;
    double2* wq = w + item * (size_t)steps * 256 + (lane >> 4) * 8 + (lane & 15);
    double2 buf[DEPTH][16];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
        for (int j = 0; j < 16; ++j) buf[d][j] = q[(size_t)d * 1024 + j * 64];
    double acc = 0.0;
    for (int t = 0; t < steps; t += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int tn = min(t + u + DEPTH - 1, steps - 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) buf[(u + DEPTH - 1) % DEPTH][j] = q[(size_t)tn * 1024 + j * 64];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += buf[u][j].x + buf[u][j].y;
            if ((lane & 15) < 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) wq[(size_t)min(t + u, steps - 1) * 256 + j * 32] = make_double2(acc, buf[u][j].x);
            }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int DEPTH, int WPB>
static float run_runs_rw(const double2* p, double2* w, int items, int steps, double* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_wave_runs_rw<DEPTH, WPB>), dim3(items / WPB), dim3(64 * WPB), 0, 0, p, w, steps, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

template <int DEPTH, int WPB>
static float run_runs(const double2* p, int items, int steps, double* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_wave_runs<DEPTH, WPB>), dim3(items / WPB), dim3(64 * WPB), 0, 0, p, steps, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const int steps = 32;
    double* out; CK(hipMalloc(&out, 8));
    for (int items : {1024, 2048, 4096}) {
        const size_t n = (size_t)items * steps * 1024;          // double2 elements
        double2* p; CK(hipMalloc(&p, n * 16)); CK(hipMemset(p, 0, n * 16));
        const double gb = n * 16 / 1e9;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_gridstride, dim3(256 * 8), dim3(256), 0, 0, p, n, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("items %4d (%.0f MB): grid-stride 2048x256 %.1f us = %.2f TB/s\n", items, gb * 1e3, best * 1e3, gb / best);
        float t;
        t = run_runs<1, 4>(p, items, steps, out); printf("   wave runs depth 1, 4 waves/WG: %.1f us = %.2f TB/s\n", t * 1e3, gb / t);
        t = run_runs<2, 4>(p, items, steps, out); printf("   wave runs depth 2, 4 waves/WG: %.1f us = %.2f TB/s\n", t * 1e3, gb / t);
        t = run_runs<3, 4>(p, items, steps, out); printf("   wave runs depth 3, 4 waves/WG: %.1f us = %.2f TB/s\n", t * 1e3, gb / t);
        t = run_runs<4, 4>(p, items, steps, out); printf("   wave runs depth 4, 4 waves/WG: %.1f us = %.2f TB/s\n", t * 1e3, gb / t);
        t = run_runs<2, 1>(p, items, steps, out); printf("   wave runs depth 2, 1 wave /WG: %.1f us = %.2f TB/s\n", t * 1e3, gb / t);
        double2* w; CK(hipMalloc(&w, n * 4)); CK(hipMemset(w, 0, n * 4));
        t = run_runs_rw<1, 4>(p, w, items, steps, out); printf("   read 16 KB + write 4 KB per step, depth 1: %.1f us = %.2f TB/s (read+write)\n", t * 1e3, gb * 1.25 / t);
        t = run_runs_rw<2, 4>(p, w, items, steps, out); printf("   read 16 KB + write 4 KB per step, depth 2: %.1f us = %.2f TB/s\n", t * 1e3, gb * 1.25 / t);
        t = run_runs_rw<3, 4>(p, w, items, steps, out); printf("   read 16 KB + write 4 KB per step, depth 3: %.1f us = %.2f TB/s\n", t * 1e3, gb * 1.25 / t);
        CK(hipFree(w));
        CK(hipFree(p));
    }
    return 0;
}
