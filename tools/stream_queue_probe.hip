// stream_queue_probe.hip -- are some streams slower at back-to-back dependent small launches?  (launch-bound runs timed right after an engine was created)
// For each of 16 streams created one after the other: 2000 dependent launches of a tiny kernel, wall time per launch; three rounds.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_touch(double* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0; }
int main() {
    double* buf = nullptr;
    (void)hipMalloc(&buf, 1 << 20);
    (void)hipMemset(buf, 0, 1 << 20);
    const int NS = 16, NL = 2000;
    std::vector<hipStream_t> st(NS);
    for (int i = 0; i < NS; ++i) (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    for (int round = 0; round < 3; ++round) {
        printf("round %d:", round);
        for (int i = 0; i < NS; ++i) {
            for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_touch, dim3(8), dim3(64), 0, st[i], buf);
            (void)hipStreamSynchronize(st[i]);
            const auto t0 = std::chrono::steady_clock::now();
            for (int w = 0; w < NL; ++w) hipLaunchKernelGGL(k_touch, dim3(8), dim3(64), 0, st[i], buf);
            (void)hipStreamSynchronize(st[i]);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / NL;
            printf(" %.2f", us);
        }
        printf("  us per launch\n");
    }
    return 0;
}
