// mfma_f64_4x4_layout.hip -- derives the lane layout of v_mfma_f64_4x4x4_4b_f64 empirically: one-hot A in lane la, one-hot
// B in lane lb, and the lane of D that receives the product.  Output: for every la the list of (lb -> ld).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(double* out) {
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, l = threadIdx.x;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)blockIdx.x * 64 + l] = d;
}

int main() {
    double* dout;
    hipMalloc(&dout, 4096 * 64 * sizeof(double));
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(64), 0, 0, dout);
    std::vector<double> h(4096 * 64);
    hipMemcpy(h.data(), dout, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    return 0;
}
