// qoc_mfma_expm.hip -- translation unit of the MFMA-path exponential kernels (qoc_mfma_expm.h) and their launcher.
#include "qoc_kernels_mfma.h"
#include "qoc_mfma_expm.h"
#include "qoc_mfma_expm_stream.h"

template <int NT>
static inline void qoc_mfma_launch_all_expm(QocMfma& mf, const QocDev& d, hipStream_t s) {
    // AUTO: NT = 2 with at least half of the 1024 SIMDs busy -> one wave per (seed, chunk) on v_mfma_f64_4x4x4 (0.92 vs 1.21 ms
    // per launch at C2 x 64); NT = 1 and small launches keep the 16x16x4 kernel (C1: 0.072 vs 0.074 ms; one C2 trajectory:
    // 0.67 vs 0.75 ms).  qoc_config.variant forces one of the three kernels (parity tests, A/B runs).
    const int v = qoc_mfma_expm_variant(mf, d);
    if (v == 4 && NT == 2) {
        constexpr int NTS = 2;
        if (d.k <= 4) hipLaunchKernelGGL((k_mfma_expm_chunk4s<NTS, 4>), dim3(d.B * mf.C), dim3(64), 0, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_expm_chunk4s<NTS, 8>), dim3(d.B * mf.C), dim3(64), 0, s, d, mf);
    }
    else if (v == 3 && NT <= 2) hipLaunchKernelGGL(k_mfma_expm_chunk4w<NT>, dim3(d.B * mf.C), dim3(64), 0, s, d, mf);
    else if (v == 2) hipLaunchKernelGGL(k_mfma_expm_chunk4<NT>, dim3(d.B * mf.C), dim3(64 * NT), 0, s, d, mf);
    else hipLaunchKernelGGL(k_mfma_expm_chunk<NT>, dim3(d.B * mf.C), dim3(64 * NT), 0, s, d, mf);
}
void qoc_mfma_launch_expm(QocMfma& mf, const QocDev& d, hipStream_t s) {
    if (mf.NT == 1) qoc_mfma_launch_all_expm<1>(mf, d, s); else if (mf.NT == 2) qoc_mfma_launch_all_expm<2>(mf, d, s); else if (mf.NT == 3) qoc_mfma_launch_all_expm<3>(mf, d, s); else qoc_mfma_launch_all_expm<4>(mf, d, s);
}
