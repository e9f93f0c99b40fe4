// qoc_mfma_expm_inplace.hip -- translation unit of k_mfma_expm_inplace (qoc_mfma_expm_inplace.h) and its launcher.
// Compiled with -mllvm -amdgpu-mfma-vgpr-form (__graft_entry__.UNIT_FLAGS): the accumulators of its products live in VGPRs.
#include "qoc_kernels_mfma.h"
#include "qoc_mfma_expm_inplace.h"

void qoc_mfma_launch_expm_inplace(QocMfma& mf, const QocDev& d, hipStream_t s) {
    const dim3 grid(d.B * mf.C), block(64);
    const bool even = (d.T & 1) == 0;
    if (d.k <= 4) { if (even) hipLaunchKernelGGL((k_mfma_expm_inplace<4, true>), grid, block, 0, s, d, mf); else hipLaunchKernelGGL((k_mfma_expm_inplace<4, false>), grid, block, 0, s, d, mf); }
    else { if (even) hipLaunchKernelGGL((k_mfma_expm_inplace<8, true>), grid, block, 0, s, d, mf); else hipLaunchKernelGGL((k_mfma_expm_inplace<8, false>), grid, block, 0, s, d, mf); }
}
