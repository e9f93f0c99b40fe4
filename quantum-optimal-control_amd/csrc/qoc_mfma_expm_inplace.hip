// qoc_mfma_expm_inplace.hip -- translation unit of k_mfma_expm_inplace (qoc_mfma_expm_inplace.h) and its launcher.
// Compiled with -mllvm -amdgpu-mfma-vgpr-form (__graft_entry__.UNIT_FLAGS): the accumulators of its products live in VGPRs.
#include "qoc_kernels_mfma.h"
#include "qoc_mfma_expm_inplace.h"

void qoc_mfma_launch_expm_inplace(QocMfma& mf, const QocDev& d, hipStream_t s) {
    const dim3 grid(d.B * mf.C), block(64);
    const bool even = (d.T & 1) == 0, s0 = d.s == 0;
    // active 4-row strips of the padded 32 x 32 matrices: ceil(n / 4) (17 <= n <= 32: 5 .. 8)
    const int qa = qoc_active_strips(d.n);
#define QOC_IP(KCv, EVv, S0v) do { if (qa == 8) hipLaunchKernelGGL((k_mfma_expm_inplace<KCv, EVv, S0v, 8>), grid, block, 0, s, d, mf); \
                                   else if (qa == 7) hipLaunchKernelGGL((k_mfma_expm_inplace<KCv, EVv, S0v, 7>), grid, block, 0, s, d, mf); \
                                   else if (qa == 6) hipLaunchKernelGGL((k_mfma_expm_inplace<KCv, EVv, S0v, 6>), grid, block, 0, s, d, mf); \
                                   else hipLaunchKernelGGL((k_mfma_expm_inplace<KCv, EVv, S0v, 5>), grid, block, 0, s, d, mf); } while (0)
    if (d.k <= 4) { if (even) { if (s0) QOC_IP(4, true, true); else QOC_IP(4, true, false); } else { if (s0) QOC_IP(4, false, true); else QOC_IP(4, false, false); } }
    else { if (even) { if (s0) QOC_IP(8, true, true); else QOC_IP(8, true, false); } else { if (s0) QOC_IP(8, false, true); else QOC_IP(8, false, false); } }
#undef QOC_IP
}
