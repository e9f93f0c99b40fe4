// qoc_gemm_wg.hip -- translation unit of k_zgemm_wg (qoc_gemm_tiles.h): plain products of large matrices on 64 x 128 workgroup tiles.
// Compiled with hipcc's default AGPR form of the MFMAs (no -amdgpu-mfma-vgpr-form, unlike qoc_engine.hip): see the comment at the kernel.
#define QOC_ZGEMM_WG_TU
#include "qoc_gemm_tiles.h"

void qoc_zgemm_wg_launch(const GemmArgs& g, unsigned blocks, hipStream_t s) {
    if (g.btrans) hipLaunchKernelGGL(k_zgemm_wg<true>, dim3(blocks), dim3(64 * ZW_WAVES), qoc_zgemm_wg_lds(), s, g);
    else hipLaunchKernelGGL(k_zgemm_wg<false>, dim3(blocks), dim3(64 * ZW_WAVES), qoc_zgemm_wg_lds(), s, g);
}
bool qoc_zgemm_wg_opt_in() {
    return hipFuncSetAttribute((const void*)k_zgemm_wg<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qoc_zgemm_wg_lds()) == hipSuccess &&
           hipFuncSetAttribute((const void*)k_zgemm_wg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qoc_zgemm_wg_lds()) == hipSuccess;
}
