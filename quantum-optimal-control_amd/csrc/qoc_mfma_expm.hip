// qoc_mfma_expm.hip -- translation unit of the MFMA-path exponential kernels (qoc_mfma_expm.h) and their launcher.
#include <cstdlib>
#include "qoc_kernels_mfma.h"
#include "qoc_mfma_expm.h"
#include "qoc_mfma_expm_stream.h"
#include "qoc_mfma_expm_pair.h"
#include "qoc_mfma_expm_rows.h"

template <int NT>
static inline void qoc_mfma_launch_all_expm(QocMfma& mf, const QocDev& d, hipStream_t s) {
    // AUTO: NT = 2 with at least half of the 1024 SIMDs busy -> one wave per (seed, chunk) on v_mfma_f64_4x4x4 (0.92 vs 1.21 ms
    // per launch at C2 x 64); NT = 1 and small launches keep the 16x16x4 kernel (C1: 0.072 vs 0.074 ms; one C2 trajectory:
    // 0.67 vs 0.75 ms).  qoc_config.variant forces one of the three kernels (parity tests, A/B runs).
    const int v = qoc_mfma_expm_variant(mf, d);
    if constexpr (NT >= 3) {
        if (v == 5) {
            // latency mode of 32 < n <= 64: K_t per slice by the row-block kernel (NT = 3: two workgroups per CU), then the row-split chains
            const size_t lds = qoc_expm_rows_lds<NT>();
            // (active inner strips ceil(n / 4) of the problem padded to 16 NT, as in the batch kernel below)
            const int qa = mf.exp_rows_qa_full ? 4 * NT : (d.n + 3) / 4;
#define QOC_ROWS_SL(KCv, QAv) do { static bool reserved = false; \
                                   if (!reserved) { hipFuncSetAttribute((const void*)k_mfma_expm_rows<NT, KCv, true, QAv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); reserved = true; } \
                                   hipLaunchKernelGGL((k_mfma_expm_rows<NT, KCv, true, QAv>), dim3(d.B * d.steps), dim3(256), lds, s, d, mf); } while (0)
#define QOC_ROWS_SL_QA(KCv) do { if (qa >= 4 * NT) QOC_ROWS_SL(KCv, 4 * NT); else if (qa == 4 * NT - 1) QOC_ROWS_SL(KCv, 4 * NT - 1); \
                                 else if (qa == 4 * NT - 2) QOC_ROWS_SL(KCv, 4 * NT - 2); else QOC_ROWS_SL(KCv, 4 * NT - 3); } while (0)
            if (d.k <= 4) QOC_ROWS_SL_QA(4); else QOC_ROWS_SL_QA(8);
#undef QOC_ROWS_SL_QA
#undef QOC_ROWS_SL
            hipLaunchKernelGGL(k_mfma_chain_rows2<NT>, dim3(d.B * mf.C * 4 * NT), dim3(64 * NT), 0, s, d, mf, (const cplx*)mf.KfD, 1, d.steps, mf.L, mf.PfD, mf.C, (const cplx*)nullptr, mf.PfT);
            hipLaunchKernelGGL(k_mfma_chain_rows2<NT>, dim3(d.B * mf.NG * 4 * NT), dim3(64 * NT), 0, s, d, mf, (const cplx*)mf.PfD, 0, mf.C, mf.G, mf.GfD, mf.NG, (const cplx*)nullptr, mf.GfT);
            return;
        }
    }
    if constexpr (NT >= 3) {
        if (v == 7) {
            const size_t lds = qoc_expm_rows_lds<NT>();
            // active 4-row strips of the problem padded to 16 NT: ceil(n / 4) (4 NT - 3 .. 4 NT); QOC_ROWS_QA_FULL=1 (experimental switch): the padded problem in full
            const int qa = mf.exp_rows_qa_full ? 4 * NT : (d.n + 3) / 4;
#define QOC_ROWS(KCv, QAv) do { static bool reserved = false;                                /* (per instance) */ \
                                if (!reserved) { hipFuncSetAttribute((const void*)k_mfma_expm_rows<NT, KCv, false, QAv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); reserved = true; } \
                                hipLaunchKernelGGL((k_mfma_expm_rows<NT, KCv, false, QAv>), dim3(d.B * mf.C), dim3(256), lds, s, d, mf); } while (0)
#define QOC_ROWS_QA(KCv) do { if (qa >= 4 * NT) QOC_ROWS(KCv, 4 * NT); else if (qa == 4 * NT - 1) QOC_ROWS(KCv, 4 * NT - 1); \
                              else if (qa == 4 * NT - 2) QOC_ROWS(KCv, 4 * NT - 2); else QOC_ROWS(KCv, 4 * NT - 3); } while (0)
            if (d.k <= 4) QOC_ROWS_QA(4); else QOC_ROWS_QA(8);
#undef QOC_ROWS_QA
#undef QOC_ROWS
            return;
        }
    }
    if (v == 5 && NT == 2) {
        // latency mode: K_t by two waves per slice, then the chunk products and the products of groups of G chunks
#define QOC_SL2(QAv) do { if (d.k <= 4) hipLaunchKernelGGL((k_mfma_expm_slice2<4, QAv>), dim3(d.B * d.steps), dim3(128), 0, s, d, mf); \
                          else hipLaunchKernelGGL((k_mfma_expm_slice2<8, QAv>), dim3(d.B * d.steps), dim3(128), 0, s, d, mf); } while (0)
        QOC_QA_SWITCH_LAT(mf.exp_lat_qa8 ? 8 : qoc_active_strips_lat(d.n), QOC_SL2);      // (QOC_LAT_QA8=1, experimental: the padded problem in full)
#undef QOC_SL2
        // (k_mfma_chain_rows2: the columns of the right operand split over the waves of a workgroup -- 9.8 -> 8.5 us per launch at C2)
        hipLaunchKernelGGL(k_mfma_chain_rows2<2>, dim3(d.B * mf.C * 8), dim3(128), 0, s, d, mf, (const cplx*)mf.KfD, 1, d.steps, mf.L, mf.PfD, mf.C, (const cplx*)nullptr, mf.PfT);
        hipLaunchKernelGGL(k_mfma_chain_rows2<2>, dim3(d.B * mf.NG * 8), dim3(128), 0, s, d, mf, (const cplx*)mf.PfD, 0, mf.C, mf.G, mf.GfD, mf.NG, (const cplx*)nullptr, mf.GfT);
    }
    else if (v == 6 && NT == 2) {
        if (d.k <= 4) hipLaunchKernelGGL(k_mfma_expm_pair<4>, dim3(d.B * mf.C), dim3(128), 0, s, d, mf);
        else hipLaunchKernelGGL(k_mfma_expm_pair<8>, dim3(d.B * mf.C), dim3(128), 0, s, d, mf);
    }
    else if (v == 8 && NT == 2) qoc_mfma_launch_expm_inplace(mf, d, s);
    else if (v == 4 && NT == 2) {
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

// final_state = P_{C-1} ... P_0 U0 and unitary_scale (tensorflow_state.py:223-225) for the latency mode, where the sweeps do not
// form them: one chain over the group products and U0, then the unpack.  Called by the engine before a read-back.
__global__ void __launch_bounds__(64) k_mfma_unpack_final(QocDev d, QocMfma mf) {
    const int b = blockIdx.x, lane = threadIdx.x, n = d.n;
    const cplx* T = mf.TfD + (size_t)b * mf.FR;
    cplx* Xf = d.Xfinal + (size_t)b * n * n;
    const int QS = 4 * mf.NT;
    for (int f = 0; f < mf.NT * QS; ++f) {
        const int cb = f / QS, q = f - cb * QS, row = 4 * q + (lane >> 4), col = 16 * cb + (lane & 15);
        if (row < n && col < n) Xf[row * n + col] = T[f * 64 + lane];
    }
    __syncthreads();
    double part = 0.0;
    for (int c = lane; c < n; c += 64) {
        cplx rs = cmake(0.0, 0.0);
        for (int a = 0; a < n; ++a) rs = cadd(rs, Xf[c * n + a]);
        part += rs.x * rs.x + rs.y * rs.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) d.uscale[b] = part / (double)n;
}
void qoc_mfma_final_state(QocMfma& mf, const QocDev& d, hipStream_t s) {
    QocDev dd = d;
    dd.skip_done = 0;                                                     // every seed's last evaluation is still in GfD
    if (mf.NT == 4) hipLaunchKernelGGL(k_mfma_chain_rows<4>, dim3(d.B * 16), dim3(64), 0, s, dd, mf, (const cplx*)mf.GfD, 0, mf.NG, mf.NG, mf.TfD, 1, (const cplx*)mf.U0fD, (cplx*)nullptr);
    else if (mf.NT == 3) hipLaunchKernelGGL(k_mfma_chain_rows<3>, dim3(d.B * 12), dim3(64), 0, s, dd, mf, (const cplx*)mf.GfD, 0, mf.NG, mf.NG, mf.TfD, 1, (const cplx*)mf.U0fD, (cplx*)nullptr);
    else hipLaunchKernelGGL(k_mfma_chain_rows<2>, dim3(d.B * 8), dim3(64), 0, s, dd, mf, (const cplx*)mf.GfD, 0, mf.NG, mf.NG, mf.TfD, 1, (const cplx*)mf.U0fD, (cplx*)nullptr);
    hipLaunchKernelGGL(k_mfma_unpack_final, dim3(d.B), dim3(64), 0, s, dd, mf);
}
