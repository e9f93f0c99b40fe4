// qoc_mfma_latency.hip -- translation unit of the latency-mode sweeps (qoc_mfma_latency.h) and their launchers.
#include <cstdlib>
#include "qoc_kernels_mfma.h"
#include "qoc_mfma_latency.h"

static size_t grad_lat_lds(int kc, int NT) { return (size_t)kc * 256 * NT * NT * sizeof(cplx) + (size_t)16 * 4 * 2 * kc * sizeof(double); }
static int grad_lat_kc(const QocDev& d) { return d.k <= 4 ? 4 : (d.k == 5 ? 5 : 8); }      // control images in LDS (16 KB each at NT = 2, 36 KB at NT = 3)

static const void* grad_lat_kernel(const QocMfma& mf, const QocDev& d) {
    const int kc = grad_lat_kc(d);
    if (mf.NT == 4) return mf.mq <= 2 ? (const void*)k_mfma_grad_lat4<2> : (const void*)k_mfma_grad_lat4<4>;
    if (mf.NT == 3) return mf.mq <= 2 ? (const void*)k_mfma_grad_lat<2, 4, 3> : (const void*)k_mfma_grad_lat<4, 4, 3>;
    if (kc == 8) return mf.mq <= 2 ? (const void*)k_mfma_grad_lat<2, 8> : (const void*)k_mfma_grad_lat<4, 8>;
    if (kc == 5) return mf.mq <= 2 ? (const void*)k_mfma_grad_lat<2, 5> : (const void*)k_mfma_grad_lat<4, 5>;
    return mf.mq <= 2 ? (const void*)k_mfma_grad_lat<2, 4> : (const void*)k_mfma_grad_lat<4, 4>;
}

int qoc_mfma_latency_setup(QocMfma& mf, const QocDev& d, std::string& msg) {
    if (hipFuncSetAttribute(grad_lat_kernel(mf, d), hipFuncAttributeMaxDynamicSharedMemorySize, (int)grad_lat_lds(mf.NT == 4 ? 2 : grad_lat_kc(d), mf.NT)) != hipSuccess) {
        msg = "MFMA path: cannot reserve LDS for the latency-mode gradient kernel";
        return -2;
    }
    return 0;
}

// the chunk offsets of the source recursion come out of the forward sweep itself when the sources need nothing but Psi (undressed forbidden levels, no speed_up)
static inline bool qoc_lat_offsets_in_sweep(const QocMfma& mf, const QocDev& d) {
    return mf.lat_src_fast && !mf.lat_dressed && !d.has_speed && d.n_forb > 0 && !mf.exp_lat_offsets_own;
}

// forward and z-free adjoint sweep side by side: 2 x (seed, chunk, group of 4 columns) workgroups of NT waves (one row tile each)
void qoc_mfma_latency_sweeps(QocMfma& mf, const QocDev& d, hipStream_t s) {
    // (with a state regulariser only the forward half: the costate needs the sources, i.e. the forward states, first)
    // (state regularisers on the batch kernels' recursion: the forward half only; on the thin source sweeps both halves, Lambda0 is used)
    const dim3 g((mf.lat_sources && !mf.lat_src_fast ? 1 : 2) * d.B * mf.C * mf.mq);
    const int co = qoc_lat_offsets_in_sweep(mf, d) ? 1 : 0;           // undressed forbidden levels: the forward role also leaves the chunk offsets of the source recursion
    if (mf.NT == 4) hipLaunchKernelGGL(k_mfma_sweep_lat<4>, g, dim3(256), 0, s, d, mf, co);
    else if (mf.NT == 3) hipLaunchKernelGGL(k_mfma_sweep_lat<3>, g, dim3(192), 0, s, d, mf, co);
    else hipLaunchKernelGGL(k_mfma_sweep_lat<2>, g, dim3(128), 0, s, d, mf, co);
    if (mf.lat_src_fast) {                                             // fidelity + state-regulariser values straight from PsiL (instead of unpack + k_loss)
        const dim3 gl(d.B * ((d.steps + 1 + 15) / 16));
#define QOC_LOSS(NTv) do { if (mf.lat_dressed) hipLaunchKernelGGL((k_mfma_loss_lat<NTv, true>), gl, dim3(1024), 0, s, d, mf); \
                           else hipLaunchKernelGGL((k_mfma_loss_lat<NTv, false>), gl, dim3(1024), 0, s, d, mf); } while (0)
        if (mf.NT == 4) QOC_LOSS(4); else if (mf.NT == 3) QOC_LOSS(3); else QOC_LOSS(2);
#undef QOC_LOSS
    }
    else if (mf.lat_sources) qoc_mfma_unpack_inter(mf, d, s);         // k_loss, the sources and the batch backward kernels read d.inter
}

// ap != nullptr: the tail of the iteration (k_finish_t<true>) runs inside, in the last workgroup of each seed
// (no bandpass regulariser: its DFT stays in the separate k_finish_t<false>)
void qoc_mfma_latency_gradient(QocMfma& mf, const QocDev& d, const QocAdamDev* ap, hipStream_t s) {
    if (mf.lat_src_fast) {
        // the source part of the costate: chunk offsets, group offsets, then the sweep that stores the total costate (k_loss has run)
        const dim3 gc(d.B * mf.C * mf.mq), gg(d.B * mf.NG * mf.mq), bs(64 * mf.NT);
        const bool have_offsets = qoc_lat_offsets_in_sweep(mf, d);     // (k_mfma_sweep_lat has left them)
#define QOC_SRC1(NTv, DRv) do { if (!have_offsets) hipLaunchKernelGGL((k_mfma_sweep_src<NTv, DRv>), gc, bs, 0, s, d, mf, 0); hipLaunchKernelGGL((k_mfma_sweep_src<NTv, DRv>), gg, bs, 0, s, d, mf, 1); \
                          hipLaunchKernelGGL((k_mfma_sweep_src<NTv, DRv>), gc, bs, 0, s, d, mf, 2); } while (0)
#define QOC_SRC(NTv) do { if (mf.lat_dressed) QOC_SRC1(NTv, true); else QOC_SRC1(NTv, false); } while (0)
        if (mf.NT == 4) QOC_SRC(4); else if (mf.NT == 3) QOC_SRC(3); else QOC_SRC(2);
#undef QOC_SRC1
#undef QOC_SRC
    }
    const int kc = grad_lat_kc(d), sl = 16 / mf.NT;                      // slices per workgroup, NT waves (row tiles) each
    const dim3 g(d.B * ((d.steps + sl - 1) / sl)), b(64 * sl * mf.NT);
    const size_t lds = grad_lat_lds(kc, mf.NT);
    const QocAdamDev a = ap ? *ap : QocAdamDev{};
    const bool local_regs = d.has_amp || d.has_env || d.has_dwdt || d.has_d2wdt2;
    const int fuse = (ap ? 1 : 0) | (mf.lat_src_fast ? 2 : 0) | (local_regs ? 4 : 0);
    if (mf.NT == 4) {
        const size_t lds4 = grad_lat_lds(2, 4);
        if (mf.mq <= 2) hipLaunchKernelGGL(k_mfma_grad_lat4<2>, g, b, lds4, s, d, mf, a, fuse);
        else hipLaunchKernelGGL(k_mfma_grad_lat4<4>, g, b, lds4, s, d, mf, a, fuse);
        return;
    }
    if (mf.NT == 3) {
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad_lat<2, 4, 3>), g, b, lds, s, d, mf, a, fuse);
        else hipLaunchKernelGGL((k_mfma_grad_lat<4, 4, 3>), g, b, lds, s, d, mf, a, fuse);
        return;
    }
#define QOC_GL(KCv) do { if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad_lat<2, KCv>), g, b, lds, s, d, mf, a, fuse); \
                         else hipLaunchKernelGGL((k_mfma_grad_lat<4, KCv>), g, b, lds, s, d, mf, a, fuse); } while (0)
    if (kc == 8) QOC_GL(8); else if (kc == 5) QOC_GL(5); else QOC_GL(4);
#undef QOC_GL
}
