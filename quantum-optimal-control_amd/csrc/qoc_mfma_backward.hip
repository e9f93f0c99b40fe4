// qoc_mfma_backward.hip -- translation unit of the MFMA-path backward sweeps / gradient kernels (qoc_mfma_backward.h), their
// launcher, and the path set-up (it reserves LDS for these kernels).
#include "qoc_kernels_mfma.h"
#include "qoc_plan_limits.h"
#include "qoc_mfma_backward.h"
#include "qoc_mfma_downup.h"

// k_mfma_downup for this problem: 4 or 5 control images in LDS, active 4-row strips of the padded propagators = ceil(n / 4)
static const void* qoc_downup_kernel(const QocDev& d) {
    const int qa = qoc_active_strips(d.n);
#define QOC_DU(KCv) (qa == 8 ? (const void*)k_mfma_downup<2, KCv, 8> : qa == 7 ? (const void*)k_mfma_downup<2, KCv, 7> : \
                     qa == 6 ? (const void*)k_mfma_downup<2, KCv, 6> : (const void*)k_mfma_downup<2, KCv, 5>)
    return d.k == 5 ? QOC_DU(5) : QOC_DU(4);
#undef QOC_DU
}

// the batch sweep k_mfma_backward3<MODE 0>; with a state regulariser (the route that takes it by default) on the active strips of K
static const void* qoc_backward3_kernel(const QocMfma& mf, const QocDev& d) {
    const bool src = d.n_forb > 0 || d.has_speed;
    const int qa = qoc_active_strips(d.n);
#define QOC_B3S(MQv, KCv) (qa == 8 ? (const void*)k_mfma_backward3<MQv, true, KCv, 0, 8> : qa == 7 ? (const void*)k_mfma_backward3<MQv, true, KCv, 0, 7> : \
                           qa == 6 ? (const void*)k_mfma_backward3<MQv, true, KCv, 0, 6> : (const void*)k_mfma_backward3<MQv, true, KCv, 0, 5>)
#define QOC_B3K(MQv, KCv) (src ? QOC_B3S(MQv, KCv) : (const void*)k_mfma_backward3<MQv, false, KCv>)
    if (d.k == 5) return mf.mq <= 2 ? QOC_B3K(2, 5) : QOC_B3K(4, 5);
    return mf.mq <= 2 ? QOC_B3K(2, 4) : QOC_B3K(4, 4);
#undef QOC_B3K
#undef QOC_B3S
}

int qoc_mfma_setup(QocMfma& mf, const QocDev& d, int chunks_req, const cplx* Hs_host,
                                 std::vector<void*>& allocs, std::string& msg) {
    // (latency mode: NT = 2 kernels only -- a smaller problem is padded to 32: one trajectory of n = 16 runs 0.083 ms like n = 17, not 0.20)
    // (latency mode of 32 < n <= 48: NT = 3 with up to four controls, the NT = 4 kernels on the problem padded to 64 with more)
    const int NT = (mf.variant == 5 && d.n <= 32) ? 2 : d.n <= 16 ? 1 : (d.n <= 32 ? 2 : ((d.n <= 48 && !(mf.variant == 5 && d.k > 4)) ? 3 : 4));
    const int FR = 256 * NT * NT;
    mf.NT = NT; mf.FR = FR;
    // both sweeps in one kernel, a wave per half chunk (k_mfma_downup; needs the z-free chunk boundaries of the scan: bnd_adj below, and
    // m <= 8: its exchange buffers and images share the LDS with the control images)
    mf.updown = NT == 2 && mf.variant != 1 && mf.variant != 5 && d.k <= 5 && d.m <= 8 && !(d.n_forb > 0 || d.has_speed) &&
                !qoc_exp_is("QOC_UPDOWN", 0);
    mf.exp_rows_qa_full = qoc_exp_env("QOC_ROWS_QA_FULL") && !qoc_exp_is("QOC_ROWS_QA_FULL", 0);
    mf.exp_lat_qa8 = qoc_exp_env("QOC_LAT_QA8") && !qoc_exp_is("QOC_LAT_QA8", 0);
    mf.exp_lat_offsets_own = qoc_exp_is("QOC_LAT_OFFSETS_IN_SWEEP", 0);
    int C = chunks_req;
    if (C <= 0) {
        // NT = 2: the default exponential kernel is ONE wave of 444 VGPRs per (seed, chunk), i.e. at most one resident wave per
        // SIMD: B*C must not exceed the 1024 SIMDs or a second, nearly empty round doubles the launch (48 seeds: C = 22 ->
        // 1056 items, 24.0k it/s; C = 21 -> 1008 items, 39.6k it/s).  Other NT: ~2 waves per SIMD.
        // Fewer than 32 seeds (NT = 2): chunks down to 8 slices keep all SIMDs busy -- 16 seeds: 0.50 ms per iteration with 63 chunks against
        // 0.62 with 32; 20 seeds: 0.52 (50 chunks) against 0.63; 24 seeds: 0.58 (42) against 0.64 -- the longer boundary recursion of the
        // sweeps costs less than the idle SIMDs of the exponential kernel.
        C = NT == 2 ? QOC_PLAN_CHUNK_ITEMS / d.Bplan : (QOC_PLAN_CHUNK_ITEMS + d.Bplan - 1) / d.Bplan;          // (the PLANNED batch: a shard of it chunks like the whole)
        if (C > (NT == 2 ? QOC_PLAN_CHUNKS_MAX_NT2 : QOC_PLAN_CHUNKS_MAX)) C = NT == 2 ? QOC_PLAN_CHUNKS_MAX_NT2 : QOC_PLAN_CHUNKS_MAX;
    }
    // latency mode (variant 5; AUTO for a handful of seeds, qoc_mfma_latency_ok): one wave per SLICE for the exponentials, short
    // chunks whose products come from k_mfma_chain_products, groups of G chunks for two-level chunk boundaries in the sweeps
    mf.latency = mf.variant == 5;
    if (mf.latency && chunks_req <= 0) C = (d.steps + 7) / 8;   // chunks of 8 slices
    if (C > d.steps) C = d.steps;
    if (C > (mf.latency ? 16 * QOC_MAXC : QOC_MAXC)) C = mf.latency ? 16 * QOC_MAXC : QOC_MAXC;
    if (C < 1) C = 1;
    int L = (d.steps + C - 1) / C;
    C = (d.steps + L - 1) / L;                       // no empty chunks
    mf.C = C; mf.L = L;
    mf.G = 0; mf.NG = 0;
    mf.lat_sources = mf.latency && (d.n_forb > 0 || d.has_speed);
    if (mf.latency) {                              // groups of ~sqrt(C) chunks, at least 8: the boundary walks are <= (NG - 1) + (G - 1) thin products
        int G = 8;
        while (G * G < C) ++G;
        mf.G = G; mf.NG = (C + G - 1) / G;
    }
    mf.mq = (d.m + 3) / 4;
    {
        double f = 1.0;
        for (int j = 0; j < 24; ++j) { if (j > 0) f *= (double)j; mf.invfact[j] = 1.0 / f; }
    }
    std::vector<cplx> hd((size_t)(d.k + 1) * FR), ht((size_t)(d.k + 1) * FR), u0(FR);
    for (int kk = 0; kk <= d.k; ++kk) {
        qoc_to_fragD(Hs_host + (size_t)kk * d.n * d.n, d.n, false, hd.data() + (size_t)kk * FR, NT);
        qoc_to_fragD(Hs_host + (size_t)kk * d.n * d.n, d.n, true, ht.data() + (size_t)kk * FR, NT);
    }
    std::vector<cplx> u0h((size_t)d.n * d.n);
    if (hipMemcpy(u0h.data(), d.U0, u0h.size() * sizeof(cplx), hipMemcpyDeviceToHost) != hipSuccess) { msg = "U0 readback failed"; return -2; }
    qoc_to_fragD(u0h.data(), d.n, false, u0.data(), NT);
    auto up = [&](cplx** dst, const std::vector<cplx>& src) -> bool {
        void* p = nullptr;
        if (hipMalloc(&p, src.size() * sizeof(cplx)) != hipSuccess) return false;
        allocs.push_back(p);
        if (hipMemcpy(p, src.data(), src.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) return false;
        *dst = (cplx*)p;
        return true;
    };
    if (!up(&mf.HfD, hd) || !up(&mf.HfT, ht) || !up(&mf.U0fD, u0)) { msg = "MFMA path: constant upload failed"; return -3; }
    {
        // k_mfma_expm_inplace evaluates the Taylor polynomial in the scaled variable S = sigma A with sigma^T = 1/T!: the polynomial is then
        // monic, its Horner start X = S + c I costs nothing but the diagonal, and sigma / 2^s is folded into a scaled copy of the Hamiltonians
        const int T = d.T < 1 ? 1 : (d.T > 22 ? 22 : d.T);
        mf.psigma = std::pow(mf.invfact[T], 1.0 / (double)T);
        for (int j = 0; j < 24; ++j) mf.pcoef[j] = mf.invfact[j] * std::pow(mf.psigma, -(double)j);
        mf.pcoef[T] = 1.0;
        const double sc = mf.psigma / (double)(1 << d.s);
        std::vector<cplx> hs(hd);
        for (auto& v : hs) { v.x *= sc; v.y *= sc; }
        if (!up(&mf.HsD, hs)) { msg = "MFMA path: constant upload failed"; return -3; }
    }
    // the large buffers are carved out of ONE allocation (placement of separate hipMallocs after earlier engines of the process
    // were freed was worth a factor 2 on the GEMM path)
    std::vector<std::pair<cplx**, size_t>> wanted;
    auto al = [&](cplx** dst, size_t count) -> bool { wanted.emplace_back(dst, (count * sizeof(cplx) + 4095) & ~(size_t)4095); return true; };
    mf.skew_c = 5 * 16;                            // 1280 B per chunk
    mf.skew_b = 3 * 16;                            //  768 B per seed
    const size_t nk = (size_t)d.B * ((size_t)d.steps * FR + (size_t)C * mf.skew_c + mf.skew_b), np = (size_t)d.B * C * FR;
    mf.store_T = !((NT == 2 || NT == 3) && mf.variant != 1) || mf.latency;     // the 4x4x4 forward sweep gathers K^T operands from fragD(K); latency mode reads transposed copies
    const bool split_grad = (NT > 2 || (NT == 2 && d.k >= 6)) && mf.variant != 1;   // k <= 5: backward3 (5 images still fit next to its pads)
    if (split_grad && !al(&mf.LamD, (size_t)d.B * d.steps * 16 * NT * 16)) { msg = "MFMA path: out of device memory"; return -3; }
    { const int kg = NT >= 4 ? 2 : 4; mf.grad_lds = (size_t)(d.k < kg ? d.k : kg) * FR * sizeof(cplx); }
    mf.grad_rt = NT >= 2 && split_grad && d.k <= 8 && !qoc_exp_is("QOC_GRAD_RT", 0);
    if (mf.grad_rt) {
        mf.grad_lds = (size_t)((d.k + 3) & ~3) * NT * 256 * sizeof(cplx);
        if (!al((cplx**)&mf.gpart, ((size_t)NT * d.B * d.k * d.steps + 1) / 2)) { msg = "MFMA path: out of device memory"; return -3; }
    }
    if (mf.updown && !al(&mf.LamL, (size_t)d.B * d.steps * NT * (mf.mq <= 2 ? 2 : 4) * 64)) { msg = "MFMA path: out of device memory"; return -3; }
    if (mf.lat_sources && !al(&mf.Goff, (size_t)d.B * mf.NG * 4 * NT * 64)) { msg = "MFMA path: out of device memory"; return -3; }
    // chunk boundaries once per seed (k_mfma_bnd_scan) for the 4x4x4 batch sweeps; the adjoint ones only where k_mfma_backward3 takes them
    // (NT = 2, k <= 5, no state regulariser: the costate is then linear in the overlap)
    const bool bnd = (NT == 2 || NT == 3) && mf.variant != 1 && !mf.latency;
    const bool bnd_adj = bnd && NT == 2 && d.k <= 5 && !(d.n_forb > 0 || d.has_speed);
    mf.updown = mf.updown && bnd_adj;
    const size_t nbnd = (size_t)d.B * C * NT * (mf.mq <= 2 ? 2 : 4) * 64;
    if ((bnd && !al(&mf.BndF, nbnd)) || (bnd_adj && !al(&mf.BndA, nbnd))) { msg = "MFMA path: out of device memory"; return -3; }
    // forbidden levels / speed_up on the thin affine sweeps of qoc_mfma_latency.h; dressed levels (up to 4 of them) take their sources from Fd
    const bool dressed = d.n_forb > 0 && d.forbid_dressed;
    mf.lat_src_fast = mf.lat_sources && !(dressed && d.n_forb > 4);
    mf.lat_dressed = mf.lat_src_fast && dressed;
    if (mf.lat_src_fast) {
        const size_t vec = (size_t)NT * (mf.mq <= 2 ? 2 : 4) * 64;
        if (!al(&mf.AoffL, (size_t)d.B * C * vec) || !al(&mf.GoffL, (size_t)d.B * mf.NG * vec) || !al(&mf.LamS, (size_t)d.B * d.steps * vec) ||
            !al((cplx**)&mf.loss_part, (size_t)d.B * (d.steps + 1))) { msg = "MFMA path: out of device memory"; return -3; }   // (loss_part: two doubles per time point)
    }
    if (mf.latency && (!al(&mf.GfD, (size_t)d.B * mf.NG * FR) || !al(&mf.GfT, (size_t)d.B * mf.NG * FR) || !al(&mf.TfD, (size_t)d.B * FR) ||
                       !al(&mf.PsiL, (size_t)d.B * d.steps * NT * (mf.mq <= 2 ? 2 : 4) * 64) ||
                       !al(&mf.LamL, (size_t)d.B * d.steps * NT * (mf.mq <= 2 ? 2 : 4) * 64))) { msg = "MFMA path: out of device memory"; return -3; }
    if (!al(&mf.KfD, nk) || (mf.store_T && !al(&mf.KfT, nk)) || !al(&mf.PfD, np) || !al(&mf.PfT, np) || !al(&mf.Aoff, (size_t)d.B * C * 4 * NT * 64)) { msg = "MFMA path: out of device memory"; return -3; }
    {
        size_t total = 0;
        for (auto& w : wanted) total += w.second;
        char* arena = nullptr;
        if (hipMalloc((void**)&arena, qoc_arena_bytes(total)) != hipSuccess) { msg = "MFMA path: out of device memory"; return -3; }
        allocs.push_back(arena);
        size_t off = 0;
        for (auto& w : wanted) { *w.first = (cplx*)(arena + off); off += w.second; }
        // column groups beyond m are never written by the sweeps and must read as zero in the gradient kernel
        const size_t xl = (size_t)d.B * d.steps * NT * (mf.mq <= 2 ? 2 : 4) * 64 * sizeof(cplx);
        if (mf.PsiL && (hipMemset(mf.PsiL, 0, xl) != hipSuccess || hipMemset(mf.LamL, 0, xl) != hipSuccess)) { msg = "MFMA path: clearing PsiL / LamL failed"; return -2; }
        // k_mfma_expm_inplace never writes the all-zero 4-row strips of a padded propagator (rows >= 4 ceil(n / 4)): they read as zero from here on
        if (NT == 2 && d.n <= 28 && (hipMemset(mf.KfD, 0, nk * sizeof(cplx)) != hipSuccess || (mf.KfT && hipMemset(mf.KfT, 0, nk * sizeof(cplx)) != hipSuccess))) {
            msg = "MFMA path: clearing the propagator buffers failed"; return -2;
        }
    }
    const size_t pads = (size_t)4 * 16 * (16 * NT + 1) * sizeof(cplx);
    const size_t hbytes = (size_t)d.k * FR * sizeof(cplx);
    mf.h_in_lds = (hbytes + pads) <= 160 * 1024;
    mf.bwd_lds = pads + (mf.h_in_lds ? hbytes : 0);
    // prefetching row-split kernel (NT = 2, k <= 5): 4 or 5 control images + the pads + row partials
    {
        const int kc = d.k == 5 ? 5 : 4;
        mf.bwd_lds3 = (size_t)kc * FR * sizeof(cplx) + (size_t)8 * 2 * 16 * B2_LDP * sizeof(cplx) + 4 * 2 * 2 * 4 * kc * sizeof(double);
    }
    const void* b3k = qoc_backward3_kernel(mf, d);
    if (mf.latency) {
        if (qoc_mfma_latency_setup(mf, d, msg) != 0) return -2;
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)d.B * sizeof(unsigned)) != hipSuccess) { msg = "MFMA path: out of device memory"; return -3; }
        allocs.push_back(p);
        if (hipMemset(p, 0, (size_t)d.B * sizeof(unsigned)) != hipSuccess) { msg = "MFMA path: clearing the arrival counters failed"; return -2; }
        mf.lat_count = (unsigned*)p;
    }
    if (NT == 2 && mf.lat_sources && d.k <= 5) {
#define QOC_B3L(MQv, KCv, MODEv) (const void*)k_mfma_backward3<MQv, true, KCv, MODEv>
        const void* k1 = d.k == 5 ? (mf.mq <= 2 ? QOC_B3L(2, 5, 1) : QOC_B3L(4, 5, 1)) : (mf.mq <= 2 ? QOC_B3L(2, 4, 1) : QOC_B3L(4, 4, 1));
        const void* k2 = d.k == 5 ? (mf.mq <= 2 ? QOC_B3L(2, 5, 2) : QOC_B3L(4, 5, 2)) : (mf.mq <= 2 ? QOC_B3L(2, 4, 2) : QOC_B3L(4, 4, 2));
#undef QOC_B3L
        if (hipFuncSetAttribute(k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess ||
            hipFuncSetAttribute(k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess) {
            msg = "MFMA path: cannot reserve LDS for the two-level backward kernel";
            return -2;
        }
    }
    if (NT == 2 && mf.updown) {
        const int kc = d.k == 5 ? 5 : 4, mqv = mf.mq <= 2 ? 2 : 4;
        mf.du_lds = (size_t)kc * FR * sizeof(cplx) + (size_t)8 * 4 * mqv * F2_LDP * sizeof(cplx) + (size_t)4 * 2 * NT * mqv * 64 * sizeof(cplx) +
                    (size_t)8 * 4 * kc * sizeof(double) + 8 * sizeof(int);          // control images, wave images, exchange buffers, row partials, flags
        if (hipFuncSetAttribute(qoc_downup_kernel(d), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.du_lds) != hipSuccess) {
            msg = "MFMA path: cannot reserve LDS for the fused sweep kernel";
            return -2;
        }
    }
    if (NT == 2 && bnd_adj) {
        const void* k3 = d.k == 5 ? (mf.mq <= 2 ? (const void*)k_mfma_backward3<2, false, 5, 3> : (const void*)k_mfma_backward3<4, false, 5, 3>)
                                  : (mf.mq <= 2 ? (const void*)k_mfma_backward3<2, false, 4, 3> : (const void*)k_mfma_backward3<4, false, 4, 3>);
        if (hipFuncSetAttribute(k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess) {
            msg = "MFMA path: cannot reserve LDS for the backward kernel with precomputed boundaries";
            return -2;
        }
    }
    if (NT == 2 && hipFuncSetAttribute(b3k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds3) != hipSuccess) {
        msg = "MFMA path: cannot reserve LDS for the prefetching backward kernel";
        return -2;
    }
    if (split_grad) {
        const void* gk = NT == 2 ? (mf.mq <= 2 ? (const void*)k_mfma_grad<2, 2> : (const void*)k_mfma_grad<2, 4>)
                       : NT == 3 ? (mf.mq <= 2 ? (const void*)k_mfma_grad<3, 2> : (const void*)k_mfma_grad<3, 4>)
                                 : (mf.mq <= 2 ? (const void*)k_mfma_grad<4, 2> : (const void*)k_mfma_grad<4, 4>);
        if (mf.grad_rt)
            gk = NT == 2 ? (mf.mq <= 2 ? (const void*)k_mfma_grad_rt<2, 2> : (const void*)k_mfma_grad_rt<2, 4>)
               : NT == 3 ? (mf.mq <= 2 ? (const void*)k_mfma_grad_rt<3, 2> : (const void*)k_mfma_grad_rt<3, 4>)
                         : (mf.mq <= 2 ? (const void*)k_mfma_grad_rt<4, 2> : (const void*)k_mfma_grad_rt<4, 4>);
        if (hipFuncSetAttribute(gk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.grad_lds) != hipSuccess) { msg = "MFMA path: cannot reserve LDS for the gradient kernel"; return -2; }
    }
    if (mf.h_in_lds) {
        const hipError_t e1 = hipFuncSetAttribute((const void*)k_mfma_backward<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        const hipError_t e2 = hipFuncSetAttribute((const void*)k_mfma_backward<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        const hipError_t e3 = hipFuncSetAttribute((const void*)k_mfma_backward<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        const hipError_t e4 = hipFuncSetAttribute((const void*)k_mfma_backward<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf.bwd_lds);
        if ((NT == 1 ? e1 : (NT == 2 ? e2 : (NT == 3 ? e3 : e4))) != hipSuccess) {
            msg = "MFMA path: cannot reserve LDS for the backward kernel";
            return -2;
        }
    }
    return 0;
}

template <int NT>
static inline void qoc_mfma_launch_all_backward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    const int items = d.B * mf.C;
    if ((d.n_forb > 0 || d.has_speed) && mf.C > 1) {
        if (NT == 2 && mf.variant != 1) {
            const int wpg = mf.lat_sources ? 1 : 4;                      // latency mode: one sweep per workgroup, i.e. per CU
#define QOC_O2(QAv) do { if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_bwd_offsets2<2, false, QAv>), dim3((items + wpg - 1) / wpg), dim3(64 * wpg), 0, s, d, mf); \
                         else hipLaunchKernelGGL((k_mfma_bwd_offsets2<4, false, QAv>), dim3((items + wpg - 1) / wpg), dim3(64 * wpg), 0, s, d, mf); } while (0)
            QOC_QA_SWITCH(qoc_active_strips(d.n), QOC_O2);
#undef QOC_O2
        } else {
            hipLaunchKernelGGL(k_mfma_bwd_offsets<NT>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        }
    }
    // NT = 2: k <= 5 -> k_mfma_backward3 (pair of waves per item, control images in LDS); k >= 6 -> costate sweep + k_mfma_grad.
    // The choice must not depend on the batch size: the gradient sums associate differently between the kernels, and a seed has
    // to evolve bit-identically whatever batch / GPU it is sharded into.  variant 1 keeps the one-wave 16x16x4 kernel (A/B).
    if (NT == 2 && mf.variant != 1) {
        if (d.k <= 5) {
            const bool src = d.n_forb > 0 || d.has_speed;
            if (mf.lat_sources) {
                // latency mode with a state regulariser: group offsets first, then the sweep with two-level affine boundaries; one pair of
                // waves per workgroup (per CU)
                const dim3 gg(d.B * mf.NG), gc(items), b1(128);
#define QOC_B3L(MQv, MODEv, GRID) do { if (d.k == 5) hipLaunchKernelGGL((k_mfma_backward3<MQv, true, 5, MODEv>), GRID, b1, mf.bwd_lds3, s, d, mf); \
                                       else hipLaunchKernelGGL((k_mfma_backward3<MQv, true, 4, MODEv>), GRID, b1, mf.bwd_lds3, s, d, mf); } while (0)
                if (mf.mq <= 2) { QOC_B3L(2, 2, gg); QOC_B3L(2, 1, gc); } else { QOC_B3L(4, 2, gg); QOC_B3L(4, 1, gc); }
#undef QOC_B3L
                return;
            }
            const dim3 g3((items + 3) / 4), b3(512);                     // 4 pairs of waves per workgroup
#define QOC_B3B(MQv) do { if (d.k == 5) hipLaunchKernelGGL((k_mfma_backward3<MQv, false, 5, 3>), g3, b3, mf.bwd_lds3, s, d, mf); \
                          else hipLaunchKernelGGL((k_mfma_backward3<MQv, false, 4, 3>), g3, b3, mf.bwd_lds3, s, d, mf); } while (0)
            if (mf.updown && !src) {
                const dim3 gd((items + 3) / 4), bd(512);                  // 4 items per workgroup, a wave per half chunk
                void* kargs[] = {(void*)&d, (void*)&mf};
                (void)hipLaunchKernel(qoc_downup_kernel(d), gd, bd, kargs, mf.du_lds, s);
            }
            else if (mf.BndA && !src) { if (mf.mq <= 2) QOC_B3B(2); else QOC_B3B(4); }
            else {
                void* kargs[] = {(void*)&d, (void*)&mf};
                (void)hipLaunchKernel(qoc_backward3_kernel(mf, d), g3, b3, kargs, mf.bwd_lds3, s);
            }
#undef QOC_B3B
            return;
        }
        {
            // k >= 6: the control images fit in LDS next to no sweep's pads; costate sweep + slice-parallel gradient kernel (4 images
            // per pass) instead of the row-split 16x16x4 sweep reading them from L2 (C2 x 64 with k = 8: 1.55 vs 1.71 ms per iteration)
#define QOC_O2F(QAv) do { if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_bwd_offsets2<2, true, QAv>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf); \
                          else hipLaunchKernelGGL((k_mfma_bwd_offsets2<4, true, QAv>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf); } while (0)
            QOC_QA_SWITCH(qoc_active_strips(d.n), QOC_O2F);
#undef QOC_O2F
            const int slices = d.B * d.steps;
            int gg = (slices + 3) / 4; if (gg > 2048) gg = 2048;
            if (mf.grad_rt) {
                if (gg > 512) gg = 512;
                if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad_rt<2, 2>), dim3(2 * gg), dim3(256), mf.grad_lds, s, d, mf);
                else hipLaunchKernelGGL((k_mfma_grad_rt<2, 4>), dim3(2 * gg), dim3(256), mf.grad_lds, s, d, mf);
                hipLaunchKernelGGL(k_mfma_grad_sum, dim3(256), dim3(256), 0, s, d, mf, 2);
            }
            else if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad<2, 2>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
            else hipLaunchKernelGGL((k_mfma_grad<2, 4>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
        }
        return;
    }
    if (NT > 2 && mf.variant != 1) {
        // n > 32: the sweep only propagates the costates, the gradients are formed slice-parallel with the control images in LDS
        // (no pad, no images; on the active strips ceil(n / 4) of the problem padded to 16 NT)
#define QOC_BWS(QAv) hipLaunchKernelGGL((k_mfma_backward<NT, false, true, QAv>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf, 0)
        switch (4 * NT - (d.n + 3) / 4) { case 3: QOC_BWS(4 * NT - 3); break; case 2: QOC_BWS(4 * NT - 2); break; case 1: QOC_BWS(4 * NT - 1); break; default: QOC_BWS(4 * NT); break; }
#undef QOC_BWS
        const int slices = d.B * d.steps;
        int gg = (slices + 3) / 4; if (gg > 1024) gg = 1024;
        constexpr int GN = NT > 2 ? NT : 3;                                  // (never launched for NT <= 2)
        if (mf.grad_rt) {
            int g2 = (slices + 3) / 4; if (g2 > 512) g2 = 512;
            if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad_rt<GN, 2>), dim3(GN * g2), dim3(256), mf.grad_lds, s, d, mf);
            else hipLaunchKernelGGL((k_mfma_grad_rt<GN, 4>), dim3(GN * g2), dim3(256), mf.grad_lds, s, d, mf);
            hipLaunchKernelGGL(k_mfma_grad_sum, dim3(256), dim3(256), 0, s, d, mf, GN);
            return;
        }
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_grad<GN, 2>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_grad<GN, 4>), dim3(gg), dim3(256), mf.grad_lds, s, d, mf);
        return;
    }
    if (mf.h_in_lds)
        hipLaunchKernelGGL((k_mfma_backward<NT, true>), dim3((items + 3) / 4), dim3(256), mf.bwd_lds, s, d, mf, 0);
    else
        hipLaunchKernelGGL((k_mfma_backward<NT, false>), dim3((items + 3) / 4), dim3(256), mf.bwd_lds, s, d, mf, 0);
}
void qoc_mfma_launch_backward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    if (mf.latency && (!mf.lat_sources || mf.lat_src_fast)) { qoc_mfma_latency_gradient(mf, d, nullptr, s); return; }
    if (mf.NT == 1) qoc_mfma_launch_all_backward<1>(mf, d, s); else if (mf.NT == 2) qoc_mfma_launch_all_backward<2>(mf, d, s); else if (mf.NT == 3) qoc_mfma_launch_all_backward<3>(mf, d, s); else qoc_mfma_launch_all_backward<4>(mf, d, s);
}

