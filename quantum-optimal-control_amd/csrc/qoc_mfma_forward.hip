// qoc_mfma_forward.hip -- translation unit of the MFMA-path forward sweeps (qoc_mfma_forward.h) and their launcher.
#include "qoc_kernels_mfma.h"
#include "qoc_mfma_forward.h"

// the NT = 2 sweep from the chunk boundaries of the scan, on the active column groups of K
static void qoc_launch_forward2_bnd(QocMfma& mf, const QocDev& d, int sw, hipStream_t s) {
#define QOC_F2(QAv) do { if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_forward2<2, 2, true, QAv>), dim3((sw + 3) / 4), dim3(256), 0, s, d, mf); \
                         else hipLaunchKernelGGL((k_mfma_forward2<2, 4, true, QAv>), dim3((sw + 3) / 4), dim3(256), 0, s, d, mf); } while (0)
    QOC_QA_SWITCH(qoc_active_strips(d.n), QOC_F2);
#undef QOC_F2
}

void qoc_mfma_launch_forward(QocMfma& mf, const QocDev& d, hipStream_t s) {
    if (mf.latency) { qoc_mfma_latency_sweeps(mf, d, s); return; }       // (final_state only when read back: qoc_mfma_final_state)
    const int items = d.B * mf.C + d.B * mf.NT;
    if (mf.BndF) {
        // chunk boundaries once per seed (forward, and the z-free adjoint ones when the backward sweep takes them), then the sweep
        const bool no_final = mf.updown || d.state_transfer;            // (state transfer has no final_state: tensorflow_state.py:244-261)
        const int MQ = mf.mq <= 2 ? 2 : 4, waves = d.B * ((mf.BndA ? 2 : 1) * MQ + (no_final ? 0 : 4 * mf.NT));     // + the column blocks of final_state
        const int flags = (mf.BndA ? 1 : 0) | (mf.updown ? 2 : 0) | (no_final ? 8 : 0);
        if (mf.NT == 2) hipLaunchKernelGGL(k_mfma_bnd_scan<2>, dim3((waves + 3) / 4), dim3(256), 0, s, d, mf, MQ, flags, (const cplx*)mf.PfT, (const cplx*)mf.PfD, mf.C);
        else hipLaunchKernelGGL(k_mfma_bnd_scan<3>, dim3((waves + 3) / 4), dim3(256), 0, s, d, mf, MQ, flags, (const cplx*)mf.PfT, (const cplx*)mf.PfD, mf.C);
        const int sw = d.B * mf.C;                                                // sweep items only: final_state comes from the scan
        if (mf.updown) {
            // the forward sweep runs inside k_mfma_downup (after the adjoint one); Psi_N for the loss came from the scan
        } else if (mf.NT == 2) {
            qoc_launch_forward2_bnd(mf, d, sw, s);
        } else {
            // (33 <= n <= 48: the active 4-column groups of the K padded to 48, ceil(n / 4) = 9 .. 12)
#define QOC_F3(QAv) do { if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_forward2<3, 2, true, QAv>), dim3((sw + 3) / 4), dim3(256), 0, s, d, mf); \
                         else hipLaunchKernelGGL((k_mfma_forward2<3, 4, true, QAv>), dim3((sw + 3) / 4), dim3(256), 0, s, d, mf); } while (0)
            switch ((d.n + 3) / 4) { case 9: QOC_F3(9); break; case 10: QOC_F3(10); break; case 11: QOC_F3(11); break; default: QOC_F3(12); break; }
#undef QOC_F3
        }
    }
    else if (mf.NT == 1) hipLaunchKernelGGL(k_mfma_forward<1>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    else if (mf.NT == 2 && mf.variant != 1) {
        // 4x4x4 sweep; like the backward choice this must not depend on the batch size (bit-identical seeds across shardings)
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_forward2<2, 2>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_forward2<2, 4>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    }
    else if (mf.NT == 3 && mf.variant != 1) {
        if (mf.mq <= 2) hipLaunchKernelGGL((k_mfma_forward2<3, 2>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
        else hipLaunchKernelGGL((k_mfma_forward2<3, 4>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    }
    else if (mf.NT == 2) hipLaunchKernelGGL(k_mfma_forward<2>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    else if (mf.NT == 3) hipLaunchKernelGGL(k_mfma_forward<3>, dim3((items + 3) / 4), dim3(256), 0, s, d, mf);
    else {
        // NT = 4: the active column groups ceil(n / 4) = 13 .. 16 of the K padded to 64
#define QOC_F4(QAv) hipLaunchKernelGGL((k_mfma_forward<4, QAv>), dim3((items + 3) / 4), dim3(256), 0, s, d, mf)
        switch ((d.n + 3) / 4) { case 13: QOC_F4(13); break; case 14: QOC_F4(14); break; case 15: QOC_F4(15); break; default: QOC_F4(16); break; }
#undef QOC_F4
    }
    if (!d.uscale_in_loss && !mf.updown && !d.state_transfer) hipLaunchKernelGGL(k_mfma_uscale, dim3(d.B), dim3(64), 0, s, d);
}

// k_mfma_downup batches: final_state = P_{C-1} ... P_0 U0 and unitary_scale of the last evaluation, when they are read back
void qoc_mfma_final_state_batch(QocMfma& mf, const QocDev& d, hipStream_t s) {
    QocDev dd = d;
    dd.skip_done = 0;                                                     // every seed's last evaluation is still in PfT
    const int MQ = mf.mq <= 2 ? 2 : 4, waves = d.B * 4 * mf.NT;
    hipLaunchKernelGGL(k_mfma_bnd_scan<2>, dim3((waves + 3) / 4), dim3(256), 0, s, dd, mf, MQ, (mf.BndA ? 1 : 0) | 4, (const cplx*)mf.PfT, (const cplx*)mf.PfD, mf.C);
    hipLaunchKernelGGL(k_mfma_uscale, dim3(d.B), dim3(64), 0, s, dd);
}

// state transfer: unitary_scale of the last evaluation from Psi_N = d.inter[steps] (the routes whose loss is formed inside a sweep kernel), on read-back
void qoc_mfma_uscale_state_transfer(const QocDev& d, hipStream_t s) {
    QocDev dd = d;
    dd.skip_done = 0;
    hipLaunchKernelGGL(k_mfma_uscale_st, dim3(d.B), dim3(64), 0, s, dd);
}

void qoc_mfma_unpack_inter(QocMfma& mf, const QocDev& d, hipStream_t s) {
    if (mf.updown) {                                                              // k_mfma_downup keeps no Psi_t: the forward sweep, now
        const int sw = d.B * mf.C;
        qoc_launch_forward2_bnd(mf, d, sw, s);
        return;
    }
    hipLaunchKernelGGL(k_mfma_unpack_inter, dim3(512), dim3(256), 0, s, d, mf, mf.mq <= 2 ? 2 : 4);
}
