// qoc_mfma_expm_pair.h -- MFMA path, exponentials K_t = matexp(A_t) + chunk products for n <= 32 batches with TWO WAVES PER SIMD
// (qoc_config.variant = 6): k_mfma_expm_pair.  Reference semantics: core/tensorflow_state.py:25-46 (get_matexp).
//
// AN EXPERIMENT THAT LOST, kept selectable for A/B runs and as the record of why: C2 x 64 runs 0.977 ms per launch with it against
// 0.829 ms for k_mfma_expm_chunk4s (AUTO never picks it).
//
// k_mfma_expm_chunk4s gives every (seed, chunk) item one wave that is alone on its SIMD (482 registers): whatever that wave does
// between two MFMAs -- the combine of a product, the Horner terms, above all the LDS stores of the next left-operand image --
// leaves the matrix pipe idle (pipe busy 72 % of the cycles; tools/mfma_two_wave_probe.hip: 10.6 ns per MFMA with the epilogue of
// a product, 7.0 without).  A second wave on the same SIMD should fill those holes with ITS MFMAs (the same probe: 8.3 ns per MFMA
// with two waves per SIMD and 16 image stores per 384 MFMAs).  That needs <= 256 registers per wave and a quarter of the LDS:
//   * a (seed, chunk) item is a workgroup of two waves, each owning a 16-column block of every matrix (the split of
//     k_mfma_expm_slice2): 192 MFMAs per product and wave, right operand and result in 8 strip registers, the left operand read
//     block by block from the item's transposed LDS image, to which each wave contributes its own 16 columns (8 ds_write_b128)
//     before one LDS barrier per product; the images are double-buffered so that no second barrier is needed;
//   * no re+im sums image (it would be the third of the LDS that makes four items per CU impossible, and a third of the store
//     traffic): the sum of a left block is one v_add_f64 per block, which the partner wave's MFMAs cover;
//   * four items per CU = 8 waves, two per SIMD, from different items, so their epilogues never coincide by construction.
// What the probe did not model is the LDS READ path: with a 16-column block per wave a 1 KB block read feeds 3 MFMAs, not 6 -- one
// ds_read_b128 per 51 pipe cycles and SIMD is 63 % of the CU's 128 B/clk, and the reads arrive late.  Timing knobs (wrong numerics,
// same instruction counts): every second block read dropped: <= 0.82 ms; no barrier: 0.886; no v_add: <= 0.93.  A split by ROWS would
// halve the block reads but needs the whole right operand (96 registers with its sums) in every wave, i.e. a strip image in LDS
// beside the transposed one and twice the stores; a full-matrix wave (6 MFMAs per block read) does not fit 256 registers.
#pragma once
#include "qoc_mfma_frag.h"
#include "qoc_mfma_expm_stream.h"     // QLDS, lds_order()

template <int KC>
__global__ void __launch_bounds__(128, 2) k_mfma_expm_pair(QocDev d, QocMfma mf) {
    constexpr int NT = 2;
    __shared__ __attribute__((aligned(16))) cplx img[2][QNP * QLDS];
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;                               // whole workgroup: no barrier yet
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    double idv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) idv[r] = dlt == 4 * r ? 1.0 : 0.0;
    const int mm = d.T >> 1;
    const bool even = (d.T & 1) == 0;
    const int nH = even ? mm - 1 : mm;
    const double p_c0 = even ? mf.invfact[2 * mm - 2] : mf.invfact[2 * mm], p_c1 = even ? mf.invfact[2 * mm - 1] : mf.invfact[2 * mm + 1];
    const double p_cT = even ? mf.invfact[d.T] : 0.0;
    struct Col { double re[QQS], im[QQS]; };                            // column block J: strip ib = rows 4 ib .. 4 ib + 3
    auto diag = [&](int ib) { return ((ib >> 2) == J) ? idv[ib & 3] : 0.0; };
    int cur = 0;
    auto publish = [&](const Col& m) {                                  // own strips of the next left operand -> image `cur`, then meet the partner
#pragma unroll
        for (int ib = 0; ib < QQS; ++ib) img[cur][(16 * J + (lane & 15)) * QLDS + 4 * ib + (lane >> 4)] = cmake(m.re[ib], m.im[ib]);
        lds_barrier();
    };
    double a[QQS], bq[QQS], cq[QQS];
    auto product = [&](const Col& p) {                                  // acc = (image cur) * p; the other image is free for the next publish
        const cplx* base = img[cur] + (lane >> 4) * QLDS + (lane & 3);
        constexpr int NS = QQS * QQS, RA = 4, RS = RA + 1;
        cplx vb[RS];
        double su[QQS];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) su[kb] = p.re[kb] + p.im[kb];
#pragma unroll
        for (int st = 0; st < RA; ++st) vb[st] = base[4 * (st / QQS) * QLDS + 4 * (st % QQS)];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int kb = st / QQS, ib = st % QQS;
            if (st + RA < NS) vb[(st + RA) % RS] = base[4 * ((st + RA) / QQS) * QLDS + 4 * ((st + RA) % QQS)];
            lds_order();
            const cplx v = vb[st % RS];
            const double vs = v.x + v.y;
            if (kb == 0) {
                a[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, p.re[kb], 0.0, 0, 0, 0);
                bq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, p.im[kb], 0.0, 0, 0, 0);
                cq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, su[kb], 0.0, 0, 0, 0);
            } else {
                a[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, p.re[kb], a[ib], 0, 0, 0);
                bq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, p.im[kb], bq[ib], 0, 0, 0);
                cq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, su[kb], cq[ib], 0, 0, 0);
            }
        }
        cur ^= 1;
    };
    Col R;
#pragma unroll
    for (int ib = 0; ib < QQS; ++ib) { R.re[ib] = diag(ib); R.im[ib] = 0.0; }
    for (int t = t0; t < t1; ++t) {
        // ---- A_t, own column block, two groups of four strips (80 registers of Hamiltonian strips at a time) ----------------------
        double ck[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) ck[kk] = kk < d.k ? d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale : 0.0;
        Col A, X;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            cplx hst[KC + 1][4];
#pragma unroll
            for (int kk = 0; kk <= KC; ++kk) {
                const cplx* H = mf.HfD + (size_t)(kk <= d.k ? kk : 0) * QFR + (J * QQS + 4 * hf) * 64 + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) hst[kk][q] = H[q * 64];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                double re = hst[0][q].x * inv_scale, im = hst[0][q].y * inv_scale;
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) { re = fma(ck[kk], hst[kk + 1][q].x, re); im = fma(ck[kk], hst[kk + 1][q].y, im); }
                A.re[4 * hf + q] = re; A.im[4 * hf + q] = im;
            }
        }
        if (d.T >= 2) {
            // ---- A2 = A * A, polynomial start ----------------------------------------------------------------------------------
            publish(A);
            product(A);
            Col A2;
#pragma unroll
            for (int ib = 0; ib < QQS; ++ib) {
                const double re = a[ib] - bq[ib], im = cq[ib] - a[ib] - bq[ib];
                A2.re[ib] = re; A2.im[ib] = im;
                X.re[ib] = fma(p_cT, re, fma(p_c1, A.re[ib], p_c0 * diag(ib)));
                X.im[ib] = fma(p_cT, im, p_c1 * A.im[ib]);
            }
            if (nH > 0) {
                publish(A2);                                            // image of A2 stays through the Horner products
                for (int i = nH - 1; i >= 0; --i) {
                    product(X);
                    cur ^= 1;                                           // same image again
                    const double d0 = mf.invfact[2 * i], d1 = mf.invfact[2 * i + 1];
#pragma unroll
                    for (int ib = 0; ib < QQS; ++ib) {
                        X.re[ib] = (a[ib] - bq[ib]) + fma(d1, A.re[ib], d0 * diag(ib));
                        X.im[ib] = fma(d1, A.im[ib], cq[ib] - a[ib] - bq[ib]);
                    }
                }
                cur ^= 1;                                               // the next publish must not overwrite A2 while the partner still reads it
            }
        } else {
#pragma unroll
            for (int ib = 0; ib < QQS; ++ib) { X.re[ib] = A.re[ib] + diag(ib); X.im[ib] = A.im[ib]; }
        }
        // ---- squarings -------------------------------------------------------------------------------------------------------
        for (int sq = 0; sq < d.s; ++sq) {
            publish(X);
            product(X);
#pragma unroll
            for (int ib = 0; ib < QQS; ++ib) { X.re[ib] = a[ib] - bq[ib]; X.im[ib] = cq[ib] - a[ib] - bq[ib]; }
        }
        // ---- K_t out, chunk product R <- K_t R -----------------------------------------------------------------------------------
        const size_t item = kitem(mf, d.steps, b, t);
#pragma unroll
        for (int ib = 0; ib < QQS; ++ib) mf.KfD[item + (J * QQS + ib) * 64 + lane] = cmake(X.re[ib], X.im[ib]);
        publish(X);
        product(R);
#pragma unroll
        for (int ib = 0; ib < QQS; ++ib) { R.re[ib] = a[ib] - bq[ib]; R.im[ib] = cq[ib] - a[ib] - bq[ib]; }
    }
    // ---- P_c out: fragD(P) from the registers, fragD(P^T) through the image ----------------------------------------------------------
    const size_t pitem = ((size_t)b * mf.C + c) * QFR;
#pragma unroll
    for (int ib = 0; ib < QQS; ++ib) mf.PfD[pitem + (J * QQS + ib) * 64 + lane] = cmake(R.re[ib], R.im[ib]);
    publish(R);
#pragma unroll
    for (int q = 0; q < QQS; ++q) mf.PfT[pitem + (J * QQS + q) * 64 + lane] = img[cur][(4 * q + (lane >> 4)) * QLDS + 16 * J + (lane & 15)];
}
