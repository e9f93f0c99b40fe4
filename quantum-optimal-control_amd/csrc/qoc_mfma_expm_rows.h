// qoc_mfma_expm_rows.h -- MFMA path, exponentials K_t = matexp(A_t) + chunk products for 32 < n <= 64 (NT = 3, 4) by FOUR waves per
// (seed, chunk) item, each owning a block of ROWS (qoc_config.variant = 7): k_mfma_expm_rows.  Reference semantics:
// core/tensorflow_state.py:25-46 (get_matexp).
//
// k_mfma_expm_chunk4<NT> gives every 16-COLUMN block of an item a wave: NT waves per workgroup (NT = 3: one SIMD of the CU idles, the
// 113 KB of images leave room for one workgroup) and, above all, every 1 KB block read of the left operand feeds only the 3 MFMAs of
// that wave's column block -- 4 waves doing that are ~90 % of the CU's LDS read bandwidth (NT = 4 sits at 50 % of the matrix peak for
// exactly that reason).  Here wave w owns the row strips 4 NT w .. 4 NT (w + 1) - 1 (rows 4 NT w .. of every column block):
//   D(ib, J) = sum_kb L(ib, kb) R(kb, J):   the left block (ib, kb) is read once and feeds 3 NT MFMAs (all column blocks J),
//                                           the right strips (kb, J) come from a second LDS image S in strip layout (lane-contiguous
//                                           1 KB reads, one per kb and J, shared by the NT row strips of the wave),
// i.e. (NT + NT) KB of LDS reads per 3 NT^2 MFMAs instead of 1.5 NT^2: 0.28 (NT = 3) / 0.17 (NT = 4) KB per MFMA against 0.5 / 0.33.
// Every wave writes its result strips to both images (T: transposed, left operand of a later product; S: strips, right operand).
// No re + im sums image: one v_add_f64 per block read (per 3 NT MFMAs).  NT = 3 double-buffers both images (155 KB) and needs one
// LDS barrier per product; NT = 4 has room for one buffer each (135 KB): a second barrier before the images are overwritten.
#pragma once
#include <type_traits>
#include "qoc_mfma_frag.h"
#include "qoc_mfma_expm_stream.h"     // QLDS, lds_order()

#ifndef QOC_ROWS_NB3
#define QOC_ROWS_NB3 1       // image buffers of NT = 3: 1 = single-buffered, two workgroups per CU; 2 = double-buffered, one
#endif
// SLICE (latency mode of n > 32): an item is one time slice -- K_t and its transposed copy only; the chunk products are left to
// k_mfma_chain_rows.
// QA = ACTIVE 4-row strips, ceil(n / 4) (4 NT - 3 .. 4 NT): every matrix of the slice is block diagonal in (active, padding) -- polynomials of the padded
// A_t, whose padding rows and columns are zero -- so the block steps over the inner indices 4 QA .. contribute nothing to the active block and are not
// run: QA / (4 NT) of the MFMAs of the padded product (n = 36: 9 / 12, n = 52: 13 / 16).  What the padding block of K_t and P_c then holds (zero
// beyond strip QA instead of the identity) never meets a non-zero vector entry: the sweeps' vectors are zero there.
template <int NT, int KC, bool SLICE = false, int QA = 4 * NT>
__global__ void __launch_bounds__(256, (NT == 3 && QOC_ROWS_NB3 == 1) ? 2 : 1) k_mfma_expm_rows(QocDev d, QocMfma mf) {
    static_assert(QA > 4 * NT - 4 && QA <= 4 * NT, "active strips of a problem padded to 16 NT");
    constexpr int NB = NT == 3 ? QOC_ROWS_NB3 : 1;                       // image buffers
    extern __shared__ __attribute__((aligned(16))) char smem_rows[];
    cplx* imgT = (cplx*)smem_rows;                                       // [NB][QNP * QLDS]   left operand: image[column][row]
    cplx* imgS = imgT + (size_t)NB * QNP * QLDS;                         // [NB][NT * QQS * 64] right operand: strip (J, kb), lane
    constexpr int TSZ = QNP * QLDS, SSZ = NT * QQS * 64;
    const int lane = threadIdx.x & 63;
    const unsigned ulane = threadIdx.x & 63u;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // row strips NT w .. NT w + NT - 1
    const int per = SLICE ? d.steps : mf.C;
    const int b = blockIdx.x / per, c = blockIdx.x - b * per;
    if (d.skip_done && d.done[b]) return;                               // whole workgroup: no barrier yet
    QOC_LAP_INIT
    const int t0 = SLICE ? c : c * mf.L, t1 = SLICE ? c + 1 : min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    int dlt = (lane & 15) - (lane >> 4);
    // identity: element (row 4 ib + lk, column 16 J + lc) with ib = NT w + r is diagonal iff 4 ib - 16 J == lc - lk
    auto diag = [&](int r, int J) { return (4 * (NT * w + r) - 16 * J == dlt) ? 1.0 : 0.0; };
    const int mm = d.T >> 1;
    const bool even = (d.T & 1) == 0;
    const int nH = even ? mm - 1 : mm;
    const double p_c0 = even ? mf.invfact[2 * mm - 2] : mf.invfact[2 * mm], p_c1 = even ? mf.invfact[2 * mm - 1] : mf.invfact[2 * mm + 1];
    const double p_cT = even ? mf.invfact[d.T] : 0.0;
    struct Rows { double re[NT][NT], im[NT][NT]; };                     // [r][J]: strip ib = NT w + r of column block J
    int tcur = 0, scur = 0;
    // own strips into the images (left: T, right: S), then meet the other three waves.  One buffer (NT = 4): the images may only be
    // overwritten when every wave has finished the product that read them -- a barrier BEFORE the stores as well.
    auto publish = [&](const Rows& ml, bool left, const Rows& mr, bool right) {   // ml -> left image T, mr -> right image S
        if (NB == 1) lds_barrier();
        cplx* T = imgT + (size_t)tcur * TSZ;
        cplx* S = imgS + (size_t)scur * SSZ;
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                if (left) T[(16 * J + (lane & 15)) * QLDS + 4 * (NT * w + r) + (lane >> 4)] = cmake(ml.re[r][J], ml.im[r][J]);
                if (right) S[(J * QQS + NT * w + r) * 64 + lane] = cmake(mr.re[r][J], mr.im[r][J]);
            }
        lds_barrier();
        QOC_LAP(1)
    };
    double a[NT][NT], bq[NT][NT], cq[NT][NT];
    // acc = (image T[tcur]) * (strips S[scur]) for the own row strips.  SAME (round 5): the right operand IS the left one (A * A, the squarings): its strips
    // are read from the T image -- strip (J, kb), lane (lk, lc) = T[column 16 J + lc][row 4 kb + lk], the conflict-free pattern of the strip stores -- and
    // the publish before it skipped the 4 NT stores into S (a third of a slice's LDS stores)
    auto product = [&](auto same) {
        constexpr bool SAME = decltype(same)::value;
        const cplx* base = imgT + (size_t)tcur * TSZ + (lane >> 4) * QLDS + (lane & 3) + 4 * NT * w;
        const cplx* sb = SAME ? imgT + (size_t)tcur * TSZ + (lane & 15) * QLDS + (lane >> 4) : imgS + (size_t)scur * SSZ + lane;
        auto strip = [&](int J, int kb) -> cplx { return SAME ? sb[16 * J * QLDS + 4 * kb] : sb[(J * QQS + kb) * 64]; };
        constexpr int NS = QA * NT, RA = 4, RS = RA + 1;                  // block steps (kb, r), kb-major, over the ACTIVE inner strips
        cplx vb[RS], rs[2][NT];
#pragma unroll
        for (int J = 0; J < NT; ++J) rs[0][J] = strip(J, 0);
#pragma unroll
        for (int st = 0; st < RA; ++st) vb[st] = base[4 * (st / NT) * QLDS + 4 * (st % NT)];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int kb = st / NT, r = st % NT;
            if (st + RA < NS) vb[(st + RA) % RS] = base[4 * ((st + RA) / NT) * QLDS + 4 * ((st + RA) % NT)];
            if (r == 0 && kb + 1 < QA) {
#pragma unroll
                for (int J = 0; J < NT; ++J) rs[(kb + 1) & 1][J] = strip(J, kb + 1);
            }
            lds_order();
            const cplx v = vb[st % RS];
            const double vs = v.x + v.y;
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                const cplx q = rs[kb & 1][J];
                const double qs = q.x + q.y;
                if (kb == 0) {
                    a[r][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, q.x, 0.0, 0, 0, 0);
                    bq[r][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, q.y, 0.0, 0, 0, 0);
                    cq[r][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, qs, 0.0, 0, 0, 0);
                } else {
                    a[r][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, q.x, a[r][J], 0, 0, 0);
                    bq[r][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, q.y, bq[r][J], 0, 0, 0);
                    cq[r][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, qs, cq[r][J], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        QOC_LAP(2)
    };
    auto flipT = [&]() { if (NB == 2) tcur ^= 1; };
    auto flipS = [&]() { if (NB == 2) scur ^= 1; };
    Rows R;
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int J = 0; J < NT; ++J) { R.re[r][J] = diag(r, J); R.im[r][J] = 0.0; }
    for (int t = t0; t < t1; ++t) {
        asm volatile("" : "+v"(dlt));            // (the diagonal masks are compares, not nine doubles hoisted out of the loop and spilled)
        // ---- A_t, own row strips, one column block at a time (NT (KC + 1) strips of the Hamiltonian stack in flight) ------------------
        double ck[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) ck[kk] = kk < d.k ? d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale : 0.0;
        Rows A, X;
        unsigned hoff[NT][NT];
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int r = 0; r < NT; ++r) { hoff[J][r] = ((unsigned)(J * QQS + NT * w + r) * 64u + ulane) * (unsigned)sizeof(cplx); asm volatile("" : "+v"(hoff[J][r])); }
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            cplx hst[KC + 1][NT];
#pragma unroll
            for (int kk = 0; kk <= KC; ++kk) {
                // a scalar base per matrix + a 32-bit byte offset per (column block, strip) that hipcc cannot see through: with the lane
                // folded into the pointer it hoists one 64-bit VGPR pair per (matrix, column block, strip) out of the slice loop and spills
                // them (236 B of scratch, every reload followed by s_waitcnt vmcnt(0): the loads of the assembly went out one at a time)
                const char* H = (const char*)(mf.HfD + (size_t)(kk <= d.k ? kk : 0) * QFR);
#pragma unroll
                for (int r = 0; r < NT; ++r) hst[kk][r] = *(const cplx*)(H + hoff[J][r]);
            }
#pragma unroll
            for (int r = 0; r < NT; ++r) {
                double re = hst[0][r].x * inv_scale, im = hst[0][r].y * inv_scale;
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) { re = fma(ck[kk], hst[kk + 1][r].x, re); im = fma(ck[kk], hst[kk + 1][r].y, im); }
                A.re[r][J] = re; A.im[r][J] = im;
            }
        }
        QOC_LAP(0)
        if (d.T >= 2) {
            // ---- A2 = A * A, polynomial start ----------------------------------------------------------------------------------
            publish(A, true, A, false);
            product(std::true_type{});
            flipT(); flipS();
            Rows A2;
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    const double re = a[r][J] - bq[r][J], im = cq[r][J] - a[r][J] - bq[r][J];
                    A2.re[r][J] = re; A2.im[r][J] = im;
                    X.re[r][J] = fma(p_cT, re, fma(p_c1, A.re[r][J], p_c0 * diag(r, J)));
                    X.im[r][J] = fma(p_cT, im, p_c1 * A.im[r][J]);
                }
            if (nH > 0) {
                for (int i = nH - 1; i >= 0; --i) {
                    publish(A2, i == nH - 1, X, true);                  // the left image of A2 stays through the Horner products
                    product(std::false_type{});
                    flipS();
                    const double d0 = mf.invfact[2 * i], d1 = mf.invfact[2 * i + 1];
#pragma unroll
                    for (int r = 0; r < NT; ++r)
#pragma unroll
                        for (int J = 0; J < NT; ++J) {
                            X.re[r][J] = (a[r][J] - bq[r][J]) + fma(d1, A.re[r][J], d0 * diag(r, J));
                            X.im[r][J] = fma(d1, A.im[r][J], cq[r][J] - a[r][J] - bq[r][J]);
                        }
                }
                flipT();                                                // the next left image must not overwrite A2 while a partner still reads it
            }
        } else {
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int J = 0; J < NT; ++J) { X.re[r][J] = A.re[r][J] + diag(r, J); X.im[r][J] = A.im[r][J]; }
        }
        // ---- squarings -------------------------------------------------------------------------------------------------------
        for (int sq = 0; sq < d.s; ++sq) {
            publish(X, true, X, false);
            product(std::true_type{});
            flipT(); flipS();
#pragma unroll
            for (int r = 0; r < NT; ++r)
#pragma unroll
                for (int J = 0; J < NT; ++J) { X.re[r][J] = a[r][J] - bq[r][J]; X.im[r][J] = cq[r][J] - a[r][J] - bq[r][J]; }
        }
        // ---- K_t out (fragD from the registers; the transposed copy through the left image), chunk product R <- K_t R ---------------
        const size_t item = kitem(mf, d.steps, b, t);
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int J = 0; J < NT; ++J) mf.KfD[item + (J * QQS + NT * w + r) * 64 + lane] = cmake(X.re[r][J], X.im[r][J]);
        QOC_LAP(3)
        publish(X, true, R, !SLICE);
        if (mf.store_T) {
            const cplx* T = imgT + (size_t)tcur * TSZ;
            for (int f = w; f < NT * QQS; f += 4) {                     // fragment (cb, q) of fragD(K^T): K^T[4 q + lk][16 cb + lc] = K[16 cb + lc][4 q + lk]
                const int cb = f / QQS, q = f - cb * QQS;
                mf.KfT[item + (size_t)f * 64 + lane] = T[(4 * q + (lane >> 4)) * QLDS + 16 * cb + (lane & 15)];
            }
        }
        if constexpr (SLICE) return;                                    // (uniform: no barrier follows)
        product(std::false_type{});
        flipT(); flipS();
#pragma unroll
        for (int r = 0; r < NT; ++r)
#pragma unroll
            for (int J = 0; J < NT; ++J) { R.re[r][J] = a[r][J] - bq[r][J]; R.im[r][J] = cq[r][J] - a[r][J] - bq[r][J]; }
        QOC_LAP(3)
    }
    // ---- P_c out: fragD(P) from the registers, fragD(P^T) through the left image ------------------------------------------------------
    const size_t pitem = ((size_t)b * mf.C + c) * QFR;
#pragma unroll
    for (int r = 0; r < NT; ++r)
#pragma unroll
        for (int J = 0; J < NT; ++J) mf.PfD[pitem + (J * QQS + NT * w + r) * 64 + lane] = cmake(R.re[r][J], R.im[r][J]);
    publish(R, true, R, false);
    {
        const cplx* T = imgT + (size_t)tcur * TSZ;
        for (int f = w; f < NT * QQS; f += 4) {
            const int cb = f / QQS, q = f - cb * QQS;
            mf.PfT[pitem + (size_t)f * 64 + lane] = T[(4 * q + (lane >> 4)) * QLDS + 16 * cb + (lane & 15)];
        }
    }
    QOC_LAP(4)
    QOC_LAP_DONE
}

template <int NT> static inline size_t qoc_expm_rows_lds() {
    return (size_t)(NT == 3 ? QOC_ROWS_NB3 : 1) * ((size_t)QNP * QLDS + (size_t)NT * QQS * 64) * sizeof(cplx);
}
