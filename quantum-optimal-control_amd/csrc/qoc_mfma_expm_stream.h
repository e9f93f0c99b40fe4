// qoc_mfma_expm_stream.h -- MFMA path, exponentials K_t = matexp(A_t) + chunk products, qoc_config.variant = 4 (default for
// n <= 32 batches): k_mfma_expm_chunk4s.  Reference semantics: core/tensorflow_state.py:25-46 (get_matexp).
//
// Same arithmetic as k_mfma_expm_chunk4w (one wave per (seed, chunk), every product on v_mfma_f64_4x4x4_4b_f64 in the
// 3-multiplication form, left operand block by block from a transposed LDS image + its re+im sums, right operand and result in
// strip registers), re-organised around two measurements (tools/mfma_cover_probe.hip, tools/expm_phase_probe.hip;
// profiles/r02_mfma_cover_probe.txt, r02_expm_phase_probe_chunk4w.txt):
//   * a wave that is alone on its SIMD cannot hide VALU work behind its own fp64 MFMAs (any VALU instruction between two MFMAs
//     costs 3-12 cycles of MFMA issue), but LDS and global memory instructions ARE free there as long as their own paths keep up;
//   * the LDS STORE path moves ~16 B per cycle per wave: writing the image of one 32x32 complex matrix + sums (24 KB) takes
//     ~1600 cycles when issued back to back -- 16 % of chunk4w's time, more than all its VALU work -- yet a ds_write_b128 per
//     4 MFMAs costs nothing.
// So the image of a product's left operand is no longer written before the product, but DURING it, strip by strip (a strip =
// 4 rows x 16 columns = one result register): the block steps run strip-major, step (strip j, k) needs only strip j of the
// image, and strip j + 1 is stored while strip j multiplies (the LDS executes one wave's operations in order, so there is no
// fence at all).  Only strip 0 is stored before the product starts, and that store is issued early in the preceding VALU batch.
// The VALU work (combine a-b / c-a-b, Horner terms, re+im sums of the new right operand) stays batched between the products.
// The assembly of A_{t+1} (5 x 16 loads from the L2-resident Hamiltonian stack) rides on the chunk-product MFMAs of slice t.
// The image row stride is 16 NT + 5 elements: conflict-free for the b128/b64 strip stores AND for the block reads (the stride
// 16 NT + 1 of the older kernels puts the four columns of a block on overlapping bank groups: 2-way conflicts on every read).
#pragma once
#include <type_traits>
#include <utility>
#include "qoc_mfma_frag.h"

#define QLDS (16 * NT + 5)        // image row stride (complex elements) of this kernel
#ifndef QOC_RING_AHEAD
#define QOC_RING_AHEAD 2          // block steps between the fetch of a left-operand block and its MFMAs
#endif

// phase timing hooks: no-ops in the product build; tools/expm_phase_probe.hip defines them to read the shader clock
#ifndef QOC_LAP
#define QOC_LAP(ph)
#define QOC_LAP_INIT
#define QOC_LAP_DONE
#endif

template <int NT> struct Sums { double v[NT][4 * NT]; };     // re + im of a strip-register matrix, [column block J][strip ib]

template <int NT>
__device__ __forceinline__ void strip_sums(const CTile (&m)[NT][NT], Sums<NT>& s) {
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int ib = 0; ib < QQS; ++ib) s.v[J][ib] = m[J][ib >> 2].re[ib & 3] + m[J][ib >> 2].im[ib & 3];
}
// strip (column block Jk, rows 4 ib .. 4 ib + 3) of m -> transposed image + sums image
template <int NT>
__device__ __forceinline__ void strip_store(cplx* img, double* imgs, int lane, const CTile (&m)[NT][NT], const Sums<NT>& s, int Jk, int ib) {
    const int o = (16 * Jk + (lane & 15)) * QLDS + 4 * ib + (lane >> 4);
    img[o] = cmake(m[Jk][ib >> 2].re[ib & 3], m[Jk][ib >> 2].im[ib & 3]);
    imgs[o] = s.v[Jk][ib];
}
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
template <class F, int... I>
__device__ __forceinline__ void qoc_for_each_step(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// acc = L * p with L read from the image.  STORE: strips 1.. of L (registers Lm, sums Ls) are written while the product runs;
// strip 0 must already be in the image.  hook(integral_constant<step>) runs at the start of every block step.
template <int NT, bool STORE, class Hook>
__device__ __forceinline__ void mm_stream(cplx* img, double* imgs, int lane, const CTile (&Lm)[NT][NT], const Sums<NT>& Ls,
                                          const CTile (&p)[NT][NT], const Sums<NT>& ps, double (&a)[NT][QQS], double (&b)[NT][QQS],
                                          double (&c)[NT][QQS], Hook&& hook) {
    constexpr int NSTRIP = NT * QQS, NS = 4 * NSTRIP;
    const cplx* base = img + (lane >> 4) * QLDS + (lane & 3);
    const double* bases = imgs + (lane >> 4) * QLDS + (lane & 3);
    constexpr int RA = QOC_RING_AHEAD, RS = RA + 1;
    cplx vb[RS]; double sb[RS];
    auto fetch = [&](int st, int slot) {              // block (rows 4 ib.., columns 4 kb..) of step st
        const int j = st >> 2, kb = 4 * (j / QQS) + (st & 3), ib = j % QQS;
        vb[slot] = base[4 * kb * QLDS + 4 * ib];
        sb[slot] = bases[4 * kb * QLDS + 4 * ib];
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < RA; ++st) fetch(st, st);
    auto step = [&](auto sc) {
        constexpr int st = decltype(sc)::value;
        constexpr int j = st >> 2, kq = st & 3, Jk = j / QQS, ib = j % QQS, kb = 4 * Jk + kq;
        hook(sc);
        if constexpr (STORE) {
            // strip o is read first by the fetch issued at step 4 o - RA: its image goes out one step earlier, its sums with that step
            qoc_for_each_step([&](auto oc) {
                constexpr int o = decltype(oc)::value + 1;
                constexpr int s128 = 4 * o - RA - 1 > 0 ? 4 * o - RA - 1 : 0, s64 = 4 * o - RA > 0 ? 4 * o - RA : 0;
                const int off = (16 * (o / QQS) + (lane & 15)) * QLDS + 4 * (o % QQS) + (lane >> 4);
                if constexpr (s128 == st) img[off] = cmake(Lm[o / QQS][(o % QQS) >> 2].re[(o % QQS) & 3], Lm[o / QQS][(o % QQS) >> 2].im[(o % QQS) & 3]);
                if constexpr (s64 == st) imgs[off] = Ls.v[o / QQS][o % QQS];
            }, std::make_integer_sequence<int, NSTRIP - 1>{});
        }
        if constexpr (st + RA < NS) fetch(st + RA, (st + RA) % RS);
        lds_order();
        const cplx v = vb[st % RS];
        const double vs = sb[st % RS];
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            const double br = p[J][kb >> 2].re[kb & 3], bi = p[J][kb >> 2].im[kb & 3], bs = ps.v[J][kb];
            if constexpr (kb == 0) {                 // first contribution to (J, ib): start the accumulators from the inline zero
                a[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br, 0.0, 0, 0, 0);
                b[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi, 0.0, 0, 0, 0);
                c[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, bs, 0.0, 0, 0, 0);
            } else {
                a[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br, a[J][ib], 0, 0, 0);
                b[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi, b[J][ib], 0, 0, 0);
                c[J][ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, bs, c[J][ib], 0, 0, 0);
            }
        }
    };
    qoc_for_each_step(step, std::make_integer_sequence<int, NS>{});
    // the epilogue's VALU batch must not be scheduled into the tail of the MFMA stream: a VALU instruction between two MFMAs of a
    // lone wave costs 3-12 cycles of MFMA issue, 4-5 in a batch of its own (profiles/r02_mfma_cover_probe.txt)
    __builtin_amdgcn_sched_barrier(0);
}

struct NoHook { template <class S> __device__ __forceinline__ void operator()(S) const {} };

// KC = controls handled by the pipelined assembly (k <= KC; surplus controls carry a zero coefficient)
template <int NT, int KC>
__global__ void __launch_bounds__(64, 1) k_mfma_expm_chunk4s(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[QNP * QLDS];
    __shared__ __attribute__((aligned(16))) double imgs[QNP * QLDS];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    constexpr int NSTRIP = NT * QQS;
    double idv[4];                                    // identity pattern of a diagonal tile: register r holds the diagonal where dlt == 4 r
#pragma unroll
    for (int r = 0; r < 4; ++r) idv[r] = dlt == 4 * r ? 1.0 : 0.0;
    const int mm = d.T >> 1;
    const bool even = (d.T & 1) == 0;
    const int nH = even ? mm - 1 : mm;                // Horner products over A2 (tensorflow_state.py:37-41 in Paterson-Stockmeyer form)
    const double p_c0 = even ? mf.invfact[2 * mm - 2] : mf.invfact[2 * mm], p_c1 = even ? mf.invfact[2 * mm - 1] : mf.invfact[2 * mm + 1];
    const double p_cT = even ? mf.invfact[d.T] : 0.0;

    QOC_LAP_INIT
    CTile R[NT][NT], A[NT][NT], X[NT][NT];
    Sums<NT> As, Xs, Rs;
    double a[NT][QQS], bb[NT][QQS], cc[NT][QQS];
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_identity<NT>(J, lane, R[J]);

    // out = combine of the three accumulator sets for strip (J, ib): re = a - b, im = c - a - b
    auto comb_re = [&](int J, int ib) { return a[J][ib] - bb[J][ib]; };
    auto comb_im = [&](int J, int ib) { return cc[J][ib] - a[J][ib] - bb[J][ib]; };

    // ---- first slice of the chunk: A_t assembled in the open ------------------------------------------------------
    {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            colblock_load<NT>(mf.HfD, J, lane, A[J]);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib) { A[J][Ib].re *= inv_scale; A[J][Ib].im *= inv_scale; }
        }
#pragma unroll 1
        for (int kk = 0; kk < d.k; ++kk) {
            const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t0] * inv_scale;
            const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                        A[J][Ib].re[r] = fma(ck, h.x, A[J][Ib].re[r]);
                        A[J][Ib].im[r] = fma(ck, h.y, A[J][Ib].im[r]);
                    }
        }
        strip_sums<NT>(A, As);
        strip_store<NT>(img, imgs, lane, A, As, 0, 0);
        lds_order();
    }
    QOC_LAP(0)

    for (int t = t0; t < t1; ++t) {
        // ---- A2 = A * A (image <- A) -----------------------------------------------------------------------------
        mm_stream<NT, true>(img, imgs, lane, A, As, A, As, a, bb, cc, NoHook{});
        QOC_LAP(1)
        {
            CTile A2[NT][NT];
            Sums<NT> A2s;
            // strip 0 of the next left operand first, so that its store drains under the rest of this VALU batch
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int ib = 0; ib < QQS; ++ib) {
                    const double re = comb_re(J, ib), im = comb_im(J, ib);
                    A2[J][ib >> 2].re[ib & 3] = re; A2[J][ib >> 2].im[ib & 3] = im;
                    const double one = ((ib >> 2) == J) ? p_c0 * idv[ib & 3] : 0.0;
                    X[J][ib >> 2].re[ib & 3] = fma(p_cT, re, fma(p_c1, A[J][ib >> 2].re[ib & 3], one));
                    X[J][ib >> 2].im[ib & 3] = fma(p_cT, im, p_c1 * A[J][ib >> 2].im[ib & 3]);
                    if (J == 0 && ib == 0) {
                        if (nH > 0) { A2s.v[0][0] = re + im; strip_store<NT>(img, imgs, lane, A2, A2s, 0, 0); }
                        else { Xs.v[0][0] = X[0][0].re[0] + X[0][0].im[0]; strip_store<NT>(img, imgs, lane, X, Xs, 0, 0); }
                        lds_order();
                    }
                }
            strip_sums<NT>(X, Xs);
            QOC_LAP(2)
            // ---- Horner over A2: X <- B_i + A2 * X; the first product writes the image of A2 -----------------------
            if (nH > 0) {
                strip_sums<NT>(A2, A2s);
                for (int i = nH - 1; i >= 0; --i) {
                    if (i == nH - 1) mm_stream<NT, true>(img, imgs, lane, A2, A2s, X, Xs, a, bb, cc, NoHook{});
                    else mm_stream<NT, false>(img, imgs, lane, A2, A2s, X, Xs, a, bb, cc, NoHook{});
                    QOC_LAP(3)
                    const double d0 = mf.invfact[2 * i], d1 = mf.invfact[2 * i + 1];
#pragma unroll
                    for (int J = 0; J < NT; ++J)
#pragma unroll
                        for (int ib = 0; ib < QQS; ++ib) {
                            const double one = ((ib >> 2) == J) ? d0 * idv[ib & 3] : 0.0;
                            X[J][ib >> 2].re[ib & 3] = comb_re(J, ib) + fma(d1, A[J][ib >> 2].re[ib & 3], one);
                            X[J][ib >> 2].im[ib & 3] = fma(d1, A[J][ib >> 2].im[ib & 3], comb_im(J, ib));
                            if (J == 0 && ib == 0 && i == 0) {      // polynomial done: X is the next left operand
                                Xs.v[0][0] = X[0][0].re[0] + X[0][0].im[0];
                                strip_store<NT>(img, imgs, lane, X, Xs, 0, 0);
                                lds_order();
                            }
                        }
                    strip_sums<NT>(X, Xs);
                    QOC_LAP(4)
                }
            }
        }
        // ---- squarings: X <- X * X (image <- X) ---------------------------------------------------------------------
        for (int sq = 0; sq < d.s; ++sq) {
            mm_stream<NT, true>(img, imgs, lane, X, Xs, X, Xs, a, bb, cc, NoHook{});
            QOC_LAP(5)
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int ib = 0; ib < QQS; ++ib) {
                    const double re = comb_re(J, ib), im = comb_im(J, ib);
                    X[J][ib >> 2].re[ib & 3] = re; X[J][ib >> 2].im[ib & 3] = im;
                    Xs.v[J][ib] = re + im;
                    if (J == 0 && ib == 0) { strip_store<NT>(img, imgs, lane, X, Xs, 0, 0); lds_order(); }
                }
            QOC_LAP(6)
        }
        // ---- K_t out; chunk product R <- K_t * R (image <- K_t), A_{t+1} assembled under these MFMAs --------------------
        const size_t item = kitem(mf, d.steps, b, t);
        strip_sums<NT>(R, Rs);
        {
            const int tn = min(t + 1, d.steps - 1);
            double ck[KC];
            const cplx* hk[KC + 1];
            hk[0] = mf.HfD;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                ck[kk] = kk < d.k ? d.u[((size_t)b * d.k + kk) * d.steps + tn] * inv_scale : 0.0;
                hk[kk + 1] = mf.HfD + (size_t)(kk < d.k ? kk + 1 : 0) * QFR;
            }
            // groups of GS strips: the loads of ALL k + 1 matrices for group g are issued at step g * SP, and one VALU batch right
            // before the loads of group g + 1 forms those strips of A_{t+1} from the staging registers (no read-modify-write of A,
            // which the register allocator parks in AGPRs during this product)
            constexpr int GS = KC <= 4 ? 2 : 1, NG = NSTRIP / GS, NS = 4 * NSTRIP, SP = (NS - 1) / NG;
            static_assert(NSTRIP % GS == 0 && SP >= 2, "assembly groups do not fit the product (NT = 2 only)");
            cplx stage[GS * (KC + 1)];
            const unsigned lane16 = (unsigned)lane;
            auto hook = [&](auto sc) {
                constexpr int st = decltype(sc)::value;
                if constexpr (st % SP == 0 && st / SP <= NG) {
                    constexpr int g = st / SP;
                    if constexpr (g >= 1) {
#pragma unroll
                        for (int e = 0; e < GS; ++e) {
                            const int j = (g - 1) * GS + e, J = j / QQS, ib = j % QQS;
                            double re = stage[e * (KC + 1)].x * inv_scale, im = stage[e * (KC + 1)].y * inv_scale;
#pragma unroll
                            for (int kk = 0; kk < KC; ++kk) {
                                re = fma(ck[kk], stage[e * (KC + 1) + kk + 1].x, re);
                                im = fma(ck[kk], stage[e * (KC + 1) + kk + 1].y, im);
                            }
                            A[J][ib >> 2].re[ib & 3] = re;
                            A[J][ib >> 2].im[ib & 3] = im;
                        }
                    }
                    if constexpr (g < NG) {
#pragma unroll
                        for (int e = 0; e < GS; ++e)
#pragma unroll
                            for (int kk = 0; kk <= KC; ++kk) {
                                const cplx* sb = hk[kk] + ((g * GS + e) & ~3) * 64;            // uniform: scalar address arithmetic
                                stage[e * (KC + 1) + kk] = sb[((g * GS + e) & 3) * 64 + lane16];
                            }
                    }
                }
            };
            mm_stream<NT, true>(img, imgs, lane, X, Xs, R, Rs, a, bb, cc, hook);
            // K_t goes out only now: on gfx950 stores share vmcnt with loads and retire in order, so a store burst issued before the
            // product would make the first assembly batch wait for 16 KB of HBM write acknowledgements (5 % of the kernel)
#pragma unroll
            for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.KfD + item, J, lane, X[J]);
            QOC_LAP(7)
        }
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int ib = 0; ib < QQS; ++ib) {
                R[J][ib >> 2].re[ib & 3] = comb_re(J, ib);
                R[J][ib >> 2].im[ib & 3] = comb_im(J, ib);
                if (J == 0 && ib == 0) {                                  // strip 0 of A_{t+1} for the next slice's first product
                    As.v[0][0] = A[0][0].re[0] + A[0][0].im[0];
                    strip_store<NT>(img, imgs, lane, A, As, 0, 0);
                    lds_order();
                }
            }
        strip_sums<NT>(A, As);
        QOC_LAP(8)
    }
    // ---- chunk product out: fragD(P_c) from the registers, fragD(P_c^T) through the image ------------------------------------
    const size_t pitem = (size_t)b * mf.C + c;
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R[J]);
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int ib = 0; ib < QQS; ++ib)
            img[(16 * J + (lane & 15)) * QLDS + 4 * ib + (lane >> 4)] = cmake(R[J][ib >> 2].re[ib & 3], R[J][ib >> 2].im[ib & 3]);
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int q = 0; q < QQS; ++q) mf.PfT[pitem * QFR + (J * QQS + q) * 64 + lane] = img[(4 * q + (lane >> 4)) * QLDS + 16 * J + (lane & 15)];
    QOC_LAP(9)
    QOC_LAP_DONE
}

// ---- chain products in fragD format (latency mode): OUT[b][i] = IN[b][i*len + len' - 1] ... IN[b][i*len], len' = min(len, count - i*len);
// `tail` (one matrix shared by all seeds, e.g. U0) multiplies from the right after the chain.  Used twice per iteration (chunk products
// P_c from the slice propagators, group products from the chunk products) and once per read-back (final_state).  in_is_K: IN is the K
// storage (kitem addressing with its skews), else a plain [B][count] array.  ROW-SPLIT over 8 waves per output:
// R <- R * M_t needs, for the rows 4w .. 4w+3 of the result, only the SAME rows of R as left operand (4x4 blocks of those rows
// against the strips of M_t): wave w therefore carries its four rows through the whole chain alone -- no shared image, no barrier.
// Its two result strips go through a wave-private 4-row transposition pad (2.5 KB + sums) to become the left blocks of the next
// product; the right operand M_t is read by every wave straight from global memory as strip registers (fragD is the strip layout),
// one matrix ahead.  48 MFMAs per product and wave instead of 384: a chain of 7 products takes ~4 us instead of ~23.
template <int NT>
__global__ void __launch_bounds__(64) k_mfma_chain_rows(QocDev d, QocMfma mf, const cplx* __restrict__ IN, int in_is_K, int count, int len,
                                                        cplx* __restrict__ OUT, int nout, const cplx* __restrict__ tail, cplx* __restrict__ OUTT) {
    constexpr bool DB = NT <= 3;                                     // the next right operand in flight (NT = 4: a matrix is 256 registers, one at a time)
    constexpr int PS = 5;                                            // pad stride (complex elements per column): conflict-free stores and block reads
    __shared__ __attribute__((aligned(16))) cplx pad[QNP * PS];
    __shared__ __attribute__((aligned(16))) double pads[QNP * PS];
    const int lane = threadIdx.x, lc = lane & 15, lk = lane >> 4;
    const int w = blockIdx.x % QQS, item = blockIdx.x / QQS;         // 4 NT row blocks of 4 rows
    const int b = item / nout, i = item - b * nout;
    if (d.skip_done && d.done[b]) return;
    const int hi = min(i * len + len, count) - 1, lo = i * len - (tail ? 1 : 0);
    auto src = [&](int t) -> const cplx* { return (tail && t < i * len) ? tail : (in_is_K ? IN + kitem(mf, d.steps, b, t) : IN + ((size_t)b * count + t) * QFR); };
    struct Mat { cplx s[NT][QQS]; };                                 // all strips of a right operand
    auto load_mat = [&](const cplx* __restrict__ F, Mat& m) {
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int kb = 0; kb < QQS; ++kb) m.s[J][kb] = F[(J * QQS + kb) * 64 + lane];
    };
    cplx r[NT];                                                      // own rows of the running product: strips (J, ib = w)
    {
        const cplx* F = src(hi);
#pragma unroll
        for (int J = 0; J < NT; ++J) r[J] = F[(J * QQS + w) * 64 + lane];
    }
    Mat M0;
    if (hi > lo) load_mat(src(hi - 1), M0);
    auto product = [&](const Mat& m) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {                               // R[4w + lk][16 J + lc] -> pad[column][row in block]
            pad[(16 * J + lc) * PS + lk] = r[J];
            pads[(16 * J + lc) * PS + lk] = r[J].x + r[J].y;
        }
        lds_order();
        cplx blk[QQS]; double bls[QQS];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) {                           // lane 16 k + 4 blk + i  <-  R[4w + i][4 kb + k]
            blk[kb] = pad[(4 * kb + lk) * PS + (lane & 3)];
            bls[kb] = pads[(4 * kb + lk) * PS + (lane & 3)];
        }
        double a[NT], bq[NT], cq[NT];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb)
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                const double br = m.s[J][kb].x, bi = m.s[J][kb].y, bs = br + bi;
                if (kb == 0) {
                    a[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].x, br, 0.0, 0, 0, 0);
                    bq[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].y, bi, 0.0, 0, 0, 0);
                    cq[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(bls[kb], bs, 0.0, 0, 0, 0);
                } else {
                    a[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].x, br, a[J], 0, 0, 0);
                    bq[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].y, bi, bq[J], 0, 0, 0);
                    cq[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(bls[kb], bs, cq[J], 0, 0, 0);
                }
            }
#pragma unroll
        for (int J = 0; J < NT; ++J) r[J] = cmake(a[J] - bq[J], cq[J] - a[J] - bq[J]);
    };
    int t = hi - 1;
    if constexpr (DB) {
        Mat M1;
        for (; t - 1 >= lo; t -= 2) {                                // the next right operand is in flight while this one multiplies
            load_mat(src(t - 1), M1); lds_order(); product(M0);
            if (t - 2 >= lo) load_mat(src(t - 2), M0);
            lds_order(); product(M1);
        }
        if (t >= lo) product(M0);
    } else {
        for (; t >= lo; --t) {
            product(M0);
            if (t - 1 >= lo) load_mat(src(t - 1), M0);
        }
    }
    cplx* out = OUT + ((size_t)b * nout + i) * QFR;
#pragma unroll
    for (int J = 0; J < NT; ++J) out[(J * QQS + w) * 64 + lane] = r[J];
    if (OUTT) {                                                      // fragD(P^T): this wave's rows are columns there (scattered 16 B stores)
        cplx* outt = OUTT + ((size_t)b * nout + i) * QFR;
        const int row = 4 * w + lk;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            const int col = 16 * J + lc;                             // P[row][col] = P^T[col][row] -> fragment (row >> 4, col >> 2), lane 16 (col & 3) + (row & 15)
            outt[((row >> 4) * QQS + (col >> 2)) * 64 + 16 * (col & 3) + (row & 15)] = r[J];
        }
    }
}

// The same chains with the COLUMNS of the right operand split over the waves of a workgroup as well: workgroup = (output, row block w),
// wave J = column block J.  A wave then loads only the strips (kb, J) of M_t -- 1/NT of the matrix, the fetch that bounds a step of
// k_mfma_chain_rows (a 16 NT^2 KB matrix per wave and step) -- and runs 3 QQS MFMAs instead of 3 NT QQS; the price is one LDS barrier
// per step: the left blocks need the four rows of R over ALL columns, so the waves exchange their strips through a double-buffered pad.
template <int NT>
__global__ void __launch_bounds__(64 * NT) k_mfma_chain_rows2(QocDev d, QocMfma mf, const cplx* __restrict__ IN, int in_is_K, int count, int len,
                                                              cplx* __restrict__ OUT, int nout, const cplx* __restrict__ tail, cplx* __restrict__ OUTT) {
    constexpr int PS = 5;                                            // pad stride (complex elements per column): conflict-free stores and block reads
    __shared__ __attribute__((aligned(16))) cplx pad[2][QNP * PS];
    const int lane = threadIdx.x & 63, lc = lane & 15, lk = lane >> 4;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = blockIdx.x % QQS, item = blockIdx.x / QQS;         // 4 NT row blocks of 4 rows
    const int b = item / nout, i = item - b * nout;
    if (d.skip_done && d.done[b]) return;                            // whole workgroup: no barrier yet
    const int hi = min(i * len + len, count) - 1, lo = i * len - (tail ? 1 : 0);
    auto src = [&](int t) -> const cplx* { return (tail && t < i * len) ? tail : (in_is_K ? IN + kitem(mf, d.steps, b, t) : IN + ((size_t)b * count + t) * QFR); };
    struct Col { cplx s[QQS]; };                                     // strips (kb, J) of a right operand
    auto load_col = [&](const cplx* __restrict__ F, Col& m) {
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) m.s[kb] = F[(J * QQS + kb) * 64 + lane];
    };
    cplx r = src(hi)[(J * QQS + w) * 64 + lane];                     // own strip of the running product: rows 4 w .., column block J
    int buf = 0;
    auto product = [&](const Col& m) {
        pad[buf][(16 * J + lc) * PS + lk] = r;                       // R[4w + lk][16 J + lc] -> pad[column][row in block]
        lds_barrier();
        double a = 0.0, bq = 0.0, cq = 0.0;
        cplx blk[QQS];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) blk[kb] = pad[buf][(4 * kb + lk) * PS + (lane & 3)];   // lane 16 k + 4 blk + i  <-  R[4w + i][4 kb + k]
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) {
            const double br = m.s[kb].x, bi = m.s[kb].y;
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].x, br, a, 0, 0, 0);
            bq = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].y, bi, bq, 0, 0, 0);
            cq = __builtin_amdgcn_mfma_f64_4x4x4f64(blk[kb].x + blk[kb].y, br + bi, cq, 0, 0, 0);
        }
        r = cmake(a - bq, cq - a - bq);
        buf ^= 1;                                                    // the other pad: nobody reads it any more (everyone passed this step's barrier)
    };
    Col M0, M1;
    if (hi > lo) load_col(src(hi - 1), M0);
    int t = hi - 1;
    for (; t - 1 >= lo; t -= 2) {                                    // the next right operand is in flight while this one multiplies
        load_col(src(t - 1), M1); lds_order(); product(M0);
        if (t - 2 >= lo) load_col(src(t - 2), M0);
        lds_order(); product(M1);
    }
    if (t >= lo) product(M0);
    cplx* out = OUT + ((size_t)b * nout + i) * QFR;
    out[(J * QQS + w) * 64 + lane] = r;
    if (OUTT) {                                                      // fragD(P^T): this wave's rows are columns there (scattered 16 B stores)
        cplx* outt = OUTT + ((size_t)b * nout + i) * QFR;
        const int row = 4 * w + lk, col = 16 * J + lc;               // P[row][col] = P^T[col][row] -> fragment (row >> 4, col >> 2), lane 16 (col & 3) + (row & 15)
        outt[((row >> 4) * QQS + (col >> 2)) * 64 + 16 * (col & 3) + (row & 15)] = r;
    }
}

// ---- latency mode: K_t by TWO waves per slice (one per 16-column block), n <= 32 -------------------------------------------------
// Left multiplication acts on column blocks independently, so wave J carries column block J of every matrix of the slice in 8 strip
// registers and needs the other wave only for the LEFT operand: both publish their strips of it into one of two LDS images (one
// barrier per product), then each runs the 64 block steps with 3 MFMAs per step.  192 instead of 384 MFMAs per product on the
// dependent chain of 6 products, half the assembly and half the epilogue per wave: 32 -> ~17 us for the 500 slices of one C2
// trajectory (two waves per slice = 1000 waves, one per SIMD).  Also forms the controls and stores fragD(K_t) and fragD(K_t^T).
// Measured per launch for one C2 trajectory: one wave per slice 34.2 us, this kernel 25.5 us, four waves per slice (column block x
// row half, two barriers per product) 27.1 us -- beyond two waves the kernel is bound by its ~4 us of cold first loads after the
// kernel boundary and by the publish / barrier / fetch latency of each product, not by its MFMAs (device-side clocks: assembly 4.2 us,
// first product 1.3 us, the other five 7.3-10 us, output 0.6 us in the four-wave variant).
// QA = active 4-row strips of the padded problem (ceil(n / 4), 5 .. 8): the all-zero row strips beyond are neither assembled nor multiplied
// nor stored (QA^2 of the 64 block steps of a product); KfD / KfT keep the zeros of set-up there.
template <int KC, int QA = 8>
__global__ void __launch_bounds__(128) k_mfma_expm_slice2(QocDev d, QocMfma mf) {
    constexpr int NT = 2;
    __shared__ __attribute__((aligned(16))) cplx img[2][QNP * QLDS];
    __shared__ __attribute__((aligned(16))) double imgs[2][QNP * QLDS];
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / d.steps, t = blockIdx.x - b * d.steps;
    if (d.skip_done && d.done[b]) return;                               // whole workgroup: no barrier yet
    QOC_LAP_INIT
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
    struct Col { double re[QQS], im[QQS], su[QQS]; };                   // column block J: strip ib = rows 4 ib .. 4 ib + 3, plus re + im
    // ---- A_t, own column block: every load first (one round trip to the L2-resident stack), controls meanwhile -------------------
    cplx hst[KC + 1][QQS];
#pragma unroll
    for (int kk = 0; kk <= KC; ++kk) {
        const cplx* H = mf.HfD + (size_t)(kk <= d.k ? kk : 0) * QFR + (J * QQS) * 64 + lane;
#pragma unroll
        for (int ib = 0; ib < QA; ++ib) hst[kk][ib] = H[ib * 64];
    }
    double ck[KC], bs0[KC], ma[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {                                   // every load first, unconditionally (clamped): ONE round trip, not k
        const int kc = kk < d.k ? kk : 0;
        bs0[kk] = d.base[((size_t)b * d.k + kc) * d.steps + t];
        ma[kk] = d.maxA[kc];
    }
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {                                   // u = maxA sin(base)   tensorflow_state.py:176-178
        const double wk = sin(bs0[kk]);
        const double uk = ma[kk] * wk;
        if (kk < d.k && J == 0 && lane == 0) { const size_t ci = ((size_t)b * d.k + kk) * d.steps + t; d.w[ci] = wk; d.u[ci] = uk; }
        ck[kk] = kk < d.k ? uk * inv_scale : 0.0;
    }
    if constexpr (QA < QQS) {                                           // rows >= 4 QA of both images read as zero (fragD(K^T) is gathered from them)
#pragma unroll
        for (int ib = QA; ib < QQS; ++ib) {
            const int o = (16 * J + (lane & 15)) * QLDS + 4 * ib + (lane >> 4);
            img[0][o] = cmake(0.0, 0.0); img[1][o] = cmake(0.0, 0.0); imgs[0][o] = 0.0; imgs[1][o] = 0.0;
        }
    }
    Col A, X;
#pragma unroll
    for (int ib = 0; ib < QA; ++ib) {
        double re = hst[0][ib].x * inv_scale, im = hst[0][ib].y * inv_scale;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) { re = fma(ck[kk], hst[kk + 1][ib].x, re); im = fma(ck[kk], hst[kk + 1][ib].y, im); }
        A.re[ib] = re; A.im[ib] = im; A.su[ib] = re + im;
    }
    QOC_LAP(0)
    int cur = 0;
    auto publish = [&](const Col& m) {                                  // own strips of the next left operand -> image `cur`, then meet the partner
#pragma unroll
        for (int ib = 0; ib < QA; ++ib) {
            const int o = (16 * J + (lane & 15)) * QLDS + 4 * ib + (lane >> 4);
            img[cur][o] = cmake(m.re[ib], m.im[ib]);
            imgs[cur][o] = m.su[ib];
        }
        lds_barrier();
        QOC_LAP(1)
    };
    double a[QQS], bq[QQS], cq[QQS];
    auto product = [&](const Col& p) {                                  // acc = (image cur) * p; the other image is free for the next publish
        const cplx* base = img[cur] + (lane >> 4) * QLDS + (lane & 3);
        const double* bases = imgs[cur] + (lane >> 4) * QLDS + (lane & 3);
        constexpr int NS = QA * QA, RA = 3, RS = RA + 1;
        cplx vb[RS]; double sb[RS];
        auto fetch = [&](int st, int slot) {
            const int kb = st / QA, ib = st % QA;
            vb[slot] = base[4 * kb * QLDS + 4 * ib];
            sb[slot] = bases[4 * kb * QLDS + 4 * ib];
        };
#pragma unroll
        for (int st = 0; st < RA; ++st) fetch(st, st);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int kb = st / QA, ib = st % QA;
            if (st + RA < NS) fetch(st + RA, (st + RA) % RS);
            lds_order();
            const cplx v = vb[st % RS];
            const double vs = sb[st % RS];
            if (kb == 0) {
                a[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, p.re[kb], 0.0, 0, 0, 0);
                bq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, p.im[kb], 0.0, 0, 0, 0);
                cq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, p.su[kb], 0.0, 0, 0, 0);
            } else {
                a[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, p.re[kb], a[ib], 0, 0, 0);
                bq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, p.im[kb], bq[ib], 0, 0, 0);
                cq[ib] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, p.su[kb], cq[ib], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
        QOC_LAP(2)
    };
    auto diag = [&](int ib) { return ((ib >> 2) == J) ? idv[ib & 3] : 0.0; };
    // ---- A2 = A * A, polynomial start ------------------------------------------------------------------------------------------
    publish(A);
    product(A);
    {
        Col A2;
#pragma unroll
        for (int ib = 0; ib < QA; ++ib) {
            const double re = a[ib] - bq[ib], im = cq[ib] - a[ib] - bq[ib];
            A2.re[ib] = re; A2.im[ib] = im; A2.su[ib] = re + im;
            X.re[ib] = fma(p_cT, re, fma(p_c1, A.re[ib], p_c0 * diag(ib)));
            X.im[ib] = fma(p_cT, im, p_c1 * A.im[ib]);
            X.su[ib] = X.re[ib] + X.im[ib];
        }
        QOC_LAP(3)
        if (nH > 0) {
            publish(A2);                                                // image of A2 stays through the Horner products
            for (int i = nH - 1; i >= 0; --i) {
                product(X);
                cur ^= 1;                                               // same image again
                const double d0 = mf.invfact[2 * i], d1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int ib = 0; ib < QA; ++ib) {
                    X.re[ib] = (a[ib] - bq[ib]) + fma(d1, A.re[ib], d0 * diag(ib));
                    X.im[ib] = fma(d1, A.im[ib], cq[ib] - a[ib] - bq[ib]);
                    X.su[ib] = X.re[ib] + X.im[ib];
                }
                QOC_LAP(3)
            }
            cur ^= 1;                                                   // the next publish must not overwrite A2 while the partner still reads it
        }
    }
    // ---- squarings ---------------------------------------------------------------------------------------------------------------
    for (int sq = 0; sq < d.s; ++sq) {
        publish(X);
        product(X);
#pragma unroll
        for (int ib = 0; ib < QA; ++ib) {
            const double re = a[ib] - bq[ib], im = cq[ib] - a[ib] - bq[ib];
            X.re[ib] = re; X.im[ib] = im; X.su[ib] = re + im;
        }
        QOC_LAP(3)
    }
    // ---- K_t out: fragD(K) from the registers, fragD(K^T) through the image ------------------------------------------------------------
    const size_t item = kitem(mf, d.steps, b, t);
#pragma unroll
    for (int ib = 0; ib < QA; ++ib) mf.KfD[item + (J * QQS + ib) * 64 + lane] = cmake(X.re[ib], X.im[ib]);
    publish(X);
#pragma unroll
    for (int q = 0; q < QA; ++q) mf.KfT[item + (J * QQS + q) * 64 + lane] = img[cur][(4 * q + (lane >> 4)) * QLDS + 16 * J + (lane & 15)];
    QOC_LAP(4)
    QOC_LAP_DONE
}
