// qoc_mfma_expm_inplace.h -- MFMA path, exponentials K_t = matexp(A_t) + chunk products, qoc_config.variant = 8 (round 3 default for
// n <= 32 batches with Taylor order >= 3): k_mfma_expm_inplace.  Reference semantics: core/tensorflow_state.py:25-46 (get_matexp).
//
// Same arithmetic units as k_mfma_expm_chunk4s (one wave per (seed, chunk), products on v_mfma_f64_4x4x4_4b_f64 in the
// 3-multiplication form, left operand block by block from a transposed LDS image + its re+im sums, right operand in strip
// registers), re-cut around what the round-2 counters said (profiles/r02_pmc_expm_variants.txt): the kernel lost a quarter of its
// time to ~1850 VALU instructions per slice, more than half of them v_accvgpr moves -- 48 accumulators parked in AGPRs and read back
// after every product, operands of the next product kept in registers only to be streamed into the image.
//
//  * ROW-STRIP-MAJOR products.  The 64 block steps of a product run row strip by row strip: group ib = the 8 steps (ib, kb = 0..7)
//    that complete rows 4 ib .. 4 ib + 3 of the result for both column blocks.  Only 6 accumulators are live (VGPRs, never AGPRs:
//    the unit is compiled with -amdgpu-mfma-vgpr-form), and they are combined right at the group boundary: re = a - b,
//    s = c - 2 b (= re + im, the sum the NEXT product needs, for free), im = s - re: 3 VALU instructions per strip.
//  * IN-PLACE IMAGE.  Group ib is the last reader of rows 4 ib .. of the left operand's image, so the result strip of that group is
//    stored into those rows (its four LDS stores ride under the next group's MFMAs, one per block step): at the end of a product
//    the image holds the next left operand.  No operand waits in registers to be streamed, no product starts with an empty
//    image: the block fetches of product N + 1 are issued during the last steps of product N and a slice is ONE stream of MFMAs.
//  * RIGHT OPERANDS COME BACK FROM THE IMAGE.  Where the next product's right operand is this product's result (squarings), its
//    strips are read back from the image into the very registers the current right operand vacates (strip kb after its last use in
//    step (7, kb)): LDS -> register moves cost no VALU instruction and no second register set.
//  * Commuting operands.  Every matrix of the polynomial is a polynomial in A_t, so the Horner step X <- A2 X + B_i is taken as
//    X <- X A2 + B_i: the changing factor is the LEFT operand (image, in place), the fixed A2 the right one (registers).  B_i =
//    d0 I + d1 A enters through the accumulators' initial values (a0 = B_i.re, c0 = B_i.re + B_i.im: 2 instead of 3 VALU per strip).
//  * MONIC polynomial.  The kernel works with S_t = sigma A_t, sigma^T = 1 / T!: in S the Taylor polynomial sum_j pcoef[j] S^j (pcoef[j] =
//    sigma^-j / j!) is monic, so its Horner start is S + c I (odd order) -- the image of S that is already in LDS with a shifted diagonal: no
//    arithmetic but on the diagonal tiles, and the strips go back as planes (nothing pairs re with im) -- or S^2 + c1 S + c0 I (even order); the
//    scale has dissolved when the last Horner product finishes (lambda_i = sigma^2i, lambda_0 = 1).  sigma / 2^s sits in a scaled copy of the
//    Hamiltonian images (QocMfma::HsD), so the assembly of S_t is k fused multiply-adds per entry and nothing else.
//  * Registers: SA = S_t as three planes (re, im, re + im; im dies after the first product), SB = A2, then the squarings' right
//    operand, then R' = K_t R (copied to R after the product); the running chunk product R is touched once per slice and is what
//    the register allocator parks in AGPRs (MFMA reads B operands from there).  K_t goes to HBM strip by strip from the epilogue of
//    the product that completes it; A_{t+1} is assembled strip by strip under the chunk product, written to the image rows that
//    product has released, and read back as the right operand of the next A * A.
#pragma once
#include <type_traits>
#include <utility>
#include "qoc_mfma_frag.h"

#define ILDS (16 * NT + 5)        // image row stride (complex elements): conflict-free strip stores and block reads (as chunk4s)

#ifndef QOC_INPLACE_PIPE
#define QOC_INPLACE_PIPE 0        // 1: combine of group ib - 1 after the first block step of group ib (two accumulator sets) instead of at the group
                                  // boundary -- measured SLOWER, 0.815 against 0.800 ms per launch: the boundary batch does not wait for the pipe (its
                                  // first instructions need the oldest results), and a batch between two MFMAs of a running group costs more issue slots
#endif
#ifndef QOC_INPLACE_STAGE2
#define QOC_INPLACE_STAGE2 0      // 1: the Hamiltonian strips of A_{t+1} are fetched TWO groups ahead of their assembly (two register sets of k + 1 strips)
#endif
#ifndef QOC_INPLACE_SWAP
#define QOC_INPLACE_SWAP 0        // 1: the chunk product writes R' into the set the squarings vacated and the two sets swap ROLES for the next slice
                                  // (time loop unrolled by two) instead of copying R' back into R
#endif
#ifndef QOC_LAP
#define QOC_LAP(ph)
#define QOC_LAP_INIT
#define QOC_LAP_DONE
#endif

namespace qoc_inplace {

template <int NT> struct Set { double re[NT][4 * NT], im[NT][4 * NT], su[NT][4 * NT]; };   // strip (J, ib): rows 4 ib .., columns 16 J ..; lane = 16 row + col
// left blocks in flight (RA = 3 steps ahead, NS % 4 == 0) + the result strip of the last completed group, whose four LDS stores are
// issued one per block step under the NEXT group's MFMAs (issued together at the group boundary they occupy the wave's LDS queue for ~200
// cycles and the block reads behind them arrive late)
template <int NT> struct Ring { cplx v[4]; double s[4]; cplx pri[NT]; double psu[NT]; double ppl[3][NT]; };   // ppl: a pending strip as planes (PLANES products)

__device__ __forceinline__ void fence() { asm volatile("" ::: "memory"); }
template <class F, int... I>
__device__ __forceinline__ void for_each(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// strip (J, ib) of the matrix in the image -> the three planes of a set (strip layout = the store pattern: conflict-free)
// PAIR: re and im by one ds_read_b128 (conflict-free with this stride; they then share a 128-bit register tuple); else two ds_read_b64 (2-way
// bank conflicts, but the planes are separate registers: the imaginary plane of A_t must be free to die after the first product)
template <int NT, bool PAIR>
__device__ __forceinline__ void load_strip(const cplx* img, const double* imgs, int lane, Set<NT>& S, int J, int ib) {
    const int o = (16 * J + (lane & 15)) * ILDS + 4 * ib + (lane >> 4);
    if constexpr (PAIR) { const cplx v = img[o]; S.re[J][ib] = v.x; S.im[J][ib] = v.y; }
    else { const double* p = (const double*)(img + o); S.re[J][ib] = p[0]; S.im[J][ib] = p[1]; }
    S.su[J][ib] = imgs[o];
}

// One product acc = (image) * P, row strip by row strip.
//   init(ibc, a, c) -> std::true_type if it preset the accumulators a[J], c[J] of this group (Horner term); std::false_type: they start from zero
//   epi(ibc, a, b, c, ori, osu): the group's accumulators are complete: combine; ori / osu = the strip that replaces rows 4 ib .. of the image
// RELOAD_OUT: the result is the NEXT product's right operand too: strip kb of P is read back from the image after its last use (step (7, kb));
// RELOAD_IN: this product's strip QS - 1 is such a read-back still to be issued (its stores are this product's first four).
// The ring holds the blocks of steps st .. st + RA - 1 on entry and those of the NEXT product's first steps on exit (every product
// of this kernel reads its left operand from the same image, whose rows 0 .. are complete long before the previous product ends).
// PLANES: the epilogue hands the image strip over as three planes (ring.ppl: re, im, re + im) that are stored by three ds_write_b64 each --
// for results that ARE registers of another matrix (X = S + c I: no VALU instruction to pair re with im); the last strip is paired for the next product.
// QA = ACTIVE 4-row strips per matrix dimension (ceil(n / 4), 5 .. 4 NT): a problem padded to 16 NT has all-zero rows and columns beyond 4 QA, and
// neither the groups that would complete those rows nor the block steps over those inner indices run -- (QA / 8)^2 of the MFMAs of the padded product
// at NT = 2.  The ring of left blocks is indexed by step & 3, so a product is NSP = NS rounded up to a multiple of 4 steps long: the steps NS .. NSP - 1
// are virtual (they exist only as positions of the ring; nothing is fetched for them and nothing multiplies).
template <int NT, int QA, bool RELOAD_IN, bool RELOAD_OUT, bool PLANES, class Init, class Epi>
__device__ __forceinline__ void product(cplx* img, double* imgs, int lane, Ring<NT>& ring, Set<NT>& P, Init&& init, Epi&& epi) {
    constexpr int QS = QA, NS = QS * QS, NSP = (NS + 3) & ~3, RA = 3;
    static_assert(QA >= 5 && QA <= 4 * NT, "active strips: the stores of a pending strip and the read-backs need five block steps per group");
    const cplx* base = img + (lane >> 4) * ILDS + (lane & 3);
    const double* bases = imgs + (lane >> 4) * ILDS + (lane & 3);
    cplx* wbase = img + (lane & 15) * ILDS + (lane >> 4);           // strip (J, ib) -> wbase[16 J ILDS + 4 ib]
    double* wbases = imgs + (lane & 15) * ILDS + (lane >> 4);
    auto fetch = [&](int st) {
        const int s2 = st % NSP, ib = s2 / QS, kb = s2 % QS;
        if (s2 >= NS) return;                                           // a virtual step
        ring.v[st & 3] = base[4 * kb * ILDS + 4 * ib];
        ring.s[st & 3] = bases[4 * kb * ILDS + 4 * ib];
    };
    // (QOC_INPLACE_PIPE = 1: the combine of group ib - 1 runs after the first block step of group ib, on the other accumulator set)
    double acc[2][3][NT];
    for_each([&](auto ibc) {
        constexpr int ib = decltype(ibc)::value, pib = (ib + QS - 1) % QS;      // pending strip: the previous group's (the previous product's last)
        constexpr int off = (QOC_INPLACE_PIPE && ib > 0) ? 1 : 0;               // its stores follow the (deferred) combine
        double (&a)[NT] = acc[ib & 1][0];
        double (&b)[NT] = acc[ib & 1][1];
        double (&c)[NT] = acc[ib & 1][2];
        constexpr bool preset = decltype(init(ibc, a, c))::value;
        init(ibc, a, c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < QS; ++kb) {
            const int st = ib * QS + kb;
            fetch(st + RA);
            fence();
            if (PLANES && ib > 0) {                                     // this product's own strips: three plane stores per column block
#pragma unroll
                for (int q = kb - off; q >= 0 && q < 3 * NT; q += QS) {     // (one per block step; with fewer steps than stores the first steps take two)
                    const int J = q / 3, w = q % 3;
                    if (w < 2) ((double*)(wbase + 16 * J * ILDS + 4 * pib))[w] = ring.ppl[w][J];
                    else wbases[16 * J * ILDS + 4 * pib] = ring.ppl[2][J];
                    fence();
                }
            } else if (kb >= off && kb < off + 2 * NT) {                // one store of the pending strip per block step
                const int q = kb - off;
                if ((q & 1) == 0) wbase[16 * (q >> 1) * ILDS + 4 * pib] = ring.pri[q >> 1];
                else wbases[16 * (q >> 1) * ILDS + 4 * pib] = ring.psu[q >> 1];
                fence();
            }
            const cplx v = ring.v[st & 3];
            const double vs = ring.s[st & 3];
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                if (kb == 0) {
                    a[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, P.re[J][kb], preset ? a[J] : 0.0, 0, 0, 0);
                    b[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, P.im[J][kb], 0.0, 0, 0, 0);
                    c[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, P.su[J][kb], preset ? c[J] : 0.0, 0, 0, 0);
                } else {
                    a[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, P.re[J][kb], a[J], 0, 0, 0);
                    b[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, P.im[J][kb], b[J], 0, 0, 0);
                    c[J] = __builtin_amdgcn_mfma_f64_4x4x4f64(vs, P.su[J][kb], c[J], 0, 0, 0);
                }
            }
            if constexpr (QOC_INPLACE_PIPE && ib > 0) {
                if (kb == 0) {                                          // the previous group's VALU batch, under no dependence on the pipe
                    __builtin_amdgcn_sched_barrier(0);
                    static_assert(!PLANES || !QOC_INPLACE_PIPE, "plane epilogues with the deferred combine are not wired up");
                    if constexpr (!PLANES) epi(std::integral_constant<int, ib - 1>{}, acc[(ib - 1) & 1][0], acc[(ib - 1) & 1][1], acc[(ib - 1) & 1][2], ring.pri, ring.psu);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (RELOAD_IN) {
                // (issued behind the step of the strip's last store; with five steps per group the strip multiplies in step 4 already: one step earlier)
                if (ib == 0 && kb == (QS > 2 * NT + 1 ? 2 * NT : 2 * NT - 1)) {   // the previous product's last strip is in the image now
                    fence();
#pragma unroll
                    for (int J = 0; J < NT; ++J) load_strip<NT, true>(img, imgs, lane, P, J, QS - 1);
                    fence();
                }
            }
            if constexpr (RELOAD_OUT) {
                if (ib == QS - 1 && kb < QS - 1) {                      // strip kb has multiplied for the last time: the new matrix's strip kb takes its registers
                    fence();
#pragma unroll
                    for (int J = 0; J < NT; ++J) load_strip<NT, true>(img, imgs, lane, P, J, kb);
                    fence();
                }
            }
        }
        // the group's VALU batch stays a batch: a lone wave's VALU instructions cost MFMA issue slots wherever they stand, least in a group
        if constexpr (!QOC_INPLACE_PIPE || ib == QS - 1) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PLANES) {
                epi(ibc, a, b, c, ring.ppl[0], ring.ppl[1], ring.ppl[2]);
                if constexpr (ib == QS - 1) {                           // the next product stores pairs
#pragma unroll
                    for (int J = 0; J < NT; ++J) { ring.pri[J] = cmake(ring.ppl[0][J], ring.ppl[1][J]); ring.psu[J] = ring.ppl[2][J]; }
                }
            } else epi(ibc, a, b, c, ring.pri, ring.psu);
            __builtin_amdgcn_sched_barrier(0);
        }
    }, std::make_integer_sequence<int, QS>{});
#pragma unroll
    for (int st = NS; st < NSP; ++st) { fetch(st + RA); fence(); }      // the virtual steps: only their look-ahead happens
}

// fragment f of a fragD matrix at uniform address F: scalar base (whole 4 KB groups of fragments) + immediate + 32-bit lane offset, so that
// no per-lane 64-bit address is ever formed (hipcc otherwise precomputes one per (matrix, strip) pair and spills them)
__device__ __forceinline__ const cplx* frag_at(const cplx* F, int f) { return F + (f & ~3) * 64; }
__device__ __forceinline__ cplx* frag_at(cplx* F, int f) { return F + (f & ~3) * 64; }

}  // namespace qoc_inplace

// KC = controls handled by the pipelined assembly (k <= KC; surplus controls carry a zero coefficient); EVEN = even Taylor order; S0 = no
// squaring (the last Horner product completes K_t and stores it: a run-time test there splits every Horner product into a basic block per group)
// QA = active 4-row strips, ceil(n / 4) (see product()): the strips QA .. 7 of K_t are never written (the buffer is cleared once at set-up: zero rows
// and columns, which is what the sweeps need of a padded propagator), those of the chunk product keep the identity they start from.
template <int KC, bool EVEN, bool S0, int QA = 8>
__global__ void __launch_bounds__(64, 1) k_mfma_expm_inplace(QocDev d, QocMfma mf) {
    using namespace qoc_inplace;
    constexpr int NT = 2, QS = 4 * NT;
    __shared__ __attribute__((aligned(16))) cplx img[QNP * ILDS];
    __shared__ __attribute__((aligned(16))) double imgs[QNP * ILDS];
    const int lane = threadIdx.x;
    const unsigned ulane = threadIdx.x;
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;
    QOC_LAP_INIT
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const int dlt = (lane & 15) - (lane >> 4);
    double idv[4];                                    // identity pattern of a diagonal tile: strip r of the tile holds the diagonal where dlt == 4 r
#pragma unroll
    for (int r = 0; r < 4; ++r) idv[r] = dlt == 4 * r ? 1.0 : 0.0;
    const int mm = d.T >> 1;
    constexpr bool even = EVEN;
    const int nH = even ? mm - 1 : mm;                // Horner products over A2 (tensorflow_state.py:37-41 in Paterson-Stockmeyer form); >= 1 here (T >= 3)
    // the polynomial in the scaled variable S = sigma A_t is monic (QocMfma::pcoef): its Horner start is S + c0 I (odd order) or S^2 + c1 S + c0 I
    const double p_c0 = even ? mf.pcoef[2 * mm - 2] : mf.pcoef[2 * mm], p_c1 = even ? mf.pcoef[2 * mm - 1] : 1.0;

    Set<NT> SA, SB, R;
    constexpr bool SWAP = QOC_INPLACE_SWAP && KC <= 4;   // (the KC = 8 instances with two slice bodies crash hipcc's AGPR-copy rewrite pass)
    Ring<NT> ring;
    auto diag = [&](int J, int ib) -> double { return (ib >> 2) == J ? idv[ib & 3] : 0.0; };
    auto no_init = [](auto, double (&)[NT], double (&)[NT]) { return std::false_type{}; };

    // chunk product starts as the identity
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int ib = 0; ib < QS; ++ib) {
            R.re[J][ib] = diag(J, ib); R.im[J][ib] = 0.0; R.su[J][ib] = diag(J, ib);
            if (SWAP && ib >= QA) { SB.re[J][ib] = diag(J, ib); SB.im[J][ib] = 0.0; SB.su[J][ib] = diag(J, ib); }
        }

    const cplx* hk[KC + 1];
    hk[0] = mf.HsD;                                   // Hamiltonian images already scaled by sigma / 2^s
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) hk[kk + 1] = mf.HsD + (size_t)(kk < d.k ? kk + 1 : 0) * QFR;
    auto coeffs = [&](int t, double (&ck)[KC]) {
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) ck[kk] = kk < d.k ? d.u[((size_t)b * d.k + kk) * d.steps + t] : 0.0;
    };
    // strip (J, ib) of S_t = sigma (H0' + sum_k u_k H_k') / 2^s from the staged (scaled) Hamiltonian strips
    auto assemble = [&](const cplx (&h)[KC + 1], const double (&ck)[KC], double& re, double& im) {
        re = h[0].x; im = h[0].y;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) { re = fma(ck[kk], h[kk + 1].x, re); im = fma(ck[kk], h[kk + 1].y, im); }
    };
    // result strip of a plain product: re = a - b, su = c - 2 b = re + im, im = su - re
    auto combine = [](double a, double bq, double cq, double& re, double& im, double& su) { re = a - bq; su = fma(-2.0, bq, cq); im = su - re; };

    // ---- first slice of the chunk: A_t assembled in the open, image + set SA ---------------------------------------------------------
    {
        double ck[KC];
        coeffs(t0, ck);
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int ib = 0; ib < QA; ++ib) {
                cplx h[KC + 1];
#pragma unroll
                for (int kk = 0; kk <= KC; ++kk) h[kk] = frag_at(hk[kk], J * QS + ib)[((J * QS + ib) & 3) * 64 + ulane];
                double re, im;
                assemble(h, ck, re, im);
                SA.re[J][ib] = re; SA.im[J][ib] = im; SA.su[J][ib] = re + im;
                if (ib == QA - 1) { ring.pri[J] = cmake(re, im); ring.psu[J] = re + im; }      // the last strip is the first product's pending one
                else {
                    const int o = (16 * J + (lane & 15)) * ILDS + 4 * ib + (lane >> 4);
                    img[o] = cmake(re, im); imgs[o] = re + im;
                }
                fence();                                  // one strip's loads at a time: hoisted together they are 320 registers
            }
        {
            const cplx* base = img + (lane >> 4) * ILDS + (lane & 3);
            const double* bases = imgs + (lane >> 4) * ILDS + (lane & 3);
#pragma unroll
            for (int st = 0; st < 3; ++st) { ring.v[st] = base[4 * st * ILDS]; ring.s[st] = bases[4 * st * ILDS]; }   // steps (0, kb = 0..2)
        }
        fence();
    }
    QOC_LAP(0)

    auto slice = [&](Set<NT>& SB, Set<NT>& R, int t) __attribute__((always_inline)) {
        cplx* Kout = mf.KfD + kitem(mf, d.steps, b, t);
        // ---- S2 = S * S -> SB;  Horner start X = S + c0 I (odd order: the strips of S with a shifted diagonal -- planes, no pairing) or
        //      S2 + c1 S + c0 I (even order) -> image (left operand of the first Horner product) ---------------------------------------------
        product<NT, QA, false, false, true>(img, imgs, lane, ring, SA, no_init,
                                        [&](auto ibc, double (&a)[NT], double (&bq)[NT], double (&cq)[NT], double (&ore)[NT], double (&oim)[NT], double (&osu)[NT]) {
            constexpr int ib = decltype(ibc)::value;
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                double re, im, su;
                combine(a[J], bq[J], cq[J], re, im, su);
                SB.re[J][ib] = re; SB.im[J][ib] = im; SB.su[J][ib] = su;
                if constexpr (even) {
                    ore[J] = re + fma(p_c1, SA.re[J][ib], p_c0 * diag(J, ib));
                    oim[J] = fma(p_c1, SA.im[J][ib], im);
                    osu[J] = ore[J] + oim[J];
                } else if ((ib >> 2) == J) {
                    ore[J] = fma(p_c0, diag(J, ib), SA.re[J][ib]); oim[J] = SA.im[J][ib]; osu[J] = fma(p_c0, diag(J, ib), SA.su[J][ib]);
                } else {
                    ore[J] = SA.re[J][ib]; oim[J] = SA.im[J][ib]; osu[J] = SA.su[J][ib];
                }
            }
        });
        QOC_LAP(1)
        // ---- Horner over A2 with the factors commuted: X <- X * A2 + (d0 I + d1 A); the last one is followed by a product that takes its
        //      right operand from the image (squaring) or needs none (s = 0: the result is K_t) ------------------------------------------------
        for (int i = nH - 1; i >= 0; --i) {
            const double d0 = mf.pcoef[2 * i], d1 = mf.pcoef[2 * i + 1];
            auto init = [&](auto ibc, double (&a)[NT], double (&cq)[NT]) {
                constexpr int ib = decltype(ibc)::value;
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    a[J] = fma(d1, SA.re[J][ib], d0 * diag(J, ib));
                    cq[J] = fma(d1, SA.su[J][ib], d0 * diag(J, ib));
                }
                return std::true_type{};
            };
            auto epi = [&](auto ibc, double (&a)[NT], double (&bq)[NT], double (&cq)[NT], cplx (&ori)[NT], double (&osu)[NT]) {
                constexpr int ib = decltype(ibc)::value;
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    double re, im, su;
                    combine(a[J], bq[J], cq[J], re, im, su);
                    ori[J] = cmake(re, im); osu[J] = su;
                    if constexpr (S0) { if (i == 0) frag_at(Kout, J * QS + ib)[((J * QS + ib) & 3) * 64 + ulane] = ori[J]; }
                }
            };
            if (i > 0) product<NT, QA, false, false, false>(img, imgs, lane, ring, SB, init, epi);
            else product<NT, QA, false, true, false>(img, imgs, lane, ring, SB, init, epi);
        }
        QOC_LAP(2)
        // ---- squarings: X <- X * X; the right operand is read back from the image into SB strip by strip ---------------------------------
        for (int sq = 0; sq < d.s; ++sq) {
            const bool kout = sq == d.s - 1;
            product<NT, QA, true, true, false>(img, imgs, lane, ring, SB, no_init,
                                    [&](auto ibc, double (&a)[NT], double (&bq)[NT], double (&cq)[NT], cplx (&ori)[NT], double (&osu)[NT]) {
                constexpr int ib = decltype(ibc)::value;
#pragma unroll
                for (int J = 0; J < NT; ++J) {
                    double re, im, su;
                    combine(a[J], bq[J], cq[J], re, im, su);
                    ori[J] = cmake(re, im); osu[J] = su;
                    if (kout) frag_at(Kout, J * QS + ib)[((J * QS + ib) & 3) * 64 + ulane] = ori[J];      // K_t, strip (J, ib)
                }
            });
        }
        QOC_LAP(3)
        // ---- chunk product R' = K_t * R -> SB (free: the squarings' right operand is not needed any more); A_{t+1} assembled strip by
        //      strip under the same MFMAs and written to the image rows the product has released ----------------------------------------------
        {
            double ck[KC];
            coeffs(min(t + 1, d.steps - 1), ck);
            constexpr int HD = QOC_INPLACE_STAGE2 ? 2 : 1;             // groups of look-ahead of the Hamiltonian strips
            cplx h[HD][NT][KC + 1];
            auto stage = [&](int ib) {
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int kk = 0; kk <= KC; ++kk) h[ib % HD][J][kk] = frag_at(hk[kk], J * QS + ib)[((J * QS + ib) & 3) * 64 + ulane];
            };
            stage(0);
            if constexpr (HD == 2) stage(1);
            Set<NT>& RN = SB;                  // R' (with the swap it stays there: the caller exchanges the roles)
            product<NT, QA, false, false, false>(img, imgs, lane, ring, R, no_init,
                                      [&](auto ibc, double (&a)[NT], double (&bq)[NT], double (&cq)[NT], cplx (&ori)[NT], double (&osu)[NT]) {
                constexpr int ib = decltype(ibc)::value;
#pragma unroll
                for (int J = 0; J < NT; ++J) {                          // A_{t+1}, strip (J, ib) -> image
                    double re, im;
                    assemble(h[ib % HD][J], ck, re, im);
                    ori[J] = cmake(re, im); osu[J] = re + im;
                }
                if constexpr (ib + HD < QA) stage(ib + HD);
#pragma unroll
                for (int J = 0; J < NT; ++J) combine(a[J], bq[J], cq[J], RN.re[J][ib], RN.im[J][ib], RN.su[J][ib]);
            });
            QOC_LAP(4)
            if constexpr (!SWAP) {
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int ib = 0; ib < QA; ++ib) { R.re[J][ib] = SB.re[J][ib]; R.im[J][ib] = SB.im[J][ib]; R.su[J][ib] = SB.su[J][ib]; }   // (strips beyond QA stay the identity)
            }
            // right operand of the next A * A: read back from the image; the last strip is still pending, i.e. in registers
            fence();
#pragma unroll
            for (int J = 0; J < NT; ++J) {
#pragma unroll
                for (int ib = 0; ib < QA - 1; ++ib) load_strip<NT, false>(img, imgs, lane, SA, J, ib);
                SA.re[J][QA - 1] = ring.pri[J].x; SA.im[J][QA - 1] = ring.pri[J].y; SA.su[J][QA - 1] = ring.psu[J];
            }
            fence();
            QOC_LAP(5)
        }
    };
    if constexpr (SWAP) {
    // the chunk product of slice t lands in the set that was "SB" during that slice: it is "R" of slice t + 1, whose A2 / squaring operand / R''
    // take the set the old chunk product has left.  (The strips beyond QA of BOTH sets hold the identity.)
    Set<NT>* Rfin = &R;
    for (int t = t0; t < t1; t += 2) {
        slice(SB, R, t); Rfin = &SB;
        if (t + 1 < t1) { slice(R, SB, t + 1); Rfin = &R; }
    }
    if (Rfin != &R) {
#pragma unroll
        for (int J = 0; J < NT; ++J)
#pragma unroll
            for (int ib = 0; ib < QS; ++ib) { R.re[J][ib] = SB.re[J][ib]; R.im[J][ib] = SB.im[J][ib]; }
    }
    } else {
        for (int t = t0; t < t1; ++t) slice(SB, R, t);
    }
    // ---- chunk product out: fragD(P_c) from the registers, fragD(P_c^T) through the image ------------------------------------------------
    const size_t pitem = (size_t)b * mf.C + c;
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int ib = 0; ib < QS; ++ib) frag_at(mf.PfD + pitem * QFR, J * QS + ib)[((J * QS + ib) & 3) * 64 + ulane] = cmake(R.re[J][ib], R.im[J][ib]);
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int ib = 0; ib < QS; ++ib) img[(16 * J + (lane & 15)) * ILDS + 4 * ib + (lane >> 4)] = cmake(R.re[J][ib], R.im[J][ib]);
    wave_lds_fence();
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int q = 0; q < QS; ++q) frag_at(mf.PfT + pitem * QFR, J * QS + q)[((J * QS + q) & 3) * 64 + ulane] = img[(4 * q + (lane >> 4)) * ILDS + 16 * J + (lane & 15)];
    QOC_LAP(6)
    QOC_LAP_DONE
}
