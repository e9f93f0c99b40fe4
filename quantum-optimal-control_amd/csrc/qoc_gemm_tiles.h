// qoc_gemm_tiles.h -- k_zgemm32: batched complex-fp64 GEMM on 32x32 output tiles (GEMM path building block).
//
// ONE wavefront per 32x32 output tile, v_mfma_f64_16x16x4_f64 with the 3-multiplication complex form (12 MFMAs per 4-deep
// k-slice, 12 independent accumulator chains); operand fragments are loaded straight from global memory in the MFMA A/B
// lane layouts (B rows are coalesced 256-byte segments; A is a 16-row x 64-byte gather that L1/L2 absorb), no LDS and no
// barriers.  Epilogues: C = alpha*op(A)*B + beta*E + gamma*I (EPI 0), per-tile dot with conj(L) (EPI 1), per-column dots
// with conj(L) (EPI 2).  Small batches split the inner dimension over SK waves of a workgroup (partials meet in LDS).
#pragma once
#include "qoc_common.h"


typedef double gd4 __attribute__((ext_vector_type(4)));
#define GMFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define QOC_TW 32          // thin (vector block) width

struct GemmArgs {
    const cplx* A; long long sA; int lda;      // left operand  (batch stride in elements; 0 = shared)
    const cplx* Bm; long long sB; int ldb;     // right operand
    cplx* C; long long sC; int ldc;            // output
    const cplx* E; long long sE; int lde;      // optional addend (nullptr = none)
    double alpha, beta, gamma;                 // C = alpha*op(A)*B + beta*E + gamma*I
    int Kdim;                                  // inner dimension (multiple of 4)
    int tiles_m, tiles_n;                      // output tiles per matrix
    int batch;
    // dot epilogue (EPI = 1): partial[batch][tile_m] = Re sum conj(L)*(A*B) over the tile
    const cplx* L; long long sL; int ldl;
    double* partial; int partial_stride;       // partial[(batch*partial_stride) + offset + tile_m]
    int partial_offset;
    int inner; long long sA2, sB2, sC2, sL2;   // inner > 0: batch index bt -> (bt / inner, bt % inner); A, Bm, C, L offsets = hi*s?2 + lo*s?
    cplx* CT; long long sCT; int ldct;         // EPI = 0, optional: also store the transpose, CT[batch][col][row]
    int ldp;                                   // EPI = 2: per-COLUMN dots, partial[batch*stride + offset + tile_m*ldp + col]
    // k_zgemm_wg only: the right operand is bt_c1 * Bm + bt_c0 * I (square Bm), formed while its chunks are staged -- the top block of a
    // Paterson-Stockmeyer recursion then never exists in memory (no k_gemm_ps_init pass: one read and one write of every matrix saved)
    int btrans; double bt_c0, bt_c1;
};

// SK wavefronts of a workgroup split the inner dimension of ONE tile (small-batch chain launches are latency-bound when
// a single wave walks all of K); partial (re, im) tiles meet in LDS (16 KB per extra wave), wave 0 runs the epilogue.
template <bool CONJT, int EPI, int SK>
__global__ void __launch_bounds__(64 * SK) k_zgemm32(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) double sk_part[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles = g.tiles_m * g.tiles_n;
    const int bt = blockIdx.x / tiles, tile = blockIdx.x - bt * tiles;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int r0 = tm * 32, c0 = tn * 32;
    const int bhi = g.inner > 0 ? bt / g.inner : 0, blo = g.inner > 0 ? bt - bhi * g.inner : bt;
    const cplx* __restrict__ A = g.A + (size_t)bhi * g.sA2 + (size_t)blo * g.sA;
    const cplx* __restrict__ Bm = g.Bm + (size_t)bhi * g.sB2 + (size_t)blo * g.sB;
    gd4 t1[2][2], t2[2][2], t3[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) { t1[I][J] = (gd4){0, 0, 0, 0}; t2[I][J] = (gd4){0, 0, 0, 0}; t3[I][J] = (gd4){0, 0, 0, 0}; }
    const int lr = lane & 15, lk = lane >> 4;
    const int kspan = g.Kdim / SK;
    for (int k0 = wv * kspan; k0 < (wv + 1) * kspan; k0 += 8) {     // two k-slices per trip: 8 loads in flight
        cplx a[2][2], b[2][2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int kk = k0 + 4 * qq + lk;
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                if (CONJT) a[qq][I] = A[(size_t)kk * g.lda + r0 + 16 * I + lr];
                else a[qq][I] = A[(size_t)(r0 + 16 * I + lr) * g.lda + kk];
            }
#pragma unroll
            for (int J = 0; J < 2; ++J) b[qq][J] = Bm[(size_t)kk * g.ldb + c0 + 16 * J + lr];
        }
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const double ar = a[qq][I].x, ai = CONJT ? -a[qq][I].y : a[qq][I].y, as = ar + ai;
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    const double br = b[qq][J].x, bi = b[qq][J].y;
                    t1[I][J] = GMFMA(ar, br, t1[I][J]);
                    t2[I][J] = GMFMA(ai, bi, t2[I][J]);
                    t3[I][J] = GMFMA(as, br + bi, t3[I][J]);
                }
            }
    }
    // combine the 3-multiplication accumulators (linear, so partial K ranges simply add)
    gd4 re[2][2], im[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J) { re[I][J] = t1[I][J] - t2[I][J]; im[I][J] = t3[I][J] - t1[I][J] - t2[I][J]; }
    if (SK > 1) {
        if (wv > 0) {
            double* dst = sk_part + (size_t)(wv - 1) * 2048 + lane;             // [32 values][64 lanes]
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[(((I * 2 + J) * 4 + r) * 2 + 0) * 64] = re[I][J][r];
                        dst[(((I * 2 + J) * 4 + r) * 2 + 1) * 64] = im[I][J][r];
                    }
        }
        __syncthreads();
        if (wv > 0) return;
        // one partial tile at a time: fully unrolled, hipcc issues the 32 (SK - 1) LDS reads of ALL partial tiles before the first add
        // (SK = 8: 448 registers wanted, 988 B of scratch per lane -- on the thin chain steps of N > 64, which are nothing but latency)
#pragma unroll 1
        for (int w = 1; w < SK; ++w) {
            const double* src = sk_part + (size_t)(w - 1) * 2048 + lane;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int J = 0; J < 2; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        re[I][J][r] += src[(((I * 2 + J) * 4 + r) * 2 + 0) * 64];
                        im[I][J][r] += src[(((I * 2 + J) * 4 + r) * 2 + 1) * 64];
                    }
        }
    }
    // D layout: register r of tile (I, J) <-> (row = r0 + 16I + (lane>>4) + 4r, col = c0 + 16J + (lane&15))
    double part = 0.0;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 16 * I + lk + 4 * r, col = c0 + 16 * J + lr;
                const double vre = re[I][J][r], vim = im[I][J][r];
                if (EPI == 0) {
                    cplx v = cmake(g.alpha * vre, g.alpha * vim);
                    if (g.E) {
                        const cplx e = g.E[(size_t)bt * g.sE + (size_t)row * g.lde + col];
                        v.x = fma(g.beta, e.x, v.x); v.y = fma(g.beta, e.y, v.y);
                    }
                    if (row == col) v.x += g.gamma;
                    g.C[(size_t)bhi * g.sC2 + (size_t)blo * g.sC + (size_t)row * g.ldc + col] = v;
                    if (g.CT) g.CT[(size_t)bt * g.sCT + (size_t)col * g.ldct + row] = v;
                } else if (EPI == 1) {
                    const cplx l = g.L[(size_t)bt * g.sL + (size_t)row * g.ldl + col];
                    part = fma(l.x, vre, part); part = fma(l.y, vim, part);      // Re(conj(l) * y)
                }
            }
    if (EPI == 1) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
        if (lane == 0) g.partial[(size_t)bt * g.partial_stride + g.partial_offset + tm] = part;
    }
    if (EPI == 2) {                                            // Re sum_rows conj(L[row][col]) * (A*B)[row][col] for every column
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            double pc = 0.0;
#pragma unroll
            for (int I = 0; I < 2; ++I)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r0 + 16 * I + lk + 4 * r, col = c0 + 16 * J + lr;
                    const cplx l = g.L[(size_t)bhi * g.sL2 + (size_t)blo * g.sL + (size_t)row * g.ldl + col];
                    pc = fma(l.x, re[I][J][r], pc); pc = fma(l.y, im[I][J][r], pc);
                }
            pc += __shfl_xor(pc, 16, 64);
            pc += __shfl_xor(pc, 32, 64);
            if (lk == 0) g.partial[(size_t)bt * g.partial_stride + g.partial_offset + (size_t)tm * g.ldp + c0 + 16 * J + lr] = pc;
        }
    }
}

// ---- k_zgemm_wg: plain products (EPI 0, no CONJT) of large matrices on v_mfma_f64_4x4x4 --------------------------------
// The 16x16x4 instruction of k_zgemm32 issues every ~103 cycles per SIMD (48 TFLOP/s); the 4x4x4 form reaches 73, but a 32x32
// tile per wave fed from L1/L2 needs ~12 TB/s of operand traffic at that rate (measured: 2x SLOWER).  Here a workgroup of 4
// waves owns a 64x128 tile of C (wave tile 32x64: 8 row blocks x 4 column strips = 96 accumulators in AGPRs) and shares the
// operands through LDS: per 16-deep k-chunk the workgroup stages A (64 x 16) and B (16 x 128) together with their re+im sums
// (a v_add_f64 inside the product loop costs as much as half an MFMA) -- fetched into registers while the previous chunk
// multiplies, written to LDS between two LDS-only barriers.  In the loop a wave reads its 4x4 blocks of A (broadcast over the
// four block lanes) through a 3-slot ring two steps ahead and the 4-row strips of B of the next k-block while this one
// multiplies; __builtin_amdgcn_sched_barrier pins that order (left alone hipcc sinks the reads next to their use).
// The kernel lives in its own translation unit (qoc_gemm_wg.hip, compiled WITHOUT -amdgpu-mfma-vgpr-form: its 96 accumulators belong in
// AGPRs, where they cost nothing until the epilogue; the rest of the GEMM path gains from the VGPR form); other units see the two host entries.
void qoc_zgemm_wg_launch(const struct GemmArgs& g, unsigned blocks, hipStream_t s);
bool qoc_zgemm_wg_opt_in();
#define ZW_KC 16           // depth of a k-chunk (32, filling the 160 KB of LDS, was slower: C5 281 vs 271 ms)
#define ZW_LDA (ZW_KC + 4) // + 4: the 16 (row, k) addresses of a block read fall into distinct banks
static inline size_t qoc_zgemm_wg_lds() { return (size_t)2 * (64 * ZW_LDA + ZW_KC * 128) * (sizeof(cplx) + sizeof(double)); }   // 159744 B of the 160 KB
#ifdef QOC_ZGEMM_WG_TU
#ifndef ZW_WAVES
#define ZW_WAVES 8         // waves per workgroup: 8 = two per SIMD, a 32 x 32 wave tile each (48 accumulators); 4 = one per SIMD, 32 x 64 (96)
#endif
// Two waves per SIMD (round 3): with one, the matrix pipe was busy 60 % of the cycles -- a lone wave's s_waitcnt / barrier time (23 % of its
// cycles, SQ_WAIT_ANY) and its ~90 VALU instructions per chunk are all exposed; a partner wave's MFMAs fill them.
template <int NJ> struct ZwB { cplx v[NJ]; double s[NJ]; };
template <bool BT>
__global__ void __launch_bounds__(64 * ZW_WAVES, 1) k_zgemm_wg(GemmArgs g) {
    constexpr int NW = ZW_WAVES, NTHR = 64 * NW, NJ = NW == 8 ? 2 : 4;        // NJ = 16-column strips of a wave tile
    extern __shared__ __attribute__((aligned(16))) char zw_lds[];
    // LDS: two image sets (double buffer), each A [64][ZW_LDA] + B [ZW_KC][128]: the complex parts of both sets, then their re + im sums
    cplx* Ai = (cplx*)zw_lds;                        // set 0: [64][ZW_LDA]
    cplx* Bi = Ai + 64 * ZW_LDA;                     //        [ZW_KC][128]
    double* As = (double*)(Ai + 2 * (64 * ZW_LDA + ZW_KC * 128));        // sums of set 0: [64][ZW_LDA]
    double* Bs = As + 64 * ZW_LDA;                   //                   [ZW_KC][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm2 = g.tiles_m >> 1, tn2 = g.tiles_n >> 2, tiles = tm2 * tn2;
    // workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give XCD x a CONTIGUOUS range of tiles, so that the
    // workgroups sharing an A row panel / a B matrix meet in one L2 (measured neutral at C5: the kernel is not L2-bound)
    const unsigned nb = gridDim.x;
    const unsigned lid = (nb & 7) == 0 ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int bt = lid / tiles, tile = lid - bt * tiles;
    const int tm = tile / tn2, tn = tile - tm * tn2;
    const int r0 = tm * 64, c0 = tn * 128;
    const int bhi = g.inner > 0 ? bt / g.inner : 0, blo = g.inner > 0 ? bt - bhi * g.inner : bt;
    const cplx* __restrict__ A = g.A + (size_t)bhi * g.sA2 + (size_t)blo * g.sA;
    const cplx* __restrict__ Bm = g.Bm + (size_t)bhi * g.sB2 + (size_t)blo * g.sB;
    const int lr = lane & 15, lk = lane >> 4, li4 = lane & 3;
    const int r0w = 32 * (wv & 1), c0w = 16 * NJ * (wv >> 1);
    // staging roles, lane-contiguous in global memory AND in LDS (a first mapping with 4 / 8 consecutive elements per thread
    // put every ds_write on 4 bank groups): element e of a thread is flat index e * NTHR + tid of the chunk,
    // A chunk 64 rows x KC k -> (row = idx / KC, k = idx % KC), B chunk KC k x 128 columns -> (k = idx / 128, column = idx % 128)
    constexpr int KC = ZW_KC, NSA = 64 * KC / NTHR, NSB = KC * 128 / NTHR, RPE = NTHR / KC, KPE = NTHR / 128;   // elements per thread; A rows / B k-rows per e
    const cplx* Ag = A + (size_t)(r0 + tid / KC) * g.lda + (tid % KC);            // + e * RPE rows
    const cplx* Bg = Bm + (size_t)(tid >> 7) * g.ldb + c0 + (tid & 127);          // + e * KPE k-rows
    double t1[8][NJ], t2[8][NJ], t3[8][NJ];
#pragma unroll
    for (int ib = 0; ib < 8; ++ib)
#pragma unroll
        for (int J = 0; J < NJ; ++J) { t1[ib][J] = 0.0; t2[ib][J] = 0.0; t3[ib][J] = 0.0; }
    const int nch = g.Kdim / KC;
    cplx sa[NSA], sb[NSB];
    int staged_ch = 0;                                          // the chunk the staged registers hold (BT: the row index of a B element decides its diagonal term)
    auto stage_fetch = [&](int ch) {
        ch = min(ch, nch - 1);
        staged_ch = ch;
#pragma unroll
        for (int e = 0; e < NSA; ++e) sa[e] = Ag[(size_t)(RPE * e) * g.lda + KC * ch];
#pragma unroll
        for (int e = 0; e < NSB; ++e) sb[e] = Bg[(size_t)(KC * ch + KPE * e) * g.ldb];
        __builtin_amdgcn_sched_barrier(0);
    };
    const cplx* arow = Ai + (r0w + li4) * ZW_LDA + lk;          // + (4 ib) * ZW_LDA + 4 kb
    const double* asrow = As + (r0w + li4) * ZW_LDA + lk;
    const cplx* bcol = Bi + lk * 128 + c0w + lr;                // + (4 kb) * 128 + 16 J
    const double* bscol = Bs + lk * 128 + c0w + lr;
    // Two image sets: while chunk ch multiplies out of set ch & 1, the staged registers of chunk ch + 1 go to the other set ONE LDS store at a
    // time under the first half of its MFMAs, the fetch of chunk ch + 2 is issued at half time into the registers just stored, and ONE
    // barrier per chunk separates "everybody has read set ch & 1 and written the other" from the next chunk.
    constexpr int SET = 64 * ZW_LDA + ZW_KC * 128;            // elements of one image set (complex part; the sums follow the two complex sets)
    cplx* const Ai0 = Ai; double* const As0 = (double*)(Ai + 2 * SET);
    auto store_one = [&](int q, int set) {                      // q = 0 .. 2 (NSA + NSB) - 1: one LDS store of the staged chunk
        cplx* ai = Ai0 + (size_t)set * SET; cplx* bi = ai + 64 * ZW_LDA;
        double* as = As0 + (size_t)set * SET; double* bs = as + 64 * ZW_LDA;
        const int e = q >> 1;
        if (e < NSA) { const int o = (RPE * e + tid / KC) * ZW_LDA + (tid % KC); if (q & 1) as[o] = sa[e].x + sa[e].y; else ai[o] = sa[e]; }
        else {
            const int f = e - NSA, o = (KPE * f + (tid >> 7)) * 128 + (tid & 127);
            cplx v = sb[f];
            if constexpr (BT) {                                 // bt_c1 * B + bt_c0 * I
                const bool dg = KC * staged_ch + KPE * f + (tid >> 7) == c0 + (tid & 127);
                v.x = fma(g.bt_c1, v.x, dg ? g.bt_c0 : 0.0); v.y = g.bt_c1 * v.y;
            }
            if (q & 1) bs[o] = v.x + v.y; else bi[o] = v;
        }
    };
    constexpr int NQ = 2 * (NSA + NSB), NSTEP = 2 * KC;        // the stores of a chunk over the first 16 of its 32 block steps
    stage_fetch(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) store_one(q, 0);
    stage_fetch(1);
    lds_barrier();
    for (int ch = 0; ch < nch; ++ch) {
        const int set = ch & 1;
        const cplx* arow_s = arow + (size_t)set * SET;
        const double* asrow_s = (const double*)((const char*)asrow + (size_t)set * SET * sizeof(double));
        const cplx* bcol_s = bcol + (size_t)set * SET;
        const double* bscol_s = (const double*)((const char*)bscol + (size_t)set * SET * sizeof(double));
        // KC/4 x 8 block steps (kb, ib) of the chunk; A blocks through a 3-slot ring two steps ahead, B strips one k-block ahead
        cplx av[3]; double as[3];
        auto load_a = [&](int st, int slot) { const int kb = st >> 3, ib = st & 7; av[slot] = arow_s[(4 * ib) * ZW_LDA + 4 * kb]; as[slot] = asrow_s[(4 * ib) * ZW_LDA + 4 * kb]; };
        auto load_bs = [&](ZwB<NJ>& b, int kb) {
#pragma unroll
            for (int J = 0; J < NJ; ++J) { b.v[J] = bcol_s[(4 * kb) * 128 + 16 * J]; b.s[J] = bscol_s[(4 * kb) * 128 + 16 * J]; }
        };
        ZwB<NJ> b0, b1;
        load_bs(b0, 0);
        load_a(0, 0);
        load_a(1, 1);
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            const int kb = st >> 3, ib = st & 7;
            if (st + 2 < NSTEP) load_a(st + 2, (st + 2) % 3);
            if (ib == 0 && kb + 1 < KC / 4) load_bs((kb & 1) ? b0 : b1, kb + 1);
            // chunk ch + 1 -> the other set, spread over the first half of the steps (clamped chunks beyond the last one are stored too: harmless)
#pragma unroll
            for (int q = (st * NQ) / (NSTEP / 2); q < ((st + 1) * NQ) / (NSTEP / 2) && q < NQ; ++q) store_one(q, set ^ 1);
            if (st == NSTEP / 2) stage_fetch(ch + 2);          // the staged registers are free again
            __builtin_amdgcn_sched_barrier(0);
            const ZwB<NJ>& b = (kb & 1) ? b1 : b0;
            const cplx a = av[st % 3];
            const double asum = as[st % 3];
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                t1[ib][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, b.v[J].x, t1[ib][J], 0, 0, 0);
                t2[ib][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, b.v[J].y, t2[ib][J], 0, 0, 0);
                t3[ib][J] = __builtin_amdgcn_mfma_f64_4x4x4f64(asum, b.s[J], t3[ib][J], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                          // set ch & 1 has been read by every wave, the other set is complete
    }
    // D strip (ib, J): lane 16 i + c16 <-> (row = r0 + r0w + 4 ib + i, col = c0 + c0w + 16 J + c16)
#pragma unroll
    for (int ib = 0; ib < 8; ++ib)
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
            const int row = r0 + r0w + 4 * ib + lk, col = c0 + c0w + 16 * J + lr;
            const double vre = t1[ib][J] - t2[ib][J], vim = t3[ib][J] - t1[ib][J] - t2[ib][J];
            cplx v = cmake(g.alpha * vre, g.alpha * vim);
            if (g.E) {
                const cplx e = g.E[(size_t)bt * g.sE + (size_t)row * g.lde + col];
                v.x = fma(g.beta, e.x, v.x); v.y = fma(g.beta, e.y, v.y);
            }
            if (row == col) v.x += g.gamma;
            g.C[(size_t)bhi * g.sC2 + (size_t)blo * g.sC + (size_t)row * g.ldc + col] = v;
            if (g.CT) g.CT[(size_t)bt * g.sCT + (size_t)col * g.ldct + row] = v;
        }
}
#endif  // QOC_ZGEMM_WG_TU
