// qoc_mfma_forward.h -- MFMA path: thin forward sweeps Psi_t = K_t Psi_{t-1} (inter vectors), final unitary, unitary_scale.
// Reference semantics: core/tensorflow_state.py:204-242.
#pragma once
#include "qoc_mfma_frag.h"

// ---- kernel F: thin forward sweep  Psi_t = K_t Psi_{t-1}  (inter vectors) + final unitary ------------------------
// grid.x = B*C sweep waves + B*2 final-unitary waves, 4 waves per workgroup, no LDS, no barriers.
// QA: active 4-column groups of the padded K / chunk products (ceil(n / 4); the vectors' rows beyond are zero) in the sweep items
template <int NT, int QA = 4 * NT>
__global__ void __launch_bounds__(256) k_mfma_forward(QocDev d, QocMfma mf) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_sweep = d.B * mf.C;
    if (item < n_sweep) {
        const int b = item / mf.C, c = item - b * mf.C;
        if (d.skip_done && d.done[b]) return;
        const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
        CTile Psi[NT];
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) v = d.Psi0[row * d.m + col];
                Psi[Ib].re[r] = v.x; Psi[Ib].im[r] = v.y;
            }
        cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
        if (c == 0) {                                                   // inter[0] = V  (tensorflow_state.py:232-233)
            for (int o = lane; o < d.n * d.m; o += 64) iv[o] = d.V[o];
        }
        AFragT<NT> A;
        for (int cc = 0; cc < c; ++cc) {                                // chunk boundary from the chunk products
            afrag_load<NT, false, QA>(mf.PfT + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT, QA>(A, Psi, acc);
            for (int Ib = 0; Ib < NT; ++Ib) Psi[Ib] = acc[Ib];
        }
        for (int t = t0; t < t1; ++t) {
            afrag_load<NT, false, QA>(mf.KfT + kitem(mf, d.steps, b, t), lane, A);
            CTile acc[NT];
            mm_colblock<NT, QA>(A, Psi, acc);
            for (int Ib = 0; Ib < NT; ++Ib) Psi[Ib] = acc[Ib];
            cplx* out = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                    if (row < d.n && col < d.m) out[row * d.m + col] = cmake(Psi[Ib].re[r], Psi[Ib].im[r]);
                }
        }
    } else if (item < n_sweep + d.B * NT) {
        // final_state = P_{C-1} ... P_0 U0 (tensorflow_state.py:223), one wave per 16-column half
        const int w = item - n_sweep, b = w / NT, J = w - b * NT;
        if (d.skip_done && d.done[b]) return;
        CTile X[NT];
        colblock_load<NT>(mf.U0fD, J, lane, X);
        AFragT<NT> A;
        for (int cc = 0; cc < mf.C; ++cc) {
            afrag_load<NT, false>(mf.PfT + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, X, acc);
            for (int Ib = 0; Ib < NT; ++Ib) X[Ib] = acc[Ib];
        }
        cplx* Xf = d.Xfinal + (size_t)b * d.n * d.n;
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = 16 * J + (lane & 15);
                if (row < d.n && col < d.n) Xf[row * d.n + col] = cmake(X[Ib].re[r], X[Ib].im[r]);
            }
    }
}

// ---- kernel F2: the thin forward sweep of NT = 2 on v_mfma_f64_4x4x4 ---------------------------------------------------
// Transposed recursion Psi_t^T = Psi_{t-1}^T K_t^T: the right operand (4 k-rows x 16 columns of K^T) is a fragD register of
// KfT as stored, the left operand a 4x4 block of Psi^T read from a wave-private LDS image (broadcast over the 4 blocks), the
// result register (I, jb) holds Psi[row 16 I + lane % 16][column 4 jb + lane / 16]: no output column is padding (a 16x16x4
// tile spends half of its columns on m = 8) -- 48 MQ MFMAs of 17 cycles per slice instead of 48 of ~100.  K_{t+1} is fetched
// while slice t multiplies.  Final-unitary waves as in k_mfma_forward.

// (No run-time branch may surround the loads of the sweep: hipcc waits for a conditional load on the spot -- 173 -> 183 us with an
// `if (mf.latency)` around two load patterns.  The latency mode has its own sweep kernel, qoc_mfma_latency.h.)
// BND: the chunk-start vectors come from k_mfma_bnd_scan (BndF) instead of a walk over the chunk products before this chunk.  Every
// chunk of a seed walked the SAME products at the same time -- an L2 hot spot that made a boundary step 5.8 us against 2.6 us for a
// slice of the sweep itself, half of the kernel's critical path at 16 chunks (profiles/r03_chunks_kernel_table.txt).
// QA: active 4-column groups of the padded K (ceil(n / 4); the columns beyond are zero): neither loaded nor multiplied.
template <int NT, int MQ, bool BND = false, int QA = 4 * NT>
__global__ void __launch_bounds__(256) k_mfma_forward2(QocDev d, QocMfma mf) {
    constexpr int LDP = 16 * NT + 1;
    __shared__ __attribute__((aligned(16))) cplx f2_img[4][16 * LDP];             // per wave: image[column j][row]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wv;
    const int n_sweep = d.B * mf.C;
    if (item < n_sweep) {
        const int c = item / d.B, b = item - c * d.B;      // chunk-major: the waves of a workgroup walk 4 different seeds
        if (d.skip_done && d.done[b]) return;
        const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
        const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
        cplx* img = f2_img[wv];
        double pre[NT][MQ], pim[NT][MQ];
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const int row = 16 * I + lc, col = 4 * jb + lk;
                cplx v = cmake(0.0, 0.0);
                if constexpr (BND) v = mf.BndF[(((size_t)b * mf.C + c) * NT * MQ + I * MQ + jb) * 64 + lane];   // the sweep's own register layout
                else if (row < d.n && col < d.m) v = d.Psi0[row * d.m + col];
                pre[I][jb] = v.x; pim[I][jb] = v.y;
            }
        cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
        if (c == 0) {                                       // inter[0] = V  (tensorflow_state.py:232-233)
            for (int o = lane; o < d.n * d.m; o += 64) iv[o] = d.V[o];
        }
        struct Frag { cplx f[NT][QQS]; };
        // the operand is M^T in strip registers, gathered from fragD(M) (no transposed copy of the 512 MB of K: the sweep is bound by
        // the HBM stream)
        auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int q = 0; q < QA; ++q)  // K^T[4q + lk][16 I + lc] = K[16 I + lc][4q + lk] gathered from fragD(K): quads of lanes (lk) read 64 contiguous bytes
                    fr.f[I][q] = F[((q >> 2) * QQS + 4 * I + (lc >> 2)) * 64 + 16 * (lc & 3) + 4 * (q & 3) + lk];
        };
        // Psi <- M Psi with M^T given by its fragD fragment
        auto product = [&](const Frag& fr) {
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) img[(4 * jb + lk) * LDP + 16 * I + lc] = cmake(pre[I][jb], pim[I][jb]);
            wave_lds_fence();
            double a[NT][MQ], bq[NT][MQ], cq[NT][MQ];
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) { a[I][jb] = 0.0; bq[I][jb] = 0.0; cq[I][jb] = 0.0; }
#pragma unroll
            for (int kb = 0; kb < QA; ++kb) {
                cplx v[MQ];
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) v[jb] = img[(4 * jb + li4) * LDP + 4 * kb + lk];   // Psi[4 kb + lk][4 jb + li4]
#pragma unroll
                for (int I = 0; I < NT; ++I) {
                    const double br = fr.f[I][kb].x, bi = fr.f[I][kb].y, bs = br + bi;
#pragma unroll
                    for (int jb = 0; jb < MQ; ++jb) {
                        a[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x, br, a[I][jb], 0, 0, 0);
                        bq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].y, bi, bq[I][jb], 0, 0, 0);
                        cq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x + v[jb].y, bs, cq[I][jb], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) { pre[I][jb] = a[I][jb] - bq[I][jb]; pim[I][jb] = cq[I][jb] - a[I][jb] - bq[I][jb]; }
        };
        // chunk-start vectors: Psi <- P_cc Psi over the chunks before this one, one matrix at a time -- the chunks of a seed walk the
        // same products, and more loads in flight only deepen that L2 hot spot (32 seeds x 32 chunks: 119 us per launch like this,
        // 170 us with two matrices ahead)
        if constexpr (!BND) {
            Frag B0;
            for (int cc = 0; cc < c; ++cc) { load_frag(mf.PfD + ((size_t)b * mf.C + cc) * QFR, B0); product(B0); }
        }
        auto step = [&](const Frag& fr, int t) {
            product(fr);
            cplx* out = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) {
                    const int row = 16 * I + lc, col = 4 * jb + lk;
                    if (row < d.n && col < d.m) out[row * d.m + col] = cmake(pre[I][jb], pim[I][jb]);
                }
        };
        const cplx* Kb = mf.KfD + kitem(mf, d.steps, b, t0);   // slices of one chunk are FR apart
        const int len = t1 - t0;
        {
#ifndef QOC_FWD_AHEAD
#define QOC_FWD_AHEAD 1
#endif
        constexpr int PD = QOC_FWD_AHEAD;                               // slices in flight ahead of the product
        Frag Kq[PD + 1];
#pragma unroll
        for (int q = 0; q < PD; ++q) load_frag(Kb + (size_t)min(q, len - 1) * mf.FR, Kq[q]);
        int t = 0;
        for (; t + PD + 1 <= len; t += PD + 1) {
#pragma unroll
            for (int q = 0; q <= PD; ++q) {
                load_frag(Kb + (size_t)min(t + q + PD, len - 1) * mf.FR, Kq[(q + PD) % (PD + 1)]); asm volatile("" ::: "memory"); step(Kq[q], t0 + t + q);
            }
        }
#pragma unroll
        for (int q = 0; q <= PD; ++q)
            if (t + q < len) step(Kq[q], t0 + t + q);
        }
    } else if (!BND && item < n_sweep + d.B * NT) {
        // final_state = P_{C-1} ... P_0 U0 (tensorflow_state.py:223), one wave per 16-column half (BND: formed by k_mfma_bnd_scan)
        const int w = item - n_sweep, b = w / NT, J = w - b * NT;
        if (d.skip_done && d.done[b]) return;
        CTile X[NT];
        colblock_load<NT>(mf.U0fD, J, lane, X);
        AFragT<NT> A;
        for (int cc = 0; cc < mf.C; ++cc) {
            afrag_load<NT, false>(mf.PfT + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT>(A, X, acc);
            for (int Ib = 0; Ib < NT; ++Ib) X[Ib] = acc[Ib];
        }
        cplx* Xf = d.Xfinal + (size_t)b * d.n * d.n;
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = 16 * J + (lane & 15);
                if (row < d.n && col < d.n) Xf[row * d.n + col] = cmake(X[Ib].re[r], X[Ib].im[r]);
            }
    }
}

// ---- chunk-boundary scan (batch kernels, NT = 2 / 3): the vectors every chunk starts from, once per seed -------------------------------
// Forward: BndF[c] = P_{c-1} ... P_0 Psi_0 (tensorflow_state.py:229-242 re-associated over the chunk products).  Adjoint, only when no
// state regulariser makes the costate affine: BndA[c] = P_{c+1}^dagger ... P_{C-1}^dagger W, the z-FREE costate at the end of chunk c (the
// terminal costate is -(2/m^2) z W, linear in the overlap z = tr(W^dagger Psi_N) that the loss kernel forms later; the backward sweep scales
// by it).  One wave per (seed, direction, block of 4 vector columns): columns are independent under left multiplication, so a step is
// 12 NT^2 MFMAs and one private LDS image; the next chunk product (lane-contiguous fragD(P^T) / conj fragD(P) strips) is fetched while this
// one multiplies.  Vectors are stored in the sweeps' register layout (row 16 I + lane % 16, column 4 jb + lane / 16).
// The same waves form final_state = P_{C-1} ... P_0 U0 (tensorflow_state.py:223), four columns of U0 per wave (role 2): in the sweep kernel
// that chain was C sequential 32 x 32 products on v_mfma_f64_16x16x4 by two waves per seed -- ~5 us per chunk, the critical path of the
// launch from ~32 chunks on (63 chunks: 405 us of which the sweep itself needs ~90).
template <int NT>
__global__ void __launch_bounds__(256) k_mfma_bnd_scan(QocDev d, QocMfma mf, int MQ, int flags, const cplx* __restrict__ PT, const cplx* __restrict__ PD, int C) {
    // flags: 1 = adjoint boundaries too, 2 = Psi_N = P_{C-1} BndF[C-1] -> d.inter[steps] (the loss needs it before k_mfma_downup runs),
    //        4 = ONLY final_state (on read-back), 8 = no final_state (k_mfma_downup batches form it when it is read back: its 8 waves per
    //        seed were 2/3 of this kernel's reads -- every wave of a seed reads every chunk product: 64 seeds x 32 chunks x 12 waves x 16 KB
    //        = 403 MB, 48 us)
    const int adjoint_too = flags & 1, psi_final = flags & 2;
    constexpr int LDP = 16 * NT + 1;
    __shared__ __attribute__((aligned(16))) cplx sc_img[4][4 * LDP];               // per wave: image[column j % 4][row]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nvec = (adjoint_too ? 2 : 1) * MQ;                                     // roles: Psi blocks, (costate blocks,) column blocks of U0
    const int per_seed = (flags & 4) ? 4 * NT : ((flags & 8) ? nvec : nvec + 4 * NT);
    const int item = blockIdx.x * 4 + wv;
    if (item >= d.B * per_seed) return;
    const int b = item / per_seed, w = item - b * per_seed + ((flags & 4) ? nvec : 0);
    const int role = w < MQ ? 0 : (w < nvec ? 1 : 2), jb = role == 2 ? w - nvec : (role == 1 ? w - MQ : w), dir = role == 1 ? 1 : 0;
    if (d.skip_done && d.done[b]) return;
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    cplx* img = sc_img[wv];
    double pre[NT], pim[NT];                                                        // (PT / PD / C: the chunk products, or the products of groups of chunks)
    {
        const cplx* V0 = role == 2 ? d.U0 : (dir ? d.W : d.Psi0);
        const int ncol = role == 2 ? d.n : d.m;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            const int row = 16 * I + lc, col = 4 * jb + lk;
            cplx v = cmake(0.0, 0.0);
            if (row < d.n && col < ncol) v = V0[row * ncol + col];
            pre[I] = v.x; pim[I] = v.y;
        }
    }
    cplx* out = (dir ? mf.BndA : mf.BndF) + (size_t)b * C * NT * MQ * 64;
    auto store = [&](int c) {
        if (role == 2 || c >= C) return;
#pragma unroll
        for (int I = 0; I < NT; ++I) out[(((size_t)c * NT + I) * MQ + jb) * 64 + lane] = cmake(pre[I], pim[I]);
    };
    struct Frag { cplx f[NT][QQS]; };
    const double sg = dir ? -1.0 : 1.0;                                             // adjoint: conj fragD(P) IS the strip operand of P^dagger
    auto load_frag = [&](int c, Frag& fr) {
        const cplx* F = (dir ? PD : PT) + ((size_t)b * C + c) * QFR;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < QQS; ++q) fr.f[I][q] = F[(I * QQS + q) * 64 + lane];
    };
    auto product = [&](const Frag& fr) {
#pragma unroll
        for (int I = 0; I < NT; ++I) img[lk * LDP + 16 * I + lc] = cmake(pre[I], pim[I]);
        wave_lds_fence();
        double a[NT], bq[NT], cq[NT];
#pragma unroll
        for (int I = 0; I < NT; ++I) { a[I] = 0.0; bq[I] = 0.0; cq[I] = 0.0; }
        cplx vv[QQS];                                                               // all block reads first: the wave is alone on its SIMD, and a
#pragma unroll                                                                      // read issued next to its use exposes the LDS latency QQS times per step
        for (int kb = 0; kb < QQS; ++kb) vv[kb] = img[li4 * LDP + 4 * kb + lk];     // X[4 kb + lk][4 jb + li4]
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) {
            const cplx v = vv[kb];
#pragma unroll
            for (int I = 0; I < NT; ++I) {
                const double br = fr.f[I][kb].x, bi = sg * fr.f[I][kb].y, bs = br + bi;
                a[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br, a[I], 0, 0, 0);
                bq[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi, bq[I], 0, 0, 0);
                cq[I] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x + v.y, bs, cq[I], 0, 0, 0);
            }
        }
#pragma unroll
        for (int I = 0; I < NT; ++I) { pre[I] = a[I] - bq[I]; pim[I] = cq[I] - a[I] - bq[I]; }
        wave_lds_fence();                                                           // the image is rewritten by the next product
    };
    // forward: c = 0 .. C-2 (result -> chunk c + 1; final_state: .. C-1); adjoint: c = C-1 .. 1 (result -> chunk c - 1); step s uses chunk cs(s)
    auto cs = [&](int s) { return dir ? C - 1 - s : s; };
    const int nst = (role == 2 || (role == 0 && psi_final)) ? C : C - 1;
    store(dir ? C - 1 : 0);
    if (nst >= 1) {
        // two chunk products in flight ahead of the one that multiplies
        Frag f0, f1, f2;
        load_frag(cs(0), f0); load_frag(cs(min(1, nst - 1)), f1);
        asm volatile("" ::: "memory");
        auto one = [&](Frag& cur, Frag& nxt, int s) {
            load_frag(cs(min(s + 2, nst - 1)), nxt); asm volatile("" ::: "memory"); product(cur); store(dir ? cs(s) - 1 : cs(s) + 1);
        };
        int s = 0;
        for (; s + 3 <= nst; s += 3) { one(f0, f2, s); one(f1, f0, s + 1); one(f2, f1, s + 2); }
        if (s < nst) { one(f0, f2, s); ++s; }
        if (s < nst) { one(f1, f0, s); ++s; }
    }
    if (role == 0 && psi_final) {
        cplx* out = d.inter + ((size_t)b * (d.steps + 1) + d.steps) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            const int row = 16 * I + lc, col = 4 * jb + lk;
            if (row < d.n && col < d.m) out[row * d.m + col] = cmake(pre[I], pim[I]);
        }
    }
    if (role == 2) {
        cplx* Xf = d.Xfinal + (size_t)b * d.n * d.n;
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            const int row = 16 * I + lc, col = 4 * jb + lk;
            if (row < d.n && col < d.n) Xf[row * d.n + col] = cmake(pre[I], pim[I]);
        }
    }
}

// unitary_scale = (1/n) sum_c |sum_a X[c][a]|^2                      tensorflow_state.py:225
__global__ void __launch_bounds__(64) k_mfma_uscale(QocDev d) {
    const int b = blockIdx.x, n = d.n, lane = threadIdx.x;
    const cplx* X = d.Xfinal + (size_t)b * n * n;
    double part = 0.0;
    for (int c = lane; c < n; c += 64) {
        cplx rs = cmake(0.0, 0.0);
        for (int a = 0; a < n; ++a) rs = cadd(rs, X[c * n + a]);
        part += rs.x * rs.x + rs.y * rs.y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
    if (lane == 0) d.uscale[b] = part / (double)n;
}

// state transfer: unitary_scale = (sum_{a,j} |Psi_N[a][j]|^2)^2 / m^2      tensorflow_state.py:335 (k_loss forms it on the routes that launch it)
__global__ void __launch_bounds__(64) k_mfma_uscale_st(QocDev d) {
    const int b = blockIdx.x, nm = d.n * d.m, lane = threadIdx.x;
    const cplx* fin = d.inter + ((size_t)b * (d.steps + 1) + d.steps) * nm;
    double nrm = 0.0;
    for (int o = lane; o < nm; o += 64) { const cplx f = fin[o]; nrm += f.x * f.x + f.y * f.y; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
    if (lane == 0) d.uscale[b] = nrm * nrm / ((double)d.m * (double)d.m);
}

// latency mode: d.inter[b][t + 1] (API layout, analysis.py:60) from PsiL[b][t]; launched when the vectors are read back
__global__ void __launch_bounds__(256) k_mfma_unpack_inter(QocDev d, QocMfma mf, int MQ) {
    const int NT = mf.NT, per = NT * MQ;
    const size_t total = (size_t)d.B * d.steps * per * 64;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(o & 63);
        const size_t rg = o >> 6;
        const int r = (int)(rg % per);
        const size_t bt = rg / per;
        const int I = r / MQ, jb = r - I * MQ, row = 16 * I + (lane & 15), col = 4 * jb + (lane >> 4);
        const size_t b = bt / d.steps, t = bt - b * d.steps;
        if (row < d.n && col < d.m) d.inter[(b * (d.steps + 1) + t + 1) * d.n * d.m + (size_t)row * d.m + col] = mf.PsiL[o];
    }
}
