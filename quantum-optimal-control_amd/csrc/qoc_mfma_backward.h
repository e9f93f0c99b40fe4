// qoc_mfma_backward.h -- MFMA path: backward sweeps Lambda_{t-1} = K_t^dagger Lambda_t (+ sources), affine chunk offsets and
// the control gradients dL/du_{k,t} = Re <Lambda_t, H_k' Psi_t>.  Reference semantics: core/tensorflow_state.py:49-65.
#pragma once
#include "qoc_mfma_frag.h"
#include "qoc_state_source.h"   // source_at()

// ---- kernel B0: affine offsets of the backward recursion when state regularisers add a source at every slice --------
// Lambda_{t-1} = K_t^dagger Lambda_t + S_{t-1} is affine; over chunk c it maps the chunk-end costate E to
// P_c^dagger E + a_c with a_c = result of running the chunk from a ZERO costate.  One wave per (seed, chunk >= 1).
template <int NT>
__global__ void __launch_bounds__(256) k_mfma_bwd_offsets(QocDev d, QocMfma mf) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (item >= d.B * mf.C) return;
    const int b = item / mf.C, c = item - b * mf.C;
    if (c == 0 || (d.skip_done && d.done[b])) return;                                  // a_0 is never used; finished seeds are frozen
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    CTile Z[NT];
#pragma unroll
    for (int Ib = 0; Ib < NT; ++Ib) { Z[Ib].re = (d4){0, 0, 0, 0}; Z[Ib].im = (d4){0, 0, 0, 0}; }
    AFragT<NT> A;
    for (int t = t1 - 1; t >= t0; --t) {
        afrag_load<NT, true>(mf.KfD + kitem(mf, d.steps, b, t), lane, A);
        CTile acc[NT];
        mm_colblock<NT>(A, Z, acc);
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                cplx sv = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) sv = source_at(d, b, t, row, col);
                Z[Ib].re[r] = acc[Ib].re[r] + sv.x; Z[Ib].im[r] = acc[Ib].im[r] + sv.y;
            }
    }
    colblock_store<NT>(mf.Aoff + ((size_t)b * mf.C + c) * (QQS * 64), 0, lane, Z);
}

// ---- kernel B0': the affine offsets of NT = 2 on v_mfma_f64_4x4x4 ------------------------------------------------------
// Same recursion as k_mfma_bwd_offsets (Z <- K_t^dagger Z + S_t from a zero costate, one wave per (seed, chunk >= 1)) in the
// transposed form of k_mfma_forward2: Z^T <- Z^T conj(K_t), right operand = the fragD(K) registers as stored (contiguous loads),
// left operand = 4x4 blocks of Z^T from a wave-private LDS image.  The source term has conditional loads (waited for on the
// spot), so it is evaluated before the next K_t is fetched.  163 -> 124 us per launch at the regularised C2 x 64.  FULL = true: the same
// sweep from the terminal costate through the chunk boundaries, storing Lambda_t for k_mfma_grad (k >= 6 controls).
// QA: active 4-row strips of the padded K (ceil(n / 4)): the all-zero strips beyond are neither loaded nor multiplied.
template <int MQ, bool FULL, int QA = 8>
__global__ void __launch_bounds__(256) k_mfma_bwd_offsets2(QocDev d, QocMfma mf) {
    constexpr int NT = 2;
    __shared__ __attribute__((aligned(16))) cplx o2_img[4][16 * F2_LDP];          // per wave: image[column j][row]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * (blockDim.x >> 6) + wv;            // 4 waves per workgroup; one in latency mode (one sweep per CU)
    if (item >= d.B * mf.C) return;
    const int b = item / mf.C, c = item - b * mf.C;
    if ((!FULL && c == 0) || (d.skip_done && d.done[b])) return;                       // a_0 is never used; finished seeds are frozen
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    cplx* img = o2_img[wv];
    double zre[2][MQ], zim[2][MQ];                                                      // (I, jb): Z[16 I + lc][4 jb + lk]
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) { zre[I][jb] = 0.0; zim[I][jb] = 0.0; }
    struct Frag { cplx f[2][8]; };
    auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int q = 0; q < QA; ++q) fr.f[I][q] = F[(I * QQS + q) * 64 + lane];        // K[4q + lk][16 I + lc]
    };
    double sre[2][MQ], sim[2][MQ];
    auto source = [&](int t) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const int row = 16 * I + lc, col = 4 * jb + lk;
                cplx sv = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) sv = source_at(d, b, t, row, col);
                sre[I][jb] = sv.x; sim[I][jb] = sv.y;
            }
    };
    const bool need_src = d.n_forb > 0 || d.has_speed;
    auto step = [&](const Frag& fr) {                                                  // Z <- K^dagger Z + S
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) img[(4 * jb + lk) * F2_LDP + 16 * I + lc] = cmake(zre[I][jb], zim[I][jb]);
        wave_lds_fence();
        double a[2][MQ], bq[2][MQ], cq[2][MQ];
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { a[I][jb] = 0.0; bq[I][jb] = 0.0; cq[I][jb] = 0.0; }
#pragma unroll
        for (int kb = 0; kb < QA; ++kb) {
            cplx v[MQ];
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) v[jb] = img[(4 * jb + li4) * F2_LDP + 4 * kb + lk];   // Z[4 kb + lk][4 jb + li4]
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                const double br = fr.f[I][kb].x, bi = -fr.f[I][kb].y, bs = br + bi;
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) {
                    a[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x, br, a[I][jb], 0, 0, 0);
                    bq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].y, bi, bq[I][jb], 0, 0, 0);
                    cq[I][jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[jb].x + v[jb].y, bs, cq[I][jb], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { zre[I][jb] = a[I][jb] - bq[I][jb] + sre[I][jb]; zim[I][jb] = cq[I][jb] - a[I][jb] - bq[I][jb] + sim[I][jb]; }
    };
    const cplx* Kb = mf.KfD + kitem(mf, d.steps, b, t0);                 // slices of one chunk are FR apart
    const int len = t1 - t0;
    Frag A, A1;
    auto zero_src = [&]() {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { sre[I][jb] = 0.0; sim[I][jb] = 0.0; }
    };
    if (FULL) {
        // FULL: the costate sweep itself (k > 4 controls: the gradients are formed by k_mfma_grad from the stored Lambda_t).
        // Terminal costate -(2/m^2) z W (+ S_steps), then E_{cc-1} = P_cc^dagger E_cc + a_cc down to the end of this chunk.
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const int row = 16 * I + lc, col = 4 * jb + lk;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) {
                    v = cscale(cmul(z, d.W[row * d.m + col]), c0);
                    if (need_src) v = cadd(v, source_at(d, b, d.steps, row, col));
                }
                zre[I][jb] = v.x; zim[I][jb] = v.y;
            }
        for (int cc = mf.C - 1; cc > c; --cc) {
            if (need_src) {
                const cplx* ao = mf.Aoff + ((size_t)b * mf.C + cc) * (QQS * 64) + 16 * (lc & 3) + lk;
#pragma unroll
                for (int I = 0; I < 2; ++I)
#pragma unroll
                    for (int jb = 0; jb < MQ; ++jb) { const cplx o = ao[(4 * I + (lc >> 2)) * 64 + 4 * jb]; sre[I][jb] = o.x; sim[I][jb] = o.y; }
            } else {
                zero_src();
            }
            load_frag(mf.PfD + ((size_t)b * mf.C + cc) * QFR, A);
            step(A);
        }
    }
    auto store_lam = [&](int t) {                                        // LamD[b][t][row][16 columns]
        if (!FULL) return;
        cplx* lo = mf.LamD + ((size_t)b * d.steps + t) * (16 * NT * 16);
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) lo[(16 * I + lc) * 16 + 4 * jb + lk] = cmake(zre[I][jb], zim[I][jb]);
    };
    // Undressed forbidden levels and speed_up need only Psi_t at this lane's own entries and one scalar per slice: fetched UNCONDITIONALLY (clamped
    // indices, masked afterwards) together with the next K_t -- source_at()'s conditional loads are waited for on the spot with vmcnt(0), which
    // drained the K prefetch in every slice (as in k_mfma_backward3, source_fast).  Dressed levels keep source_at().
    const bool fast_src = need_src && !d.forbid_dressed;
    struct SrcIn { cplx own[2][MQ]; cplx zt; };
    double wrow[2] = {0.0, 0.0};                                         // sum of 2 a_f over the forbidden levels equal to this lane's rows (loop invariant)
    cplx wown[2][MQ];
    const double speed_coef = (need_src && d.has_speed) ? -d.a_speed * d.su_resid[b] * 2.0 / ((double)d.m * (double)d.m) : 0.0;
    if (fast_src) {
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            for (int f = 0; f < d.n_forb; ++f) wrow[I] += (16 * I + lc == d.forb_state[f]) ? 2.0 * d.forb_a[f] : 0.0;
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) wown[I][jb] = d.W[min(16 * I + lc, d.n - 1) * d.m + min(4 * jb + lk, d.m - 1)];
        }
    }
    auto fetch_src = [&](SrcIn& si, int t) {
        const cplx* psi = d.inter + ((size_t)b * (d.steps + 1) + max(t, 1)) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) si.own[I][jb] = psi[min(16 * I + lc, d.n - 1) * d.m + min(4 * jb + lk, d.m - 1)];
        si.zt = *(d.has_speed ? d.ztau + (size_t)b * (d.steps + 1) + max(t, 1) : d.zfin + b);
    };
    auto apply_src = [&](const SrcIn& si, int t) {
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const cplx phi = si.own[I][jb];
                const double w = wrow[I] * (phi.x * phi.x + phi.y * phi.y);
                cplx sv = cscale(phi, w);
                const cplx zw = cscale(cmul(si.zt, wown[I][jb]), speed_coef);
                sv.x += d.has_speed ? zw.x : 0.0; sv.y += d.has_speed ? zw.y : 0.0;
                const bool ok = t > 0 && 16 * I + lc < d.n && 4 * jb + lk < d.m;
                sre[I][jb] = ok ? sv.x : 0.0; sim[I][jb] = ok ? sv.y : 0.0;
            }
    };
    auto src_or_zero = [&](int t) { if (need_src && t > 0) source(max(t, 1)); else zero_src(); };
    load_frag(Kb + (size_t)(len - 1) * mf.FR, A);
    int i = 0;                                                           // step i handles slice t = t1 - 1 - i
    if (fast_src) {
        SrcIn s0, s1;
        fetch_src(s0, t1 - 1);
        for (; i + 2 <= len; i += 2) {
            fetch_src(s1, t1 - 2 - i); load_frag(Kb + (size_t)(len - 2 - i) * mf.FR, A1); asm volatile("" ::: "memory"); store_lam(t1 - 1 - i); apply_src(s0, t1 - 1 - i); step(A);
            fetch_src(s0, t1 - 3 - i); load_frag(Kb + (size_t)max(len - 3 - i, 0) * mf.FR, A); asm volatile("" ::: "memory"); store_lam(t1 - 2 - i); apply_src(s1, t1 - 2 - i); step(A1);
        }
        if (i < len) { store_lam(t1 - 1 - i); apply_src(s0, t1 - 1 - i); step(A); }
    } else {
    for (; i + 2 <= len; i += 2) {
        src_or_zero(t1 - 1 - i); load_frag(Kb + (size_t)(len - 2 - i) * mf.FR, A1); asm volatile("" ::: "memory"); store_lam(t1 - 1 - i); step(A);
        src_or_zero(t1 - 2 - i); load_frag(Kb + (size_t)max(len - 3 - i, 0) * mf.FR, A); asm volatile("" ::: "memory"); store_lam(t1 - 2 - i); step(A1);
    }
    if (i < len) { src_or_zero(t1 - 1 - i); store_lam(t1 - 1 - i); step(A); }
    }
    if (FULL) return;
    cplx* out = mf.Aoff + ((size_t)b * mf.C + c) * (QQS * 64);             // D-layout 16x16x4 column block 0
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)                                     // all 16 columns: the 16x16x4 backward kernels read the whole block
            out[(4 * I + (lc >> 2)) * 64 + 16 * (lc & 3) + 4 * jb + lk] = jb < MQ ? cmake(zre[I][jb < MQ ? jb : 0], zim[I][jb < MQ ? jb : 0]) : cmake(0.0, 0.0);
}

// ---- kernel B: thin backward sweep  Lambda_{t-1} = K_t^dagger Lambda_t  + control gradients ----------------------
// dL/du_{k,t} = Re sum_ab H_k'[a][b] Q_t[a][b],  Q_t = conj(Lambda_t) Psi_t^T  (rank-m outer product on the MFMA),
// which equals Re <Lambda_t, H_k' Psi_t> of the reference's matexp_op_grad (tensorflow_state.py:61-63).
// 4 waves per workgroup; LDS: fragD image of the k control Hamiltonians (shared) + one transposition pad per wave.
// QA: active 4-row strips of the padded K / chunk products (ceil(n / 4)): the costate's rows beyond are zero, so those strips are neither loaded nor multiplied.
template <int NT, bool H_IN_LDS, bool SPLIT = false, int QA = 4 * NT>
__global__ void __launch_bounds__(256) k_mfma_backward(QocDev d, QocMfma mf, int single_chunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cplx* Hl = (cplx*)smem;                                                     // [k] fragD(H_k')
    cplx* pad = (cplx*)(smem + (H_IN_LDS ? (size_t)d.k * QFR * sizeof(cplx) : 0)) + (size_t)wv * 16 * QLDR;
    if (H_IN_LDS) {
        for (int o = threadIdx.x; o < d.k * QFR; o += blockDim.x) Hl[o] = mf.HfD[QFR + o];
        __syncthreads();
    }
    const cplx* Hsrc = H_IN_LDS ? Hl : (mf.HfD + QFR);
    const int CC = single_chunk ? 1 : mf.C;
    const int item = blockIdx.x * 4 + wv;
    if (item >= d.B * CC) return;
    const int b = item / CC, c = item - b * CC;
    if (d.skip_done && d.done[b]) return;
    const int t0 = single_chunk ? 0 : c * mf.L, t1 = single_chunk ? d.steps : min(t0 + mf.L, d.steps);
    const bool need_src = d.n_forb > 0 || d.has_speed;
    // terminal costate: -(2/m^2) z W (+ S_steps)
    CTile Lam[NT];
    {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                cplx v = cmake(0.0, 0.0);
                if (row < d.n && col < d.m) {
                    v = cscale(cmul(z, d.W[row * d.m + col]), c0);
                    if (need_src) v = cadd(v, source_at(d, b, d.steps, row, col));
                }
                Lam[Ib].re[r] = v.x; Lam[Ib].im[r] = v.y;
            }
    }
    AFragT<NT> A;
    if (!single_chunk) {
        for (int cc = mf.C - 1; cc > c; --cc) {                          // Lambda at the end of this chunk
            afrag_load<NT, true, QA>(mf.PfD + ((size_t)b * mf.C + cc) * QFR, lane, A);
            CTile acc[NT];
            mm_colblock<NT, QA>(A, Lam, acc);
            for (int Ib = 0; Ib < NT; ++Ib) Lam[Ib] = acc[Ib];
            if (need_src) {                                              // E_{cc-1} = P_cc^dagger E_cc + a_cc
                CTile off[NT];
                colblock_load<NT>(mf.Aoff + ((size_t)b * mf.C + cc) * (QQS * 64), 0, lane, off);
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib) { Lam[Ib].re += off[Ib].re; Lam[Ib].im += off[Ib].im; }
            }
        }
    }
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    for (int t = t1 - 1; t >= t0; --t) {
        if constexpr (SPLIT) {
            // costates only: Lambda_t goes to LamD[b][t][row][16 columns] and k_mfma_grad forms the gradients slice-parallel
            cplx* lam_out = mf.LamD + ((size_t)b * d.steps + t) * (16 * NT * 16);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lam_out[(16 * Ib + (lane >> 4) + 4 * r) * 16 + (lane & 15)] = cmake(Lam[Ib].re[r], Lam[Ib].im[r]);
        } else {
        // ---- Q = conj(Lambda_t) Psi_t^T, 3-multiplication form:  Qr = T1 + T2, Qi = T3 - T1 + T2 with
        //      T1 = Lr Pr, T2 = Li Pi, T3 = (Lr - Li)(Pr + Pi) ------------------------------------------------------
        lds_put_colblock<NT>(pad, 0, lane, Lam);                             // wave-private image: pad[j][row]
        double lr[NT][4], li[NT][4], pr[NT][4], pi[NT][4];
        const cplx* psi = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lr[I][q] = 0.0; li[I][q] = 0.0; pr[I][q] = 0.0; pi[I][q] = 0.0;
                if (q < mf.mq) {
                    const int row = 16 * I + (lane & 15), j = 4 * q + (lane >> 4);
                    const cplx lv = pad[j * QLDR + row];
                    lr[I][q] = lv.x; li[I][q] = lv.y;
                    if (row < d.n && j < d.m) {
                        const cplx pv = psi[row * d.m + j];
                        pr[I][q] = pv.x; pi[I][q] = pv.y;
                    }
                }
            }
        double g[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) g[kk] = 0.0;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int Jp = 0; Jp < NT; ++Jp) {
                d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q < mf.mq) {
                        t1v = QMFMA(lr[I][q], pr[Jp][q], t1v);
                        t2v = QMFMA(li[I][q], pi[Jp][q], t2v);
                        t3v = QMFMA(lr[I][q] - li[I][q], pr[Jp][q] + pi[Jp][q], t3v);
                    }
                }
                const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (kk >= d.k) continue;
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = Hsrc[(size_t)kk * QFR + (Jp * QQS + 4 * I + r) * 64 + lane];
                        acc = fma(h.x, qr[r], acc);
                        acc = fma(-h.y, qi[r], acc);
                    }
                    g[kk] += acc;
                }
            }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk >= d.k) continue;
            double v = g[kk];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) d.dLdu[((size_t)b * d.k + kk) * d.steps + t] = v;
        }
        }
        if (t == 0) break;
        // ---- Lambda_{t-1} = K_t^dagger Lambda_t (+ S_{t-1}) ---------------------------------------------------------
        afrag_load<NT, true, QA>(mf.KfD + kitem(mf, d.steps, b, t), lane, A);
        CTile acc[NT];
        mm_colblock<NT, QA>(A, Lam, acc);
        for (int Ib = 0; Ib < NT; ++Ib) Lam[Ib] = acc[Ib];
        if (need_src) {
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * Ib + (lane >> 4) + 4 * r, col = lane & 15;
                    if (row < d.n && col < d.m) {
                        const cplx sv = source_at(d, b, t, row, col);
                        Lam[Ib].re[r] += sv.x; Lam[Ib].im[r] += sv.y;
                    }
                }
        }
    }
}

#define B2_LDP 17          // row stride of the 16-row costate images of k_mfma_backward3

// ---- kernel G: control gradients of all slices in parallel (n > 32) ---------------------------------------------------
// For NT = 3/4 the fragD images of the control Hamiltonians (36 / 64 KB each) no longer fit in LDS next to the transposition
// pads of the sweep, and a sweep that reads them from L2 at every slice is bound by that stream (2.0 of 6.9 ms at n = 48 x 64).
// The sweep (k_mfma_backward<NT, false, true>) therefore only propagates the costates and stores Lambda_t; this kernel, with
// nothing but the images of up to 4 controls in LDS, forms Q = conj(Lambda_t) Psi_t^T and dL/du_{k,t} = Re sum_ab H_k'[a,b] Q[a,b]
// for every (seed, slice) independently: one wave per slice, the next slice's operands fetched while this one multiplies.
template <int NT, int MQ>
__global__ void __launch_bounds__(256) k_mfma_grad(QocDev d, QocMfma mf) {
    constexpr int KG = NT >= 4 ? 2 : 4;                                         // control images per pass: 2 x 64 KB or 4 x 36 KB of LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Hl = (cplx*)smem;                                                     // [<= KG] fragD(H_k')
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lk = lane >> 4, lc = lane & 15;
    const int total = d.B * d.steps, stride = gridDim.x * 4;
    struct Ops { double lr[NT][MQ], li[NT][MQ], pr[NT][MQ], pi[NT][MQ]; };
    auto fetch = [&](Ops& o, int s) {
        s = min(s, total - 1);
        const int b = s / d.steps, t = s - b * d.steps;
        const cplx* lam = mf.LamD + (size_t)s * (16 * NT * 16);
        const cplx* psi = d.inter + ((size_t)b * (d.steps + 1) + t + 1) * d.n * d.m;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                const int j = 4 * q + lk;
                const cplx lv = lam[(16 * I + lc) * 16 + j];
                // rows >= n / columns >= m: clamped, finite, unmasked (they meet zero columns of Lambda / zero padding of H')
                const cplx pv = psi[min(16 * I + lc, d.n - 1) * d.m + min(j, d.m - 1)];
                o.lr[I][q] = lv.x; o.li[I][q] = lv.y; o.pr[I][q] = pv.x; o.pi[I][q] = pv.y;
            }
        asm volatile("" ::: "memory");
    };
    for (int k0 = 0; k0 < d.k; k0 += KG) {                                   // controls in groups of <= KG images
        const int kn = min(KG, d.k - k0);
        __syncthreads();
        for (int o = threadIdx.x; o < kn * QFR; o += blockDim.x) Hl[o] = mf.HfD[(size_t)(1 + k0) * QFR + o];
        __syncthreads();
        auto contract = [&](const Ops& o, int s) {
            if (s >= total) return;
            const int b = s / d.steps, t = s - b * d.steps;
            if (d.skip_done && d.done[b]) return;
            double g[KG];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) g[kk] = 0.0;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int Jp = 0; Jp < NT; ++Jp) {
                    // the H_k' entries of this tile do not depend on the MFMAs: all their LDS reads go out as one batch ahead of them
                    // (left to itself hipcc reads them one at a time with every wait exposed: n = 48 x 64 seeds 307 us per launch,
                    // NT * NT * 4 KG reads of ~100 cycles each per slice)
                    // (NT = 4, two images per pass: batching made it slower, 1.37 -> 1.58 ms; left to the compiler there)
                    constexpr bool BATCH = NT <= 3;
                    cplx hq[KG][4];
                    if constexpr (BATCH) {
#pragma unroll
                        for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                            for (int r = 0; r < 4; ++r) hq[kk][r] = Hl[(size_t)(kk < kn ? kk : 0) * QFR + (Jp * QQS + 4 * I + r) * 64 + lane];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < MQ; ++q) {
                        t1v = QMFMA(o.lr[I][q], o.pr[Jp][q], t1v);
                        t2v = QMFMA(o.li[I][q], o.pi[Jp][q], t2v);
                        t3v = QMFMA(o.lr[I][q] - o.li[I][q], o.pr[Jp][q] + o.pi[Jp][q], t3v);
                    }
                    const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
                    for (int kk = 0; kk < KG; ++kk) {
                        if (kk >= kn) continue;
                        double acc = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            cplx h;
                            if constexpr (BATCH) h = hq[kk][r];
                            else h = Hl[(size_t)kk * QFR + (Jp * QQS + 4 * I + r) * 64 + lane];
                            acc = fma(h.x, qr[r], acc);
                            acc = fma(-h.y, qi[r], acc);
                        }
                        g[kk] += acc;
                    }
                }
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                if (kk >= kn) continue;
                double v = g[kk];
                v += dpp_xor<1>(v); v += dpp_xor<2>(v); v += dpp_xor<4>(v); v += dpp_xor<8>(v);
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (lane == 0) d.dLdu[((size_t)b * d.k + k0 + kk) * d.steps + t] = v;
            }
        };
        Ops o0, o1;
        int s = blockIdx.x * 4 + wv;
        fetch(o0, s);
        for (; s < total; s += 2 * stride) {
            fetch(o1, s + stride); contract(o0, s);
            fetch(o0, s + 2 * stride); contract(o1, s + stride);
        }
    }
}

// ---- kernel G2: the same contraction split by ROW TILES of the control Hamiltonians (n > 32 batches) -------------------------
// k_mfma_grad at NT = 4 passes the 64 KB images through LDS two at a time (every slice fetched and multiplied once per pass) and its
// fully unrolled 16-tile body allocates badly (1742 accvgpr moves, 784 B of scratch per lane: 1.37 ms per launch at n = 64 x 64
// seeds, ~10x its MFMA time).  A workgroup here owns ONE 16-row tile h of every H_k' (k x NT x 4 KB of LDS: all controls of k <= 8
// resident at once, two workgroups per CU at k <= 4) and forms, for its slices, the NT tiles Q[h, Jp] = conj(Lambda_t[h]) Psi_t[Jp]^T
// and the partial sums Re sum_{a in tile h, b} H_k'[a,b] Q[a,b]; blockIdx.x % NT = h, so the NT workgroups of a slice run side by
// side and share its vectors through L2.  The NT partials per (seed, control, slice) go to gpart[h] and are added in fixed order by
// k_mfma_grad_sum (deterministic: no atomics).  A tile's 4 x 4 image reads leave as one batch ahead of its MFMAs.
template <int NT, int MQ>
__global__ void __launch_bounds__(256, 2) k_mfma_grad_rt(QocDev d, QocMfma mf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Hl = (cplx*)smem;                                                     // [K4][NT col blocks][4 strips][64]: rows 16h .. 16h + 15 of fragD(H_k'), zero beyond k
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lk = lane >> 4, lc = lane & 15;
    const int h = blockIdx.x % NT, g = blockIdx.x / NT;
    const int K4 = (d.k + 3) & ~3;
    for (int o = threadIdx.x; o < K4 * NT * 256; o += blockDim.x) {
        const int ln = o & 63, r = (o >> 6) & 3, Jp = (o >> 8) % NT, kk = o / (NT * 256);
        Hl[o] = kk < d.k ? mf.HfD[(size_t)(1 + kk) * QFR + (Jp * QQS + 4 * h + r) * 64 + ln] : cmake(0.0, 0.0);
    }
    __syncthreads();
    const int total = d.B * d.steps, stride = (gridDim.x / NT) * 4;
    struct Ops { double lr[MQ], li[MQ], pr[NT][MQ], pi[NT][MQ]; };
    auto fetch = [&](Ops& o, int s) {
        s = min(s, total - 1);
        const int b = s / d.steps, t = s - b * d.steps;
        const cplx* lam = mf.LamD + (size_t)s * (16 * NT * 16);
        const cplx* psi = d.inter + ((size_t)b * (d.steps + 1) + t + 1) * d.n * d.m;
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const int j = 4 * q + lk;
            const cplx lv = lam[(16 * h + lc) * 16 + j];
            o.lr[q] = lv.x; o.li[q] = lv.y;
#pragma unroll
            for (int Jp = 0; Jp < NT; ++Jp) {
                // rows >= n / columns >= m: clamped, finite, unmasked (they meet zero columns of Lambda / zero padding of H')
                const cplx pv = psi[min(16 * Jp + lc, d.n - 1) * d.m + min(j, d.m - 1)];
                o.pr[Jp][q] = pv.x; o.pi[Jp][q] = pv.y;
            }
        }
        asm volatile("" ::: "memory");
    };
    double* part = mf.gpart + (size_t)h * d.B * d.k * d.steps;
    // One slice: (A) the NT tiles Q[h, Jp] from the operands in registers, (B) the operands of the wave's NEXT slice requested into the
    // same registers, (C) the contraction with the images, tile by tile, while those loads are in flight.
    Ops o;
    int s = g * 4 + wv;
    fetch(o, s);
    for (; s < total; s += stride) {
        const int b = s / d.steps, t = s - b * d.steps;
        d4 qr[NT], qi[NT];
#pragma unroll
        for (int Jp = 0; Jp < NT; ++Jp) {
            d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                t1v = QMFMA(o.lr[q], o.pr[Jp][q], t1v);
                t2v = QMFMA(o.li[q], o.pi[Jp][q], t2v);
                t3v = QMFMA(o.lr[q] - o.li[q], o.pr[Jp][q] + o.pi[Jp][q], t3v);
            }
            qr[Jp] = t1v + t2v; qi[Jp] = t3v - t1v + t2v;                          // Re, Im of conj(lambda) psi^T, tile (h, Jp)
            __builtin_amdgcn_sched_barrier(0);                                    // (tile by tile: 3 accumulators alive, not 3 NT)
        }
        fetch(o, s + stride);
        if (d.skip_done && d.done[b]) continue;
        double gk[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) gk[kk] = 0.0;
#pragma unroll
        for (int Jp = 0; Jp < NT; ++Jp)
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {                                      // two controls at a time: 8 image reads in one batch
                if (2 * kp >= K4) continue;                                      // (uniform)
                cplx hq[2][4];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) hq[kk][r] = Hl[(((2 * kp + kk) * NT + Jp) * 4 + r) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        gk[2 * kp + kk] = fma(hq[kk][r].x, qr[Jp][r], gk[2 * kp + kk]);
                        gk[2 * kp + kk] = fma(-hq[kk][r].y, qi[Jp][r], gk[2 * kp + kk]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk >= d.k) continue;
            double v = gk[kk];
            v += dpp_xor<1>(v); v += dpp_xor<2>(v); v += dpp_xor<4>(v); v += dpp_xor<8>(v);
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane == 0) part[((size_t)b * d.k + kk) * d.steps + t] = v;
        }
    }
}
// dL/du = the NT row-tile partials of k_mfma_grad_rt, added in fixed order
__global__ void __launch_bounds__(256) k_mfma_grad_sum(QocDev d, QocMfma mf, int NT) {
    const size_t per = (size_t)d.k * d.steps, tot = (size_t)d.B * per;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < tot; o += (size_t)gridDim.x * blockDim.x) {
        if (d.skip_done && d.done[o / per]) continue;
        double v = mf.gpart[o];
        for (int h = 1; h < NT; ++h) v += mf.gpart[(size_t)h * tot + o];
        d.dLdu[o] = v;
    }
}

// ---- kernel B3: the row-split backward sweep of NT = 2 with every per-slice latency taken off the dependent chain ---------
// A pair of waves per (seed, chunk) item, one 16-row tile of the costate each, exchanged through double-buffered LDS images
// (8 waves = 4 items per workgroup share one LDS image of the control Hamiltonians).  Against its predecessor (k_mfma_backward2:
// same split on 16x16x4, branchy loop, __syncthreads; removed):
//  * the slice loop is branch-free (finished / out-of-range steps run on clamped addresses and only their store is
//    masked), so hipcc keeps counted vmcnt waits, and the K_t^dagger fragment and Psi_t of the NEXT slice are fetched at the
//    top of each step into a second register set (2x unrolled rotation): the predecessor exposed two HBM round trips per slice
//    (Psi before the Q tiles, K before the costate product: ~5 of its 7.5 us per slice);
//  * the chunk-boundary recursion prefetches P_{cc-1} the same way and computes every step unconditionally (select);
//  * the workgroup barrier orders LDS only (lds_barrier), so the prefetch stays in flight across it;
//  * control gradients: 16-lane DPP butterflies, the 4 row partials of both waves go through LDS and lane kk of wave h = 0
//    adds the 8 partials of control kk (was: six ds_bpermute levels per control).
// Used for k <= 4 controls without state regularisers (no per-slice source term); anything else keeps backward2.
// MODE (latency mode with a state regulariser, QocMfma::lat_sources -- the chunks are 8 slices short there, so the chunk-boundary
// recursion E_{cc-1} = P_cc^dagger E_cc + a_cc would be ~60 dependent products): 1 = two-level boundaries -- whole groups of G chunks
// first, E <- G_g^dagger E + A_g with the group products GfD of k_mfma_chain_rows and the group offsets Goff, then the chunks of the
// own group; 2 = the pass that forms those group offsets: an item is a (seed, group), the recursion runs over the chunks of the
// group from a zero costate and its result goes to Goff; no slices.  MODE 0 (batch kernels) has neither branch in its code.
// MODE 3 (batch kernels without a state regulariser, round 3): no boundary recursion at all -- the z-free costate at the end of the chunk
// comes from k_mfma_bnd_scan (BndA) and is scaled by -(2/m^2) z here.  In MODE 0 every item ran C - 1 boundary steps on the chunk products
// that all chunks of its seed read at the same time: 6.8 us per step against 2.0 us per slice, 100 of the kernel's 165 us at 16 chunks.
// QA: active 4-row strips of the padded K / chunk products (ceil(n / 4)); the zero strips beyond are neither loaded nor multiplied.
template <int MQ, bool SRC, int KC = 4, int MODE = 0, int QA = 8>
__global__ void __launch_bounds__(512) k_mfma_backward3(QocDev d, QocMfma mf) {
    constexpr int NT = 2;                                                       // KC = control images in LDS: 4, or 5 (k = 5 still fits the 160 KB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wv & 1, pair = wv >> 1;
    cplx* Hl = (cplx*)smem;                                                     // [KC] fragD(H_k'), zero beyond k
    cplx* pads = Hl + (size_t)KC * QFR;                                         // [8 waves][2 buffers][16 * B2_LDP]
    double* gpart = (double*)(pads + 8 * 2 * 16 * B2_LDP);                     // [4 pairs][2 buffers][2 waves][4 rows][KC]
    if (MODE != 2) for (int o = threadIdx.x; o < KC * QFR; o += blockDim.x) Hl[o] = o < d.k * QFR ? mf.HfD[QFR + o] : cmake(0.0, 0.0);
    // costate images of the pair: image[buffer][column j][row % 16] (row stride B2_LDP), rows 0..15 in pad_lo, 16..31 in pad_hi
    cplx* mypad = pads + (size_t)wv * 2 * 16 * B2_LDP;
    const cplx* pad_lo = pads + (size_t)(2 * pair) * 2 * 16 * B2_LDP;
    const cplx* pad_hi = pads + (size_t)(2 * pair + 1) * 2 * 16 * B2_LDP;
    const int item = blockIdx.x * (blockDim.x >> 7) + pair;           // 4 pairs per workgroup in the batch kernels
    const int n_items = d.B * (MODE == 2 ? mf.NG : mf.C);                // MODE 2: c is a GROUP index
    const bool item_ok = item < n_items;
    const int c = item_ok ? item / d.B : 0, b = item_ok ? item - c * d.B : 0;   // chunk-major, as in k_mfma_forward2
    const bool active = item_ok && !(d.skip_done && d.done[b]);
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    // own part of the costate as the D operand of v_mfma_f64_4x4x4 on the TRANSPOSED recursion
    //   Lambda_{t-1}^T = Lambda_t^T conj(K_t):  register jb, lane 16 i + 4 blk + j  <->  Lambda[row 16h + 4 blk + j][column 4 jb + i],
    // so that the right operand (4 k-rows x 16 columns of conj(K)) is a fragD register exactly as the 16x16x4 kernels store it,
    // the left operand is a 4x4 block of Lambda^T read from the LDS image (broadcast over blk), and no output column is padding
    // (a 16x16x4 tile spends half of its columns on m = 8): 24 MQ MFMAs of 17 cycles instead of 24 of ~100.
    double ore[MQ], oim[MQ];
    {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) {
            const int row = 16 * h + lc, col = 4 * jb + lk;
            cplx v = cmake(0.0, 0.0);
            if constexpr (MODE == 3) {
                static_assert(MODE != 3 || !SRC, "precomputed boundaries are z-free: no state regulariser");
                v = cscale(cmul(z, mf.BndA[(((size_t)b * mf.C + c) * NT * MQ + h * MQ + jb) * 64 + lane]), c0);
            } else if (MODE != 2 && row < d.n && col < d.m) {
                v = cscale(cmul(z, d.W[row * d.m + col]), c0);
                if (SRC) v = cadd(v, source_at(d, b, d.steps, row, col));
            }
            ore[jb] = v.x; oim[jb] = v.y;
        }
    }
    struct Frag { cplx f[8]; };
    struct PsiReg { double pr[2][MQ], pi[2][MQ]; cplx own[MQ]; cplx zt; };   // own / zt: inputs of the source term (SRC only)
    auto put_own = [&](int bf) {
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) mypad[(bf * 16 + 4 * jb + lk) * B2_LDP + lc] = cmake(ore[jb], oim[jb]);
    };
    auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
        for (int q = 0; q < QA; ++q) fr.f[q] = F[(h * QQS + q) * 64 + lane];
    };
    // (ore, oim) <- rows of tile h of M^dagger Lambda, M given by its fragD fragment, Lambda by the image `bf` of the pair
    auto dagger_product = [&](const Frag& fr, int bf, double (&nre)[MQ], double (&nim)[MQ]) {
        double a[MQ], bq[MQ], cq[MQ];
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) { a[jb] = 0.0; bq[jb] = 0.0; cq[jb] = 0.0; }
#pragma unroll
        for (int kb = 0; kb < QA; ++kb) {
            const cplx* src = (kb < 4 ? pad_lo : pad_hi) + (size_t)bf * 16 * B2_LDP + 4 * (kb & 3) + lk;
            const double br = fr.f[kb].x, bi = -fr.f[kb].y, bs = br + bi;
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) {
                const cplx v = src[(4 * jb + li4) * B2_LDP];                  // Lambda[4 kb + lk][4 jb + li4]
                a[jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x, br, a[jb], 0, 0, 0);
                bq[jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.y, bi, bq[jb], 0, 0, 0);
                cq[jb] = __builtin_amdgcn_mfma_f64_4x4x4f64(v.x + v.y, bs, cq[jb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) { nre[jb] = a[jb] - bq[jb]; nim[jb] = cq[jb] - a[jb] - bq[jb]; }
    };
    put_own(0);
    lds_barrier();
    int buf = 0;
    // ---- costate at the end of this chunk: E_{cc-1} = P_cc^dagger E_cc, uniform trip count, result kept only while cc > c ----
    {
        const cplx* offs = mf.Aoff + (size_t)b * mf.C * (QQS * 64);   // offsets of the steps in flight: per chunk, or per group (Goff)
        auto bstep = [&](const Frag& fr, bool keep, int cc) {
            double nre[MQ], nim[MQ];
            cplx off[MQ];
            if (SRC) {                                               // E_{cc-1} = P_cc^dagger E_cc + a_cc; a_cc is a D-layout 16x16x4 column block
                const cplx* ao = offs + (size_t)cc * (QQS * 64) + (4 * h + (lc >> 2)) * 64 + 16 * (lc & 3) + lk;
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) off[jb] = ao[4 * jb];
            }
            dagger_product(fr, buf, nre, nim);
            if (SRC) {
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) { nre[jb] += off[jb].x; nim[jb] += off[jb].y; }
            }
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { ore[jb] = keep ? nre[jb] : ore[jb]; oim[jb] = keep ? nim[jb] : oim[jb]; }
            put_own(buf ^ 1);
            lds_barrier();
            buf ^= 1;
        };
        // n_steps matrices base[idx_of(s)], s = 0 .. n_steps - 1, the next one fetched while this one multiplies; the trip count is
        // the same for every wave of the workgroup (bstep holds a barrier)
        auto bsteps = [&](const cplx* base, int n_steps, auto idx_of, auto keep_of) {
            if (n_steps <= 0) return;
            Frag f0, f1;
            load_frag(base + (size_t)idx_of(0) * QFR, f0);
            int s = 0;
            for (; s + 2 <= n_steps; s += 2) {
                load_frag(base + (size_t)idx_of(s + 1) * QFR, f1); asm volatile("" ::: "memory"); bstep(f0, keep_of(s), idx_of(s));
                load_frag(base + (size_t)idx_of(min(s + 2, n_steps - 1)) * QFR, f0); asm volatile("" ::: "memory"); bstep(f1, keep_of(s + 1), idx_of(s + 1));
            }
            if (s < n_steps) bstep(f0, keep_of(s), idx_of(s));
        };
        const int C = mf.C;
        if constexpr (MODE == 0) {
            bsteps(mf.PfD + (size_t)b * C * QFR, C - 1, [&](int s) { return C - 1 - s; }, [&](int s) { return C - 1 - s > c; });
        } else if constexpr (MODE == 3) {
            (void)C;
        } else if constexpr (MODE == 1) {
            const int G = mf.G, NG = mf.NG, g = c / G;
            offs = mf.Goff + (size_t)b * NG * (QQS * 64);
            bsteps(mf.GfD + (size_t)b * NG * QFR, NG - 1, [&](int s) { return NG - 1 - s; }, [&](int s) { return NG - 1 - s > g; });
            offs = mf.Aoff + (size_t)b * C * (QQS * 64);
            bsteps(mf.PfD + (size_t)b * C * QFR, G - 1, [&](int s) { return min(g * G + G - 1 - s, C - 1); }, [&](int s) { const int cc = g * G + G - 1 - s; return cc < C && cc > c; });
        } else {
            // group offset A_g: from a zero costate over ALL chunks of group c (= g), last to first
            const int G = mf.G, g = c;
            bsteps(mf.PfD + (size_t)b * C * QFR, G, [&](int s) { return min(g * G + G - 1 - s, C - 1); }, [&](int s) { return g * G + G - 1 - s < C; });
            if (active) {
                cplx* go = mf.Goff + ((size_t)b * mf.NG + g) * (QQS * 64) + (4 * h + (lc >> 2)) * 64 + 16 * (lc & 3) + lk;
#pragma unroll
                for (int jb = 0; jb < MQ; ++jb) go[4 * jb] = cmake(ore[jb], oim[jb]);
            }
            return;
        }
    }
    // ---- slices of the chunk, last to first ------------------------------------------------------------------------------
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    const int prow0 = min(lc, d.n - 1), prow1 = min(16 + lc, d.n - 1);
    auto fetch = [&](Frag& fr, PsiReg& ps, int i) {                          // operands of step i (slice t = t1 - 1 - i), clamped
        const int t = max(t1 - 1 - i, 0);
        load_frag(mf.KfD + kitem(mf, d.steps, b, t), fr);
        const cplx* psi = iv + (size_t)(t + 1) * d.n * d.m;
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const int j = 4 * q + lk, jc = min(j, d.m - 1);
            // out-of-range (row >= n, column >= m) entries read a clamped, finite element and need no mask: they only meet
            // the zero columns of Lambda (j >= m) or the zero padding of H_k' (row >= n); a masked load would be made
            // conditional by hipcc and waited for on the spot, draining the K prefetch with it
            const cplx p0 = psi[prow0 * d.m + jc], p1 = psi[prow1 * d.m + jc];
            ps.pr[0][q] = p0.x; ps.pi[0][q] = p0.y;
            ps.pr[1][q] = p1.x; ps.pi[1][q] = p1.y;
            if (SRC) ps.own[q] = (psi - (size_t)d.n * d.m)[(h ? prow1 : prow0) * d.m + jc];    // Psi_t at this lane's costate entries
        }
        if (SRC) ps.zt = *(d.has_speed ? d.ztau + (size_t)b * (d.steps + 1) + t : d.zfin + b);
        asm volatile("" ::: "memory");
    };
    // Source S_t of the state regularisers at this lane's costate entries (row 16h + lc, column 4 jb + lk).  Undressed forbidden
    // levels and speed_up need only Psi_t at those same entries and one scalar per slice, which fetch() brings in with the other
    // operands (unconditional loads, no wait on the spot).  A dressed forbidden level needs a whole column of Psi_t and falls
    // back to source_at(): its loads are conditional (hipcc waits for them on the spot, draining vmcnt), so that call sits BEFORE
    // the next operands are fetched -- what is in flight then are this step's operands, which are needed now anyway.
    const bool fast_src = !d.forbid_dressed;
    cplx wown[MQ];
    const double speed_coef = (SRC && d.has_speed) ? -d.a_speed * d.su_resid[b] * 2.0 / ((double)d.m * (double)d.m) : 0.0;
    if (SRC) {
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) wown[jb] = d.W[min(16 * h + lc, d.n - 1) * d.m + min(4 * jb + lk, d.m - 1)];
    }
    double wrow = 0.0;                      // this lane's row: sum of 2 a_f over the forbidden levels equal to it (loop invariant)
    if (SRC) for (int f = 0; f < d.n_forb; ++f) wrow += (16 * h + lc == d.forb_state[f]) ? 2.0 * d.forb_a[f] : 0.0;
    auto source_fast = [&](const PsiReg& ps, int i, double (&fre)[MQ], double (&fim)[MQ]) {
        const int t = t1 - 1 - i, row = 16 * h + lc;
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) {
            const cplx phi = ps.own[jb];
            const double pop = phi.x * phi.x + phi.y * phi.y;
            const double w = wrow * pop;
            cplx sv = cscale(phi, w);
            const cplx zw = cscale(cmul(ps.zt, wown[jb]), speed_coef);
            sv.x += d.has_speed ? zw.x : 0.0; sv.y += d.has_speed ? zw.y : 0.0;
            const bool ok = t > 0 && row < d.n && 4 * jb + lk < d.m;
            fre[jb] = ok ? sv.x : 0.0; fim[jb] = ok ? sv.y : 0.0;
        }
    };
    double sre[MQ], sim[MQ];
    auto source = [&](int i) {
        if (!SRC || fast_src) return;
        const int t = t1 - 1 - i, tc = max(t, 1);
#pragma unroll
        for (int jb = 0; jb < MQ; ++jb) {
            const int row = 16 * h + lc, col = 4 * jb + lk;
            cplx sv = cmake(0.0, 0.0);
            if (t > 0 && row < d.n && col < d.m) sv = source_at(d, b, tc, row, col);
            sre[jb] = sv.x; sim[jb] = sv.y;
        }
    };
    auto step = [&](const Frag& fr, const PsiReg& ps, int i) {
        const int t = t1 - 1 - i;
        const bool live = active && t >= t0;
        // ---- Q tiles (h, 0..1) = conj(Lambda_t)[rows of tile h] Psi_t^T and the contraction with H_k' ---------------------
        double lr[MQ], li[MQ];
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const cplx lv = mypad[(buf * 16 + 4 * q + lk) * B2_LDP + lc];       // Lambda[16h + lc][4q + lk]
            lr[q] = lv.x; li[q] = lv.y;
        }
        double g[KC];
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) g[kk] = 0.0;
#pragma unroll
        for (int Jp = 0; Jp < 2; ++Jp) {
            d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                t1v = QMFMA(lr[q], ps.pr[Jp][q], t1v);
                t2v = QMFMA(li[q], ps.pi[Jp][q], t2v);
                t3v = QMFMA(lr[q] - li[q], ps.pr[Jp][q] + ps.pi[Jp][q], t3v);
            }
            const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx hv = Hl[(size_t)kk * QFR + (Jp * QQS + 4 * h + r) * 64 + lane];
                    acc = fma(hv.x, qr[r], acc);
                    acc = fma(-hv.y, qi[r], acc);
                }
                g[kk] += acc;
            }
        }
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {                                       // sum over the 16 lanes of a DPP row
            g[kk] += dpp_xor<1>(g[kk]); g[kk] += dpp_xor<2>(g[kk]); g[kk] += dpp_xor<4>(g[kk]); g[kk] += dpp_xor<8>(g[kk]);
        }
        if (lc == 0) {
            double* gp = gpart + ((((size_t)pair * 2 + buf) * 2 + h) * 4 + lk) * KC;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) gp[kk] = g[kk];
        }
        // ---- Lambda_{t-1} = K_t^dagger Lambda_t ------------------------------------------------------------------------------
        dagger_product(fr, buf, ore, oim);
        if (SRC) {
            if (fast_src) source_fast(ps, i, sre, sim);
#pragma unroll
            for (int jb = 0; jb < MQ; ++jb) { ore[jb] += sre[jb]; oim[jb] += sim[jb]; }
        }
        put_own(buf ^ 1);
        lds_barrier();
        if (live && h == 0 && lane < d.k) {
            const double* gp = gpart + ((size_t)pair * 2 + buf) * 2 * 4 * KC + lane;
            double sum = 0.0;
#pragma unroll
            for (int x = 0; x < 8; ++x) sum += gp[x * KC];
            d.dLdu[((size_t)b * d.k + lane) * d.steps + t] = sum;
        }
        buf ^= 1;
    };
    Frag k0, k1;
    PsiReg p0, p1;
    fetch(k0, p0, 0);
    for (int i = 0; i < mf.L; i += 2) {
        source(i);     fetch(k1, p1, i + 1); step(k0, p0, i);
        source(i + 1); fetch(k0, p0, i + 2); step(k1, p1, i + 1);
    }
}

