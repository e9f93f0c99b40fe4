// qoc_mfma_latency.h -- the sweeps of the latency mode of the MFMA path (qoc_config.variant = 5; AUTO for a handful of control sets
// of a 16 < n <= 32 unitary problem -- the reference's own use is ONE control set per Grape() call, main_grape/grape.py:106-109).
//
// With one trajectory the batch kernels are a chain of dependent sweeps: forward over the slices (tensorflow_state.py:204-227),
// then the costate back over them (the custom gradient of tensorflow_state.py:77-133).  Without a state regulariser the costate
//     Lambda_t = c0 z Lambda0_t,   Lambda0_t = (K_N ... K_{t+1})^dagger W,   c0 = -2/m^2,  z = tr(W^dagger Psi_N)
// is LINEAR in the overlap z, so the z-free costate Lambda0 does not wait for the forward sweep: both sweeps run side by side in
// ONE launch (k_mfma_sweep_lat: 2 x chunks x column groups workgroups of NT waves), each storing its vectors in its own register layout,
// and the gradient
//     dL/du_{k,t} = c0 Re( conj(z) G_{k,t} ),   G_{k,t} = tr( Lambda0_t^dagger H_k' Psi_t )
// is one slice-parallel kernel (k_mfma_grad_lat: two waves per slice) that also forms z and the loss.  Against the pair of waves
// per chunk that walked costate and gradient together (k_mfma_backward3, 35 us for one C2 trajectory) the dependent chain is
// shorter by one whole sweep.
#pragma once
#include "qoc_mfma_frag.h"
#include "qoc_kernels_finish.h"

// One workgroup per (role, seed, chunk, group of 4 columns), wave I of it = rows 16 I .. 16 I + 15 of the vectors.  role 0: Psi_t = K_t
// Psi_{t-1} from Psi0, stores PsiL[t] = Psi after slice t; role 1: Lambda0_{t-1} = K_t^dagger Lambda0_t from W, stores LamL[t] = Lambda0
// BEFORE K_t^dagger is applied (the costate that meets Psi_t in the gradient).  Both walk two-level chunk boundaries first (products of
// groups of G chunks, then chunk products), the forward one upwards from the start of the pulse, the adjoint one downwards from its
// end.  The roles differ in pointers, index directions and one sign only -- no branch surrounds a load (hipcc would wait for such a
// load on the spot):
//   forward operand  M = K:         strip (I, kb) of lane (lk, lc) = K[16 I + lc][4 kb + lk]       = fragD(K^T)(I, kb)   (KfT / PfT / GfT)
//   adjoint operand  M = K^dagger:  conj(K[4 kb + lk][16 I + lc])                                  = conj fragD(K)(I, kb) (KfD / PfD / GfD)
// X <- M X needs, for row tile I of the result, only the strips (I, kb) of the operand -- 1/NT of the matrix, lane-contiguous 1 KB
// loads, the fetch that bounds a step -- and 3 QQS MFMAs; the left blocks need X over ALL rows, so the waves publish their 16 rows to
// a double-buffered LDS image and meet at one barrier per step.  (With one wave carrying all row tiles, 16 NT^2 KB and 3 NT QQS MFMAs
// per step, the launch was 18.5 instead of 13 us at C2, and at NT = 4 a 256-register matrix left no room for a prefetch.)
// chunk_offsets (undressed forbidden levels only -- their sources are elementwise in this layout and need nothing but Psi): the forward
// role goes back DOWN its chunk from a zero costate with the sources of the states it has just formed and leaves the chunk offset a_c (AoffL):
// role 0 of k_mfma_sweep_src, one launch less (one C2 trajectory with dwdt + two forbidden levels: 0.126 -> 0.121 ms per iteration).
template <int NT>
__global__ void __launch_bounds__(64 * NT) k_mfma_sweep_lat(QocDev d, QocMfma mf, int chunk_offsets) {
    constexpr int LDP = 16 * NT + 1;
    __shared__ __attribute__((aligned(16))) cplx img[2][4 * LDP];                 // image[buffer][column j][row] of the workgroup's 4 columns
    const int lane = threadIdx.x & 63;
    const int I = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cs = mf.mq;
    const int n_sweep = d.B * mf.C * cs;
    const int adj = __builtin_amdgcn_readfirstlane((int)blockIdx.x >= n_sweep ? 1 : 0);
    const int w = (int)blockIdx.x - adj * n_sweep;
    const int item = w / cs, jq0 = w - item * cs;
    const int c = item / d.B, b = item - c * d.B;
    if (d.skip_done && d.done[b]) return;                                         // whole workgroup: no barrier yet
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps), len = t1 - t0;
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    double sg = adj ? -1.0 : 1.0;
    double pre, pim;
    {
        const cplx* X0 = adj ? d.W : d.Psi0;
        const int row = 16 * I + lc, col = 4 * jq0 + lk;
        cplx v = cmake(0.0, 0.0);
        if (row < d.n && col < d.m) v = X0[row * d.m + col];
        pre = v.x; pim = v.y;
    }
    cplx* iv = d.inter + (size_t)b * (d.steps + 1) * d.n * d.m;
    if (!adj && c == 0 && jq0 == 0 && I == 0) {                                   // inter[0] = V  (tensorflow_state.py:232-233)
        for (int o = lane; o < d.n * d.m; o += 64) iv[o] = d.V[o];
    }
    struct Frag { cplx f[QQS]; };                                                 // strips (I, kb) of the operand
    auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
        for (int q = 0; q < QQS; ++q) fr.f[q] = F[(I * QQS + q) * 64 + lane];
    };
    int buf = 0;
    auto product = [&](const Frag& fr) {
        img[buf][lk * LDP + 16 * I + lc] = cmake(pre, pim);
        lds_barrier();
        double a = 0.0, bq = 0.0, cq = 0.0;
        cplx v[QQS];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) v[kb] = img[buf][li4 * LDP + 4 * kb + lk];   // X[4 kb + lk][column li4 of the group]
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) {
            const double br = fr.f[kb].x, bi = fr.f[kb].y, bs = fma(sg, bi, br);
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(v[kb].x, br, a, 0, 0, 0);
            bq = __builtin_amdgcn_mfma_f64_4x4x4f64(v[kb].y, bi, bq, 0, 0, 0);
            cq = __builtin_amdgcn_mfma_f64_4x4x4f64(v[kb].x + v[kb].y, bs, cq, 0, 0, 0);
        }
        pre = fma(-sg, bq, a); pim = fma(-sg, bq, cq - a);
        buf ^= 1;                                                                 // the other image: everyone has passed this step's barrier
    };
    {
        const int G = mf.G, g = c / G, C = mf.C, NG = mf.NG;
        const int cend = min(g * G + G, C) - 1;
        const int n_grp = adj ? NG - 1 - g : g, n_ch = adj ? cend - c : c - g * G, n_bnd = n_grp + n_ch;
        const int g_first = adj ? NG - 1 : 0, c_first = adj ? cend : g * G, dir = adj ? -1 : 1;
        const cplx* Gb = (adj ? mf.GfD : mf.GfT) + (size_t)b * NG * QFR;
        const cplx* Pb = (adj ? mf.PfD : mf.PfT) + (size_t)b * C * QFR;
        auto bnd_ptr = [&](int i) -> const cplx* {
            i = min(i, n_bnd - 1);
            return i < n_grp ? Gb + (size_t)(g_first + dir * i) * QFR : Pb + (size_t)(c_first + dir * (i - n_grp)) * QFR;
        };
        if (n_bnd > 0) {
            constexpr int PD = 2;
            Frag Bq[PD + 1];
#pragma unroll
            for (int q = 0; q < PD; ++q) load_frag(bnd_ptr(q), Bq[q]);
            int i = 0;
            for (; i + PD + 1 <= n_bnd; i += PD + 1) {
#pragma unroll
                for (int q = 0; q <= PD; ++q) {
                    load_frag(bnd_ptr(i + q + PD), Bq[(q + PD) % (PD + 1)]); asm volatile("" ::: "memory"); product(Bq[q]);
                }
            }
#pragma unroll
            for (int q = 0; q <= PD; ++q)
                if (i + q < n_bnd) product(Bq[q]);
        }
    }
    const int MQs = mf.mq <= 2 ? 2 : 4;
    cplx* XL = (adj ? mf.LamL : mf.PsiL) + (size_t)b * d.steps * (NT * MQs) * 64;
    auto store = [&](int t) { XL[((size_t)t * (NT * MQs) + I * MQs + jq0) * 64 + lane] = cmake(pre, pim); };
    auto step = [&](const Frag& fr, int i) {
        const int t = adj ? t1 - 1 - i : t0 + i;
        if (adj) store(t);
        if (!adj || i + 1 < len) product(fr);
        if (!adj) {
            store(t);
            if (t + 1 == d.steps) {
                const int row = 16 * I + lc, col = 4 * jq0 + lk;
                if (row < d.n && col < d.m) (iv + (size_t)(t + 1) * d.n * d.m)[row * d.m + col] = cmake(pre, pim);
            }
        }
    };
    const cplx* Kb = (adj ? mf.KfD : mf.KfT) + kitem(mf, d.steps, b, t0);
    auto k_ptr = [&](int i) -> const cplx* {
        i = min(i, len - 1);
        return Kb + (size_t)(adj ? len - 1 - i : i) * mf.FR;
    };
    const double psr = pre, psi_ = pim;                                           // the chunk-start vector: the state the source of slice t0 is formed from
    Frag K0, K1;
    load_frag(k_ptr(0), K0);
    int i = 0;
    for (; i + 2 <= len; i += 2) {
        load_frag(k_ptr(i + 1), K1); asm volatile("" ::: "memory"); step(K0, i);
        load_frag(k_ptr(i + 2), K0); asm volatile("" ::: "memory"); step(K1, i + 1);
    }
    if (i < len) step(K0, i);
    if (adj || !chunk_offsets) return;
    // ---- chunk offset a_c = sum over the chunk of (K_{t0}^dagger ... K_{t-1}^dagger) S_t: X <- K_t^dagger X + S_t from X = 0, t = t1 - 1 .. t0, with
    //      S_t[row][col] = (sum of 2 a_f over the forbidden levels f == row) |psi|^2 psi, psi = the state BEFORE slice t (regularization_functions.py:71-95,
    //      qoc_state_source.h) -- this lane's own entry of the vectors the sweep above has just stored (the chunk-start vector for t0)
    {
        const int row = 16 * I + lc, col = 4 * jq0 + lk;
        const bool inside = row < d.n && col < d.m;
        double wrow = 0.0;
        for (int f = 0; f < d.n_forb; ++f) wrow += (row == d.forb_state[f]) ? 2.0 * d.forb_a[f] : 0.0;
        const cplx* mine = XL + (size_t)(I * MQs + jq0) * 64 + lane;             // + t (NT MQs) 64: this lane's entry of Psi after slice t
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the stores of the sweep above (this very lane's) have reached memory
        auto source = [&](int t) -> cplx {                                        // tau = t: the state before slice t; 0 at tau = 0
            const cplx p = t > t0 ? mine[(size_t)(t - 1) * (NT * MQs) * 64] : cmake(psr, psi_);
            const double pop = p.x * p.x + p.y * p.y;
            return (t > 0 && inside) ? cscale(p, wrow * pop) : cmake(0.0, 0.0);
        };
        const cplx* Ka = mf.KfD + kitem(mf, d.steps, b, t0);
        sg = -1.0;
        pre = 0.0; pim = 0.0;
        load_frag(Ka + (size_t)(len - 1) * mf.FR, K0);
        for (int j = 0; j < len; ++j) {
            const int t = t1 - 1 - j;
            const cplx add = source(t);
            if (j + 1 < len) load_frag(Ka + (size_t)(len - 2 - j) * mf.FR, K1);
            asm volatile("" ::: "memory");
            product(K0);
            pre += add.x; pim += add.y;
            if (j + 1 < len) K0 = K1;
        }
        mf.AoffL[((size_t)b * mf.C + c) * (NT * MQs * 64) + (size_t)(I * MQs + jq0) * 64 + lane] = cmake(pre, pim);
    }
}

// Fidelity and state-regulariser values of the latency mode with sources, straight from the register-layout vectors (k_loss reads the API
// layout and walks all time points in ONE workgroup per seed: 20 us + 5 us of unpacking for a C2 trajectory).  One wave per time point
// tau = 0 .. steps (tau = 0: the initial vectors V; tau >= 1: PsiL[tau - 1]), 16 per workgroup: the overlap z_tau = sum_j <w_j, psi_j(tau)>
// (tensorflow_state.py:282-333; speed_up uses every tau, regularization_functions.py:88-95) and the forbidden-level term
// sum_f a_f/2 |psi_f|^4 (:71-85, undressed levels).  The last workgroup of a seed to finish (arrival counter) adds the per-tau
// partials in a fixed order and publishes z, the loss, reg_state, su_resid.
// DRESS (dressed forbidden levels, at most 4): the amplitudes phi_f[col] = <dressed level f | psi_col(tau)> are formed here -- a lane adds its
// rows, a 16-lane butterfly the rest -- and 2 a_f |phi|^2 phi goes to QocDev::Fd for the source sweeps (k_dress_amplitudes's job on the other paths).
template <int NT, bool DRESS = false>
__global__ void __launch_bounds__(1024) k_mfma_loss_lat(QocDev d, QocMfma mf) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npts = d.steps + 1, nblk = (npts + 15) / 16;
    const int b = blockIdx.x / nblk, cb = blockIdx.x - b * nblk;
    if (d.skip_done && d.done[b]) return;
    const int tau = cb * 16 + wv;
    const int lk = lane >> 4, lc = lane & 15;
    const int MQs = mf.mq <= 2 ? 2 : 4, per_vec = NT * MQs * 64;
    const double mm = (double)d.m * (double)d.m;
    double* part = mf.loss_part + (size_t)b * npts * 2;
    if (tau < npts) {
        double zr = 0.0, zi = 0.0, fb = 0.0;
        const cplx* pl = mf.PsiL + ((size_t)b * d.steps + max(tau - 1, 0)) * per_vec + lane;
        cplx phi[4][4];                                                           // DRESS: [level f][column quad jq], this lane's rows
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) phi[f][jq] = cmake(0.0, 0.0);
#pragma unroll
        for (int I = 0; I < NT; ++I) {
            const int row = 16 * I + lc;
            cplx vs[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) vs[f] = (DRESS && f < d.n_forb && row < d.n) ? d.Vs[row * d.n + d.forb_state[f]] : cmake(0.0, 0.0);
#pragma unroll
            for (int jq = 0; jq < 4; ++jq) {
                if (jq >= MQs) continue;
                const int g = I * MQs + jq, col = 4 * jq + lk;
                const bool inside = row < d.n && col < d.m;
                cplx psi = pl[g * 64];
                if (tau == 0) psi = inside ? d.V[row * d.m + col] : cmake(0.0, 0.0);
                const cplx w = inside ? d.W[row * d.m + col] : cmake(0.0, 0.0);
                zr += psi.x * w.x + psi.y * w.y;        // psi * conj(w)
                zi += psi.y * w.x - psi.x * w.y;
                if (DRESS) {
                    if (inside) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) cfma_conj(phi[f][jq], vs[f], psi);
                    }
                } else {
                    double wf = 0.0;
                    for (int f = 0; f < d.n_forb; ++f) wf += (row == d.forb_state[f]) ? 0.5 * d.forb_a[f] : 0.0;
                    const double pop = inside ? psi.x * psi.x + psi.y * psi.y : 0.0;
                    fb += wf * pop * pop;
                }
            }
        }
        if (DRESS) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int jq = 0; jq < 4; ++jq) {
                    if (jq >= MQs || f >= d.n_forb) continue;                    // (uniform)
                    cplx v = phi[f][jq];
                    v.x += dpp_xor<1>(v.x); v.y += dpp_xor<1>(v.y); v.x += dpp_xor<2>(v.x); v.y += dpp_xor<2>(v.y);
                    v.x += dpp_xor<4>(v.x); v.y += dpp_xor<4>(v.y); v.x += dpp_xor<8>(v.x); v.y += dpp_xor<8>(v.y);
                    const int col = 4 * jq + lk;
                    if (lc == 0 && col < d.m) {
                        const double pop = v.x * v.x + v.y * v.y;
                        d.Fd[(((size_t)b * npts + tau) * d.n_forb + f) * d.m + col] = cscale(v, 2.0 * d.forb_a[f] * pop);
                        fb += d.forb_a[f] * 0.5 * pop * pop;
                    }
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { zr += __shfl_xor(zr, off, 64); zi += __shfl_xor(zi, off, 64); fb += __shfl_xor(fb, off, 64); }
        if (lane == 0) {
            d.ztau[(size_t)b * npts + tau] = cmake(zr, zi);
            part[2 * tau] = (zr * zr + zi * zi) / mm;
            part[2 * tau + 1] = fb;
        }
    }
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(mf.lat_count + b, 1u) == (unsigned)(nblk - 1) ? 1 : 0;
        if (last) { mf.lat_count[b] = 0u; __threadfence(); }
    }
    __syncthreads();
    if (!last || wv != 0) return;
    double sv = 0.0, sf = 0.0;                                                    // fixed order: lane l adds tau = l, l + 64, ...; then the butterfly
    for (int t = lane; t < npts; t += 64) { sv += part[2 * t]; sf += part[2 * t + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sv += __shfl_xor(sv, off, 64); sf += __shfl_xor(sf, off, 64); }
    if (lane == 0) {
        const cplx z = d.ztau[(size_t)b * npts + d.steps];
        double reg_state = 0.0;
        if (d.has_speed) {
            const double resid = (double)npts - sv;
            d.su_resid[b] = resid;
            reg_state += d.a_speed * 0.5 * resid * resid;
        }
        if (d.n_forb > 0) reg_state += sf;
        d.zfin[b] = z;
        d.loss[b] = 1.0 - (z.x * z.x + z.y * z.y) / mm;
        d.reg_state[b] = reg_state;
    }
}

// ---- state regularisers (undressed forbidden levels, speed_up) in the latency mode ----------------------------------------------------
// The costate splits as Lambda_t = c0 z Lambda0_t + LambdaS_t: Lambda0 is the z-free sweep above; the source part obeys the affine
// recursion LambdaS_{t-1} = K_t^dagger LambdaS_t + S_t, LambdaS_{N-1} = S_N, with S_t = d(state regularisers)/dPsi at time t (source_at,
// qoc_state_source.h) -- elementwise in the sweeps' register layout: an undressed forbidden level and speed_up need Psi (PsiL[t-1], the
// state after slice t-1), W and one scalar per time step at the lane's own entry only.  Thin affine sweeps in three passes, the same
// workgroup shape as k_mfma_sweep_lat (a wave per 16-row tile, a workgroup per 4 columns):
//   role 0: every chunk from a zero costate over its slices            -> chunk offsets  a_c  (AoffL)
//   role 1: every group from a zero costate over its chunks, + a_c     -> group offsets  A_g  (GoffL)
//   role 2: from S_N down over whole groups (G_g^dagger X + A_g), the chunks of the own group (P_c^dagger X + a_c), then the slices of the
//           chunk, storing the TOTAL costate c0 z Lambda0_t + LambdaS_t (LamS) that k_mfma_grad_lat contracts with Psi_t.
// Against the pair-of-waves kernels of the batch path on the same chunks (k_mfma_bwd_offsets2 28 us, k_mfma_backward3<MODE 2> 8,
// <MODE 1> 39, which also carry the gradient work) the three passes are thin products with a vector add per step.
template <int NT, bool DRESS = false>
__global__ void __launch_bounds__(64 * NT) k_mfma_sweep_src(QocDev d, QocMfma mf, int role) {
    constexpr int LDP = 16 * NT + 1;
    __shared__ __attribute__((aligned(16))) cplx img[2][4 * LDP];
    const int lane = threadIdx.x & 63;
    const int I = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cs = mf.mq, per = role == 1 ? mf.NG : mf.C;
    const int item = (int)blockIdx.x / cs, jq0 = (int)blockIdx.x - item * cs;
    const int c = item / d.B, b = item - c * d.B;                                 // c: chunk (roles 0, 2) or group (role 1)
    if (c >= per || (d.skip_done && d.done[b])) return;                           // whole workgroup: no barrier yet
    const int lk = lane >> 4, lc = lane & 15, li4 = lane & 3;
    const int row = 16 * I + lc, col = 4 * jq0 + lk;
    const bool inside = row < d.n && col < d.m;
    const int MQs = mf.mq <= 2 ? 2 : 4, slot = (I * MQs + jq0) * 64 + lane, per_vec = NT * MQs * 64;
    double wrow = 0.0;                                                            // sum of 2 a_f over the forbidden levels equal to this lane's row
    for (int f = 0; f < d.n_forb; ++f) wrow += (row == d.forb_state[f]) ? 2.0 * d.forb_a[f] : 0.0;
    const cplx wown = inside ? d.W[row * d.m + col] : cmake(0.0, 0.0);
    const double speed_coef = d.has_speed ? -d.a_speed * d.su_resid[b] * 2.0 / ((double)d.m * (double)d.m) : 0.0;
    const cplx* psil = mf.PsiL + (size_t)b * d.steps * per_vec + slot;
    const cplx* ztau = d.has_speed ? d.ztau + (size_t)b * (d.steps + 1) : d.zfin + b;   // (a valid address either way: no branch around the load)
    const int zstride = d.has_speed ? 1 : 0;
    // DRESS (dressed forbidden levels, n_forb <= 4): S_tau[row][col] = sum_f Vs[row][st_f] Fd[tau][f][col] with the per-(tau, f, col)
    // amplitudes k_loss left in Fd (qoc_state_source.h) -- the rotation's row entries are loaded once, a step reads n_forb values
    cplx vsf[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) vsf[f] = (DRESS && f < d.n_forb && inside) ? d.Vs[row * d.n + d.forb_state[f]] : cmake(0.0, 0.0);
    const int nfm = DRESS ? d.n_forb * d.m : 0, colc = min(col, d.m - 1);
    const cplx* fdb = DRESS ? d.Fd + (size_t)b * (d.steps + 1) * nfm + colc : nullptr;
    auto source = [&](int tau) -> cplx {                                          // S_tau at this lane's entry, tau = 1 .. steps (0 for tau <= 0)
        const cplx psi = DRESS ? cmake(0.0, 0.0) : psil[(size_t)max(tau - 1, 0) * per_vec];
        const cplx zt = ztau[(size_t)tau * zstride];
        const double pop = psi.x * psi.x + psi.y * psi.y;
        cplx sv = cscale(psi, wrow * pop);
        if (DRESS) {
            const cplx* fd = fdb + (size_t)max(tau, 0) * nfm;
#pragma unroll
            for (int f = 0; f < 4; ++f) cfma(sv, vsf[f], fd[(size_t)min(f, d.n_forb - 1) * d.m]);   // (f >= n_forb: a valid address, a zero factor)
        }
        const cplx zw = cscale(cmul(zt, wown), speed_coef);
        sv.x += zw.x; sv.y += zw.y;
        return (tau > 0 && inside) ? sv : cmake(0.0, 0.0);
    };
    double pre = 0.0, pim = 0.0;
    struct Frag { cplx f[QQS]; };
    auto load_frag = [&](const cplx* __restrict__ F, Frag& fr) {
#pragma unroll
        for (int q = 0; q < QQS; ++q) fr.f[q] = F[(I * QQS + q) * 64 + lane];
    };
    int buf = 0;
    auto product = [&](const Frag& fr, const cplx add) {                          // X <- M^dagger X + add, M given by fragD(M)
        img[buf][lk * LDP + 16 * I + lc] = cmake(pre, pim);
        lds_barrier();
        double a = 0.0, bq = 0.0, cq = 0.0;
        cplx v[QQS];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) v[kb] = img[buf][li4 * LDP + 4 * kb + lk];
#pragma unroll
        for (int kb = 0; kb < QQS; ++kb) {
            const double br = fr.f[kb].x, bi = fr.f[kb].y, bs = br - bi;
            a = __builtin_amdgcn_mfma_f64_4x4x4f64(v[kb].x, br, a, 0, 0, 0);
            bq = __builtin_amdgcn_mfma_f64_4x4x4f64(v[kb].y, bi, bq, 0, 0, 0);
            cq = __builtin_amdgcn_mfma_f64_4x4x4f64(v[kb].x + v[kb].y, bs, cq, 0, 0, 0);
        }
        pre = a + bq + add.x; pim = cq - a + bq + add.y;
        buf ^= 1;
    };
    cplx* AoffL = mf.AoffL + (size_t)b * mf.C * per_vec + slot;
    cplx* GoffL = mf.GoffL + (size_t)b * mf.NG * per_vec + slot;
    const int G = mf.G, C = mf.C, NG = mf.NG;
    Frag F0, F1;
    if (role == 1) {
        // group offset: from zero over the chunks of group c, last to first
        const int clast = min(c * G + G, C) - 1, cfirst = c * G, nst = clast - cfirst + 1;
        const cplx* Pb = mf.PfD + (size_t)b * C * QFR;
        load_frag(Pb + (size_t)clast * QFR, F0);
        for (int s = 0; s < nst; ++s) {
            const int cc = clast - s;
            const cplx add = AoffL[(size_t)cc * per_vec];
            if (s + 1 < nst) load_frag(Pb + (size_t)(cc - 1) * QFR, F1);
            asm volatile("" ::: "memory");
            product(F0, add);
            if (s + 1 < nst) F0 = F1;
        }
        GoffL[(size_t)c * per_vec] = cmake(pre, pim);
        return;
    }
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps), len = t1 - t0;
    if (role == 2) {
        const cplx st = source(d.steps);                                          // terminal: LambdaS at the last slice = S_N
        pre = st.x; pim = st.y;
        const int g = c / G, clast = min(g * G + G, C) - 1;
        const cplx* Gb = mf.GfD + (size_t)b * NG * QFR;
        const cplx* Pb = mf.PfD + (size_t)b * C * QFR;
        const int n_grp = NG - 1 - g, n_ch = clast - c, n_bnd = n_grp + n_ch;
        auto bnd_mat = [&](int i) -> const cplx* { i = min(i, n_bnd - 1); return i < n_grp ? Gb + (size_t)(NG - 1 - i) * QFR : Pb + (size_t)(clast - (i - n_grp)) * QFR; };
        auto bnd_add = [&](int i) -> const cplx* { return i < n_grp ? GoffL + (size_t)(NG - 1 - i) * per_vec : AoffL + (size_t)(clast - (i - n_grp)) * per_vec; };
        if (n_bnd > 0) load_frag(bnd_mat(0), F0);
        for (int i = 0; i < n_bnd; ++i) {
            const cplx add = *bnd_add(i);
            if (i + 1 < n_bnd) load_frag(bnd_mat(i + 1), F1);
            asm volatile("" ::: "memory");
            product(F0, add);
            if (i + 1 < n_bnd) F0 = F1;
        }
    }
    // slices of the chunk, last to first: role 0 accumulates the offset, role 2 stores the total costate before every step
    const cplx z = d.zfin[b];
    const double c0 = -2.0 / ((double)d.m * (double)d.m);
    const cplx* lam0 = mf.LamL + (size_t)b * d.steps * per_vec + slot;
    cplx* lams = mf.LamS + (size_t)b * d.steps * per_vec + slot;
    const cplx* Kb = mf.KfD + kitem(mf, d.steps, b, t0);
    load_frag(Kb + (size_t)(len - 1) * mf.FR, F0);
    for (int i = 0; i < len; ++i) {
        const int t = t1 - 1 - i;
        if (role == 2) {
            const cplx l0 = lam0[(size_t)t * per_vec];
            const cplx zl = cscale(cmul(z, l0), c0);
            lams[(size_t)t * per_vec] = cmake(zl.x + pre, zl.y + pim);
            if (i + 1 == len) break;                                              // (the step over slice t0 would give the boundary of the chunk below)
        }
        const cplx add = source(t);
        if (i + 1 < len) load_frag(Kb + (size_t)(len - 2 - i) * mf.FR, F1);
        asm volatile("" ::: "memory");
        product(F0, add);
        if (i + 1 < len) F0 = F1;
    }
    if (role == 0) AoffL[(size_t)c * per_vec] = cmake(pre, pim);
}

// Gradient of the latency mode: two waves (row tiles h = 0, 1 of the costate) per time slice, 8 slices per workgroup, the control
// images H_k' (fragD layout, as k_mfma_backward3 holds them) staged in LDS once per workgroup.
//   Q = conj(Lambda0_t) Psi_t^T  (tiles (h, 0..1) on v_mfma_f64_16x16x4),  G_k = sum_{r,c} H_k'[r][c] Q[r][c]  (complex),
//   dL/du_{k,t} = c0 (Re z Re G_k + Im z Im G_k)                                   tensorflow_state.py:77-133 (first-order gradient)
// Every wave forms the overlap z = sum_j <w_j, psi_j(T)> itself (256 elements; tensorflow_state.py:282-333); the first workgroup
// of a seed publishes z and the loss (k_loss otherwise).
// fuse: the LAST workgroup of a seed to finish (a counter per seed) runs the tail of the iteration -- chain rule, stop rule, Adam
// (finish_body, qoc_kernels_finish.h) -- instead of a separate single-workgroup launch whose start-up and first round trips were
// 13 us of an 85 us iteration.  Only without pulse regularisers (their branch-heavy variant stays a kernel of its own).
template <int MQ, int KC, int NT = 2>
__global__ void __launch_bounds__(1024) k_mfma_grad_lat(QocDev d, QocMfma mf, QocAdamDev ap, int fuse) {
    constexpr int SL = 16 / NT;                                                   // slices per workgroup: NT waves (row tiles) each
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Hl = (cplx*)smem;                                                       // [KC] fragD(H_k'), zero beyond k
    double* gpart = (double*)(Hl + (size_t)KC * QFR);                             // [SL][NT tiles][4 rows][2 (re, im)][KC]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wv % NT, sl = wv / NT;
    const int nblk = (d.steps + SL - 1) / SL;
    const int b = blockIdx.x / nblk, cb = blockIdx.x - b * nblk;
    if (d.skip_done && d.done[b]) return;
    const int t = cb * SL + sl, tc = min(t, d.steps - 1);
    const bool live = t < d.steps;
    const int lk = lane >> 4, lc = lane & 15;
    // operands of the slice: costate rows of tile h, state rows of both tiles, all lane-contiguous
    const bool total = (fuse & 2) != 0;          // state regularisers: LamS holds the TOTAL costate c0 z Lambda0 + LambdaS (k_mfma_sweep_src); k_loss formed z and the loss
    double lr[MQ], li[MQ], pr[NT][MQ], pi[NT][MQ];
    {
        const cplx* ll = (total ? mf.LamS : mf.LamL) + ((size_t)b * d.steps + tc) * (NT * MQ) * 64 + lane;
        const cplx* pl = mf.PsiL + ((size_t)b * d.steps + tc) * (NT * MQ) * 64 + lane;
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            const cplx lv = ll[(h * MQ + q) * 64];
            lr[q] = lv.x; li[q] = lv.y;
#pragma unroll
            for (int Jp = 0; Jp < NT; ++Jp) { const cplx pv = pl[(Jp * MQ + q) * 64]; pr[Jp][q] = pv.x; pi[Jp][q] = pv.y; }
        }
    }
    double zr = 0.0, zi = 0.0;
    {
        const cplx* fin = d.inter + ((size_t)b * (d.steps + 1) + d.steps) * d.n * d.m;
        for (int o = lane; o < d.n * d.m; o += 64) {
            const cplx f = fin[o], wv2 = d.W[o];
            zr += f.x * wv2.x + f.y * wv2.y;
            zi += f.y * wv2.x - f.x * wv2.y;
        }
    }
    for (int o = threadIdx.x; o < KC * QFR; o += blockDim.x) Hl[o] = o < d.k * QFR ? mf.HfD[QFR + o] : cmake(0.0, 0.0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { zr += __shfl_xor(zr, off, 64); zi += __shfl_xor(zi, off, 64); }
    if (!total && cb == 0 && wv == 0 && lane == 0) {
        d.zfin[b] = cmake(zr, zi);
        d.loss[b] = 1.0 - (zr * zr + zi * zi) / ((double)d.m * (double)d.m);
        d.reg_state[b] = 0.0;
    }
    __syncthreads();
    double gr[KC], gi[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) { gr[kk] = 0.0; gi[kk] = 0.0; }
#pragma unroll
    for (int Jp = 0; Jp < NT; ++Jp) {
        d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            t1v = QMFMA(lr[q], pr[Jp][q], t1v);
            t2v = QMFMA(li[q], pi[Jp][q], t2v);
            t3v = QMFMA(lr[q] - li[q], pr[Jp][q] + pi[Jp][q], t3v);
        }
        const d4 qr = t1v + t2v, qi = t3v - t1v + t2v;                            // Re, Im of conj(lambda) psi
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            double ar = 0.0, ai = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const cplx hv = Hl[(size_t)kk * QFR + (Jp * QQS + 4 * h + r) * 64 + lane];
                ar = fma(hv.x, qr[r], ar); ar = fma(-hv.y, qi[r], ar);
                ai = fma(hv.x, qi[r], ai); ai = fma(hv.y, qr[r], ai);
            }
            gr[kk] += ar; gi[kk] += ai;
        }
    }
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {                                             // sum over the 16 lanes of a DPP row
        gr[kk] += dpp_xor<1>(gr[kk]); gr[kk] += dpp_xor<2>(gr[kk]); gr[kk] += dpp_xor<4>(gr[kk]); gr[kk] += dpp_xor<8>(gr[kk]);
        gi[kk] += dpp_xor<1>(gi[kk]); gi[kk] += dpp_xor<2>(gi[kk]); gi[kk] += dpp_xor<4>(gi[kk]); gi[kk] += dpp_xor<8>(gi[kk]);
    }
    if (lc == 0) {
        double* gp = gpart + (((size_t)sl * NT + h) * 4 + lk) * 2 * KC;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) { gp[kk] = gr[kk]; gp[KC + kk] = gi[kk]; }
    }
    __syncthreads();
    if (live && h == 0 && lane < d.k) {
        const double* gp = gpart + (size_t)sl * (4 * NT) * 2 * KC + lane;
        double sr = 0.0, si = 0.0;
#pragma unroll
        for (int x = 0; x < 4 * NT; ++x) { sr += gp[x * 2 * KC]; si += gp[x * 2 * KC + KC]; }
        const double c0 = -2.0 / ((double)d.m * (double)d.m);
        d.dLdu[((size_t)b * d.k + lane) * d.steps + t] = total ? sr : c0 * (zr * sr + zi * si);
    }
    if (fuse & 1) {
        __shared__ double red[34];
        __shared__ int last;
        __syncthreads();                                                          // every store of the workgroup has been issued and waited for
        if (threadIdx.x == 0) {
            __threadfence();                                                      // dLdu, loss: visible device-wide before the arrival is counted
            last = atomicAdd(mf.lat_count + b, 1u) == (unsigned)(nblk - 1) ? 1 : 0;
            if (last) { mf.lat_count[b] = 0u; __threadfence(); }                  // (reset for the next evaluation: stream order)
        }
        __syncthreads();
        if (!last) return;
        if (fuse & 4) finish_body<1>(d, ap, b, red); else finish_body<0>(d, ap, b, red);   // (4: local pulse regularisers present)
    }
}

// The same for 48 < n <= 64 (NT = 4): the control images are 64 KB each, so they pass through LDS two at a time; the Q tiles of a wave
// (row tile h against the four column tiles) stay in registers across the passes.  Four slices per workgroup, four waves per slice.
template <int MQ>
__global__ void __launch_bounds__(1024) k_mfma_grad_lat4(QocDev d, QocMfma mf, QocAdamDev ap, int fuse) {
    constexpr int NT = 4, SL = 4, KG = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* Hl = (cplx*)smem;                                                       // [KG] fragD(H_k') of the current pass
    double* gpart = (double*)(Hl + (size_t)KG * QFR);                             // [SL][NT tiles][4 rows][2 (re, im)][KG]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = wv % NT, sl = wv / NT;
    const int nblk = (d.steps + SL - 1) / SL;
    const int b = blockIdx.x / nblk, cb = blockIdx.x - b * nblk;
    if (d.skip_done && d.done[b]) return;
    const int t = cb * SL + sl, tc = min(t, d.steps - 1);
    const bool live = t < d.steps;
    const int lk = lane >> 4, lc = lane & 15;
    const bool total = (fuse & 2) != 0;          // state regularisers: LamS holds the total costate, k_loss formed z and the loss
    d4 qr[NT], qi[NT];
    {
        const cplx* ll = (total ? mf.LamS : mf.LamL) + ((size_t)b * d.steps + tc) * (NT * MQ) * 64 + lane;
        const cplx* pl = mf.PsiL + ((size_t)b * d.steps + tc) * (NT * MQ) * 64 + lane;
        double lr[MQ], li[MQ];
#pragma unroll
        for (int q = 0; q < MQ; ++q) { const cplx lv = ll[(h * MQ + q) * 64]; lr[q] = lv.x; li[q] = lv.y; }
#pragma unroll
        for (int Jp = 0; Jp < NT; ++Jp) {
            d4 t1v = {0, 0, 0, 0}, t2v = {0, 0, 0, 0}, t3v = {0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < MQ; ++q) {
                const cplx pv = pl[(Jp * MQ + q) * 64];
                t1v = QMFMA(lr[q], pv.x, t1v);
                t2v = QMFMA(li[q], pv.y, t2v);
                t3v = QMFMA(lr[q] - li[q], pv.x + pv.y, t3v);
            }
            qr[Jp] = t1v + t2v; qi[Jp] = t3v - t1v + t2v;                          // Re, Im of conj(lambda) psi
        }
    }
    double zr = 0.0, zi = 0.0;
    {
        const cplx* fin = d.inter + ((size_t)b * (d.steps + 1) + d.steps) * d.n * d.m;
        for (int o = lane; o < d.n * d.m; o += 64) {
            const cplx f = fin[o], wv2 = d.W[o];
            zr += f.x * wv2.x + f.y * wv2.y;
            zi += f.y * wv2.x - f.x * wv2.y;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { zr += __shfl_xor(zr, off, 64); zi += __shfl_xor(zi, off, 64); }
    if (!total && cb == 0 && wv == 0 && lane == 0) {
        d.zfin[b] = cmake(zr, zi);
        d.loss[b] = 1.0 - (zr * zr + zi * zi) / ((double)d.m * (double)d.m);
        d.reg_state[b] = 0.0;
    }
    const double c0 = -2.0 / ((double)d.m * (double)d.m);
    for (int k0 = 0; k0 < d.k; k0 += KG) {
        __syncthreads();                                                          // the readers of the previous pass are done
        for (int o = threadIdx.x; o < KG * QFR; o += blockDim.x) Hl[o] = (size_t)k0 * QFR + o < (size_t)d.k * QFR ? mf.HfD[(size_t)(1 + k0) * QFR + o] : cmake(0.0, 0.0);
        __syncthreads();
        double gr[KG], gi[KG];
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) { gr[kk] = 0.0; gi[kk] = 0.0; }
#pragma unroll
        for (int Jp = 0; Jp < NT; ++Jp)
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                double ar = 0.0, ai = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx hv = Hl[(size_t)kk * QFR + (Jp * QQS + 4 * h + r) * 64 + lane];
                    ar = fma(hv.x, qr[Jp][r], ar); ar = fma(-hv.y, qi[Jp][r], ar);
                    ai = fma(hv.x, qi[Jp][r], ai); ai = fma(hv.y, qr[Jp][r], ai);
                }
                gr[kk] += ar; gi[kk] += ai;
            }
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            gr[kk] += dpp_xor<1>(gr[kk]); gr[kk] += dpp_xor<2>(gr[kk]); gr[kk] += dpp_xor<4>(gr[kk]); gr[kk] += dpp_xor<8>(gr[kk]);
            gi[kk] += dpp_xor<1>(gi[kk]); gi[kk] += dpp_xor<2>(gi[kk]); gi[kk] += dpp_xor<4>(gi[kk]); gi[kk] += dpp_xor<8>(gi[kk]);
        }
        if (lc == 0) {
            double* gp = gpart + (((size_t)sl * NT + h) * 4 + lk) * 2 * KG;
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) { gp[kk] = gr[kk]; gp[KG + kk] = gi[kk]; }
        }
        __syncthreads();
        if (live && h == 0 && lane < KG && k0 + lane < d.k) {
            const double* gp = gpart + (size_t)sl * (4 * NT) * 2 * KG + lane;
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int x = 0; x < 4 * NT; ++x) { sr += gp[x * 2 * KG]; si += gp[x * 2 * KG + KG]; }
            d.dLdu[((size_t)b * d.k + k0 + lane) * d.steps + t] = total ? sr : c0 * (zr * sr + zi * si);
        }
    }
    if (fuse & 1) {
        __shared__ double red[34];
        __shared__ int last;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            last = atomicAdd(mf.lat_count + b, 1u) == (unsigned)(nblk - 1) ? 1 : 0;
            if (last) { mf.lat_count[b] = 0u; __threadfence(); }
        }
        __syncthreads();
        if (!last) return;
        if (fuse & 4) finish_body<1>(d, ap, b, red); else finish_body<0>(d, ap, b, red);   // (4: local pulse regularisers present)
    }
}
