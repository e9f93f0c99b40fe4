// qoc_gemm_ts.h -- TIME-AXIS sharding of ONE large trajectory over the GPUs of a node ("alternative for C5": n = 512, 2000 slices).
//
// The GEMM path cuts the pulse into NC chunks of S slices; what couples the chunks is only the chunk products P_c.  Rank r of G owns the
// chunks [cb[r], cb[r+1]) and per iteration
//   A  forms K_t and the product tree of ITS slices (the n^3 work: 1/G of it),                       tensorflow_state.py:25-46
//   B  multiplies its chunk products into one rank product R_r,
//   X1 all-gathers the G rank products (RCCL, device to device on the engine's stream: G x N^2 complex -- 4 MB each at n = 512),
//   C  walks the rank products: its start vectors [X | Psi] = R_{r-1} ... R_0 [U0 | Psi0], and -- every rank for itself -- the whole product,
//      i.e. final_state and Psi_N, hence the overlap z and the terminal costate (no second exchange for them),        :204-242, 323-333
//   D  sweeps its chunks forward, E  backward from R_{r+1}^+ ... R_{G-1}^+ Lambda_N, forming the gradients of ITS slices,         :49-65
//   X2 all-reduces (sum) the gradient array, of which every rank filled its own columns (k x steps doubles),
//   F  runs the regulariser / Adam tail on the whole pulse -- identically on every rank: the ranks stay in lock step without a broadcast.
// Two small collectives per iteration sit on the data path; everything else is the unsharded path on a sub-range of slices.
//
// time_rank = -1 EMULATES all G ranks inside one engine on one GPU (phases A, B, D, E looped over r, the exchanges are no-ops on the shared
// buffers): the decomposition -- index ranges, rank products, the two chains over them -- is then testable against the CPU restatement on a one-GPU box;
// the real mode differs by the two RCCL calls only.  Scope: unitary mode, one control set, no state regulariser, N >= 128 with an even number
// of row tiles, m <= 8 (the wide gradient product), G <= NC.  Every rank allocates the full-size buffers (C5: ~42 GB of 288).
#pragma once

struct qoc_comm;
void qoc_comm_detach(qoc_comm* c);                                                                // qoc_comm.h: an engine lets go of its communicator
int qoc_ts_all_gather(qoc_comm* c, void* buf, size_t doubles_per_rank, hipStream_t s);      // qoc_comm.h: in place, rank r's block at buf + r * count
int qoc_ts_all_reduce_sum(qoc_comm* c, double* buf, size_t doubles, hipStream_t s);

// chunk-start vectors of the chunks [c_first, c_first + c_count): chunk c_first from the thin block of `Ystart`, the others from the per-step slots
__global__ void __launch_bounds__(256) k_ts_take_bnd(QocDev d, const cplx* __restrict__ Ys, const cplx* __restrict__ Ystart, cplx* __restrict__ Psibnd,
                                                     int N, int xw, int c_first, int c_count) {
    const int ld = xw + QOC_TW;
    const size_t per = (size_t)N * QOC_TW, slot = (size_t)N * ld;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)c_count * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t ci = o / per, e = o - ci * per;
        const int c = c_first + (int)ci;
        const int row = (int)(e / QOC_TW), col = (int)(e - (size_t)row * QOC_TW);
        const cplx* Y = ci == 0 ? Ystart : Ys + (size_t)c * slot;
        Psibnd[(size_t)c * per + e] = Y[(size_t)row * ld + xw + col];
    }
}
// inter[t + 1] (API layout) from interP[t] for the slices [t_first, t_first + t_count)
__global__ void __launch_bounds__(256) k_ts_unpad_inter(QocDev d, const cplx* __restrict__ interP, int N, int t_first, int t_count) {
    const size_t nm = (size_t)d.n * d.m, per = (size_t)N * QOC_TW;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)t_count * nm; o += (size_t)gridDim.x * blockDim.x) {
        const size_t ti = o / nm, e = o - ti * nm;
        const size_t t = (size_t)t_first + ti;
        const int row = (int)(e / d.m), col = (int)(e - (size_t)row * d.m);
        d.inter[(t + 1) * nm + e] = interP[t * per + (size_t)row * QOC_TW + col];
    }
}
// Psi_N = the thin block of the whole product applied to [U0 | Psi0]: inter[steps], what k_loss forms the overlap from
__global__ void __launch_bounds__(256) k_ts_set_psi_final(QocDev d, const cplx* __restrict__ Y, int N, int xw) {
    const int ld = xw + QOC_TW;
    const size_t nm = (size_t)d.n * d.m;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nm; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / d.m), col = (int)(e - (size_t)row * d.m);
        d.inter[(size_t)d.steps * nm + e] = Y[(size_t)row * ld + xw + col];
    }
}
// LamP[(c + 1) S - 1] = Ebnd[c] for the chunks [c_first, c_first + c_count)
__global__ void __launch_bounds__(256) k_ts_set_chunk_ends(cplx* __restrict__ LamP, const cplx* __restrict__ Ebnd, int N, int S, int c_first, int c_count) {
    const size_t per = (size_t)N * QOC_TW;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)c_count * per; o += (size_t)gridDim.x * blockDim.x) {
        const size_t ci = o / per, e = o - ci * per;
        const size_t c = (size_t)c_first + ci;
        LamP[(c * S + (S - 1)) * per + e] = Ebnd[c * per + e];
    }
}
__global__ void __launch_bounds__(256) k_ts_copy(cplx* __restrict__ dst, const cplx* __restrict__ src, size_t count) {
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < count; o += (size_t)gridDim.x * blockDim.x) dst[o] = src[o];
}

// read-back of inter_vecs on one rank of a real run: every rank holds the time points of ITS slices (and all of them inter[0], inter[steps]);
// what it does not own is cleared, then the ranks sum -- every time point has exactly one owner
__global__ void __launch_bounds__(256) k_ts_clear_foreign(QocDev d, int t_first, int t_count, int keep_ends) {
    const size_t nm = (size_t)d.n * d.m;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < (size_t)(d.steps + 1) * nm; o += (size_t)gridDim.x * blockDim.x) {
        const int tp = (int)(o / nm);                                  // time point: inter[tp], tp = t + 1 for slice t; 0 = the initial vectors
        const bool ends = tp == 0 || tp == d.steps;
        const bool mine = ends ? keep_ends != 0 : (tp - 1 >= t_first && tp - 1 < t_first + t_count);
        if (!mine) d.inter[o] = cmake(0.0, 0.0);
    }
}
static inline int qoc_gemm_ts_gather_inter(QocGemm& gm, const QocDev& d, hipStream_t s) {
    if (gm.ts_G <= 0 || gm.ts_rank < 0) return 0;                      // (an emulating engine holds every slice)
    const int c0 = gm.ts_cb[gm.ts_rank], c1 = gm.ts_cb[gm.ts_rank + 1];
    const int t0 = c0 * gm.S, t1 = min(c1 * gm.S, d.steps);
    hipLaunchKernelGGL(k_ts_clear_foreign, dim3(2048), dim3(256), 0, s, d, t0, t1 - t0, gm.ts_rank == 0 ? 1 : 0);
    return qoc_ts_all_reduce_sum(gm.ts_comm, (double*)d.inter, (size_t)2 * (d.steps + 1) * d.n * d.m, s);
}

static inline bool qoc_gemm_ts_supported(const QocGemm& gm, const QocDev& d, int G, std::string& why) {
    if (d.state_transfer) { why = "unitary mode only"; return false; }
    if (d.B != 1) { why = "one control set (restart batches shard over seeds: parallel_seeds)"; return false; }
    if (d.n_forb > 0 || d.has_speed) { why = "no state regulariser (the affine costate needs a second exchange)"; return false; }
    if (gm.persistent || gm.direct || gm.wideW <= 0) { why = "N >= 128 with an even number of 32-row tiles and m <= 8 (the GEMM path's large-matrix route)"; return false; }
    if (G < 1 || G > gm.NC) { why = "1 <= time_shards <= number of chunks"; return false; }
    return true;
}

static inline void qoc_gemm_ts_ranges(QocGemm& gm, int G) {
    gm.ts_cb.resize(G + 1);
    for (int r = 0; r <= G; ++r) gm.ts_cb[r] = (int)(((long long)gm.NC * r) / G);
}

// phases A + B for rank r: K_t and the tree of its slices, its rank product into Rall[r]
static inline void qoc_gemm_ts_expm(QocGemm& gm, const QocDev& d, int r, hipStream_t s) {
    const int N = gm.N, S = gm.S;
    const size_t NN = (size_t)N * N;
    const int c0 = gm.ts_cb[r], c1 = gm.ts_cb[r + 1];
    const size_t i0 = (size_t)c0 * S, i1 = min((size_t)c1 * S, (size_t)d.steps);   // (the padded slices of the last chunk keep the identity of set-up)
    if (i1 > i0) qoc_gemm_expm_products(gm, d, s, i0, i1 - i0);
    qoc_gemm_tree(gm, d, s, (size_t)c0 * S, (size_t)(c1 - c0) * S);
    const cplx* Pc = qoc_gemm_chunk_products(gm);
    cplx* out = gm.ts_Rall + (size_t)r * NN;
    if (c1 - c0 == 1) {
        hipLaunchKernelGGL(k_ts_copy, dim3(gemm_grid(NN)), dim3(256), 0, s, out, Pc + (size_t)c0 * NN, NN);
        return;
    }
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = g.ldb = g.ldc = N; g.Kdim = N; g.tiles_m = g.tiles_n = N / 32; g.batch = 1; g.alpha = 1.0;
    const cplx* cur = Pc + (size_t)c0 * NN;
    for (int c = c0 + 1; c < c1; ++c) {                           // R <- P_c R (later chunk on the left)
        cplx* dst = c == c1 - 1 ? out : gm.ts_Rtmp + (size_t)((c - c0) & 1) * NN;
        g.A = Pc + (size_t)c * NN; g.Bm = cur; g.C = dst;
        qoc_gemm_launch(gm, false, 0, g, s);
        cur = dst;
    }
}

// phase C, the part every rank does for itself: Yr[q + 1] = R_q Yr[q] from [U0 | Psi0]; final_state, unitary_scale, Psi_N
static inline void qoc_gemm_ts_prefix(QocGemm& gm, const QocDev& d, int G, hipStream_t s) {
    const int N = gm.N, xw = N, ld = xw + QOC_TW;
    const size_t NN = (size_t)N * N, yslot = (size_t)N * ld;
    hipLaunchKernelGGL(k_gemm_chain_init, dim3(gemm_grid((size_t)N * ld)), dim3(256), 0, s, d, gm.ts_Yr, gm.Psibnd, N, gm.NC, xw);   // Yr[0], Psibnd[0], inter[0]
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.ldb = g.ldc = ld; g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = ld / 32; g.batch = 1; g.alpha = 1.0;
    for (int q = 0; q < G; ++q) {
        g.A = gm.ts_Rall + (size_t)q * NN; g.Bm = gm.ts_Yr + (size_t)q * yslot; g.C = gm.ts_Yr + (size_t)(q + 1) * yslot;
        qoc_gemm_launch(gm, false, 0, g, s);
    }
    hipLaunchKernelGGL(k_gemm_take_final, dim3(1), dim3(1024), 0, s, d, (const cplx*)(gm.ts_Yr + (size_t)G * yslot), N);
    hipLaunchKernelGGL(k_ts_set_psi_final, dim3(gemm_grid((size_t)d.n * d.m)), dim3(256), 0, s, d, (const cplx*)(gm.ts_Yr + (size_t)G * yslot), N, xw);
}

// phases C (own boundaries) + D for rank r: chunk-start vectors of its chunks, forward sweeps, inter_vecs of its slices
static inline void qoc_gemm_ts_forward(QocGemm& gm, const QocDev& d, int r, hipStream_t s) {
    const int N = gm.N, xw = N, ld = xw + QOC_TW, S = gm.S;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW, yslot = (size_t)N * ld;
    const int c0 = gm.ts_cb[r], c1 = gm.ts_cb[r + 1], nc = c1 - c0;
    const cplx* Pc = qoc_gemm_chunk_products(gm);
    const cplx* Ystart = gm.ts_Yr + (size_t)r * yslot;
    // Psi at the start of the rank's chunks: the thin block of Ystart, then Psibnd[c + 1] = P_c Psibnd[c] on the m vectors alone (the wide X beside them is
    // only needed in the prefix over the RANK products, qoc_gemm_ts_prefix; final_state comes from there)
    hipLaunchKernelGGL(k_ts_take_bnd, dim3(gemm_grid(thin)), dim3(256), 0, s, d, (const cplx*)gm.Y0, Ystart, gm.Psibnd, N, xw, c0, 1);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.ldb = g.ldc = QOC_TW; g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = 1; g.batch = 1; g.alpha = 1.0;
    for (int c = c0; c + 1 < c1; ++c) {                           // (the last chunk's end is the next rank's start)
        g.A = Pc + (size_t)c * NN; g.Bm = gm.Psibnd + (size_t)c * thin; g.C = gm.Psibnd + (size_t)(c + 1) * thin;
        qoc_gemm_launch(gm, false, 0, g, s);
    }
    GemmArgs h;
    memset(&h, 0, sizeof h);
    h.lda = N; h.sA = (long long)NN * S; h.ldb = h.ldc = QOC_TW; h.Kdim = N; h.tiles_m = N / 32; h.tiles_n = 1;
    h.batch = nc; h.alpha = 1.0; h.sC = (long long)thin * S;
    const size_t i0 = (size_t)c0 * S;
    for (int j = 0; j < S; ++j) {                                 // Psi_{cS+j} = K_{cS+j} Psi_{cS+j-1}: the rank's chunks together, one launch per j
        h.A = gm.K + (i0 + j) * NN;
        if (j == 0) { h.Bm = gm.Psibnd + (size_t)c0 * thin; h.sB = (long long)thin; }
        else { h.Bm = gm.interP + (i0 + j - 1) * thin; h.sB = (long long)thin * S; }
        h.C = gm.interP + (i0 + j) * thin;
        qoc_gemm_launch(gm, false, 0, h, s);
    }
    // inter[t + 1] for the rank's slices t -- except inter[steps]: that one is Psi_N of the whole product on every rank (qoc_gemm_ts_prefix)
    const int t0 = (int)i0, t1 = min((int)(i0 + (size_t)nc * S), d.steps);
    const int cnt = t1 - t0 - (t1 == d.steps ? 1 : 0);
    if (cnt > 0) hipLaunchKernelGGL(k_ts_unpad_inter, dim3(gemm_grid((size_t)cnt * d.n * d.m)), dim3(256), 0, s, d, (const cplx*)gm.interP, N, t0, cnt);
}

// phase E, the part every rank does for itself: terminal costate, then Er[q] = R_q^dagger Er[q + 1] (the costate at the END of rank q - 1's slices)
static inline void qoc_gemm_ts_suffix(QocGemm& gm, const QocDev& d, int G, hipStream_t s) {
    const int N = gm.N, NC = gm.NC;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    hipLaunchKernelGGL(k_gemm_sources, dim3(gemm_grid(thin)), dim3(256), 0, s, d, gm.SrcP, gm.Ebnd, N, gm.SP, NC, QOC_TW);        // Ebnd[NC - 1] = -(2 / m^2) z W
    hipLaunchKernelGGL(k_ts_copy, dim3(gemm_grid(thin)), dim3(256), 0, s, gm.ts_Er + (size_t)G * thin, (const cplx*)(gm.Ebnd + (size_t)(NC - 1) * thin), thin);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.ldb = g.ldc = QOC_TW; g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = 1; g.batch = 1; g.alpha = 1.0;
    for (int q = G - 1; q >= 1; --q) {
        g.A = gm.ts_Rall + (size_t)q * NN; g.Bm = gm.ts_Er + (size_t)(q + 1) * thin; g.C = gm.ts_Er + (size_t)q * thin;
        qoc_gemm_launch(gm, true, 0, g, s);
    }
}

// phase E for rank r: chunk-end costates of its chunks, backward sweeps, gradients of its slices (into its columns of dLdu)
static inline void qoc_gemm_ts_backward(QocGemm& gm, const QocDev& d, int r, hipStream_t s) {
    const int N = gm.N, S = gm.S;
    const size_t NN = (size_t)N * N, thin = (size_t)N * QOC_TW;
    const int c0 = gm.ts_cb[r], c1 = gm.ts_cb[r + 1], nc = c1 - c0;
    const cplx* Pc = qoc_gemm_chunk_products(gm);
    hipLaunchKernelGGL(k_ts_copy, dim3(gemm_grid(thin)), dim3(256), 0, s, gm.Ebnd + (size_t)(c1 - 1) * thin, (const cplx*)(gm.ts_Er + (size_t)(r + 1) * thin), thin);
    GemmArgs g;
    memset(&g, 0, sizeof g);
    g.lda = N; g.ldb = g.ldc = QOC_TW; g.Kdim = N; g.tiles_m = N / 32; g.tiles_n = 1; g.batch = 1; g.alpha = 1.0;
    for (int c = c1 - 1; c > c0; --c) {                           // E_{c-1} = P_c^dagger E_c
        g.A = Pc + (size_t)c * NN; g.Bm = gm.Ebnd + (size_t)c * thin; g.C = gm.Ebnd + (size_t)(c - 1) * thin;
        qoc_gemm_launch(gm, true, 0, g, s);
    }
    hipLaunchKernelGGL(k_ts_set_chunk_ends, dim3(gemm_grid((size_t)nc * thin)), dim3(256), 0, s, gm.LamP, (const cplx*)gm.Ebnd, N, S, c0, nc);
    const size_t i0 = (size_t)c0 * S;
    GemmArgs w;
    memset(&w, 0, sizeof w);
    w.lda = N; w.sA = (long long)NN * S; w.ldb = w.ldc = QOC_TW; w.sB = w.sC = (long long)thin * S;
    w.Kdim = N; w.tiles_m = N / 32; w.tiles_n = 1; w.batch = nc; w.alpha = 1.0;
    for (int j = S - 1; j >= 1; --j) {                            // Lambda_{cS+j-1} = K_{cS+j}^dagger Lambda_{cS+j}
        w.A = gm.K + (i0 + j) * NN; w.Bm = gm.LamP + (i0 + j) * thin; w.C = gm.LamP + (i0 + j - 1) * thin;
        qoc_gemm_launch(gm, true, 0, w, s);
    }
    // gradients of the slices [t0, t1): one wide product for all controls (qoc_kernels_gemm.h, "gradients of large problems")
    const int t0 = (int)i0, t1 = min((int)(i0 + (size_t)nc * S), d.steps);
    if (t1 <= t0) return;
    const int cnt = t1 - t0, W = (int)((((size_t)cnt * QOC_WIDE_MV + 127) / 128) * 128);
    GemmArgs h;
    memset(&h, 0, sizeof h);
    h.A = gm.HsP + NN; h.sA = (long long)NN; h.lda = N;
    h.Bm = gm.wideP; h.sB = 0; h.ldb = W;
    h.C = gm.wideC; h.sC = (long long)N * W; h.ldc = W;
    h.Kdim = N; h.tiles_m = N / 32; h.tiles_n = W / 32; h.batch = d.k; h.alpha = 1.0;
    hipLaunchKernelGGL(k_gemm_to_wide, dim3(gemm_grid((size_t)cnt * N * QOC_WIDE_MV)), dim3(256), 0, s, d, (const cplx*)(gm.interP + (size_t)t0 * thin),
                       (const cplx*)(gm.LamP + (size_t)t0 * thin), gm.wideP, gm.wideL, N, W, cnt);
    qoc_gemm_launch(gm, false, 0, h, s);
    hipLaunchKernelGGL(k_gemm_dot_wide, dim3((unsigned)(((size_t)d.k * cnt + 3) / 4)), dim3(256), 0, s, d, 0, (const cplx*)gm.wideC, (const cplx*)gm.wideL, N, W, t0, cnt);
}

// one evaluation of a time-sharded engine up to (not including) the regulariser / Adam tail.  `loss` = launch_loss of the engine.
template <class Loss, class ProfBegin, class ProfEnd>
static inline int qoc_gemm_ts_evaluate(QocGemm& gm, const QocDev& d, hipStream_t s, Loss&& loss, ProfBegin&& prof_begin, ProfEnd&& prof_end) {
    const int G = gm.ts_G;
    int r0 = gm.ts_rank < 0 ? 0 : gm.ts_rank, r1 = gm.ts_rank < 0 ? G : gm.ts_rank + 1;
    const size_t NN = (size_t)gm.N * gm.N;
#ifdef QOC_DEBUG     // timing experiments only (tools/c5_time_sharded.py rank-time): an emulating engine runs ONE rank's share and no exchange -- results are garbage
    if (gm.ts_rank < 0) if (const char* only = getenv("QOC_TS_ONLY_RANK")) { r0 = atoi(only); r1 = r0 + 1; }
#endif
    int rc = prof_begin();
    if (rc) return rc;
    for (int r = r0; r < r1; ++r) qoc_gemm_ts_expm(gm, d, r, s);
    rc = prof_end();
    if (rc) return rc;
    if (gm.ts_rank >= 0 && (rc = qoc_ts_all_gather(gm.ts_comm, gm.ts_Rall, 2 * NN, s))) return rc;           // X1: the rank products
    qoc_gemm_ts_prefix(gm, d, G, s);
    for (int r = r0; r < r1; ++r) qoc_gemm_ts_forward(gm, d, r, s);
    loss();
    qoc_gemm_ts_suffix(gm, d, G, s);
    if (gm.ts_rank >= 0 && hipMemsetAsync(d.dLdu, 0, (size_t)d.k * d.steps * sizeof(double), s) != hipSuccess) return 1;
    for (int r = r0; r < r1; ++r) qoc_gemm_ts_backward(gm, d, r, s);
    if (gm.ts_rank >= 0 && (rc = qoc_ts_all_reduce_sum(gm.ts_comm, d.dLdu, (size_t)d.k * d.steps, s))) return rc;   // X2: the gradient columns
    return 0;
}
