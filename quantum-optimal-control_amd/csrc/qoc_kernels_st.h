// qoc_kernels_st.h -- state-transfer mode (matvecexp_op) for n <= 64, m <= 4: one workgroup per seed, the per-slice
// generator B_t = H0' + sum_k u_k H_k' lives in REGISTERS (thread (i, q) owns row i, columns 4e+q, e < 16 (interleaved in the quad: 64 contiguous bytes per quad per load)), vectors live
// in LDS, mat-vecs are 16 complex MACs per thread + a quad (4-lane) reduction.  The control-Hamiltonian stack is
// streamed from L2 once per slice per direction (coalesced 256 B per thread, 1 KB per row); in the backward pass the
// same stream also feeds the gradient inner products  dL/du_k = Re sum_ij conj(lambda_i) H_k'[i][j] psi_j.
//
// Reference: core/tensorflow_state.py:77-97 (get_matvecexp), :100-133 (matvecexp_op_grad), :244-261 (the slice loop).
#pragma once
#include "qoc_common.h"

#define ST_E 16      // matrix elements per thread
#define ST_MV 4      // max vectors (kernels are instantiated for 1, 2 and 4 vector slots)
#define ST_N 64      // max Hilbert-space dimension

static inline bool st_fused_supported(const QocDev& d) { return d.state_transfer && d.n <= ST_N && d.m <= ST_MV && d.k <= 8; }

// Bm[e] = sign*(H0'[i][j0+e] + sum_k u_k H_k'[i][j0+e]); optionally g[kk] += Re(H_k'[i][j0+e] * Mrow[e]).
// The k+1 matrices are streamed two at a time (32 independent 16-byte loads in flight per thread); loads are
// unconditional from clamped addresses + select: a per-element `if (valid) load` makes hipcc branch around every load.
template <bool WITH_GRAD>
__device__ __forceinline__ void st_assemble(const QocDev& d, int b, int t, int i, int j0, bool rowvalid, double sign,
                                            cplx Bm[ST_E], const cplx Mrow[ST_E], double g[8]) {
    const int n = d.n, nn = n * n;
    const int ic = min(i, n - 1);
    int col[ST_E];
    bool ok[ST_E];
#pragma unroll
    for (int e = 0; e < ST_E; ++e) { col[e] = min(4 * e + j0, n - 1); ok[e] = rowvalid && (4 * e + j0 < n); Bm[e] = cmake(0.0, 0.0); }
    const double* ub = d.u + (size_t)b * d.k * d.steps + t;
#pragma unroll 1
    for (int mm = 0; mm <= d.k; mm += 2) {
        const int m1 = min(mm + 1, d.k);
        const cplx* __restrict__ Ha = d.Hs + (size_t)mm * nn + (size_t)ic * n;
        const cplx* __restrict__ Hb = d.Hs + (size_t)m1 * nn + (size_t)ic * n;
        cplx ha[ST_E], hb[ST_E];
#pragma unroll
        for (int e = 0; e < ST_E; ++e) { ha[e] = Ha[col[e]]; hb[e] = Hb[col[e]]; }
        const double ca = sign * (mm == 0 ? 1.0 : ub[(size_t)(mm - 1) * d.steps]);
        const double cb = (mm + 1 <= d.k) ? sign * ub[(size_t)mm * d.steps] : 0.0;
        double ga = 0.0, gb = 0.0;
#pragma unroll
        for (int e = 0; e < ST_E; ++e) {
            const double ax = ok[e] ? ha[e].x : 0.0, ay = ok[e] ? ha[e].y : 0.0;
            const double bx = ok[e] ? hb[e].x : 0.0, by = ok[e] ? hb[e].y : 0.0;
            Bm[e].x = fma(ca, ax, Bm[e].x); Bm[e].y = fma(ca, ay, Bm[e].y);
            Bm[e].x = fma(cb, bx, Bm[e].x); Bm[e].y = fma(cb, by, Bm[e].y);
            if (WITH_GRAD) {
                ga = fma(ax, Mrow[e].x, ga); ga = fma(-ay, Mrow[e].y, ga);
                gb = fma(bx, Mrow[e].x, gb); gb = fma(-by, Mrow[e].y, gb);
            }
        }
        if (WITH_GRAD) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (mm >= 1 && q == mm - 1) g[q] += ga;                  // matrix mm   is control kk = mm - 1
                if (mm + 1 <= d.k && q == mm) g[q] += gb;                // matrix mm+1 is control kk = mm
            }
        }
    }
}

// y[i][:] = sum_j Bm[i][j] v[j][:] for this thread's row; result valid in all 4 lanes of the quad.
// No per-element predicates: Bm is zero outside the matrix and the padded rows of v are zero.
template <int MV>
__device__ __forceinline__ void st_matvec(const cplx Bm[ST_E], const cplx* __restrict__ v, int j0, cplx acc[MV]) {
#pragma unroll
    for (int jv = 0; jv < MV; ++jv) acc[jv] = cmake(0.0, 0.0);
#pragma unroll
    for (int e = 0; e < ST_E; ++e) {
#pragma unroll
        for (int jv = 0; jv < MV; ++jv) cfma(acc[jv], Bm[e], v[(4 * e + j0) * MV + jv]);
    }
#pragma unroll
    for (int jv = 0; jv < MV; ++jv) {
        acc[jv].x += __shfl_xor(acc[jv].x, 1, 64); acc[jv].y += __shfl_xor(acc[jv].y, 1, 64);
        acc[jv].x += __shfl_xor(acc[jv].x, 2, 64); acc[jv].y += __shfl_xor(acc[jv].y, 2, 64);
    }
}

// out = sum_{j<T} Bm^j v / j!  (v = vec[cur] on entry, out[] holds v on entry for the owner lane)
template <int MV>
__device__ __forceinline__ void st_taylor(const QocDev& d, const cplx Bm[ST_E], cplx (*vec)[ST_N * MV], int& cur,
                                          int i, int q, int j0, bool rowvalid, cplx out[MV]) {
    double fact = 1.0;
    for (int ii = 1; ii < d.T; ++ii) {
        cplx acc[MV];
        st_matvec<MV>(Bm, vec[cur], j0, acc);
        fact *= (double)ii;
        const double inv = 1.0 / fact;
        if (q == 0 && rowvalid) {
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) {
                vec[cur ^ 1][i * MV + jv] = acc[jv];                            // psi_n = H psi_n       :94 / :130
                out[jv].x = fma(acc[jv].x, inv, out[jv].x); out[jv].y = fma(acc[jv].y, inv, out[jv].y);   // += psi_n/factorial :95 / :131
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// MV = number of vector slots (>= m; extra slots carry zeros)
template <int MV>
__global__ void __launch_bounds__(256) k_st_fwd_fused(QocDev d) {
    __shared__ __attribute__((aligned(16))) cplx vec[2][ST_N * MV];
    const int b = blockIdx.x, tid = threadIdx.x, i = tid >> 2, q = tid & 3, j0 = q;   // j0 = first owned column; owned columns are 4e + j0
    const int n = d.n, m = d.m, nm = n * m;
    const bool rowvalid = i < n;
    cplx* iv = d.inter + (size_t)b * (d.steps + 1) * nm;
    cplx out[MV];
    int cur = 0;
    for (int o = tid; o < 2 * ST_N * MV; o += 256) (&vec[0][0])[o] = cmake(0.0, 0.0);
    __syncthreads();
#pragma unroll
    for (int jv = 0; jv < MV; ++jv) out[jv] = cmake(0.0, 0.0);
    if (q == 0 && rowvalid) {
#pragma unroll
        for (int jv = 0; jv < MV; ++jv)
            if (jv < m) { out[jv] = d.V[i * m + jv]; vec[0][i * MV + jv] = out[jv]; iv[i * m + jv] = out[jv]; }
    }
    __syncthreads();
    cplx Bm[ST_E];
    double gdummy[8];
    for (int t = 0; t < d.steps; ++t) {
        st_assemble<false>(d, b, t, i, j0, rowvalid, 1.0, Bm, Bm, gdummy);
        st_taylor<MV>(d, Bm, vec, cur, i, q, j0, rowvalid, out);
        if (q == 0 && rowvalid) {
            cplx* o = iv + (size_t)(t + 1) * nm;
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) {
                if (jv < m) o[i * m + jv] = out[jv];
                vec[cur][i * MV + jv] = out[jv];
            }
        }
        __syncthreads();
    }
}

template <int MV>
__global__ void __launch_bounds__(256) k_st_bwd_fused(QocDev d) {
    __shared__ __attribute__((aligned(16))) cplx vec[2][ST_N * MV];
    __shared__ __attribute__((aligned(16))) cplx psiv[ST_N * MV];
    __shared__ double red[4][8];
    const int b = blockIdx.x, tid = threadIdx.x, i = tid >> 2, q = tid & 3, j0 = q;   // j0 = first owned column; owned columns are 4e + j0
    const int n = d.n, m = d.m, nm = n * m;
    const bool rowvalid = i < n;
    const bool need_src = d.n_forb > 0 || d.has_speed;
    const cplx* iv = d.inter + (size_t)b * (d.steps + 1) * nm;
    cplx lam[MV];
    int cur = 0;
    for (int o = tid; o < 2 * ST_N * MV; o += 256) (&vec[0][0])[o] = cmake(0.0, 0.0);
    for (int o = tid; o < ST_N * MV; o += 256) psiv[o] = cmake(0.0, 0.0);
    __syncthreads();
#pragma unroll
    for (int jv = 0; jv < MV; ++jv) lam[jv] = cmake(0.0, 0.0);
    if (q == 0 && rowvalid) {
        const cplx z = d.zfin[b];
        const double c0 = -2.0 / ((double)m * (double)m);
#pragma unroll
        for (int jv = 0; jv < MV; ++jv)
            if (jv < m) {
                cplx v = cscale(cmul(z, d.W[i * m + jv]), c0);
                if (need_src) v = cadd(v, source_at(d, b, d.steps, i, jv));
                lam[jv] = v;
                vec[0][i * MV + jv] = v;
            }
    }
    cplx Bm[ST_E], Mrow[ST_E];
    for (int t = d.steps - 1; t >= 0; --t) {
        if (q == 0 && rowvalid) {
            const cplx* p = iv + (size_t)(t + 1) * nm;
#pragma unroll
            for (int jv = 0; jv < MV; ++jv)
                if (jv < m) psiv[i * MV + jv] = p[i * m + jv];
        }
        __syncthreads();
        // M[i][j] = sum_jv conj(lambda[i][jv]) psi[j][jv]   (zero outside the matrix: padded lambda/psi rows are zero)
        {
            cplx lrow[MV];
            const int ic = min(i, ST_N - 1);
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) lrow[jv] = vec[cur][ic * MV + jv];
#pragma unroll
            for (int e = 0; e < ST_E; ++e) {
                cplx acc = cmake(0.0, 0.0);
#pragma unroll
                for (int jv = 0; jv < MV; ++jv) cfma_conj(acc, lrow[jv], psiv[(4 * e + j0) * MV + jv]);
                Mrow[e] = acc;
            }
        }
        double g[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) g[kk] = 0.0;
        st_assemble<true>(d, b, t, i, j0, rowvalid, -1.0, Bm, Mrow, g);            // H = sum (-uks) H_all   :121-123
        // workgroup reduction of the k control gradients                                                     :112-114
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk >= d.k) continue;
            double v = g[kk];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((tid & 63) == 0) red[tid >> 6][kk] = v;
        }
        __syncthreads();
        if (tid < d.k) d.dLdu[((size_t)b * d.k + tid) * d.steps + t] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (t == 0) break;
        st_taylor<MV>(d, Bm, vec, cur, i, q, j0, rowvalid, lam);
        if (q == 0 && rowvalid) {
#pragma unroll
            for (int jv = 0; jv < MV; ++jv) {
                if (need_src && jv < m) lam[jv] = cadd(lam[jv], source_at(d, b, t, i, jv));
                vec[cur][i * MV + jv] = lam[jv];
            }
        }
        // the barrier at the top of the next slice orders this write before its readers
    }
}

static inline void st_fused_launch(const QocDev& d, hipStream_t s, bool forward) {
    const int mv = d.m <= 1 ? 1 : (d.m <= 2 ? 2 : 4);
    if (forward) {
        if (mv == 1) hipLaunchKernelGGL(k_st_fwd_fused<1>, dim3(d.B), dim3(256), 0, s, d);
        else if (mv == 2) hipLaunchKernelGGL(k_st_fwd_fused<2>, dim3(d.B), dim3(256), 0, s, d);
        else hipLaunchKernelGGL(k_st_fwd_fused<4>, dim3(d.B), dim3(256), 0, s, d);
    } else {
        if (mv == 1) hipLaunchKernelGGL(k_st_bwd_fused<1>, dim3(d.B), dim3(256), 0, s, d);
        else if (mv == 2) hipLaunchKernelGGL(k_st_bwd_fused<2>, dim3(d.B), dim3(256), 0, s, d);
        else hipLaunchKernelGGL(k_st_bwd_fused<4>, dim3(d.B), dim3(256), 0, s, d);
    }
}
