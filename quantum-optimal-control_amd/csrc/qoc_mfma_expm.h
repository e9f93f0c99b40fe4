// qoc_mfma_expm.h -- MFMA path: the three kernels of the exponentials K_t = matexp(A_t) + chunk products (qoc_config.variant).
// Reference semantics: core/tensorflow_state.py:25-46 (get_matexp).
#pragma once
#include "qoc_mfma_frag.h"

// ---- kernel E: K_t = matexp for every t of one chunk + chunk product P_c ---------------------------------------
// One workgroup = 2 waves = the two 16-column halves of the matrices of chunk (b, c).
template <int NT>
__global__ void __launch_bounds__(64 * NT, (NT <= 2 ? 2 : 1)) k_mfma_expm_chunk(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[2][QNP * QLDR];
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;   // a finished seed keeps the results of its last evaluation (whole workgroup: no barrier yet)
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);     // identity: tile Ib == J, register r, lanes with dlt == 4r
    int flip = 0;
    CTile R[NT];
    colblock_identity<NT>(J, lane, R);
    for (int t = t0; t < t1; ++t) {
        // ---- A_t = (H0' + sum_k u_k H_k') / 2^s : left-operand fragments + this wave's column block -----------
        AFragT<NT> A;
        CTile P[NT];
        {
            afrag_load<NT, false>(mf.HfT, lane, A);
            colblock_load<NT>(mf.HfD, J, lane, P);
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int q = 0; q < QQS; ++q) { A.re[I][q] *= inv_scale; A.im[I][q] *= inv_scale; }
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib) { P[Ib].re *= inv_scale; P[Ib].im *= inv_scale; }
#pragma unroll 1
            for (int kk = 0; kk < d.k; ++kk) {
                const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
                const cplx* __restrict__ HT = mf.HfT + (size_t)(kk + 1) * QFR;
                const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
                for (int I = 0; I < NT; ++I)
#pragma unroll
                    for (int q = 0; q < QQS; ++q) {
                        const cplx h = HT[(I * QQS + q) * 64 + lane];
                        A.re[I][q] = fma(ck, h.x, A.re[I][q]);
                        A.im[I][q] = fma(ck, h.y, A.im[I][q]);
                    }
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                        P[Ib].re[r] = fma(ck, h.x, P[Ib].re[r]);
                        P[Ib].im[r] = fma(ck, h.y, P[Ib].im[r]);
                    }
            }
        }
        // ---- order-T Taylor polynomial sum_{j<=T} A^j/j! (tensorflow_state.py:37-41), Paterson-Stockmeyer form in
        //      A2 = A*A with blocks B_i = c_{2i} I + c_{2i+1} A:  S = B_m ; S = B_i + A2*S  -> 1 + ceil(T/2) - 1 products
        //      instead of T-1 (T=5: 3 instead of 4).  c_j = 1/j! from mf.invfact.
        if (d.T >= 2) {
            CTile AJ[NT];
            for (int Ib = 0; Ib < NT; ++Ib) AJ[Ib] = P[Ib];
            CTile A2J[NT];
            mm_colblock<NT>(A, AJ, A2J);
            lds_put_colblock<NT>(img[flip], 16 * J, lane, A2J);
            __syncthreads();
            lds_get_afrag<NT>(img[flip], lane, A);                      // A now holds the left-operand fragments of A2
            flip ^= 1;
            const int mm = d.T >> 1;
            int i;
            if ((d.T & 1) == 0) {                                   // top block is c_T I: S = B_{m-1} + c_T A2
                const double c0 = mf.invfact[2 * mm - 2], c1 = mf.invfact[2 * mm - 1], cT = mf.invfact[d.T];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + cT * A2J[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + cT * A2J[Ib].im[r];
                    }
                i = mm - 2;
            } else {                                                // S = B_m = c_{2m} I + c_{2m+1} A
                const double c0 = mf.invfact[2 * mm], c1 = mf.invfact[2 * mm + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r];
                    }
                i = mm - 1;
            }
            for (; i >= 0; --i) {
                CTile acc[NT];
                mm_colblock<NT>(A, P, acc);
                const double c0 = mf.invfact[2 * i], c1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + acc[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + acc[Ib].im[r];
                    }
            }
        } else {                                                    // T == 1: I + A
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[Ib].re[r] += (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
        }
        // ---- squaring: M <- M*M, s times (:43-44); the left operand comes back through the LDS image -----------
        for (int sq = 0; sq < d.s; ++sq) {
            lds_put_colblock<NT>(img[flip], 16 * J, lane, P);
            __syncthreads();
            lds_get_afrag<NT>(img[flip], lane, A);
            flip ^= 1;
            CTile acc[NT];
            mm_colblock<NT>(A, P, acc);
            for (int Ib = 0; Ib < NT; ++Ib) P[Ib] = acc[Ib];
        }
        // ---- K_t out (both operand forms); running chunk product R <- K_t R ----------------------------------------
        const size_t item = kitem(mf, d.steps, b, t);
        colblock_store<NT>(mf.KfD + item, J, lane, P);
        lds_put_colblock<NT>(img[flip], 16 * J, lane, P);
        __syncthreads();
        lds_get_afrag<NT>(img[flip], lane, A);
        flip ^= 1;
        if (mf.store_T) afrag_store_half<NT>(mf.KfT + item, J, lane, A);
        CTile acc[NT];
        mm_colblock<NT>(A, R, acc);
        for (int Ib = 0; Ib < NT; ++Ib) R[Ib] = acc[Ib];
    }
    const size_t pitem = (size_t)b * mf.C + c;
    colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R);
    lds_put_colblock<NT>(img[flip], 16 * J, lane, R);
    __syncthreads();
    AFragT<NT> A;
    lds_get_afrag<NT>(img[flip], lane, A);
    afrag_store_half<NT>(mf.PfT + pitem * QFR, J, lane, A);
}

// Two-wave variant on the 4x4x4 instruction (qoc_config.variant = 2): k_mfma_expm_chunk with every product done by mm_colblock4.  No A-operand fragments exist any
// more: A_t is assembled in strip layout only (half the Hamiltonian loads) and every left operand is an LDS image.
template <int NT>
__global__ void __launch_bounds__(64 * NT, (NT <= 2 ? 2 : 1)) k_mfma_expm_chunk4(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[2][QNP * QLDR];
    constexpr bool SUMS = NT == 3;     // NT = 4: 2 x 64 x 65 x 24 B do not fit the 160 KB; NT = 2: the extra 17 KB cost a workgroup per CU (1.28 vs 0.95 ms)
    __shared__ __attribute__((aligned(16))) double imgs[2][SUMS ? QNP * QLDR : 1];   // re + im of the image, summed once by its writer (a v_add_f64 per block in the product loop costs ~7 cycles of MFMA issue)
    const int lane = threadIdx.x & 63;
    const int J = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    int flip = 0;
    CTile R[NT];
    colblock_identity<NT>(J, lane, R);
    for (int t = t0; t < t1; ++t) {
        CTile P[NT];
        colblock_load<NT>(mf.HfD, J, lane, P);
#pragma unroll
        for (int Ib = 0; Ib < NT; ++Ib) { P[Ib].re *= inv_scale; P[Ib].im *= inv_scale; }
#pragma unroll 1
        for (int kk = 0; kk < d.k; ++kk) {
            const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
            const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                    P[Ib].re[r] = fma(ck, h.x, P[Ib].re[r]);
                    P[Ib].im[r] = fma(ck, h.y, P[Ib].im[r]);
                }
        }
        if (d.T >= 2) {
            CTile AJ[NT];
            for (int Ib = 0; Ib < NT; ++Ib) AJ[Ib] = P[Ib];
            lds_put_colblock_sum<NT, SUMS>(img[flip], imgs[flip], 16 * J, lane, AJ);
            __syncthreads();
            CTile A2J[NT];
            mm_colblock4<NT, SUMS>(img[flip], imgs[flip], lane, AJ, A2J);
            flip ^= 1;
            lds_put_colblock_sum<NT, SUMS>(img[flip], imgs[flip], 16 * J, lane, A2J);
            __syncthreads();
            const cplx* a2img = img[flip];                          // stays valid through the Horner steps (no put until then)
            const double* a2imgs = imgs[flip];
            flip ^= 1;
            const int mm = d.T >> 1;
            int i;
            if ((d.T & 1) == 0) {
                const double c0 = mf.invfact[2 * mm - 2], c1 = mf.invfact[2 * mm - 1], cT = mf.invfact[d.T];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + cT * A2J[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + cT * A2J[Ib].im[r];
                    }
                i = mm - 2;
            } else {
                const double c0 = mf.invfact[2 * mm], c1 = mf.invfact[2 * mm + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r];
                    }
                i = mm - 1;
            }
            for (; i >= 0; --i) {
                CTile acc[NT];
                mm_colblock4<NT, SUMS>(a2img, a2imgs, lane, P, acc);
                const double c0 = mf.invfact[2 * i], c1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[Ib].re[r] = one + c1 * AJ[Ib].re[r] + acc[Ib].re[r];
                        P[Ib].im[r] = c1 * AJ[Ib].im[r] + acc[Ib].im[r];
                    }
            }
        } else {
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) P[Ib].re[r] += (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
        }
        // img[flip] is the buffer the A image lived in: every wave passed the barrier after reading it
        for (int sq = 0; sq < d.s; ++sq) {
            lds_put_colblock_sum<NT, SUMS>(img[flip], imgs[flip], 16 * J, lane, P);
            __syncthreads();
            CTile acc[NT];
            mm_colblock4<NT, SUMS>(img[flip], imgs[flip], lane, P, acc);
            flip ^= 1;
            for (int Ib = 0; Ib < NT; ++Ib) P[Ib] = acc[Ib];
        }
        const size_t item = kitem(mf, d.steps, b, t);
        colblock_store<NT>(mf.KfD + item, J, lane, P);
        lds_put_colblock_sum<NT, SUMS>(img[flip], imgs[flip], 16 * J, lane, P);
        __syncthreads();
        if (mf.store_T) lds_store_fragT_half<NT>(img[flip], mf.KfT + item, J, lane);
        CTile acc[NT];
        mm_colblock4<NT, SUMS>(img[flip], imgs[flip], lane, R, acc);
        flip ^= 1;
        for (int Ib = 0; Ib < NT; ++Ib) R[Ib] = acc[Ib];
    }
    const size_t pitem = (size_t)b * mf.C + c;
    colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R);
    lds_put_colblock_sum<NT, SUMS>(img[flip], imgs[flip], 16 * J, lane, R);
    __syncthreads();
    lds_store_fragT_half<NT>(img[flip], mf.PfT + pitem * QFR, J, lane);
}

template <int NT>
__global__ void __launch_bounds__(64, 1) k_mfma_expm_chunk4w(QocDev d, QocMfma mf) {
    __shared__ __attribute__((aligned(16))) cplx img[QNP * QLDR];
    __shared__ __attribute__((aligned(16))) double imgs[QNP * QLDR];
    const int lane = threadIdx.x;
    const int b = blockIdx.x / mf.C, c = blockIdx.x - b * mf.C;
    if (d.skip_done && d.done[b]) return;
    const int t0 = c * mf.L, t1 = min(t0 + mf.L, d.steps);
    const double inv_scale = 1.0 / (double)(1 << d.s);
    const int dlt = (lane & 15) - (lane >> 4);
    CTile R[NT][NT];
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_identity<NT>(J, lane, R[J]);
    auto put_all = [&](const CTile (&m)[NT][NT]) {
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            lds_put_colblock<NT>(img, 16 * J, lane, m[J]);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    imgs[(16 * J + (lane & 15)) * QLDR + 16 * Ib + (lane >> 4) + 4 * r] = m[J][Ib].re[r] + m[J][Ib].im[r];
        }
        wave_lds_fence();
    };
    for (int t = t0; t < t1; ++t) {
        CTile P[NT][NT];
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            colblock_load<NT>(mf.HfD, J, lane, P[J]);
#pragma unroll
            for (int Ib = 0; Ib < NT; ++Ib) { P[J][Ib].re *= inv_scale; P[J][Ib].im *= inv_scale; }
        }
#pragma unroll 1
        for (int kk = 0; kk < d.k; ++kk) {
            const double ck = d.u[((size_t)b * d.k + kk) * d.steps + t] * inv_scale;
            const cplx* __restrict__ HD = mf.HfD + (size_t)(kk + 1) * QFR;
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const cplx h = HD[(J * QQS + 4 * Ib + r) * 64 + lane];
                        P[J][Ib].re[r] = fma(ck, h.x, P[J][Ib].re[r]);
                        P[J][Ib].im[r] = fma(ck, h.y, P[J][Ib].im[r]);
                    }
        }
        if (d.T >= 2) {
            CTile AJ[NT][NT], A2J[NT][NT];
#pragma unroll
            for (int J = 0; J < NT; ++J)
                for (int Ib = 0; Ib < NT; ++Ib) AJ[J][Ib] = P[J][Ib];
            put_all(AJ);
            mm_full4<NT>(img, imgs, lane, AJ, A2J);
            wave_lds_fence();
            put_all(A2J);
            const int mm = d.T >> 1;
            int i;
            double c0, c1, cT = 0.0;
            if ((d.T & 1) == 0) { c0 = mf.invfact[2 * mm - 2]; c1 = mf.invfact[2 * mm - 1]; cT = mf.invfact[d.T]; i = mm - 2; }
            else { c0 = mf.invfact[2 * mm]; c1 = mf.invfact[2 * mm + 1]; i = mm - 1; }
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double one = (Ib == J && dlt == 4 * r) ? c0 : 0.0;
                        P[J][Ib].re[r] = one + c1 * AJ[J][Ib].re[r] + cT * A2J[J][Ib].re[r];
                        P[J][Ib].im[r] = c1 * AJ[J][Ib].im[r] + cT * A2J[J][Ib].im[r];
                    }
            for (; i >= 0; --i) {
                CTile acc[NT][NT];
                mm_full4<NT>(img, imgs, lane, P, acc);
                const double d0 = mf.invfact[2 * i], d1 = mf.invfact[2 * i + 1];
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double one = (Ib == J && dlt == 4 * r) ? d0 : 0.0;
                            P[J][Ib].re[r] = one + d1 * AJ[J][Ib].re[r] + acc[J][Ib].re[r];
                            P[J][Ib].im[r] = d1 * AJ[J][Ib].im[r] + acc[J][Ib].im[r];
                        }
            }
            wave_lds_fence();
        } else {
#pragma unroll
            for (int J = 0; J < NT; ++J)
#pragma unroll
                for (int Ib = 0; Ib < NT; ++Ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) P[J][Ib].re[r] += (Ib == J && dlt == 4 * r) ? 1.0 : 0.0;
        }
        for (int sq = 0; sq < d.s; ++sq) {
            put_all(P);
            CTile acc[NT][NT];
            mm_full4<NT>(img, imgs, lane, P, acc);
            wave_lds_fence();
#pragma unroll
            for (int J = 0; J < NT; ++J)
                for (int Ib = 0; Ib < NT; ++Ib) P[J][Ib] = acc[J][Ib];
        }
        const size_t item = kitem(mf, d.steps, b, t);
#pragma unroll
        for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.KfD + item, J, lane, P[J]);
        put_all(P);
        if (mf.store_T) {
#pragma unroll
            for (int J = 0; J < NT; ++J) lds_store_fragT_half<NT>(img, mf.KfT + item, J, lane);
        }
        CTile acc[NT][NT];
        mm_full4<NT>(img, imgs, lane, R, acc);
        wave_lds_fence();
#pragma unroll
        for (int J = 0; J < NT; ++J)
            for (int Ib = 0; Ib < NT; ++Ib) R[J][Ib] = acc[J][Ib];
    }
    const size_t pitem = (size_t)b * mf.C + c;
#pragma unroll
    for (int J = 0; J < NT; ++J) colblock_store<NT>(mf.PfD + pitem * QFR, J, lane, R[J]);
    put_all(R);
#pragma unroll
    for (int J = 0; J < NT; ++J) lds_store_fragT_half<NT>(img, mf.PfT + pitem * QFR, J, lane);
}

