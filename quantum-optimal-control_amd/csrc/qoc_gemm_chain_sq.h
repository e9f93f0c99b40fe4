// qoc_gemm_chain_sq.h -- the direct state-transfer chain on the generator AND its square (round 5): k_gemm_taylor_chain_sq.
// Reference semantics: core/tensorflow_state.py:88-96 (psi <- sum_{j<T} B^j psi / j!), :118-131 (lambda <- sum_{j<T} (-B)^j lambda / j!).
//
// k_gemm_taylor_chain_dpp walks the reference's own recursion: T - 1 DEPENDENT 64 x 64 mat-vecs per slice, each ~280 ticks of DPP FMAs plus a
// ~360-tick four-wave LDS exchange that nothing can hide (tools/chain_micro_probe.hip: write + barrier + one read alone are 243 ticks).  One chain
// per CU leaves nothing to overlap the exchange with, so the only way down is FEWER dependent mat-vecs:
//     P(B) x = sum_i (B^2)^i (c_{2i} x + c_{2i+1} B x),      c_j = 1 / j!  (j < T)
// i.e. v = B x, then Horner over B^2 on ONE vector: r = u_I; r <- B^2 r + u_i (i = I - 1 .. 0), u_i = c_{2i} x + c_{2i+1} v, I = ceil(T / 2) - 1:
// 1 + I dependent mat-vecs instead of T - 1 (T = 10: 5 instead of 9).  B^2 is not squared per slice (64000 matrix products at C3 x 64): with
// B = A_0 + sum_k u_k A_k it is the quadratic form  B^2 = M_0 + sum_k u_k M_k + sum_{k<=l} u_k u_l M_kl  in (k + 1)(k + 2) / 2 constant matrices
// (M_0 = A_0^2, M_k = A_0 A_k + A_k A_0, M_kl = A_k A_l + A_l A_k, M_kk = A_k^2) that k_gemm_assemble_sq combines exactly like the generator
// itself.  Anti-Hermitian generators only (the packed image of qoc_gemm_chain_dpp.h): B^2 is then Hermitian and is packed the same way, the
// mirrored blocks flip the sign of the IMAGINARY part.  A slice is [B packed | B^2 packed] = 5120 entries (80 KB).
// The backward chain needs no second kernel body: (-B)^2 = B^2, so only the odd coefficients change sign.
// The square costs HBM bytes (80 instead of 40 KB per slice, written once, read twice), so the scheme pays while the chains are latency-bound:
// up to ~128 control sets (QocGemm::sq_chain); larger batches stay on k_gemm_taylor_chain_dpp<packed>, which is bound by the generator traffic.
#pragma once
// (included by qoc_gemm_chains.h after qoc_gemm_chain_dpp.h)

// The generator itself is used once per slice: four-multiplication form without a sum plane, as four plain sums s1 = sum ar xr, s2 = sum ai xi,
// s3 = sum ar xi, s4 = sum ai xr.  A lane of a MIRRORED block holds conj(B[c][r]) where it needs -conj: the sign of its real parts rides on the two
// closing operations (re = sg s1 - s2, im = sg s3 + s4, sg = -1 on mirrored lanes) instead of on 16 sign flips per slice.
#define QOC_SQ_CMAC4(J) \
    asm volatile("v_fmac_f64_dpp %0, %4, %6 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, %5, %7 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %2, %5, %6 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %3, %4, %7 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4) : "v"(xr), "v"(xi), "v"(a[J].x), "v"(a[J].y))
__device__ __forceinline__ void dpp_matvec16_plain(const cplx (&a)[16], double xr, double xi, double sg, double& re, double& im) {
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
    asm volatile("s_nop 1" : "+v"(xr), "+v"(xi));          // VALU write of the entry -> DPP read: two wait states
    QOC_SQ_CMAC4(0); QOC_SQ_CMAC4(1); QOC_SQ_CMAC4(2); QOC_SQ_CMAC4(3); QOC_SQ_CMAC4(4); QOC_SQ_CMAC4(5); QOC_SQ_CMAC4(6); QOC_SQ_CMAC4(7);
    QOC_SQ_CMAC4(8); QOC_SQ_CMAC4(9); QOC_SQ_CMAC4(10); QOC_SQ_CMAC4(11); QOC_SQ_CMAC4(12); QOC_SQ_CMAC4(13); QOC_SQ_CMAC4(14); QOC_SQ_CMAC4(15);
    re = fma(sg, s1, -s2); im = fma(sg, s3, s4);
}
// The square in Gauss's three-multiplication form (qoc_gemm_chain_dpp.h): k1 += (ar + ai) xr, k2 += ar (xi - xr), k3 += ai (xr + xi).  A mirrored lane holds
// B2[c][r] where it needs the conjugate: its sum plane is ar - ai (formed once per slice) and k3 enters with the other sign -- re = k1 - sg k3, im = k1 + k2.
#define QOC_SQ_CMAC3(J) \
    asm volatile("v_fmac_f64_dpp %0, %3, %6 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, %4, %7 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %2, %5, %8 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(k1), "+v"(k2), "+v"(k3) : "v"(xr), "v"(xd), "v"(xs), "v"(sa[J]), "v"(a[J].x), "v"(a[J].y))
__device__ __forceinline__ void dpp_matvec16_square(const cplx (&a)[16], const double (&sa)[16], double xr, double xi, double sg, double& re, double& im) {
    double k1 = 0.0, k2 = 0.0, k3 = 0.0;
    double xd = xi - xr, xs = xr + xi;
    asm volatile("s_nop 1" : "+v"(xr), "+v"(xd), "+v"(xs));
    QOC_SQ_CMAC3(0); QOC_SQ_CMAC3(1); QOC_SQ_CMAC3(2); QOC_SQ_CMAC3(3); QOC_SQ_CMAC3(4); QOC_SQ_CMAC3(5); QOC_SQ_CMAC3(6); QOC_SQ_CMAC3(7);
    QOC_SQ_CMAC3(8); QOC_SQ_CMAC3(9); QOC_SQ_CMAC3(10); QOC_SQ_CMAC3(11); QOC_SQ_CMAC3(12); QOC_SQ_CMAC3(13); QOC_SQ_CMAC3(14); QOC_SQ_CMAC3(15);
    re = fma(-sg, k3, k1); im = k1 + k2;
}

// the generator / square loads of the next slice that term q of a slice with I Horner steps issues: 8 per term while terms are left, the rest in the last
__host__ __device__ constexpr int qoc_sq_loads_before(int q, int I) { return q <= I ? (8 * q < 32 ? 8 * q : 32) : 32; }

// I = ceil(T / 2) - 1 Horner steps over B^2 (compile time: the schedule of the prefetch loads and hipcc's s_waitcnt vmcnt are then exact)
template <int I>
__device__ __forceinline__ void taylor_chain_sq_body(const ChainArgs& a, int b, cplx (*part)[4][64], const double* tinv) {
    constexpr int GE = QOC_DPP_PK_ELEMS;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int idx = 16 * w + (l & 15);                       // the vector entry this lane carries (replicated over the four rows of 16 lanes)
    const bool owner = l < 16;
    const int rb = l >> 4, li = l & 15;
    const bool mirrored = rb < w;                            // packed image: see qoc_gemm_chain_dpp.h
    const unsigned koff = mirrored ? (unsigned)((w * (w + 1) / 2 + rb) * 256 + li * 16) : (unsigned)((rb * (rb + 1) / 2 + w) * 256 + li);
    const unsigned kstr = mirrored ? 1u : 16u;
    const double sg = mirrored ? -1.0 : 1.0;                 // B: -conj (real part), B^2: conj (imaginary part) -- see the two mat-vec forms above
    auto kidx = [&](int c) -> unsigned { return (koff + (unsigned)c * kstr) & 4095u; };
    const cplx* Kp = a.K + b * a.sKb;
    const cplx* Ep = a.E + b * a.sEb + (size_t)idx * (a.ldE > 0 ? a.ldE : QOC_TW);
    cplx* Op = a.Out + b * a.sOb + (size_t)idx * a.ldO;
    cplx* O2 = a.Out2 ? a.Out2 + b * a.sO2b + idx : nullptr;
    const bool owner2 = owner && a.Out2 && idx < a.n2;
    cplx xv = cmake(0.0, 0.0);
    if (a.X0) xv = a.X0[b * a.sXb + (size_t)idx * QOC_TW];
    if (a.store_initial && owner) *(Op - a.sOs) = xv;
    const int last = a.len - 1;
    // c_{2i}, +-c_{2i+1} (zero beyond the series; the backward chain applies P(-B): the odd coefficients change sign), uniform: scalar registers
    double ce[I + 1], co[I + 1];
#pragma unroll
    for (int i = 0; i <= I; ++i) {
        const double ve = tinv[2 * i], vo = 2 * i + 1 < a.nterms ? (a.sign < 0.0 ? -tinv[2 * i + 1] : tinv[2 * i + 1]) : 0.0;
        ce[i] = __hiloint2double((int)__builtin_amdgcn_readfirstlane((unsigned)__double2hiint(ve)), (int)__builtin_amdgcn_readfirstlane((unsigned)__double2loint(ve)));
        co[i] = __hiloint2double((int)__builtin_amdgcn_readfirstlane((unsigned)__double2hiint(vo)), (int)__builtin_amdgcn_readfirstlane((unsigned)__double2loint(vo)));
    }
    cplx* const my_part = &part[0][w][l];
    const cplx* const rd_part = &part[0][0][idx];
    int cur = 0;
    // the four waves' partial sums of a row meet in LDS (one write, one barrier, four reads; the two buffers alternate)
    auto exchange = [&](double pr, double pi, double& xr, double& xi) {
        my_part[cur * 256] = cmake(pr, pi);
        lds_barrier();
        const cplx s0 = rd_part[cur * 256], s1 = rd_part[cur * 256 + 64], s2 = rd_part[cur * 256 + 128], s3 = rd_part[cur * 256 + 192];
        xr = (s0.x + s1.x) + (s2.x + s3.x);
        xi = (s0.y + s1.y) + (s2.y + s3.y);
        cur ^= 1;
    };
    // slice j on (X, S) = (B_j, B_j^2) with addend e; meanwhile (B, B^2, e) of slice jn go into (Xn, Sn, en)
    auto step = [&](int j, const cplx (&X)[16], const cplx (&S)[16], const cplx& e, cplx (&Xn)[16], cplx (&Sn)[16], cplx& en, int jn) {
        const int jc = min(jn, last);
        const cplx* kj = Kp + (long long)jc * a.sKs;
        const cplx* kj2 = kj + GE;                                   // (its own scalar base: both runs of loads keep the saddr form)
        en = Ep[(long long)jc * a.sEs];
        auto prefetch = [&](int q) {                                 // the loads of term q (compile-time bounds)
#pragma unroll
            for (int g = qoc_sq_loads_before(q, I); g < qoc_sq_loads_before(q + 1, I); ++g) {
                if (g < 16) Xn[g] = kj[kidx(g)]; else Sn[g - 16] = kj2[kidx(g - 16)];
            }
        };
        prefetch(0);
        double pr, pi, vr, vi;
        dpp_matvec16_plain(X, xv.x, xv.y, sg, pr, pi);               // v = B x
        exchange(pr, pi, vr, vi);
        double sS[16];                                               // (three-multiplication form) re + im of the square's entries
#pragma unroll
        for (int c = 0; c < 16; ++c) sS[c] = fma(sg, S[c].y, S[c].x);
        double rr = fma(co[I], vr, ce[I] * xv.x), ri = fma(co[I], vi, ce[I] * xv.y);    // u_I
#pragma unroll
        for (int i = I - 1; i >= 0; --i) {                           // r <- B^2 r + u_i
            prefetch(I - i);
            dpp_matvec16_square(S, sS, rr, ri, sg, pr, pi);
            double tr, ti;
            exchange(pr, pi, tr, ti);
            rr = tr + fma(co[i], vr, ce[i] * xv.x);
            ri = ti + fma(co[i], vi, ce[i] * xv.y);
        }
        if constexpr (I == 0) prefetch(1);
        xv = cmake(rr + e.x, ri + e.y);
        if (owner) Op[(long long)j * a.sOs] = xv;
        if (owner2) O2[(long long)j * a.sO2s] = xv;
    };
    if (a.len > 0) {
        cplx X0[16], S0[16], X1[16], S1[16], e0, e1;
        {
            const cplx* kj = Kp;
#pragma unroll
            for (int c = 0; c < 16; ++c) { X0[c] = kj[kidx(c)]; S0[c] = kj[GE + kidx(c)]; }
            e0 = Ep[0];
        }
        int j = 0;
        for (; j + 2 <= a.len; j += 2) {
            step(j, X0, S0, e0, X1, S1, e1, j + 1);
            step(j + 1, X1, S1, e1, X0, S0, e0, j + 2);
        }
        if (j < a.len) step(j, X0, S0, e0, X1, S1, e1, j + 1);
    }
    if (a.Fin && owner) a.Fin[b * a.sFb + (size_t)idx * QOC_TW] = xv;
}

// Two argument sets in one launch, as k_gemm_taylor_chain_dpp.  I = ceil(nterms / 2) - 1 in 1 .. 6 (3 <= nterms <= 14; the host checks).
__global__ void __launch_bounds__(256) k_gemm_taylor_chain_sq(ChainArgs a0, ChainArgs a1, int nb0) {
    __shared__ __attribute__((aligned(16))) cplx part[2][4][64];
    __shared__ double tinv[64];
    const bool second = (int)blockIdx.x >= nb0;
    const ChainArgs a = second ? a1 : a0;
    const int b = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    if (threadIdx.x < 64) {
        double fact = 1.0;
        for (int ii = 2; ii <= (int)threadIdx.x; ++ii) fact *= (double)ii;   // the running factorial of :92-95
        tinv[threadIdx.x] = 1.0 / fact;
    }
    lds_barrier();
    switch ((a.nterms + 1) / 2 - 1) {
        case 1: taylor_chain_sq_body<1>(a, b, part, tinv); break;
        case 2: taylor_chain_sq_body<2>(a, b, part, tinv); break;
        case 3: taylor_chain_sq_body<3>(a, b, part, tinv); break;
        case 4: taylor_chain_sq_body<4>(a, b, part, tinv); break;
        case 5: taylor_chain_sq_body<5>(a, b, part, tinv); break;
        default: taylor_chain_sq_body<6>(a, b, part, tinv); break;
    }
}
static inline bool qoc_sq_chain_terms_ok(int nterms) { return nterms >= 3 && nterms <= 14; }
