// qoc_gemm_chain_dpp.h -- the direct state-transfer chain at N = 64, one state vector (BASELINE config 3): k_gemm_taylor_chain_dpp.
// Reference semantics: core/tensorflow_state.py:88-96 (forward recursion psi_n = B psi_n, psi += psi_n / n!), :118-131 (backward, -B).
//
// One workgroup of four waves per seed walks all slices; a step is T - 1 DEPENDENT 64 x 64 complex mat-vecs, so what counts is the latency of
// one mat-vec, not its throughput.  k_gemm_taylor_chain spends ~1100 cycles on one: eight ds_read_b128 of the vector per thread, 64 FMAs, a
// 26-instruction DPP butterfly over 8 lanes, LDS write, barrier.  Here the generator arrives TRANSPOSED (column c = 1 KB contiguous:
// k_gemm_assemble_rows on the transposed Hamiltonian stack, QocGemm::HsPT), wave w owns the columns 16 w .. 16 w + 15 and lane = row: the 16
// vector entries of the wave live one per lane inside every row of 16 lanes and reach the FMAs through the double-precision DPP control
// row_newbcast:j (v_fmac_f64_dpp, gfx90a+): a complex MAC is four VOP2 instructions, no LDS read of the vector, no butterfly.  What is left
// of the exchange: the four waves' partial sums of a row meet in LDS -- one ds_write_b128 per lane, one barrier, four ds_read_b128, three
// complex adds.  The sign of (sign * B) rides on the source modifiers of the DPP operand.  Measured with tools/taylor_dpp_probe.hip:
// 0.28 us per mat-vec against 0.46.
// The next-but-one generator (64 KB) is fetched two loads per Taylor term instead of as one burst of 16 per wave: the burst stalled the
// four waves ~1500 cycles per slice on the vector-memory issue queue (probe: "between inner loops").
// Round 5 -- PACKED generators (PK): with Hermitian Hamiltonians every generator B = -i dt H is anti-Hermitian, B[r][c] = -conj(B[c][r]), so only
// the 16 x 16 blocks on and below the block diagonal are stored: 10 blocks of 256 entries = 2560 entries (40 KB) per slice instead of 4096 (64 KB),
// block (R, C), R >= C, at (R (R + 1) / 2 + C) * 256, column-major inside the block.  Wave w (block column w) loads the lanes of the row blocks
// R >= w as before (column j of the block: 16 lanes = 256 B contiguous) and the lanes of the row blocks R < w from the MIRRORED block (w, R): lane i
// reads the 16 consecutive entries i * 16 + j of that block and flips the sign of the real part.  The mirrored reads hit lines wave R of the same
// workgroup fetches for the same slice (L2), so the chain's HBM reads and the assembly's HBM writes both drop to 5/8: C3 x 64 moves 7.9 instead of
// 12.6 GB per iteration, x 256 31.5 instead of 50.4.
#pragma once
// (included by qoc_gemm_chains.h after ChainArgs and before the launchers)
#define QOC_DPP_PK_ELEMS 2560     // entries of one packed generator

#ifndef QOC_DPP_GAUSS
#define QOC_DPP_GAUSS 1         // 1: complex MAC in the three-multiplication form (48 instead of 64 DPP FMAs per mat-vec); 0: the four-multiplication form
#endif
#if QOC_DPP_GAUSS
// Gauss's form with the sums on the operands that are cheap to combine: k1 += (ar + ai) xr, k2 += ar (xi - xr), k3 += ai (xr + xi);
// re = k1 - k3, im = k1 + k2.  The matrix side costs one add per entry and SLICE (the plane sa = ar + ai of the current generator, formed when
// the slice starts: 16 adds against 9 x 16 FMAs saved), the vector side two adds per term; three independent accumulator chains.
#define QOC_DPP_CMAC_(NS, J) \
    asm volatile("v_fmac_f64_dpp %0, " NS "%3, %6 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, " NS "%4, %7 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %2, " NS "%5, %8 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(k1), "+v"(k2), "+v"(k3) : "v"(xr), "v"(xd), "v"(xs), "v"(sa[J]), "v"(a[J].x), "v"(a[J].y))
#define QOC_DPP_CMAC(J) do { if constexpr (NEG) QOC_DPP_CMAC_("-", J); else QOC_DPP_CMAC_("", J); } while (0)

// partial sums of this wave's 16 columns: lane = row; (xr, xi) = the vector entry 16 w + (lane & 15); sa[J] = a[J].x + a[J].y
// CW = the columns this wave multiplies (16, or fewer on the active columns of a padded problem: 10 / 12 / 14)
template <bool NEG, int CW = 16>
__device__ __forceinline__ void dpp_matvec16(const cplx (&a)[CW], const double (&sa)[CW], double xr, double xi, double& re, double& im) {
    double k1 = 0.0, k2 = 0.0, k3 = 0.0;
    double xd = xi - xr, xs = xr + xi;
    asm volatile("s_nop 1" : "+v"(xr), "+v"(xd), "+v"(xs));          // VALU write of the entry -> DPP read: two wait states
    QOC_DPP_CMAC(0); QOC_DPP_CMAC(1); QOC_DPP_CMAC(2); QOC_DPP_CMAC(3); QOC_DPP_CMAC(4); QOC_DPP_CMAC(5); QOC_DPP_CMAC(6); QOC_DPP_CMAC(7);
    QOC_DPP_CMAC(8); QOC_DPP_CMAC(9);
    if constexpr (CW > 10) { QOC_DPP_CMAC(10); QOC_DPP_CMAC(11); }
    if constexpr (CW > 12) { QOC_DPP_CMAC(12); QOC_DPP_CMAC(13); }
    if constexpr (CW > 14) { QOC_DPP_CMAC(14); QOC_DPP_CMAC(15); }
    re = k1 - k3; im = k1 + k2;
}
#else
#define QOC_DPP_CMAC_(NS0, NS1, NS2, NS3, J) \
    asm volatile("v_fmac_f64_dpp %0, " NS0 "%2, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, " NS1 "%3, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %0, " NS2 "%3, %5 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t" \
                 "v_fmac_f64_dpp %1, " NS3 "%2, %5 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" \
                 : "+v"(re), "+v"(im) : "v"(xr), "v"(xi), "v"(a[J].x), "v"(a[J].y))
// (re, im) += +-(a[J] * x_J):  re += ar xr - ai xi,  im += ar xi + ai xr
#define QOC_DPP_CMAC(J) do { if constexpr (NEG) QOC_DPP_CMAC_("-", "-", "", "-", J); else QOC_DPP_CMAC_("", "", "-", "", J); } while (0)

// partial sums of this wave's 16 columns: lane = row; (xr, xi) = the vector entry 16 w + (lane & 15)
template <bool NEG, int CW = 16>
__device__ __forceinline__ void dpp_matvec16(const cplx (&a)[CW], const double (&)[CW], double xr, double xi, double& re, double& im) {
    re = 0.0; im = 0.0;
    asm volatile("s_nop 1" : "+v"(xr), "+v"(xi));          // VALU write of the entry -> DPP read: two wait states
    QOC_DPP_CMAC(0); QOC_DPP_CMAC(1); QOC_DPP_CMAC(2); QOC_DPP_CMAC(3); QOC_DPP_CMAC(4); QOC_DPP_CMAC(5); QOC_DPP_CMAC(6); QOC_DPP_CMAC(7);
    QOC_DPP_CMAC(8); QOC_DPP_CMAC(9);
    if constexpr (CW > 10) { QOC_DPP_CMAC(10); QOC_DPP_CMAC(11); }
    if constexpr (CW > 12) { QOC_DPP_CMAC(12); QOC_DPP_CMAC(13); }
    if constexpr (CW > 14) { QOC_DPP_CMAC(14); QOC_DPP_CMAC(15); }
}

#endif

// TS: the number of Taylor terms when it is known at compile time (10: C3 and the reference's examples; the guards of the unrolled terms and the LDS
// reads of 1 / ii! go away), 0 = a.nterms (any), -1 = a.nterms >= 9 (the eight unrolled terms all run: no guards between them either)
#ifndef QOC_DPP_NO_STATIC
#define QOC_DPP_NO_STATIC 0     // 1 (A/B builds): ten terms through the a.nterms >= 9 instance
#endif
// CW (round 5): columns per wave.  16 = the whole padded matrix; 10 / 12 / 14 = a problem of at most 4 CW levels padded to 64 -- wave w owns the columns
// CW w .. CW w + CW - 1 (the columns from n on are zero and the last waves simply have fewer non-zero ones), the lanes (lane & 15) < CW of every row of 16
// lanes carry the vector entries of those columns, and a slice's generator is its first 4 CW columns: 256 CW entries (full column-major image only).
template <bool NEG, int TS, bool PK, int CW = 16>
__device__ __forceinline__ void taylor_chain_dpp_body(const ChainArgs& a, int b, cplx (*part)[4][64], const double* tinv) {
    static_assert(CW == 16 || !PK, "the packed image has 16-column blocks");
    static_assert(CW >= 10 && CW <= 16 && CW % 2 == 0, "columns per wave: 10, 12, 14 or 16");
    constexpr int N = 64;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int idx = min(CW * w + (l & 15), N - 1);           // the vector entry this lane carries (replicated over the four rows of 16 lanes; lanes beyond CW: unused)
    const bool owner = l < CW;
    // scalar base + 32-bit lane offset: the loads take the saddr form, no 64-bit VALU address arithmetic between the FMAs of the chain
    const cplx* Kp = a.K + b * a.sKb;
    // full image: column 16 w + c of the transposed generator = + c * N; packed image: see the header (row block R = l >> 4 of block column w)
    const int rb = l >> 4, li = l & 15;
    const bool mirrored = PK && rb < w;
    const unsigned koff = !PK ? (unsigned)(CW * w) * N + l
                          : (mirrored ? (unsigned)((w * (w + 1) / 2 + rb) * 256 + li * 16) : (unsigned)((rb * (rb + 1) / 2 + w) * 256 + li));
    const unsigned kstr = !PK ? (unsigned)N : (mirrored ? 1u : 16u);
    // entry c of this lane's run: the mask tells the compiler that the byte offset fits the 32-bit lane offset of the saddr load form
    auto kidx = [&](int c) -> unsigned { return PK ? ((koff + (unsigned)c * kstr) & 4095u) : koff + (unsigned)c * kstr; };
    const int flip = mirrored ? (int)0x80000000 : 0;                        // -conj: the sign bit of the real part
    const cplx* Ep = a.E + b * a.sEb + (size_t)idx * (a.ldE > 0 ? a.ldE : QOC_TW);
    cplx* Op = a.Out + b * a.sOb + (size_t)idx * a.ldO;
    cplx* O2 = a.Out2 ? a.Out2 + b * a.sO2b + idx : nullptr;
    const bool owner2 = owner && a.Out2 && idx < a.n2;
    cplx xv = cmake(0.0, 0.0);
    if (a.X0) xv = a.X0[b * a.sXb + (size_t)idx * QOC_TW];
    if (a.store_initial && owner) *(Op - a.sOs) = xv;
    const int last = a.len - 1;
    const int nterms = TS > 0 ? TS : a.nterms;
    double inv_s[9];                                            // 1 / ii! of the unrolled terms, uniform: scalar registers
#pragma unroll
    for (int ii = 1; ii <= 8; ++ii) {
        const double v = tinv[ii];
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)__double2loint(v)), hi = __builtin_amdgcn_readfirstlane((unsigned)__double2hiint(v));
        inv_s[ii] = __hiloint2double((int)hi, (int)lo);
    }
    cplx* const my_part = &part[0][w][l];
    const cplx* const rd_part = &part[0][0][idx];
    int cur = 0;
    // one Taylor term: v <- (sign B) v, out += v / ii!
    auto term = [&](const cplx (&k)[CW], const double (&sk)[CW], int ii, double inv, double& outr, double& outi) {
        double pr, pi;
        dpp_matvec16<NEG, CW>(k, sk, xv.x, xv.y, pr, pi);
        my_part[cur * 256] = cmake(pr, pi);
        lds_barrier();
        const cplx s0 = rd_part[cur * 256], s1 = rd_part[cur * 256 + 64], s2 = rd_part[cur * 256 + 128], s3 = rd_part[cur * 256 + 192];
        xv.x = (s0.x + s1.x) + (s2.x + s3.x);
        xv.y = (s0.y + s1.y) + (s2.y + s3.y);
        outr = fma(xv.x, inv, outr); outi = fma(xv.y, inv, outi);      // psi += psi_n / factorial     :95 / :131
        cur ^= 1;                                                         // the next term writes the other buffer: one barrier per term
    };
    // slice j on the generator k with addend e; meanwhile the generator / addend of slice jn go into kn / en, two columns per term
    auto step = [&](int j, cplx (&k)[CW], const cplx& e, cplx (&kn)[CW], cplx& en, int jn) {
        const int jc = min(jn, last);
        const cplx* kj = Kp + (long long)jc * a.sKs;
        en = Ep[(long long)jc * a.sEs];
        double outr = xv.x, outi = xv.y;
        double sk[CW];                                              // (three-multiplication form) re + im of the slice's generator entries
        if constexpr (PK) {
#pragma unroll
            for (int c = 0; c < CW; ++c) k[c].x = __hiloint2double(__double2hiint(k[c].x) ^ flip, __double2loint(k[c].x));
        }
#pragma unroll
        for (int c = 0; c < CW; ++c) sk[c] = QOC_DPP_GAUSS ? k[c].x + k[c].y : 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (2 * g < CW) { kn[2 * g] = kj[kidx(2 * g)]; kn[2 * g + 1] = kj[kidx(2 * g + 1)]; }
            if (TS != 0 || g + 1 < nterms) term(k, sk, g + 1, inv_s[g + 1], outr, outi);
        }
        for (int ii = 9; ii < nterms; ++ii) {
            double inv = tinv[ii & 63];
            if (ii >= 64) { double fact = 1.0; for (int q = 2; q <= ii; ++q) fact *= (double)q; inv = 1.0 / fact; }
            term(k, sk, ii, inv, outr, outi);
        }
        xv = cmake(outr + e.x, outi + e.y);
        if (owner) Op[(long long)j * a.sOs] = xv;
        if (owner2) O2[(long long)j * a.sO2s] = xv;
    };
    if (a.len > 0) {
        cplx k0[CW], k1[CW], k2[CW], e0, e1, e2;
        {
            const cplx* kj = Kp;
#pragma unroll
            for (int c = 0; c < CW; ++c) k0[c] = kj[kidx(c)];
            e0 = Ep[0];
            const int jc = min(1, last);
            kj = Kp + (long long)jc * a.sKs;
#pragma unroll
            for (int c = 0; c < CW; ++c) k1[c] = kj[kidx(c)];
            e1 = Ep[(long long)jc * a.sEs];
        }
        int j = 0;
        for (; j + 3 <= a.len; j += 3) {
            step(j, k0, e0, k2, e2, j + 2);
            step(j + 1, k1, e1, k0, e0, j + 3);
            step(j + 2, k2, e2, k1, e1, j + 4);
        }
        if (j < a.len) step(j, k0, e0, k2, e2, j + 2);
        if (j + 1 < a.len) step(j + 1, k1, e1, k0, e0, j + 3);
    }
    if (a.Fin && owner) a.Fin[b * a.sFb + (size_t)idx * QOC_TW] = xv;
}

// Two argument sets in one launch, as k_gemm_taylor_chain: workgroups 0 .. nb0 - 1 run a0, the rest a1 (forward and z-free backward chain
// side by side when there is no state regulariser)
template <bool PK, int CW = 16>
__global__ void __launch_bounds__(256) k_gemm_taylor_chain_dpp(ChainArgs a0, ChainArgs a1, int nb0) {
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
    // (the active-column instances CW < 16 have no body of their own for ten terms: the >= 9 one takes them)
    if (a.nterms == 10 && !QOC_DPP_NO_STATIC && CW == 16) { if (a.sign < 0.0) taylor_chain_dpp_body<true, 10, PK, CW>(a, b, part, tinv); else taylor_chain_dpp_body<false, 10, PK, CW>(a, b, part, tinv); }
    else if (a.nterms >= 9) { if (a.sign < 0.0) taylor_chain_dpp_body<true, -1, PK, CW>(a, b, part, tinv); else taylor_chain_dpp_body<false, -1, PK, CW>(a, b, part, tinv); }
    else if (a.sign < 0.0) taylor_chain_dpp_body<true, 0, PK, CW>(a, b, part, tinv); else taylor_chain_dpp_body<false, 0, PK, CW>(a, b, part, tinv);
}
