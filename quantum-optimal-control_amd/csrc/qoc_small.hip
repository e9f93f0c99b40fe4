// qoc_small.hip -- host side of QOC_PATH_SMALL: which instance of k_small_iter runs a problem, its exchange buffers, the launch.
#include "qoc_small_kernel.h"

#include <cmath>
#include <cstdio>

#include "qoc_plan_limits.h"
#include "qoc_small_instances.h"

// (instantiated in qoc_small_a / _b / _c.hip)
#define QOC_SMALL_DECL(N, L, R, S) \
    extern template __global__ void qsm::k_small_iter<N, L, R, false, true>(QocDev, QocAdamDev, QocSmallDev); \
    extern template __global__ void qsm::k_small_iter<N, L, R, true, true>(QocDev, QocAdamDev, QocSmallDev);
#define QOC_SMALL_DECL1(N, L, R, S) \
    extern template __global__ void qsm::k_small_iter<N, L, R, false, false>(QocDev, QocAdamDev, QocSmallDev); \
    extern template __global__ void qsm::k_small_iter<N, L, R, true, false>(QocDev, QocAdamDev, QocSmallDev);
QOC_SMALL_INSTANCES_A(QOC_SMALL_DECL) QOC_SMALL_INSTANCES_B(QOC_SMALL_DECL) QOC_SMALL_INSTANCES_C(QOC_SMALL_DECL)
QOC_SMALL_INSTANCES_A1(QOC_SMALL_DECL1) QOC_SMALL_INSTANCES_B1(QOC_SMALL_DECL1)

namespace {

typedef void (*small_kernel_t)(QocDev, QocAdamDev, QocSmallDev);

struct Instance { int N, L, R; small_kernel_t fn[2]; bool ok[2]; bool lds_opted[2]; bool single; };   // ok: [without, with] a state regulariser (instances that spill are out); single: a build for one workgroup per control set

#define QOC_SMALL_ROW(N, L, R, S) { N, L, R, { qsm::k_small_iter<N, L, R, false, true>, S ? qsm::k_small_iter<N, L, R, true, true> : (small_kernel_t) nullptr }, { true, S != 0 }, { false, false }, false },
// (one-workgroup builds: a row of their own -- `single` -- that choose() takes for G = 1 only)
#define QOC_SMALL_ROW1(N, L, R, S) { N, L, R, { qsm::k_small_iter<N, L, R, false, false>, S ? qsm::k_small_iter<N, L, R, true, false> : (small_kernel_t) nullptr }, { true, S != 0 }, { false, false }, true },
Instance g_inst[] = { QOC_SMALL_INSTANCES_A1(QOC_SMALL_ROW1) QOC_SMALL_INSTANCES_B1(QOC_SMALL_ROW1) QOC_SMALL_INSTANCES_A(QOC_SMALL_ROW) QOC_SMALL_INSTANCES_B(QOC_SMALL_ROW) QOC_SMALL_INSTANCES_C(QOC_SMALL_ROW) };
constexpr int N_INST = sizeof(g_inst) / sizeof(g_inst[0]);

int padded_n(int n) {
    static const int sizes[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12};
    for (int s : sizes) if (n <= s) return s;
    return 0;
}

int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// microseconds of one iteration on this mapping (a model: 4 N^2 DPP FMAs of 4 cycles per product at ~2.4 GHz, R / 16 waves sharing a SIMD,
// ~1.5 us per exchange between workgroups): only used to rank the (L, G) candidates of one problem and for AUTO's threshold
double model_us(const QocDev& d, int N, int R, int L, int G, bool src) {
    const int Teff = d.state_transfer ? d.T - 1 : d.T;
    // (n = 2: a slice is a few hundred instructions between LDS round trips -- a second wave per SIMD hides them instead of competing: C1 8.4 us on 32 rows, 9.3 on 16;
    // n = 3, 4: the two waves do compete, and the 32-row builds spill -- n = 4 x 40 slices: 17.9 us on 32 rows, 13.6 on 16, 12.2 on 4 workgroups of 16)
    // (5 <= n <= 8: the split form -- 2 n^2 DPP FMAs per lane, the hand-over and the padding: 0.65 of the 4 n^2 form, measured)
    const double prod = (N >= 5 && N <= 8 ? 2.6 : 4.0) * N * N * 5.9 / 2400.0, share = N <= 4 ? (R >= 32 ? (N <= 2 ? 1.15 : 1.7) : 0.9) : (R <= 16 ? 1.0 : R / 16.0);
    // (state regularisers: besides two more products per slice, the element-wise state terms and sources of three passes -- what a slice of a SMALL system mostly
    // costs then: a qutrit with a forbidden level, 100 slices: 29 us on one workgroup x 8 slices per row, 22 on 7 workgroups x 1)
    const double per_slice = ((Teff > 1 ? Teff - 1 : 0) + d.s + (src ? 6.0 : 4.0)) * prod + 0.15 + (src ? (N <= 4 ? 0.6 : N <= 8 ? 0.3 : 0.0) : 0.0);
    const int LR = ilog2_ceil(R), LG = ilog2_ceil(G);
    double us = share * (L * per_slice + (src ? 4.0 : 2.0) * LR * (prod + 0.1));
    if (G > 1) us += (src ? 4.0 : 2.0) * 1.5 + (src ? 4.0 : 1.0) * LG * (prod + 0.1);        // (no state regulariser: range reduction, half the levels of tree + walk)
    if (d.has_band) us += 1.2e-4 * d.k * (double)d.steps * d.steps;      // direct DFT of the pulse and back: 2 k steps^2 terms over the workgroup's four SIMDs
    return 1.45 * (us + 1.5);            // (measured / modelled: 1.4 - 1.5 over n = 2 .. 12, profiles/r06_small_n_latency.txt)
}

struct Choice { int inst = -1, G = 0; double us = 1e30; };

// control sets x workgroups per control set that may spin on each other: all of them must be resident at once, one per CU -- half the CUs of the device
// the engine is created on, at most QOC_SMALL_WG_BUDGET (MI355X: 256 CUs -> 128)
int wg_budget() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return cus / 2 < QOC_SMALL_WG_BUDGET ? cus / 2 : QOC_SMALL_WG_BUDGET;
}

// G_req > 0: the caller pins the workgroups per control set (qoc_config.chunks; the tests run every exchange on short pulses with it)
Choice choose(const QocDev& d, bool src, int G_req, int R_req = 0) {
    Choice best;
    const int N = padded_n(d.n), budget = wg_budget();
    if (!N) return best;
    for (int i = 0; i < N_INST; ++i) {
        const Instance& in = g_inst[i];
        if (in.N != N || !in.ok[src ? 1 : 0] || (R_req > 0 && in.R != R_req)) continue;
        const int cap = in.R * in.L;
        int G = (d.steps + cap - 1) / cap;
        if (G_req > 0) { if (G > G_req) continue; G = G_req; }
        if (G > QOC_SMALL_MAXG || (in.single && G > 1) || (d.has_band && G > 1)) continue;       // (the bandpass DFT needs the whole pulse in one workgroup)
        if (G > 1 && ((long long)d.Bplan * G > budget || (long long)d.B * G > 2 * budget)) continue;     // (planned batch; and never more resident-or-deadlocked workgroups than CUs)
        const int Gp = 1 << ilog2_ceil(G);
        const QocSmallLayout lo = qoc_small_layout(N, in.R, in.L, d.k, d.m, Gp, src, d.has_band != 0);
        if ((size_t)lo.total * 16 > 160 * 1024) continue;
        const double us = model_us(d, N, in.R, in.L, G, src);
        if (us < best.us) { best.inst = i; best.G = G; best.us = us; }
    }
    return best;
}

bool is_src(const QocDev& d) { return d.n_forb > 0 || d.has_speed; }

}  // namespace

bool qoc_small_supported(const QocDev& d, bool antiherm, int G_req, int R_req, std::string* why) {
    auto no = [&](const char* w) { if (why) *why = w; return false; };
    const int Teff = d.state_transfer ? d.T - 1 : d.T;
    if (d.n > 12) return no("n <= 12");
    if (d.m > d.n || d.m > 16) return no("m <= n");
    if (d.k > 8) return no("k <= 8");
    if (Teff < 0 || Teff > 30) return no("a Taylor order of at most 30");
    if (d.n_forb > QOC_SMALL_NF) return no("at most 4 forbidden levels");
    if (d.state_transfer && !antiherm) return no("exactly anti-Hermitian generators in state transfer");
    if (choose(d, is_src(d), G_req, R_req).inst < 0)
        return no("a pulse that fits 32 workgroups per control set -- and their product trees 160 KB of LDS -- and, with several workgroups per set, a batch that fits the chip");
    return true;
}

bool qoc_small_auto(const QocDev& d, bool antiherm) {
    if (!qoc_small_supported(d, antiherm, 0, 0, nullptr)) return false;
    const Choice c = choose(d, is_src(d), 0);
    // one or a few control sets: the other paths cost >= 42 us per iteration whatever n (profiles/r04_latency_sizes.txt); batches of small
    // systems: the MFMA batch kernels pad to 16 x 16 tiles (C1 x 64: 78 us)
    const double long_pulse = QOC_PLAN_SMALL_MODEL_US_BASE + 1e-3 * QOC_PLAN_SMALL_MODEL_NS_PER_SLICE * d.steps;
    const double limit = is_src(d) ? QOC_PLAN_SMALL_MAX_MODEL_US_SRC : (long_pulse > QOC_PLAN_SMALL_MAX_MODEL_US ? long_pulse : QOC_PLAN_SMALL_MAX_MODEL_US);
    return c.us <= limit && d.Bplan <= QOC_PLAN_SMALL_MAX_SETS;
}

int qoc_small_setup(QocSmall& sm, const QocDev& d, bool antiherm, int G_req, int R_req, std::vector<void*>& allocs, std::string& msg) {
    std::string why;
    if (G_req < 0 || G_req > QOC_SMALL_MAXG) { msg = "the workgroup-resident path takes 1 .. 32 workgroups per control set (chunks)"; return -1; }
    if (!qoc_small_supported(d, antiherm, G_req, R_req, &why)) { msg = "the workgroup-resident path needs " + why; return -1; }
    sm.src = is_src(d);
    const Choice c = choose(d, sm.src, G_req, R_req);
    const Instance& in = g_inst[c.inst];
    sm.N = in.N; sm.L = in.L; sm.R = in.R; sm.G = c.G; sm.inst = c.inst;
    QocSmallDev& sd = sm.sd;
    sd.G = c.G; sd.LG = ilog2_ceil(c.G); sd.Gp = 1 << sd.LG;
    sd.Teff = d.state_transfer ? d.T - 1 : d.T;
    sd.iters = 1;
    const QocSmallLayout lo = qoc_small_layout(sm.N, sm.R, sm.L, d.k, d.m, sd.Gp, sm.src, d.has_band != 0);
    sm.lds_bytes = (size_t)lo.total * 16;
    const int NN2 = 2 * sm.N * sm.N;
    sd.xa_stride = NN2 + 36; sd.xb_stride = 8; sd.xs_stride = 4 + 2 * NN2;
    const size_t BG = (size_t)d.B * c.G;
    const size_t words = BG * 4 + 64 + (size_t)d.B;
    const size_t bytes = (BG * (2 * sd.xa_stride + sd.xb_stride + sd.xs_stride)) * sizeof(double) + words * sizeof(unsigned);
    sd.xa_parity = (long long)(BG * sd.xa_stride);
    char* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { msg = "hipMalloc of the exchange buffers failed"; return -3; }
    allocs.push_back(p);
    if (hipMemset(p, 0, bytes) != hipSuccess) { msg = "clearing the exchange buffers failed"; return -2; }
    sd.xA = (double*)p; sd.xB = sd.xA + 2 * BG * sd.xa_stride; sd.xS = sd.xB + BG * sd.xb_stride;
    sd.flags = (unsigned*)(sd.xS + BG * sd.xs_stride);
    sd.err = sd.flags + BG * 4;
    sd.final_valid = (int*)(sd.err + 64);
    sm.B = d.B;
    sm.flag_bytes = BG * 4 * sizeof(unsigned);
    const int s = sm.src ? 1 : 0;
    if (sm.lds_bytes > 64 * 1024 && !g_inst[c.inst].lds_opted[s]) {
        if (hipFuncSetAttribute((const void*)in.fn[s], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            msg = "cannot reserve LDS for the workgroup-resident kernel"; return -2;
        }
        g_inst[c.inst].lds_opted[s] = true;
    }
    sm.on = true;
    return 0;
}

int qoc_small_launch(QocSmall& sm, const QocDev& d, const QocAdamDev& ap, int iters, hipStream_t s, std::string& msg) {
    if (sm.inst < 0 || sm.inst >= N_INST) { msg = "no kernel instance"; return -1; }
    QocSmallDev sd = sm.sd;
    sd.iters = ap.mode == 1 ? iters : 1;
    if (sm.G > 1 && hipMemsetAsync(sd.flags, 0, sm.flag_bytes, s) != hipSuccess) { msg = "clearing the exchange flags failed"; return -2; }
    hipLaunchKernelGGL(g_inst[sm.inst].fn[sm.src ? 1 : 0], dim3((unsigned)(d.B * sm.G)), dim3((unsigned)(sm.R * 16)), sm.lds_bytes, s, d, ap, sd);
    return 0;
}

bool qoc_small_final_valid(const QocSmall& sm, hipStream_t s) {
    std::vector<int> v((size_t)sm.B, 0);
    if (hipStreamSynchronize(s) != hipSuccess) return false;
    if (hipMemcpy(v.data(), sm.sd.final_valid, v.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
    for (int x : v) if (!x) return false;
    return true;
}
