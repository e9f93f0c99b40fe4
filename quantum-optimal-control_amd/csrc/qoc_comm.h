// qoc_comm.h -- seed-parallel multi-GPU exchange behind the C ABI: one RCCL communicator per process (one process per GPU).
//
// The reference is single-device (main_grape/grape.py:106-109); random restarts are independent, so the ONLY exchange of the
// sharded run is an all-gather of per-seed scalars (final fidelities) and an optional broadcast of the winner's pulse
// (SURVEY.md 8e).  The collectives run on the ENGINE's stream, straight from the engine's device-resident scalar arrays: the
// all-gather is ordered after the iterations already enqueued, with no host synchronisation in between.
//
// librccl is opened at run time (dlopen) from the ROCm tree this library's HIP runtime comes from, so that a single-GPU user
// needs no RCCL at all and a process that also holds PyTorch's private RCCL/HIP copies never mixes the two runtimes.
// Included by qoc_engine.hip after the definition of qoc_engine.
#pragma once
#include <dlfcn.h>

#include <mutex>

namespace qoc_rccl {
// the slice of rccl.h used here (rccl/rccl.h:40-43,187,220,260,339,450,467,591,611,678); ABI-stable NCCL 2.x signatures
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
typedef int Result;                  // ncclSuccess = 0
enum { kDouble = 8, kSum = 0, kMax = 2 };
struct Api {
    void* lib = nullptr;
    Result (*GetUniqueId)(UniqueId*) = nullptr;
    Result (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    const char* (*GetErrorString)(Result) = nullptr;
    Result (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    Result (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    Result (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    std::string where;
};
static Api g_api;
static std::mutex g_api_mu;

static int load(std::string& err) {
    std::lock_guard<std::mutex> lock(g_api_mu);
    if (g_api.lib) return 0;
    std::vector<std::string> candidates;
    const char* forced = getenv("QOC_RCCL_LIBRARY");       // an override is exclusive: no silent fall-through to another librccl
    if (forced) candidates.push_back(forced);
    Dl_info info;
    if (!forced && dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {       // the ROCm tree of OUR HIP runtime
        std::string dir(info.dli_fname);
        // A process that imported PyTorch BEFORE this library has resolved our libamdhip64.so.7 to PyTorch's private copy (same
        // soname), and PyTorch's private librccl faults when driven from outside torch (observed: SIGSEGV in ncclGetUniqueId).
        // Refuse instead of crashing: the RCCL transport belongs to torch-free processes (bench.py, GrapeSharded(comm=...));
        // a torch process passes dist= and uses torch.distributed.
        if (dir.find("/torch/lib/") != std::string::npos) {
            err = "this process runs on PyTorch's private HIP runtime (" + dir + "): import quantum_optimal_control before torch, or "
                  "use the torch.distributed transport (dist=), or set QOC_RCCL_LIBRARY";
            return -1;
        }
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            candidates.push_back(dir.substr(0, slash) + "/librccl.so.1");
            candidates.push_back(dir.substr(0, slash) + "/librccl.so");
        }
    }
    if (!forced) {
        candidates.push_back("/opt/rocm/lib/librccl.so.1");
        candidates.push_back("librccl.so.1");
    }
    std::string tried;
    for (const std::string& c : candidates) {
        void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* why = dlerror();               // ONE call: dlerror() clears the message it returns
            tried += c + " (" + (why ? why : "?") + "); ";
            continue;
        }
        Api a;
        a.lib = h;
        a.where = c;
#define QOC_SYM(field, name) *(void**)(&a.field) = dlsym(h, name)
        QOC_SYM(GetUniqueId, "ncclGetUniqueId"); QOC_SYM(CommInitRank, "ncclCommInitRank"); QOC_SYM(CommDestroy, "ncclCommDestroy");
        QOC_SYM(GetErrorString, "ncclGetErrorString"); QOC_SYM(AllGather, "ncclAllGather"); QOC_SYM(AllReduce, "ncclAllReduce");
        QOC_SYM(Broadcast, "ncclBroadcast");
#undef QOC_SYM
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !a.AllGather || !a.AllReduce || !a.Broadcast) {
            tried += c + " (missing nccl symbols); ";
            dlclose(h);
            continue;
        }
        g_api = a;
        return 0;
    }
    err = "librccl not loadable: " + tried;
    return -1;
}
}  // namespace qoc_rccl

struct qoc_comm {
    qoc_rccl::Comm comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;      // collectives that do not belong to an engine (timings, barrier, winner broadcast)
    double* stage = nullptr;           // device staging: [send | recv]
    size_t stage_doubles = 0;
    int attached = 0;                  // time-sharded engines that enqueue collectives on this communicator (qoc_set_time_comm): it outlives them
};
// (called by qoc_destroy, which is compiled before this header)
void qoc_comm_detach(qoc_comm* c) { if (c && c->attached > 0) --c->attached; }

#define RCCL_TRY(expr)                                                                                                    \
    do {                                                                                                                  \
        qoc_rccl::Result r_ = (expr);                                                                                     \
        if (r_ != 0) return fail(QOC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, qoc_rccl::g_api.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

static int comm_stage(qoc_comm* c, size_t doubles) {
    if (doubles <= c->stage_doubles) return QOC_OK;
    if (c->stage) HIP_TRY(hipFree(c->stage));
    c->stage = nullptr; c->stage_doubles = 0;
    const size_t want = (doubles + 1023) & ~(size_t)1023;
    HIP_TRY(hipMalloc((void**)&c->stage, want * sizeof(double)));
    c->stage_doubles = want;
    return QOC_OK;
}

// the two data-path collectives of a time-sharded engine (qoc_gemm_ts.h), device to device on the engine's stream, no host synchronisation
int qoc_ts_all_gather(qoc_comm* c, void* buf, size_t doubles_per_rank, hipStream_t s) {
    if (!c) return fail(QOC_ERR_INVALID, "time-sharded engine without a communicator (qoc_set_time_comm)");
    RCCL_TRY(qoc_rccl::g_api.AllGather((const char*)buf + (size_t)c->rank * doubles_per_rank * sizeof(double), buf, doubles_per_rank, qoc_rccl::kDouble, c->comm, s));
    return QOC_OK;
}
int qoc_ts_all_reduce_sum(qoc_comm* c, double* buf, size_t doubles, hipStream_t s) {
    if (!c) return fail(QOC_ERR_INVALID, "time-sharded engine without a communicator (qoc_set_time_comm)");
    RCCL_TRY(qoc_rccl::g_api.AllReduce(buf, buf, doubles, qoc_rccl::kDouble, qoc_rccl::kSum, c->comm, s));
    return QOC_OK;
}

extern "C" {

int qoc_comm_unique_id(void* id128) {
    if (!id128) return fail(QOC_ERR_INVALID, "qoc_comm_unique_id: null output");
    std::string err;
    if (qoc_rccl::load(err)) return fail(QOC_ERR_HIP, "qoc_comm_unique_id: %s", err.c_str());
    qoc_rccl::UniqueId id;
    RCCL_TRY(qoc_rccl::g_api.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof id.internal);
    return QOC_OK;
}

/* the local preconditions of qoc_comm_create, without any collective: librccl loadable, device index valid, device usable.  Every
 * rank checks them BEFORE anybody enters ncclCommInitRank (a rank that failed here would leave the others blocked in it). */
int qoc_comm_probe(int32_t device) {
    std::string err;
    if (qoc_rccl::load(err)) return fail(QOC_ERR_HIP, "qoc_comm_probe: %s", err.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(QOC_ERR_HIP, "qoc_comm_probe: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(QOC_ERR_INVALID, "qoc_comm_probe: device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return fail(QOC_ERR_HIP, "qoc_comm_probe: hipStreamCreate failed on device %d", device);
    hipStreamDestroy(st);
    return QOC_OK;
}

int qoc_comm_create(const void* id128, int32_t world, int32_t rank, int32_t device, qoc_comm_handle* out) {
    if (!id128 || !out) return fail(QOC_ERR_INVALID, "qoc_comm_create: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(QOC_ERR_INVALID, "qoc_comm_create: rank %d of world %d", rank, world);
    std::string err;
    if (qoc_rccl::load(err)) return fail(QOC_ERR_HIP, "qoc_comm_create: %s", err.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(QOC_ERR_HIP, "qoc_comm_create: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(QOC_ERR_INVALID, "qoc_comm_create: device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));          // RCCL binds the communicator to the current device: one process per GPU
    qoc_comm* c = new qoc_comm();
    c->world = world; c->rank = rank; c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(QOC_ERR_HIP, "qoc_comm_create: hipStreamCreate failed"); }
    qoc_rccl::UniqueId id;
    memcpy(id.internal, id128, sizeof id.internal);
    const qoc_rccl::Result r = qoc_rccl::g_api.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
        hipStreamDestroy(c->stream);
        delete c;
        return fail(QOC_ERR_HIP, "ncclCommInitRank(world %d, rank %d, device %d) failed: %s", world, rank, device, qoc_rccl::g_api.GetErrorString(r));
    }
    *out = c;
    return QOC_OK;
}

int qoc_comm_destroy(qoc_comm_handle c) {
    if (!c) return QOC_OK;
    if (c->attached > 0)
        return fail(QOC_ERR_STATE, "qoc_comm_destroy: %d time-sharded engine(s) still use this communicator (qoc_set_time_comm): destroy them first", c->attached);
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm) qoc_rccl::g_api.CommDestroy(c->comm);
    if (c->stage) hipFree(c->stage);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return QOC_OK;
}

int qoc_comm_world(qoc_comm_handle c) { return c ? c->world : QOC_ERR_INVALID; }
int qoc_comm_rank(qoc_comm_handle c) { return c ? c->rank : QOC_ERR_INVALID; }
const char* qoc_comm_library(void) { return qoc_rccl::g_api.where.c_str(); }

#define CHECK_C(c) if (!(c)) return fail(QOC_ERR_INVALID, "null communicator"); HIP_TRY(hipSetDevice((c)->device))

/* all-gather of one per-seed scalar array of the engine (which: 0 loss, 1 reg_loss, 2 grad_squared, 3 unitary_scale), enqueued
 * on the engine's stream behind whatever iterations are already there.  `width` >= n_seeds of every rank (ranks may own
 * different numbers of seeds; the tail of a rank's row is zero).  out: [world][width] on the host. */
int qoc_comm_all_gather_scalar(qoc_comm_handle c, qoc_handle e, int32_t which, int32_t width, double* out) {
    CHECK_C(c);
    if (!e || !out) return fail(QOC_ERR_INVALID, "qoc_comm_all_gather_scalar: null argument");
    if (e->cfg.device != c->device) return fail(QOC_ERR_INVALID, "qoc_comm_all_gather_scalar: engine on device %d, communicator on %d", e->cfg.device, c->device);
    const QocDev& d = e->d;
    if (width < d.B) return fail(QOC_ERR_INVALID, "qoc_comm_all_gather_scalar: width %d < n_seeds %d", width, d.B);
    const double* src = which == 0 ? d.loss : which == 1 ? d.reg_loss : which == 2 ? d.g2 : which == 3 ? d.uscale : nullptr;
    if (!src) return fail(QOC_ERR_INVALID, "qoc_comm_all_gather_scalar: unknown scalar %d", which);
    if (which == 3) TRY(refresh_final(e));          // latency mode forms unitary_scale only on read-back (as qoc_get_scalars does)
    TRY(comm_stage(c, (size_t)width * (1 + c->world)));
    double* send = c->stage;
    double* recv = c->stage + width;
    HIP_TRY(hipMemsetAsync(send, 0, (size_t)width * sizeof(double), e->stream));
    HIP_TRY(hipMemcpyAsync(send, src, (size_t)d.B * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    RCCL_TRY(qoc_rccl::g_api.AllGather(send, recv, (size_t)width, qoc_rccl::kDouble, c->comm, e->stream));
    HIP_TRY(hipMemcpyAsync(out, recv, (size_t)width * c->world * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QOC_OK;
}

/* host-buffer collectives on the communicator's own stream (timings, barrier, the winner's pulse) */
int qoc_comm_all_gather_f64(qoc_comm_handle c, const double* send_host, int32_t count, double* recv_host) {
    CHECK_C(c);
    if (!send_host || !recv_host || count < 1) return fail(QOC_ERR_INVALID, "qoc_comm_all_gather_f64: bad argument");
    TRY(comm_stage(c, (size_t)count * (1 + c->world)));
    double* send = c->stage;
    double* recv = c->stage + count;
    HIP_TRY(hipMemcpyAsync(send, send_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(qoc_rccl::g_api.AllGather(send, recv, (size_t)count, qoc_rccl::kDouble, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(recv_host, recv, (size_t)count * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QOC_OK;
}

int qoc_comm_all_reduce_max_f64(qoc_comm_handle c, double* inout_host, int32_t count) {
    CHECK_C(c);
    if (!inout_host || count < 1) return fail(QOC_ERR_INVALID, "qoc_comm_all_reduce_max_f64: bad argument");
    TRY(comm_stage(c, (size_t)count));
    HIP_TRY(hipMemcpyAsync(c->stage, inout_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(qoc_rccl::g_api.AllReduce(c->stage, c->stage, (size_t)count, qoc_rccl::kDouble, qoc_rccl::kMax, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(inout_host, c->stage, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QOC_OK;
}

int qoc_comm_broadcast_f64(qoc_comm_handle c, double* buf_host, int64_t count, int32_t root) {
    CHECK_C(c);
    if (!buf_host || count < 1 || root < 0 || root >= c->world) return fail(QOC_ERR_INVALID, "qoc_comm_broadcast_f64: bad argument");
    TRY(comm_stage(c, (size_t)count));
    if (c->rank == root) HIP_TRY(hipMemcpyAsync(c->stage, buf_host, (size_t)count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(qoc_rccl::g_api.Broadcast(c->stage, c->stage, (size_t)count, qoc_rccl::kDouble, root, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(buf_host, c->stage, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QOC_OK;
}

int qoc_set_time_comm(qoc_handle e, qoc_comm_handle c) {
    if (!e || !c) return fail(QOC_ERR_INVALID, "qoc_set_time_comm: null argument");
    if (e->path != QOC_PATH_GEMM || e->gm.ts_G <= 0 || e->gm.ts_rank < 0) return fail(QOC_ERR_INVALID, "qoc_set_time_comm: the engine is not one rank of a time-sharded run");
    if (c->world != e->gm.ts_G || c->rank != e->gm.ts_rank) return fail(QOC_ERR_INVALID, "qoc_set_time_comm: communicator rank %d of %d, engine rank %d of %d", c->rank, c->world, e->gm.ts_rank, e->gm.ts_G);
    if (c->device != e->cfg.device) return fail(QOC_ERR_INVALID, "qoc_set_time_comm: engine on device %d, communicator on %d", e->cfg.device, c->device);
    if (e->gm.ts_comm != c) { qoc_comm_detach(e->gm.ts_comm); ++c->attached; }
    e->gm.ts_comm = c;
    return QOC_OK;
}

int qoc_comm_barrier(qoc_comm_handle c) {
    double one = 1.0;
    return qoc_comm_all_reduce_max_f64(c, &one, 1);
}

}  // extern "C"
