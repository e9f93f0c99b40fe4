/*
 * qoc.h -- C ABI of the MI355X-native GRAPE engine (libqoc_hip.so).
 *
 * The reference (SchusterLab/quantum-optimal-control, Python + TensorFlow 1.x) has no FFI: its "operator API" is the
 * set of tensors that core/run_session.py and core/analysis.py fetch from the TensorflowState graph.
 * Every entry point below replaces one of those fetches / feeds; the file:line cited is the reference call site,
 * relative to /root/reference/quantum_optimal_control/.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer; the engine copies inputs to HBM at
 *     qoc_create() and owns device memory until qoc_destroy(); outputs are written into caller buffers.
 *   - complex arrays are C-order interleaved (re, im) float64, i.e. numpy complex128.
 *   - all per-seed arrays have a leading n_seeds dimension (independent random-restart control sets that share the
 *     Hamiltonians; the reference has exactly one seed per Grape() call).
 *   - every function returns QOC_OK (0) or a negative status; qoc_last_error() gives the message (thread-local).
 *   - a handle is not thread-safe; one HIP stream per handle; one process per GPU.
 */
#ifndef QOC_H
#define QOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QOC_OK 0
#define QOC_ERR_INVALID (-1)   /* bad argument / unsupported configuration */
#define QOC_ERR_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define QOC_ERR_NOMEM (-3)
#define QOC_ERR_STATE (-4)     /* call sequence error (e.g. adam step before any evaluation) */

#define QOC_PATH_AUTO 0
#define QOC_PATH_GENERIC 1     /* any n: workgroup-cooperative complex-fp64 products from HBM/L2 */
#define QOC_PATH_MFMA 2        /* n <= 64: register-resident MFMA chain kernels (v_mfma_f64_4x4x4 / 16x16x4; AUTO for n <= 32 batches); unitary mode, and state
                                * transfer with exactly anti-Hermitian generators (K_t = sum_{j < T} A^j / j! + the same thin sweeps; AUTO up to n = 48) */
#define QOC_PATH_ST_FUSED 3    /* state transfer, n <= 64, m <= 4: register-resident generator, LDS vectors */
#define QOC_PATH_GEMM 4        /* any n, m <= 32, both modes: fused LDS exponentials + product tree + persistent thin chains (n <= 64),
                                * batched tiled MFMA GEMM launches above; state transfer by propagators or, with chunks = 1, directly */
#define QOC_PATH_SMALL 5       /* n <= 12, m <= n, k <= 8, no bandpass, <= 4 forbidden levels: the WHOLE iteration (and qoc_iterate's / qoc_run_adam's loop)
                                * inside one launch -- a row of 16 lanes per time slice, matrices column-per-lane in registers, plain complex-fp64 FMAs
                                * (v_fmac_f64_dpp), product tree in LDS, several workgroups per control set for long pulses (csrc/qoc_small.h); AUTO for
                                * one or a few control sets of the sizes the reference is used at (qubits, qutrits, two / three transmons) */

typedef struct qoc_engine* qoc_handle;

/* Static problem description == what SystemParameters hands to TensorflowState
 * (core/system_parameters.py:12-86 -> core/tensorflow_state.py:13-15). */
typedef struct qoc_config {
    int32_t n;              /* state_num                       system_parameters.py:165 */
    int32_t k;              /* ops_len                         system_parameters.py:202 */
    int32_t steps;          /*                                 system_parameters.py:164 */
    int32_t m;              /* len(states_concerned_list)      system_parameters.py:171 */
    int32_t taylor_terms;   /* exp_terms                       system_parameters.py:226-230 */
    int32_t scaling;        /* scaling (squarings)             system_parameters.py:226-230 */
    int32_t state_transfer; /* 0: unitary (matexp_op chain), 1: state transfer (matvecexp_op)  tensorflow_state.py:374-383 */
    int32_t n_seeds;        /* >= 1 */
    double dt;              /* total_time / steps */
    double total_time;
    /* get_reg_loss terms (core/regularization_functions.py:7-97): presence flag + coefficient (un-normalised,
     * the engine divides by steps as the reference does). */
    int32_t has_amplitude, has_envelope, has_dwdt, has_d2wdt2, has_speed_up, has_bandpass;
    double c_amplitude, c_envelope, c_dwdt, c_d2wdt2, c_speed_up, c_bandpass;
    int32_t band_lo, band_hi;   /* band_id = (band*total_time).astype(int)   regularization_functions.py:61 */
    int32_t n_forbidden;        /* len(forbidden_coeff_list)                 regularization_functions.py:71-85 */
    int32_t forbid_dressed;     /* rotate inter_vecs by sort_ev(v_c)^dagger  regularization_functions.py:73-80 */
    int32_t device;             /* HIP device ordinal */
    int32_t path;               /* QOC_PATH_* */
    int32_t chunks;             /* MFMA path: time chunks per seed (0 = auto).  GEMM path, state transfer: 1 = direct route (Taylor
                                 * mat-vec chains, any H), > 1 = propagator route (needs anti-Hermitian generators), 0 = auto.
                                 * Workgroup-resident path (QOC_PATH_SMALL): workgroups per control set (0 = auto) */
    int32_t variant;            /* MFMA path, kernel family: 0 = auto, 1 = v_mfma_f64_16x16x4 everywhere (exponentials by one wave per
                                 * 16-column block of a chunk, one-wave sweeps), 2 = v_mfma_f64_4x4x4 exponentials by one wave per
                                 * 16-column block, 3 = v_mfma_f64_4x4x4 exponentials by one wave per chunk (n <= 32; n > 32: same
                                 * as 2), 4 = as 3 with the left-operand image written under the product's own MFMAs (the round-2
                                 * default), 5 = latency mode (n <= 64; n <= 16 is padded to 32; auto
                                 * for one or a few control sets, decided by seeds x time slices: exponentials
                                 * per time slice, forward and z-free adjoint sweep side by side, slice-parallel gradient),
                                 * 6 = v_mfma_f64_4x4x4 exponentials by two waves per chunk, two waves
                                 * per SIMD (n <= 32; slower than 4, kept for A/B runs), 7 = n > 32: v_mfma_f64_4x4x4
                                 * exponentials by four waves per chunk, a block of rows each (auto for n > 32), 8 = as 4 with
                                 * row-strip-major products whose result rewrites the left-operand image in place (n <= 32, Taylor
                                 * order >= 3; auto for n <= 32 batches since round 3); 2..8 (and auto) run the n <= 32 sweeps on
                                 * v_mfma_f64_4x4x4 as well.
                                 * GEMM path requested explicitly (path = QOC_PATH_GEMM), direct state-transfer route at n in 33..64 with one
                                 * state vector and Hermitian H: 2 = the squared-generator Taylor chain (csrc/qoc_gemm_chain_sq.h; opt-in,
                                 * measured slower than the default chain), other values = the default chain.
                                 * Workgroup-resident path (QOC_PATH_SMALL): rows of 16 lanes per workgroup, 8 / 16 / 32 (0 = auto; A/B runs) */
    int32_t plan_seeds;         /* 0, or the batch size AUTO plans for instead of n_seeds: path, kernel family, chunk count and split
                                 * factors are derived from it, so that a restart evolves bit-identically whether it runs in one engine
                                 * of `plan_seeds` control sets or in a shard of it (GrapeSharded passes restarts / GPUs of the node) */
    int32_t time_shards;        /* 0: off.  G >= 1: ONE large trajectory sharded along the TIME axis over G ranks (the
                                 * alternative for config 5): rank r computes the propagators, sweeps and gradients of its run of time
                                 * chunks; two collectives per iteration (all-gather of G rank products, all-reduce of the gradient
                                 * array) on the engine's stream.  GEMM path, unitary mode, one control set, no state regulariser,
                                 * n > 96, m <= 8 (csrc/qoc_gemm_ts.h); the reference has no counterpart (single device,
                                 * main_grape/grape.py:106-109) */
    int32_t time_rank;          /* this engine's rank 0 .. time_shards - 1 (give it its communicator: qoc_set_time_comm), or -1: all
                                 * ranks emulated inside this one engine on one GPU (how the decomposition is tested) */
    int32_t reserved[3];
} qoc_config;

/* Adam loop hyper-parameters == Convergence (core/convergence.py:16-49). */
typedef struct qoc_adam_params {
    double rate;                /* convergence['rate']                 default 0.01  */
    double learning_rate_decay; /* convergence['learning_rate_decay']  default 2500   */
    double conv_target;         /* convergence['conv_target']          default 1e-8   */
    double min_grad;            /* convergence['min_grad']             default 1e-25  */
    int32_t max_iterations;     /* convergence['max_iterations']       default 5000   */
    int32_t poll_every;         /* host checks the device-side done flags every this many iterations (>=1) */
} qoc_adam_params;

/* ---- lifetime ---------------------------------------------------------------------------------------------------
 * qoc_create replaces TensorflowState(sys_para).build_graph() + tf.Session init
 * (main_grape/grape.py:113-114, core/run_session.py:27-29): constants are copied to HBM once.
 *   Hs   [(k+1)][n][n] complex : -i*dt*H0, -i*dt*Hops[k]     (matrix_list[:k+1] un-embedded, system_parameters.py:197-204)
 *   U0   [n][n] complex        : initial unitary              (tensorflow_state.py:162); ignored in state transfer
 *   V    [n][m] complex        : initial vectors as columns   (tensorflow_state.py:150-156)
 *   W    [n][m] complex        : target vectors as columns, U_target*V in unitary mode (tensorflow_state.py:158-166)
 *   maxA [k]                   : ops_max_amp                  (tensorflow_state.py:178)
 *   one_minus_gauss [k][steps] : envelope constant            (tensorflow_state.py:146-147); may be NULL if !has_envelope
 *   forbidden_states [n_forbidden], forbidden_coeffs [n_forbidden] (regularization_functions.py:81); any length
 *   Vs   [n][n] complex        : sort_ev(v_c, dressed_id), NULL unless forbid_dressed
 */
int qoc_create(const qoc_config* cfg, const double* Hs, const double* U0, const double* V, const double* W,
               const double* maxA, const double* one_minus_gauss, const int32_t* forbidden_states,
               const double* forbidden_coeffs, const double* Vs, qoc_handle* out);
int qoc_destroy(qoc_handle h);

/* ---- the trainable variable -------------------------------------------------------------------------------------
 * ops_weight_base [n_seeds][k][steps]  (tensorflow_state.py:174; ops_weight_base.assign, run_session.py:121).
 * qoc_set_base also resets the Adam slots and the per-seed iteration counters / done flags. */
int qoc_set_base(qoc_handle h, const double* base);
int qoc_get_base(qoc_handle h, double* base);

/* ---- one evaluation == session.run([grad_pack, loss, reg_loss, unitary_scale, grad_squared])
 * (run_session.py:53-54 and get_error :119-127).  Arrays are [n_seeds]; grad is [n_seeds][k][steps] =
 * d reg_loss / d ops_weight_base with the reference's first-order GRAPE gradient (tensorflow_state.py:49-65,
 * 100-133); any output pointer may be NULL. */
int qoc_eval(qoc_handle h, double* loss, double* reg_loss, double* grad_squared, double* unitary_scale,
             double* grad);

/* ---- one optimizer application == session.run([optimizer], {learning_rate: lr}) (run_session.py:66-69):
 * TF1 Adam (beta1 .9, beta2 .999, eps 1e-8 outside the bias correction) on the gradient of the last evaluation
 * (the reference recomputes it at the same parameters, which is numerically the same).  lr is [n_seeds]. */
int qoc_adam_step(qoc_handle h, const double* lr);

/* ---- device-resident optimisation loop == run_session.start_adam_optimizer (run_session.py:47-69), per seed:
 * evaluate; stop if loss < conv_target or grad_squared < min_grad or iterations >= max_iterations; otherwise
 * iterations += 1, lr = rate*exp(-iterations/decay), Adam step.  Stop decisions are taken on the device per seed
 * (finished seeds freeze); the host only polls.  iterations_out is [n_seeds] (may be NULL). */
int qoc_run_adam(qoc_handle h, const qoc_adam_params* p, int32_t* iterations_out);

/* Enqueue exactly `iters` loop iterations (same per-seed semantics as qoc_run_adam) without any host
 * synchronisation; used by bench.py between two qoc_sync() calls. */
int qoc_iterate(qoc_handle h, const qoc_adam_params* p, int32_t iters);
int qoc_sync(qoc_handle h);

/* Last evaluation's per-seed scalars [n_seeds] each (any pointer may be NULL). */
int qoc_get_scalars(qoc_handle h, double* loss, double* reg_loss, double* grad_squared, double* unitary_scale,
                    int32_t* iterations, int32_t* done);

/* ---- read-back == Analysis (core/analysis.py:18-41) and run_session.Get_uks (run_session.py:112-117) ----------
 * uks   [n_seeds][k][steps]          : maxA[k] * sin(base) of the current variable
 * Uf    [n_seeds][n][n] complex      : RtoCMat(final_state) of the last evaluation (unitary mode only)
 * inter [n_seeds][steps+1][n][m] cplx: inter_vecs (tau = 0 is the initial vectors) of the last evaluation */
int qoc_get_uks(qoc_handle h, double* uks);
/* uks the LAST EVALUATION ran on (inside the Adam loop the variable has already moved one step further; the reference
 * logs loss, final_state and uks of the same evaluation, run_session.py:75-91,130-138). */
int qoc_get_uks_evaluated(qoc_handle h, double* uks);
int qoc_get_final_unitary(qoc_handle h, double* Uf);
int qoc_get_inter_vecs(qoc_handle h, double* inter);

/* ---- measurement hooks (bench.py) -------------------------------------------------------------------------------
 * With profiling enabled every launch of the dominant kernel is bracketed by hipEvents on the engine's stream.
 * qoc_profile_read returns the number of bracketed launches and their summed duration in milliseconds. */
int qoc_profile_enable(qoc_handle h, int32_t on);
int qoc_profile_read(qoc_handle h, const char** kernel_name, int64_t* launches, double* total_ms);
/* Elapsed milliseconds (hipEvents on the engine stream) around `iters` iterations, after a sync. */
int qoc_time_iterations(qoc_handle h, const qoc_adam_params* p, int32_t iters, double* elapsed_ms);

/* ---- multi-GPU: seed-parallel sharding, one process per GPU (the reference is single-device,
 * main_grape/grape.py:106-109, and optimises ONE control set per call, core/system_parameters.py:272-284) -------------
 * Restarts are independent, so the data path has no collective.  The only exchange is an all-gather of per-seed scalars
 * (final fidelities) over RCCL/xGMI and an optional broadcast of the winner's pulse.  librccl is opened at run time
 * (dlopen, from the ROCm tree of this library's HIP runtime; QOC_RCCL_LIBRARY overrides), so single-GPU use needs no RCCL.
 * The 128-byte id is created on rank 0 and handed to the other ranks by the launcher (any side channel; the Python host
 * uses a file rendezvous, quantum_optimal_control/parallel_seeds.py). */
typedef struct qoc_comm* qoc_comm_handle;
#define QOC_COMM_ID_BYTES 128
int qoc_comm_unique_id(void* id128);
/* The local preconditions of qoc_comm_create (librccl loadable, device index valid, a stream can be made on it), no collective:
 * every rank checks them and the ranks agree on the outcome BEFORE anybody enters ncclCommInitRank, which would otherwise
 * wait forever for a rank that failed earlier. */
int qoc_comm_probe(int32_t device);
int qoc_comm_create(const void* id128, int32_t world, int32_t rank, int32_t device, qoc_comm_handle* out);
int qoc_comm_destroy(qoc_comm_handle c);
int qoc_comm_world(qoc_comm_handle c);
int qoc_comm_rank(qoc_comm_handle c);
const char* qoc_comm_library(void);        /* path of the librccl in use ("" before the first communicator) */
/* All-gather of one per-seed scalar array of the engine, device to device on the ENGINE's stream (ordered behind the
 * iterations already enqueued, no host synchronisation before the collective).  which: 0 loss, 1 reg_loss,
 * 2 grad_squared, 3 unitary_scale.  width >= n_seeds on every rank (rows are zero padded).  out: [world][width] (host). */
int qoc_comm_all_gather_scalar(qoc_comm_handle c, qoc_handle h, int32_t which, int32_t width, double* out);
/* Host-buffer collectives on the communicator's own stream: recv is [world][count]. */
int qoc_comm_all_gather_f64(qoc_comm_handle c, const double* send, int32_t count, double* recv);
int qoc_comm_all_reduce_max_f64(qoc_comm_handle c, double* inout, int32_t count);
int qoc_comm_broadcast_f64(qoc_comm_handle c, double* buf, int64_t count, int32_t root);
int qoc_comm_barrier(qoc_comm_handle c);

/* Time-sharded engines (qoc_config.time_shards >= 1, time_rank >= 0): the communicator whose ranks hold the other time shards; world and
 * rank must equal time_shards / time_rank.  The engine's iterations then contain RCCL calls: every rank must enqueue the same iterations
 * (and qoc_get_inter_vecs is a collective too: each rank holds the time points of its own slices, the call sums them over the ranks).
 * Lifetime: the communicator must outlive the engine -- qoc_comm_destroy returns QOC_ERR_STATE while an engine still holds it; qoc_destroy
 * of the engine releases it. */
int qoc_set_time_comm(qoc_handle h, qoc_comm_handle c);

/* ---- introspection ---------------------------------------------------------------------------------------------*/
int qoc_path_in_use(qoc_handle h);        /* the QOC_PATH_* the engine resolved AUTO to */
int qoc_chunks_in_use(qoc_handle h);
/* One line of key=value pairs naming what AUTO resolved to (path, kernel family of the exponentials, chunk count, sweep kernels / route):
 * no counterpart in the reference (TensorFlow places its ops itself); the dispatch test tests/test_auto_plan.py reads it. */
int qoc_plan_describe(qoc_handle h, char* buf, int32_t len);
int qoc_device_count(void);
int qoc_device_info(int32_t device, char* name, int32_t name_len, int32_t* compute_units, int64_t* hbm_bytes);
/* hipDeviceCanAccessPeer(device, peer): 1 when `device` can address the memory of `peer` directly (xGMI / PCIe peer-to-peer), which is what RCCL's
 * device-to-device transports between two ranks of a node need; tools/multi_gpu_selftest.py prints the matrix.  The reference runs on one device
 * (main_grape/grape.py:106-109) and has no counterpart. */
int qoc_device_peer_access(int32_t device, int32_t peer, int32_t* can_access);
const char* qoc_last_error(void);
const char* qoc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QOC_H */
