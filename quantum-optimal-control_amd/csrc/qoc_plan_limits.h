// qoc_plan_limits.h -- every MEASURED number of AUTO's dispatch table in one place.
// qoc_create (qoc_engine.hip: plan_for) and the chunking of the MFMA path (qoc_mfma_backward.hip) include it; tests/test_auto_plan.py PARSES it (one
// `#define NAME integer` per line) for its restatement of the table -- the rules exist twice, on purpose (an independent restatement catches an edit
// of the control flow), the numbers once.  Where each one was measured: the comments beside plan_for and profiles/r0N_*sweep*.txt.
#pragma once

// ---- latency mode of the MFMA path: taken while control sets x time slices <= limit and control sets <= max ------------------------------
#define QOC_PLAN_LAT_WORK 4608              // n <= 16                                        (r03_latency_sweep.txt, r04_small_n_latency.txt)
#define QOC_PLAN_LAT_WORK_SRC 4096          // ... with a state regulariser; also the cap of the 16 < n <= 32 limit with one
#define QOC_PLAN_LAT_WORK_PER_STRIP 512     // 16 < n <= 32: limit = this x max(5, ceil(n / 4))   (r04_padded_latency_sweep.txt)
#define QOC_PLAN_LAT_WORK_PER_STRIP_SRC 768 // ... with a state regulariser: min(LAT_WORK_SRC, this x strips)
#define QOC_PLAN_LAT_WORK_NT3 16384         // 32 < n <= 48 with k <= 4                        (r02_mid_n_sweep.txt)
#define QOC_PLAN_LAT_WORK_NT4 4096          // 48 < n <= 64, and 32 < n <= 48 with k > 4
#define QOC_PLAN_LAT_SETS_N16 6             // control sets, n <= 16 (unitary)
#define QOC_PLAN_LAT_SETS_N16_ST 8          // n <= 16, state transfer
#define QOC_PLAN_LAT_SETS_N32 16            // 16 < n <= 32
#define QOC_PLAN_LAT_SETS_N32_ST_WIDE 4     // 16 < n <= 32, state transfer from 25 levels on (ceil(n / 4) >= 7)
#define QOC_PLAN_LAT_SETS_NT3 8
#define QOC_PLAN_LAT_SETS_NT4 4
#define QOC_PLAN_LAT_MIN_SLICES 64          // shorter pulses never take it
#define QOC_PLAN_LAT_SINGLE_MAX_SLICES 8192 // ONE control set always does, up to this pulse length
// ---- GEMM path instead of the MFMA batch kernels ------------------------------------------------------------------------------------------
#define QOC_PLAN_NT3_MIN_SETS 8             // 32 < n <= 48: MFMA batch kernels from this many control sets on
#define QOC_PLAN_NT4_MIN_SETS_K4 32         // 48 < n <= 64 with k <= 4
#define QOC_PLAN_NT4_MIN_SETS 64            // 48 < n <= 64 with more controls
#define QOC_PLAN_GEMM_SMALL_MIN_SLICES 100  // 16 < n <= 32: GEMM route for up to GEMM_SMALL_Q<strips> control sets of at least this many slices
#define QOC_PLAN_GEMM_SMALL_Q5 2
#define QOC_PLAN_GEMM_SMALL_Q6 3
#define QOC_PLAN_GEMM_SMALL_Q7 5
#define QOC_PLAN_GEMM_SMALL_Q8 7
#define QOC_PLAN_GEMM_SMALL_SRC_Q5 5        // ... with a state regulariser
#define QOC_PLAN_GEMM_SMALL_SRC_Q6 6
#define QOC_PLAN_GEMM_SMALL_SRC_Q78 8
#define QOC_PLAN_GEMM_SMALL_ST_WIDE 8       // state transfer from 25 levels on
// ---- state transfer: direct Taylor chains of the GEMM path ---------------------------------------------------------------------------------
#define QOC_PLAN_ST_BIG_N32 112             // n <= 32 on the MFMA path: the direct chains from this many control sets on ...
#define QOC_PLAN_ST_BIG_N32_MIN_LEVELS 20   // ... of MORE than this many levels
#define QOC_PLAN_ST_BIG_N32_MIN_LEVELS_SRC 28
#define QOC_PLAN_ST_BIG_N64 48              // 32 < n <= 48 on the MFMA path
#define QOC_PLAN_ST_BIG_N64_SRC 112
#define QOC_PLAN_ST_BIG_DPP 20              // ... with ONE state vector (k_gemm_taylor_chain_dpp on the active columns: r05_st_direct_sweep.txt; 32 in round 4)
#define QOC_PLAN_ST_BIG_DPP_SRC 48
#define QOC_PLAN_ST_DIRECT_N32 112          // GEMM path: direct route instead of the propagator route from this many control sets on, n <= 32
#define QOC_PLAN_ST_DIRECT_N64 48           // n > 32
#define QOC_PLAN_ST_DIRECT_DPP 12           // n > 32 with one state vector                     (r04_c3_route_sweep_gauss.txt)
#define QOC_PLAN_ST_DIRECT_DPP_SRC 22
// ---- chunking of the MFMA batch kernels ----------------------------------------------------------------------------------------------------
#define QOC_PLAN_CHUNK_ITEMS 1024           // control sets x chunks: one wave per SIMD of the exponential kernel
#define QOC_PLAN_CHUNKS_MAX_NT2 64
#define QOC_PLAN_CHUNKS_MAX 32
// ---- workgroup-resident path (QOC_PATH_SMALL, n <= 16): taken while the modelled iteration time and the planned batch stay below ------------
#define QOC_PLAN_SMALL_MAX_MODEL_US 52      // microseconds per iteration by qoc_small.hip: model_us (the other paths: 42 - 67 up to 500 slices)   (r06_small_n_latency.txt)
#define QOC_PLAN_SMALL_MODEL_US_BASE 36     // ... or, for long pulses, 0.9 x what the latency mode of the MFMA path costs: 40 us + 0.04 us per slice
#define QOC_PLAN_SMALL_MODEL_NS_PER_SLICE 36    //     (n = 8 / 9 x 1000 slices: 83 / 89 us there, 32 / 49 here)
#define QOC_PLAN_SMALL_MAX_MODEL_US_SRC 85  // ... with a state regulariser (the other paths: 80 - 120)
#define QOC_PLAN_SMALL_MAX_SETS 256         // control sets (one workgroup each when the pulse fits one)
