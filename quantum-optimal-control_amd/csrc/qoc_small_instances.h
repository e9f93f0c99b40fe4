// qoc_small_instances.h -- which (n, slices per row, rows per workgroup) instances of k_small_iter exist.  X(N, L, R, SRC_TOO): SRC_TOO = 1 when the instance with the
// state-regulariser flow compiles without spilling (kernel-resource-usage: ScratchSize 0).  qoc_small.hip builds its table from the list; qoc_small_a / _a1 / _b / _c.hip
// instantiate their share of it (four translation units compile side by side).
#pragma once
// n <= 4, builds that may run several workgroups per control set (MM = true): 16 rows (one wave per SIMD; the exponentials of a long pulse are issue-bound)
#define QOC_SMALL_INSTANCES_A(X) \
    X(2, 1, 16, 1) X(2, 2, 16, 1) X(2, 4, 16, 1) X(3, 1, 16, 1) X(3, 2, 16, 1) X(3, 4, 16, 1) X(4, 1, 16, 1) X(4, 2, 16, 1) X(4, 4, 16, 1)
// n <= 4, builds for ONE workgroup per control set (MM = false: no exchange code, no deferred stop rule): 32 rows (a second wave per SIMD hides the LDS round trips of
// a slice of a few hundred instructions: C1) and 16 rows x 4 / 8 slices for pulses of up to 128 slices
#define QOC_SMALL_INSTANCES_A1(X) \
    X(2, 1, 32, 1) X(2, 2, 32, 1) X(2, 4, 32, 0) X(3, 1, 32, 1) X(3, 2, 32, 0) X(3, 4, 32, 0) X(4, 1, 32, 0) X(4, 2, 32, 0) \
    X(2, 4, 16, 1) X(2, 8, 16, 1) X(3, 4, 16, 1) X(3, 8, 16, 1) X(4, 4, 16, 1) X(4, 8, 16, 0)
// 5 <= n <= 8: 16 rows (one wave per SIMD: a slice is thousands of instructions)
#define QOC_SMALL_INSTANCES_B(X) \
    X(5, 1, 16, 1) X(5, 2, 16, 1) X(5, 4, 16, 1) X(6, 1, 16, 1) X(6, 2, 16, 1) X(6, 4, 16, 1) X(7, 1, 16, 1) X(7, 2, 16, 1) X(7, 4, 16, 0) \
    X(8, 1, 16, 1) X(8, 2, 16, 1) X(8, 4, 16, 0)
// ... and their builds for ONE workgroup per control set (pulses of up to 64 slices)
#define QOC_SMALL_INSTANCES_B1(X) \
    X(5, 2, 16, 1) X(5, 4, 16, 1) X(6, 2, 16, 1) X(6, 4, 16, 1) X(7, 2, 16, 1) X(7, 4, 16, 1) X(8, 2, 16, 1) X(8, 4, 16, 1)
// 8 < n <= 12; n > 10: a product tree of 16 rows (x 2 with the offsets of a state regulariser) does not always fit 160 KB beside the Hamiltonians: 8 rows.
// (n = 13 .. 16 stay on the MFMA path: a column-per-lane instance of 16 levels needs more than the 512 registers of a wave, and a 16 x 16 MFMA tile has no padding there)
#define QOC_SMALL_INSTANCES_C(X) \
    X(9, 1, 16, 1) X(9, 2, 16, 1) X(10, 1, 16, 1) X(10, 2, 16, 0) X(12, 1, 16, 1) X(12, 1, 8, 1)
