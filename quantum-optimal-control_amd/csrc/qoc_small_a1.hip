// qoc_small_a1.hip -- the one-workgroup builds (MM = false) of the n <= 4 instances of k_small_iter (csrc/qoc_small_instances.h, list A1); the host side is qoc_small.hip.
// n <= 4: no statement of these builds has a spill copy in front of its DPP read (tools/dpp_hazard_scan.py checks the objects): no padding
#define QOC_SMALL_DPP_PAD 0
#include "qoc_small_kernel.h"
#include "qoc_small_instances.h"
#define QOC_SMALL_DEF1(N, L, R, S) \
    template __global__ void qsm::k_small_iter<N, L, R, false, false>(QocDev, QocAdamDev, QocSmallDev); \
    template __global__ void qsm::k_small_iter<N, L, R, true, false>(QocDev, QocAdamDev, QocSmallDev);
QOC_SMALL_INSTANCES_A1(QOC_SMALL_DEF1)
