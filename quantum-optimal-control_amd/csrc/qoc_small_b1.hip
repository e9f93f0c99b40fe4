// qoc_small_b1.hip -- the one-workgroup builds (MM = false) of the 5 <= n <= 8 instances of k_small_iter (csrc/qoc_small_instances.h, list B1); the host side is qoc_small.hip.
#include "qoc_small_kernel.h"
#include "qoc_small_instances.h"
#define QOC_SMALL_DEF1(N, L, R, S) \
    template __global__ void qsm::k_small_iter<N, L, R, false, false>(QocDev, QocAdamDev, QocSmallDev); \
    template __global__ void qsm::k_small_iter<N, L, R, true, false>(QocDev, QocAdamDev, QocSmallDev);
QOC_SMALL_INSTANCES_B1(QOC_SMALL_DEF1)
