// placeholder replaced below by the register-resident MFMA path
#pragma once
#include <string>
#include <vector>
#include "qoc_common.h"
struct QocMfma { int C = 1; };
static inline bool qoc_mfma_supported(const QocDev&) { return false; }
static inline int qoc_mfma_setup(QocMfma&, const QocDev&, int, const cplx*, std::vector<void*>&, std::string& msg) { msg = "not built"; return -1; }
static inline void qoc_mfma_launch_expm(QocMfma&, const QocDev&, hipStream_t) {}
static inline void qoc_mfma_launch_forward(QocMfma&, const QocDev&, hipStream_t) {}
static inline void qoc_mfma_launch_backward(QocMfma&, const QocDev&, hipStream_t) {}
