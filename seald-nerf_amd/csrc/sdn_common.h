// Shared device/host helpers for libsdn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/sdn_hip.h"

#define SDN_WAVE 64

static inline int sdn_launch_status() {
    hipError_t e = hipGetLastError();
    return (int)e;
}

template <typename T>
static inline T sdn_div_up(T a, T b) { return (a + b - 1) / b; }

// float(exp(double(x))): agrees with the oracle's correctly rounded exp except for
// double-rounding ties (~1e-9 of inputs); the compositing kernels are HBM/latency
// bound, the fp64 polynomial is free next to their loads.
__device__ __forceinline__ float sdn_exp_cr(float x) { return (float)exp((double)x); }
