// us_common.h -- shared helpers for the gfx950 kernels behind include/unislam_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/unislam_hip.h"
#include "../../include/unislam_hip_experiments.h"   // declarations only; the definitions are compiled with -DUS_EXPERIMENTS

#define US_WAVE 64

void us_set_error(const char* fmt, ...);

#define US_REQUIRE(cond, code, ...)                    \
    do {                                               \
        if (!(cond)) {                                 \
            us_set_error(__VA_ARGS__);                 \
            return (code);                             \
        }                                              \
    } while (0)

// launch check: kernel launches are asynchronous; this only catches configuration errors
#define US_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            us_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return (int)e__;                                                       \
        }                                                                          \
    } while (0)

static inline int64_t us_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- wavefront (64 lanes) reductions / scans on DPP-lowered shuffles ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
