// dcl_common.h -- shared helpers for libdcl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dcl_hip.h"

void dcl_set_error(const char *fmt, ...);
void dcl_note_kernel(const char *fmt, ...);     // dcl_capi.cpp: name of the kernel an entry point chose (dcl_trace_kernels)

#define DCL_CHECK_ARG(cond, msg)                              \
    do {                                                      \
        if (!(cond)) {                                        \
            dcl_set_error("%s: %s", __func__, msg);           \
            return DCL_EINVAL;                                \
        }                                                     \
    } while (0)

#define DCL_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            dcl_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return (int)e_;                                                       \
        }                                                                         \
    } while (0)

static inline int dcl_round_up(int x, int m) { return (x + m - 1) / m * m; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 64-lane wavefront reductions (gfx950: wave64 only)
__device__ inline float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = max(v, __shfl_xor(v, o, 64));
    return v;
}
