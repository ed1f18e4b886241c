// Internal helpers shared by the gfx950 kernels of liblnn_hip.so.  Not part of the public C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/lnn_hip.h"
#include "lnn_debug.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define LNN_WAVE 64

void lnn_set_error(const char* fmt, ...);
// CUs the persistent MFMA kernels size their grids for: the device's CU count, or the budget set by lnn_debug_set_cu_budget
// (measurements only)
int lnn_cu_budget(int device_cus);

#define LNN_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            lnn_set_error(__VA_ARGS__);        \
            return LNN_ERR_BAD_ARG;            \
        }                                      \
    } while (0)

#define LNN_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            lnn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return LNN_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

static inline bool lnn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int lnn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int lnn_round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- device helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum of NV values per thread (256-thread blocks); result valid in thread 0.
template <int NV, int NT = 256>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* >= NV*(NT/64) floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < NT / 64; ++w) s += smem[w * NV + i];
            v[i] = s;
        }
    }
}
