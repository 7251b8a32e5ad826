// Shared helpers for the gfx950 kernels of libvgpt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/vgpt.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VGPT_EXPORT extern "C" __attribute__((visibility("default")))
#define WAVE 64

void vgpt_set_error(const char* fmt, ...);

#define VGPT_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            vgpt_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define VGPT_CHECK_LAUNCH(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            vgpt_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return VGPT_ERR_HIP;                                                  \
        }                                                                         \
    } while (0)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Combine a value with the one held by the lane 32 away (the two halves of a wave) without an LDS round trip:
// v_permlane32_swap hands every lane both halves' values.
__device__ __forceinline__ float half_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        // x * rcp(1 + exp(-x)): v_rcp_f32 (1 ulp) instead of the IEEE division's correction sequence (half the
        // instructions of the gated GEMM's epilogue); the result is rounded to bf16 right after
        case VGPT_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
        case VGPT_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
        case VGPT_ACT_GELU_TANH: {
            float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.0f + tanhf(u));
        }
        default: return x;
    }
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
