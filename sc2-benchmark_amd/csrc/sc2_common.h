// Shared helpers for the gfx950 kernels of libsc2amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sc2_bottleneck.h"

void sc2_set_error(const char *fmt, ...);

#define SC2_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            sc2_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define SC2_CHECK_LAUNCH()                                              \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) {                                        \
            sc2_set_error("hip launch failed: %s", hipGetErrorString(e__)); \
            return SC2_ERR_LAUNCH;                                      \
        }                                                               \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
    return __builtin_bit_cast(float, (uint32_t)h << 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16_bits(lo) | ((uint32_t)f32_to_bf16_bits(hi) << 16);
}
