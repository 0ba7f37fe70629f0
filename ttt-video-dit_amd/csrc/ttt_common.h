// Shared device helpers for the TTT scan kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttt {

typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round-to-nearest-even (NaN kept quiet)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct Act;
template <> struct Act<float> {
    static __device__ __forceinline__ float ld(const float* p, size_t i) { return p[i]; }
    static __device__ __forceinline__ void st(float* p, size_t i, float v) { p[i] = v; }
};
template <> struct Act<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
    static __device__ __forceinline__ void st(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }
};

// tanh-GELU pieces (reference ops/utils.py:51-54 constants)
constexpr float GELU_A = 0.79788456f;
constexpr float GELU_C = 0.044715f;
constexpr float GELU_3AC = 0.1070322243f;

__device__ __forceinline__ float fast_tanh(float u) {
    // tanh(u) = 1 - 2/(1+exp(2u)); exact enough in fp32 and saturates cleanly
    float e = __expf(2.0f * u);
    return 1.0f - 2.0f / (1.0f + e);
}
__device__ __forceinline__ void gelu_and_grad(float x, float& y, float& dy) {
    float x2 = x * x;
    float t = fast_tanh(GELU_A * x * (1.0f + GELU_C * x2));
    y = 0.5f * x * (1.0f + t);
    dy = 0.5f * x * ((1.0f - t * t) * (GELU_A + GELU_3AC * x2)) + 0.5f * (1.0f + t);
}
__device__ __forceinline__ float gelu_only(float x) {
    float t = fast_tanh(GELU_A * x * (1.0f + GELU_C * x * x));
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_grad(float x) {
    float y, dy;
    gelu_and_grad(x, y, dy);
    return dy;
}
__device__ __forceinline__ float gelu_grad2(float x) {  // second derivative
    float x2 = x * x;
    float t = fast_tanh(GELU_A * x * (1.0f + GELU_C * x2));
    float du = GELU_A + GELU_3AC * x2;
    float d2u = 2.0f * GELU_3AC * x;
    float s = 1.0f - t * t;
    return s * du + 0.5f * x * (s * d2u - 2.0f * t * s * du * du);
}

}  // namespace ttt
