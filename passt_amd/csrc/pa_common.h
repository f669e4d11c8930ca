// Shared device/host helpers for the gfx950 kernels of libpasst_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "passt_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define PA_WAVE 64

namespace pa {

int set_hip_error(hipError_t e);  // records the error text for pa_last_hip_error(); returns PA_ELAUNCH

inline int check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PA_OK : set_hip_error(e);
}

__host__ __device__ inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- scalar conversions -----------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---- wave reductions (64 lanes) ---------------------------------------------------------
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

// ---- exact (erf) GELU, nn.GELU default approximate='none' (models/passt.py:286) -------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. f32 round-off class) on hardware rcp/exp:
// ~12 VALU ops instead of libm erff's branchy ~30, and GELU' reuses the SAME exponential for the pdf.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& ex) {
    const float z = fabsf(x) * 0.70710678118654752440f;           // |x| / sqrt(2)
    ex = __expf(-z * z);                                          // exp(-x^2 / 2)
    const float t = __frcp_rn(1.0f + 0.3275911f * z);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t +
                        0.254829592f) * t;
    const float erf_abs = 1.0f - poly * ex;                       // erf(|x|/sqrt2)
    cdf = 0.5f + copysignf(0.5f * erf_abs, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, ex;
    gelu_parts(x, cdf, ex);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, ex;
    gelu_parts(x, cdf, ex);
    return cdf + x * 0.39894228040143267794f * ex;
}

// ---- XCD-aware, bijective block-id remap (guide T1): consecutive logical ids share an XCD ----
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

}  // namespace pa
