// Shared device/host helpers for the gfx950 kernels of libpasst_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "passt_amd.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: set it on every device this process launches
// on (keyed on hipGetDevice(), decided on the device's first launch; `state` is one static array per launch site).  Returns false
// where the device refuses it -- the launch that follows then fails loudly, or the caller takes another kernel.
inline bool lds_attr_on_this_device(const void* kernel, int bytes, signed char (&state)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[dev] == 0) {
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) (void)hipGetLastError();
        state[dev] = e == hipSuccess ? 1 : -1;
    }
    return state[dev] > 0;
}

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
// Phi(x) through erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7: f32 round-off class) on the
// hardware v_rcp_f32 / v_exp_f32 (no IEEE division, no libm erff):
//   q(x) = 0.5 * poly(t) * exp(-x^2/2),  t = 1 / (1 + p |x| / sqrt2)      ( = 1 - Phi(|x|) )
//   gelu(x)  = relu(x) - |x| q          gelu'(x) = Phi(x) + x phi(x),  Phi(x) = x >= 0 ? 1 - q : q
// ~14 / ~18 VALU ops per element; the GELU' pdf reuses the same exponential.
__device__ __forceinline__ float gelu_q(float ax, float& ex) {
    ex = __builtin_amdgcn_exp2f(ax * ax * -0.72134752044448170368f);          // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    return poly * t * ex;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float ex;
    const float ax = fabsf(x);
    return fmaf(-ax, gelu_q(ax, ex), fmaxf(x, 0.0f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float ex;
    const float q = gelu_q(fabsf(x), ex);
    const float cdf = x >= 0.0f ? 1.0f - q : q;
    return fmaf(x * 0.39894228040143267794f, ex, cdf);
}

// ---- GELU for the bf16 epilogues: odd minimax polynomials on the clamped argument, two elements per
// v_pk_fma_f32 and no transcendental (quarter-rate) instruction.  The fc1 / dgrad-fc2 GEMM epilogues are VALU
// bound (65536 outputs per CU per tile), so this is ~2.7x cheaper than the erf form above.  Coefficients and
// error bounds from tools/gelu_fit.py: |Phi err| <= 1.3e-5, |gelu' err| <= 1.5e-4 in f32 evaluation -- 1/30 and
// 1/10 of a bf16 half-ulp of the stored results.  The f32 parity path keeps gelu_erf / gelu_erf_grad.
//   Phi(x)   = 1/2 + xc P(xc^2),   gelu'(x) = 1/2 + xc Q(xc^2),   xc = clamp(x, -4.25, 4.25)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 gelu_clamp2(f32x2 x) {
    return f32x2{__builtin_amdgcn_fmed3f(x[0], -4.25f, 4.25f), __builtin_amdgcn_fmed3f(x[1], -4.25f, 4.25f)};
}
// r02: evaluated with SCALAR f32 FMAs (PA_GELU_PACKED = 0).  On gfx950 a wave64 v_fma_f32 issues in 2 cycles (SIMD-32), so
// packed f32 math has no throughput advantage per element, and v_pk_fma_f32 measured slower than two v_fma_f32 (the GELU
// epilogue ran ~13.5k cycles per 256x256 tile against a ~6.6k VALU floor; MI355X_MICROARCH.md prices 1 v_pk_fma_f32 at
// +22 cycles over 2 v_fma_f32 next to MFMAs).  gemm.hip is compiled with -fno-slp-vectorize so that the compiler does
// not re-pack the scalar chains.
#ifndef PA_GELU_PACKED
#define PA_GELU_PACKED 0
#endif
__device__ __forceinline__ float gelu_phi_fast1(float x) {       // Phi(x)
    const float xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f), t = xc * xc;
    float p = fmaf(5.564853169e-11f, t, -5.327756179e-09f);
    p = fmaf(p, t, 2.255428225e-07f);
    p = fmaf(p, t, -5.626429780e-06f);
    p = fmaf(p, t, 9.341873147e-05f);
    p = fmaf(p, t, -1.108561126e-03f);
    p = fmaf(p, t, 9.815971666e-03f);
    p = fmaf(p, t, -6.634449192e-02f);
    p = fmaf(p, t, 3.989023391e-01f);
    return fmaf(xc, p, 0.5f);
}
__device__ __forceinline__ float gelu_grad_fast1(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.25f, 4.25f), t = xc * xc;
    float q = fmaf(-3.426787246e-11f, t, 3.552477616e-09f);
    q = fmaf(q, t, -1.634916764e-07f);
    q = fmaf(q, t, 4.432675237e-06f);
    q = fmaf(q, t, -7.936289004e-05f);
    q = fmaf(q, t, 9.965486026e-04f);
    q = fmaf(q, t, -9.040460278e-03f);
    q = fmaf(q, t, 5.909254817e-02f);
    q = fmaf(q, t, -2.653926090e-01f);
    q = fmaf(q, t, 7.977564352e-01f);
    return fmaf(xc, q, 0.5f);
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
#if PA_GELU_PACKED
    const f32x2 xc = gelu_clamp2(x), t = xc * xc;
    f32x2 p = pk_fma(pk_splat(5.564853169e-11f), t, pk_splat(-5.327756179e-09f));
    p = pk_fma(p, t, pk_splat(2.255428225e-07f));
    p = pk_fma(p, t, pk_splat(-5.626429780e-06f));
    p = pk_fma(p, t, pk_splat(9.341873147e-05f));
    p = pk_fma(p, t, pk_splat(-1.108561126e-03f));
    p = pk_fma(p, t, pk_splat(9.815971666e-03f));
    p = pk_fma(p, t, pk_splat(-6.634449192e-02f));
    p = pk_fma(p, t, pk_splat(3.989023391e-01f));
    return x * pk_fma(xc, p, pk_splat(0.5f));
#else
    return f32x2{x[0] * gelu_phi_fast1(x[0]), x[1] * gelu_phi_fast1(x[1])};
#endif
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
#if PA_GELU_PACKED
    const f32x2 xc = gelu_clamp2(x), t = xc * xc;
    f32x2 q = pk_fma(pk_splat(-3.426787246e-11f), t, pk_splat(3.552477616e-09f));
    q = pk_fma(q, t, pk_splat(-1.634916764e-07f));
    q = pk_fma(q, t, pk_splat(4.432675237e-06f));
    q = pk_fma(q, t, pk_splat(-7.936289004e-05f));
    q = pk_fma(q, t, pk_splat(9.965486026e-04f));
    q = pk_fma(q, t, pk_splat(-9.040460278e-03f));
    q = pk_fma(q, t, pk_splat(5.909254817e-02f));
    q = pk_fma(q, t, pk_splat(-2.653926090e-01f));
    q = pk_fma(q, t, pk_splat(7.977564352e-01f));
    return pk_fma(xc, q, pk_splat(0.5f));
#else
    return f32x2{gelu_grad_fast1(x[0]), gelu_grad_fast1(x[1])};
#endif
}
// 8-element forms used by the epilogues: T selects the exact (f32 parity) or the fast (bf16) evaluation
template <typename T> __device__ __forceinline__ void gelu8(const float (&x)[8], float (&g)[8]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = gelu_erf(x[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 r = gelu_fast2(f32x2{x[e], x[e + 1]});
            g[e] = r[0]; g[e + 1] = r[1];
        }
    }
}
template <typename T> __device__ __forceinline__ void mul_gelu_grad8(float (&v)[8], const float (&x)[8]) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad(x[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 r = f32x2{v[e], v[e + 1]} * gelu_grad_fast2(f32x2{x[e], x[e + 1]});
            v[e] = r[0]; v[e + 1] = r[1];
        }
    }
}

// ---- 8 consecutive elements <-> 8 floats (16-byte vectors; N % 8 == 0 is an API precondition)
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *(bf16x8*)p = o;
}
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
    const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}

// ---- XCD-aware, bijective block-id remap (guide T1): consecutive logical ids share an XCD ----
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

}  // namespace pa
