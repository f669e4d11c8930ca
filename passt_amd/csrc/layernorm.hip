// LayerNorm forward / backward over the last dimension, one 64-lane wave per row.
// Reference: nn.LayerNorm in Block (models/passt.py:369,373,378-379, eps 1e-6 from :426), the final
// norm (:450,:570) and head.0 (:463, eps 1e-5).  HBM-bound: every row is read once with float4
// loads, kept in registers for the two-pass mean/variance (torch's biased variance), written once.
#include <algorithm>
#include <cstdlib>

#include "pa_common.h"

namespace pa {

static constexpr int LN_MAXV = 8;       // float4 per lane -> D <= 2048 (kernels are instantiated for 1,2,3,4,8)
static constexpr int LN_BWD_BLOCKS = 1024;

// Cache policy (A/B build knobs, round 6 sweep profiles/r06_cache_policy.txt): PA_LN_NT_LD = the row operands that are read for
// the last time here (x in the forward; dy, x, dres in the backward) as non-temporal loads, PA_LN_NT_ST bit 0 = the f32 outputs
// (dx), bit 1 = the low-precision outputs (y, dx_lp: the next GEMM's A operand) as non-temporal stores
#ifndef PA_LN_NT_LD
#define PA_LN_NT_LD 0
#endif
#ifndef PA_LN_NT_ST
#define PA_LN_NT_ST 0
#endif
__device__ __forceinline__ float4 ldrow4(const float* p) {
#if PA_LN_NT_LD
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_nontemporal_load((const f4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *(const float4*)p;
#endif
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float4& v) {
#if PA_LN_NT_ST & 2
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, (f4*)p);
#else
    *(float4*)p = v;
#endif
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float4& v) {
    bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
#if PA_LN_NT_ST & 2
    __builtin_nontemporal_store(o, (bf16x4*)p);
#else
    *(bf16x4*)p = o;
#endif
}
template <typename T> __device__ __forceinline__ float4 load4(const T* p);
template <> __device__ __forceinline__ float4 load4<float>(const float* p) { return ldrow4(p); }
template <> __device__ __forceinline__ float4 load4<bf16>(const bf16* p) {
#if PA_LN_NT_LD
    const bf16x4 v = __builtin_nontemporal_load((const bf16x4*)p);
#else
    const bf16x4 v = *(const bf16x4*)p;
#endif
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float* xr = x + (int64_t)row * D;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) { v[i] = ldrow4(xr + 4 * c); s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
    }
    const float mu = wave_sum(s) / (float)D;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            s2 += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float r = rsqrtf(wave_sum(s2) / (float)D + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = r;
    }
    T* yr = y + (int64_t)row * D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = *(const float4*)(gamma + 4 * c), b = *(const float4*)(beta + 4 * c);
            float4 o;
            o.x = (v[i].x - mu) * r * g.x + b.x; o.y = (v[i].y - mu) * r * g.y + b.y;
            o.z = (v[i].z - mu) * r * g.z + b.z; o.w = (v[i].w - mu) * r * g.w + b.w;
            store4<T>(yr + 4 * c, o);
        }
    }
}

// dx = dres + rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat));  per-block partial
// column sums of dy*xhat (dgamma) and dy (dbeta) go to ws[block][2][D].
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ dres,
                                                     float* __restrict__ dx, T* __restrict__ dx_lp,
                                                     float* __restrict__ ws, int M, int D) {
    // ws[block][3][D]: partial column sums of dy*xhat (dgamma), dy (dbeta) and of the OUTPUT dx (the bias gradient
    // of the Linear that produced this LayerNorm's input: proj for norm2, the previous block's fc2 for norm1)
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][3][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    float4 g[MAXV], ag[MAXV], ab[MAXV], ac[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        g[i] = c < nv ? *(const float4*)(gamma + 4 * c) : make_float4(0, 0, 0, 0);
        ag[i] = make_float4(0, 0, 0, 0);
        ab[i] = make_float4(0, 0, 0, 0);
        ac[i] = make_float4(0, 0, 0, 0);
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mu = mean[row], r = rstd[row];
        const int64_t base = (int64_t)row * D;
        float4 xh[MAXV], gy[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 xv = ldrow4(x + base + 4 * c);
                const float4 d = load4<T>(dy + base + 4 * c);
                xh[i] = make_float4((xv.x - mu) * r, (xv.y - mu) * r, (xv.z - mu) * r, (xv.w - mu) * r);
                gy[i] = make_float4(d.x * g[i].x, d.y * g[i].y, d.z * g[i].z, d.w * g[i].w);
                s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
                s2 += (gy[i].x * xh[i].x + gy[i].y * xh[i].y) + (gy[i].z * xh[i].z + gy[i].w * xh[i].w);
                ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
            }
        }
        const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o;
                o.x = r * (gy[i].x - c1 - xh[i].x * c2); o.y = r * (gy[i].y - c1 - xh[i].y * c2);
                o.z = r * (gy[i].z - c1 - xh[i].z * c2); o.w = r * (gy[i].w - c1 - xh[i].w * c2);
                if (dres) {
                    const float4 dr = ldrow4(dres + base + 4 * c);
                    o.x += dr.x; o.y += dr.y; o.z += dr.z; o.w += dr.w;
                }
                {
#if PA_LN_NT_ST & 1
                    typedef float f4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(f4{o.x, o.y, o.z, o.w}, (f4*)(dx + base + 4 * c));
#else
                    *(float4*)(dx + base + 4 * c) = o;
#endif
                }
                if (dx_lp) store4<T>(dx_lp + base + 4 * c, o);
                ac[i].x += o.x; ac[i].y += o.y; ac[i].z += o.z; ac[i].w += o.w;
            }
        }
    }
    // block reduce of the 4 waves' partials, then one partial row per block
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            *(float4*)(red + (wave * 3 + 0) * D + 4 * c) = ag[i];
            *(float4*)(red + (wave * 3 + 1) * D + 4 * c) = ab[i];
            *(float4*)(red + (wave * 3 + 2) * D + 4 * c) = ac[i];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * D; k += 256) {
        const int which = k / D, c = k - which * D;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * 3 + which) * D + c];
        ws[(int64_t)blockIdx.x * 3 * D + k] = s;
    }
}

// out[which][c] (+)= sum_b ws[b][which][c], which = dgamma | dbeta | dcol(optional);  block = 16 columns x 16 row
// groups (many small blocks: the partials are only a few MB, parallelism matters more than coalescing width)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblk, int D,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dcol, int accumulate) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + cx;   // index into [3][D]
    const int nk = dcol ? 3 * D : 2 * D;
    float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (k < nk) {
        int b = ry;
        for (; b + 48 < nblk; b += 64) {          // 4 loads in flight
            s += ws[(int64_t)b * 3 * D + k];
            s1 += ws[(int64_t)(b + 16) * 3 * D + k];
            s2 += ws[(int64_t)(b + 32) * 3 * D + k];
            s3 += ws[(int64_t)(b + 48) * 3 * D + k];
        }
        for (; b < nblk; b += 16) s += ws[(int64_t)b * 3 * D + k];
    }
    s = (s + s1) + (s2 + s3);
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && k < nk) {
        s = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) s += red[y][cx];
        float* out = k < D ? dgamma + k : (k < 2 * D ? dbeta + (k - D) : dcol + (k - 2 * D));
        *out = (accumulate ? *out : 0.f) + s;
    }
}

}  // namespace pa

using namespace pa;

extern "C" int pa_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, int dtype,
                                float* mean, float* rstd, int M, int D, float eps, void* stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || D <= 0) return PA_EINVAL;
    if (D % 4 || D > LN_MAXV * 256) return PA_EUNSUPPORTED;
    dim3 grid((unsigned)cdiv(M, 4));
    if (dtype != PA_BF16 && dtype != PA_F32) return PA_EINVAL;
    const int nvl = (int)cdiv(D, 256);
#define PA_LN_FWD(V)                                                                                              \
    do {                                                                                                          \
        if (dtype == PA_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16, V>), grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16*)y, mean, rstd, M, D, eps); \
        else hipLaunchKernelGGL((ln_fwd_kernel<float, V>), grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (float*)y, mean, rstd, M, D, eps); \
    } while (0)
    if (nvl <= 1) PA_LN_FWD(1); else if (nvl == 2) PA_LN_FWD(2); else if (nvl == 3) PA_LN_FWD(3);
    else if (nvl == 4) PA_LN_FWD(4); else PA_LN_FWD(8);
#undef PA_LN_FWD
    return check_launch();
}

// workgroups of the backward kernel (= partial rows its finishing reduction reads): at most 1024, at least two rows per wave
// (round 5, ESC-50 at batch 12, M = 4 236: 530 workgroups instead of 1 024 -- half the partial rows to write and reduce -- is
// + 0.9 % on that step; four rows per wave - 2 %: too few waves in flight).  PA_LN_BWD_BLOCKS / PA_LN_BWD_ROWS_PER_WAVE
// (environment, read once): A/B knobs
static int ln_bwd_blocks(int M) {
    static const int cap = [] { const char* e = getenv("PA_LN_BWD_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : LN_BWD_BLOCKS; }();
    static const int rpw = [] { const char* e = getenv("PA_LN_BWD_ROWS_PER_WAVE"); return e && atoi(e) > 0 ? atoi(e) : 2; }();
    return (int)std::max<int64_t>(1, std::min<int64_t>(cap, cdiv(M, 4 * rpw)));
}

extern "C" int64_t pa_layernorm_bwd_ws_floats(int M, int D) { return (int64_t)ln_bwd_blocks(M) * 3 * D; }

extern "C" int pa_layernorm_bwd_rows(int M) { return M > 0 ? ln_bwd_blocks(M) : 0; }

static int ln_bwd_launch(const void* dy, int dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* dres, float* dx, void* dx_lp, float* ws, int M, int D, hipStream_t st);

extern "C" int pa_layernorm_bwd_partial(const void* dy, int dtype, const float* x, const float* gamma,
                                        const float* mean, const float* rstd, const float* dres, float* dx,
                                        void* dx_lp, float* ws, int M, int D, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !ws || M <= 0 || D <= 0) return PA_EINVAL;
    if (D % 4 || D > LN_MAXV * 256) return PA_EUNSUPPORTED;
    if (dtype != PA_BF16 && dtype != PA_F32) return PA_EINVAL;
    return ln_bwd_launch(dy, dtype, x, gamma, mean, rstd, dres, dx, dx_lp, ws, M, D, (hipStream_t)stream);
}

extern "C" int pa_layernorm_bwd(const void* dy, int dtype, const float* x, const float* gamma,
                                const float* mean, const float* rstd, const float* dres, float* dx,
                                void* dx_lp, float* dgamma, float* dbeta, float* dcolsum, int accumulate, float* ws,
                                int M, int D, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !ws || M <= 0 || D <= 0) return PA_EINVAL;
    if (D % 4 || D > LN_MAXV * 256) return PA_EUNSUPPORTED;
    if (dtype != PA_BF16 && dtype != PA_F32) return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc = ln_bwd_launch(dy, dtype, x, gamma, mean, rstd, dres, dx, dx_lp, ws, M, D, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((unsigned)cdiv(3 * D, 16)), dim3(256), 0, st, ws, ln_bwd_blocks(M), D, dgamma, dbeta, dcolsum, accumulate);
    return check_launch();
}

static int ln_bwd_launch(const void* dy, int dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* dres, float* dx, void* dx_lp, float* ws, int M, int D, hipStream_t st) {
    const int nblk = ln_bwd_blocks(M);
    const size_t lds = (size_t)12 * D * sizeof(float);
    const int nvl = (int)cdiv(D, 256);
#define PA_LN_BWD(V)                                                                                              \
    do {                                                                                                          \
        if (dtype == PA_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16, V>), dim3(nblk), dim3(256), lds, st, (const bf16*)dy, x, gamma, mean, rstd, dres, dx, (bf16*)dx_lp, ws, M, D); \
        else hipLaunchKernelGGL((ln_bwd_kernel<float, V>), dim3(nblk), dim3(256), lds, st, (const float*)dy, x, gamma, mean, rstd, dres, dx, (float*)dx_lp, ws, M, D); \
    } while (0)
    if (nvl <= 1) PA_LN_BWD(1); else if (nvl == 2) PA_LN_BWD(2); else if (nvl == 3) PA_LN_BWD(3);
    else if (nvl == 4) PA_LN_BWD(4); else PA_LN_BWD(8);
#undef PA_LN_BWD
    return check_launch();
}
