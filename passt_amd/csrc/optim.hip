// Training-step glue of the caller: spectrogram mixup (ex_audioset.py:173-177, helpers/mixup.py),
// AdamW (ex_audioset.py:104-109) and plain SGD (model_speed_test, ex_audioset.py:392) on flat f32
// buffers.  Pure HBM streaming kernels: float4 grid-stride loops.
#include <algorithm>

#include "pa_common.h"

namespace pa {

__global__ void mixup_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm, const float* __restrict__ lam,
                             float* __restrict__ out, int64_t per4) {
    const int b = blockIdx.y;
    const float l = lam[b];
    const float4* xa = (const float4*)x + (int64_t)b * per4;
    const float4* xb = (const float4*)x + (int64_t)perm[b] * per4;
    float4* o = (float4*)out + (int64_t)b * per4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = xa[i], c = xb[i];
        o[i] = make_float4(a.x * l + c.x * (1.f - l), a.y * l + c.y * (1.f - l), a.z * l + c.z * (1.f - l), a.w * l + c.w * (1.f - l));
    }
}

__global__ void mixup_scalar_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm, const float* __restrict__ lam,
                                    float* __restrict__ out, int64_t per) {
    const int b = blockIdx.y;
    const float l = lam[b];
    const float* xa = x + (int64_t)b * per;
    const float* xb = x + (int64_t)perm[b] * per;
    float* o = out + (int64_t)b * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = xa[i] * l + xb[i] * (1.f - l);
}

// One definite sequence of roundings (no contraction left to the compiler: it fused a * b + c differently in the kernels this is
// inlined into -- pa_adamw and pa_adamw_stage differed by an ulp): the two moment updates are ONE fused multiply-add each, nothing
// else is fused.  pa_adamw, pa_adamw_dev and pa_adamw_stage are bit-identical to each other by construction.
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                          float wd, float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
    const float pi = p * (1.f - lr * wd);
    const float gm = (1.f - b1) * g, gv = ((1.f - b2) * g) * g;
    m = __builtin_fmaf(b1, m, gm);
    v = __builtin_fmaf(b2, v, gv);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    const float step = (lr / bc1) * (m / denom);
    p = pi - step;
}
// 16-byte accesses (n4 = n / 4 when all four pointers are 16-byte aligned, else 0) + scalar tail
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n4, int64_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n4; i += stride) {
        f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 gv = ((const f32x4*)g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];
            adamw_one(pe, gv[e], me, ve, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
            pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
    }
    for (int64_t i = n4 * 4 + t0; i < n; i += stride) adamw_one(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, bc2_sqrt);
}

// the same update with the hyper-parameters of the step read from device memory ([lr, beta1, beta2, eps, weight_decay,
// 1 - beta1^t, sqrt(1 - beta2^t)]): a launch whose arguments never change, i.e. one that can sit in a captured hipGraph
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n4, int64_t n, const float* __restrict__ hy) {
    const float lr = hy[0], b1 = hy[1], b2 = hy[2], eps = hy[3], wd = hy[4], bc1 = hy[5], bc2_sqrt = hy[6];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n4; i += stride) {
        f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 gv = ((const f32x4*)g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];
            adamw_one(pe, gv[e], me, ve, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
            pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
    }
    for (int64_t i = n4 * 4 + t0; i < n; i += stride) adamw_one(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, bc2_sqrt);
}

// ---- AdamW + weight staging in ONE pass (round 6).  The step used to end with adamw_kernel writing every updated parameter and
// the next forward beginning with stage_weights_kernel reading all of them back to make the GEMM-ready copies (bf16 W[out][in]
// for the forward, bf16 W^T[in][out] for the input gradients): one extra 345 MB read + a launch per step.  Here the optimizer
// launch of a bucket walks a descriptor table of the bucket's parameters: a matrix with copies is processed in 64 x 64 tiles
// (256 B of each row per 16 lanes; the updated values go out as f32, as the straight cast, and -- through an LDS tile -- as the
// transposed cast), everything else in runs of 4096 elements.  Same adamw_one() as adamw_kernel: parameters, moments and both
// copies are bit-identical to adamw_kernel + stage_weights_kernel (tests/test_gpu_model.py).
struct AdamwHyper { float lr, b1, b2, eps, wd, bc1, bc2_sqrt; };
// Cache policy of the matrix path (A/B build knob, round 6 sweep profiles/r06_cache_policy.txt): bit 0 = parameters, moments and
// gradients (each touched once per step) as non-temporal loads / stores, bit 1 = the GEMM-ready copies as non-temporal stores
#ifndef PA_OPT_NT
#ifdef PA_NO_CACHE_POLICY
#define PA_OPT_NT 0
#else
#define PA_OPT_NT 1        // (with the slab loads of the finishing reduction: -0.2 % of the step; the copies: +- 0)
#endif
#endif
template <typename V> __device__ __forceinline__ V opt_ld(const V* p) {
#if PA_OPT_NT & 1
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename V> __device__ __forceinline__ void opt_st(V* p, const V& v) {
#if PA_OPT_NT & 1
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <typename V> __device__ __forceinline__ void opt_st_copy(V* p, const V& v) {
#if PA_OPT_NT & 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

template <typename TO>
__global__ __launch_bounds__(256) void adamw_stage_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, const pa_adamw_stage_desc* __restrict__ descs,
                                                          int n_desc, const float* __restrict__ hy_dev, AdamwHyper hv) {
    __shared__ float tile[64][65];
    __shared__ int s_entry;
    const int bid = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {                // one wave scans the (short) table: the last entry with tile_begin <= bid
        int found = -1;
        for (int e = tid; e < n_desc; e += 64)
            if (descs[e].tile_begin <= bid) found = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) found = max(found, __shfl_xor(found, o, 64));
        if (tid == 0) s_entry = found;
    }
    __syncthreads();
    const pa_adamw_stage_desc d = descs[s_entry];
    AdamwHyper h = hv;
    if (hy_dev) { h.lr = hy_dev[0]; h.b1 = hy_dev[1]; h.b2 = hy_dev[2]; h.eps = hy_dev[3]; h.wd = hy_dev[4]; h.bc1 = hy_dev[5]; h.bc2_sqrt = hy_dev[6]; }
    float* pp = p + d.offset;
    const float* gp = g + d.offset;
    float* mp = m + d.offset;
    float* vp = v + d.offset;
    const int t = bid - d.tile_begin;
    const bool aligned = ((((uintptr_t)pp | (uintptr_t)gp | (uintptr_t)mp | (uintptr_t)vp) & 15) == 0);
    if (!d.dst && !d.dst_t) {
        // a parameter without copies (biases, LayerNorm, positional tables, the f32 head): a run of 4096 consecutive elements
        const int64_t n = (int64_t)d.rows * d.cols, i0 = (int64_t)t * 4096, i1 = min(i0 + 4096, n);
        if (aligned) {
            for (int64_t i = i0 + tid * 4; i < i1; i += 1024) {
                if (i + 4 <= i1) {
                    f32x4 pv = *(f32x4*)(pp + i), mv = *(f32x4*)(mp + i), vv = *(f32x4*)(vp + i);
                    const f32x4 gv = *(const f32x4*)(gp + i);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = pv[e], me = mv[e], ve = vv[e];
                        adamw_one(pe, gv[e], me, ve, h.lr, h.b1, h.b2, h.eps, h.wd, h.bc1, h.bc2_sqrt);
                        pv[e] = pe; mv[e] = me; vv[e] = ve;
                    }
                    *(f32x4*)(pp + i) = pv; *(f32x4*)(mp + i) = mv; *(f32x4*)(vp + i) = vv;
                } else {
                    for (int64_t k = i; k < i1; ++k) adamw_one(pp[k], gp[k], mp[k], vp[k], h.lr, h.b1, h.b2, h.eps, h.wd, h.bc1, h.bc2_sqrt);
                }
            }
        } else {
            for (int64_t i = i0 + tid; i < i1; i += 256) adamw_one(pp[i], gp[i], mp[i], vp[i], h.lr, h.b1, h.b2, h.eps, h.wd, h.bc1, h.bc2_sqrt);
        }
        return;
    }
    const int tiles_c = (d.cols + 63) >> 6;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    TO* dst = (TO*)d.dst;
    TO* dst_t = (TO*)d.dst_t;
    if (((d.rows | d.cols) & 3) == 0 && aligned) {
        // every PaSST weight: 16-byte accesses, 256 contiguous bytes of a row per 16 lanes
        const int lr = tid >> 4, l4 = (tid & 15) * 4;
        // all sixteen loads of the thread's four row pieces first (the stores below would otherwise fence the next piece's loads:
        // the four pointers alias as far as the compiler knows), then the arithmetic and the stores
        f32x4 pv[4], mv[4], vv[4], gv[4];
        bool in[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + lr + 16 * i, c = c0 + l4;
            in[i] = r < d.rows && c < d.cols;
            const int64_t at = in[i] ? (int64_t)r * d.cols + c : 0;
            pv[i] = opt_ld((const f32x4*)(pp + at)); mv[i] = opt_ld((const f32x4*)(mp + at)); vv[i] = opt_ld((const f32x4*)(vp + at)); gv[i] = opt_ld((const f32x4*)(gp + at));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + lr + 16 * i, c = c0 + l4;
            if (in[i]) {
                const int64_t at = (int64_t)r * d.cols + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pv[i][e], me = mv[i][e], ve = vv[i][e];
                    adamw_one(pe, gv[i][e], me, ve, h.lr, h.b1, h.b2, h.eps, h.wd, h.bc1, h.bc2_sqrt);
                    pv[i][e] = pe; mv[i][e] = me; vv[i][e] = ve;
                }
                opt_st((f32x4*)(pp + at), pv[i]); opt_st((f32x4*)(mp + at), mv[i]); opt_st((f32x4*)(vp + at), vv[i]);
                if (dst) {
                    if constexpr (sizeof(TO) == 2) opt_st_copy((bf16x4*)(dst + at), bf16x4{(bf16)pv[i][0], (bf16)pv[i][1], (bf16)pv[i][2], (bf16)pv[i][3]});
                    else *(f32x4*)(dst + at) = pv[i];
                }
            } else {
                pv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[lr + 16 * i][l4 + k] = pv[i][k];
        }
        if (!dst_t) return;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + lr + 16 * i, r = r0 + l4;
            if (c < d.cols && r < d.rows) {
                const f32x4 w = {tile[l4][lr + 16 * i], tile[l4 + 1][lr + 16 * i], tile[l4 + 2][lr + 16 * i], tile[l4 + 3][lr + 16 * i]};
                if constexpr (sizeof(TO) == 2) opt_st_copy((bf16x4*)(dst_t + (int64_t)c * d.rows + r), bf16x4{(bf16)w[0], (bf16)w[1], (bf16)w[2], (bf16)w[3]});
                else *(f32x4*)(dst_t + (int64_t)c * d.rows + r) = w;
            }
        }
        return;
    }
    const int tx = tid & 63, ty = tid >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const bool ok = r < d.rows && c < d.cols;
        float w = 0.f;
        if (ok) {
            const int64_t at = (int64_t)r * d.cols + c;
            adamw_one(pp[at], gp[at], mp[at], vp[at], h.lr, h.b1, h.b2, h.eps, h.wd, h.bc1, h.bc2_sqrt);
            w = pp[at];
            if (dst) dst[at] = from_f32<TO>(w);
        }
        tile[i][tx] = w;
    }
    if (!dst_t) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < d.cols && r < d.rows) dst_t[(int64_t)c * d.rows + r] = from_f32<TO>(tile[tx][i]);
    }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] -= lr * g[i];
}

// mean over the L samples of the padded / truncated, gain-scaled clip; one 1024-thread block per clip
__global__ __launch_bounds__(1024) void wave_mean_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ len,
                                                         const float* __restrict__ amp, float* __restrict__ mean, int64_t L) {
    __shared__ float part[16];
    const int b = blockIdx.x;
    const int64_t n = min((int64_t)(len ? len[b] : ldx), L);
    const float* xb = x + (int64_t)b * ldx;
    float s = 0.f;
    for (int64_t t = threadIdx.x; t < n; t += 1024) s += xb[t];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float v = threadIdx.x < 16 ? part[threadIdx.x] : 0.f;
        v = wave_sum(v);
        if (threadIdx.x == 0) mean[b] = v * (amp ? amp[b] : 1.f) / (float)L;
    }
}

__global__ __launch_bounds__(256) void wave_mix_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ len,
                                                       const float* __restrict__ amp, const int32_t* __restrict__ shift,
                                                       const int32_t* __restrict__ partner, const float* __restrict__ lam,
                                                       const float* __restrict__ mean, float* __restrict__ out, int64_t L) {
    const int b = blockIdx.y;
    const int p = partner ? partner[b] : -1;
    auto clip = [&](int c, int64_t t) {             // sample t of clip c after gain, pad/truncate and roll
        int64_t src = t - (shift ? shift[c] : 0);
        src %= L;
        if (src < 0) src += L;
        const int64_t n = min((int64_t)(len ? len[c] : ldx), L);
        return src < n ? x[(int64_t)c * ldx + src] * (amp ? amp[c] : 1.f) : 0.f;
    };
    const float l = p >= 0 ? fmaxf(lam[b], 1.f - lam[b]) : 1.f;
    const float mb = p >= 0 ? mean[b] : 0.f, mp = p >= 0 ? mean[p] : 0.f;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < L; t += (int64_t)gridDim.x * blockDim.x) {
        float v = clip(b, t);
        if (p >= 0) v = (v - mb) * l + (clip(p, t) - mp) * (1.f - l);
        out[(int64_t)b * L + t] = v;
    }
}

__global__ void swa_kernel(float* __restrict__ avg, const float* __restrict__ p, int64_t n4, int64_t n, float inv) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 a = ((f32x4*)avg)[i];
        const f32x4 w = ((const f32x4*)p)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = inv == 1.0f ? w[e] : a[e] + (w[e] - a[e]) * inv;
        ((f32x4*)avg)[i] = a;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        avg[i] = inv == 1.0f ? p[i] : avg[i] + (p[i] - avg[i]) * inv;
}

}  // namespace pa

using namespace pa;

extern "C" int pa_mixup(const float* x, const int32_t* perm, const float* lam, float* out, int B, int64_t per_sample, void* stream) {
    if (!x || !perm || !lam || !out || B <= 0 || per_sample <= 0) return PA_EINVAL;
    if (per_sample % 4) {   // e.g. the (B, 527) target matrix: scalar path
        dim3 grid((unsigned)std::min<int64_t>(cdiv(per_sample, 256), 64), (unsigned)B);
        hipLaunchKernelGGL(mixup_scalar_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, perm, lam, out, per_sample);
        return check_launch();
    }
    const int64_t per4 = per_sample / 4;
    dim3 grid((unsigned)std::min<int64_t>(cdiv(per4, 256), 64), (unsigned)B);
    hipLaunchKernelGGL(mixup_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, perm, lam, out, per4);
    return check_launch();
}

extern "C" int pa_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return PA_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const int blocks = (int)std::min<int64_t>(cdiv(std::max<int64_t>(n4, 1), 256), 8192);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sqrtf(bc2));
    return check_launch();
}

extern "C" int pa_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, void* stream) {
    if (!p || !g || !m || !v || !hyper || n <= 0) return PA_EINVAL;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const int blocks = (int)std::min<int64_t>(cdiv(std::max<int64_t>(n4, 1), 256), 8192);
    hipLaunchKernelGGL(adamw_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n4, n, hyper);
    return check_launch();
}

extern "C" int pa_adamw_stage(float* p, const float* g, float* m, float* v, const pa_adamw_stage_desc* descs, int n_desc,
                              int total_items, int dtype, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              const float* hyper_dev, void* stream) {
    if (!p || !g || !m || !v || !descs || n_desc < 1 || total_items < 1 || (!hyper_dev && step < 1)) return PA_EINVAL;
    if (dtype != PA_BF16 && dtype != PA_F32) return PA_EINVAL;
    AdamwHyper h{lr, beta1, beta2, eps, weight_decay, 1.f, 1.f};
    if (!hyper_dev) {
        h.bc1 = 1.f - powf(beta1, (float)step);
        h.bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    }
    if (dtype == PA_BF16)
        hipLaunchKernelGGL(adamw_stage_kernel<bf16>, dim3(total_items), dim3(256), 0, (hipStream_t)stream, p, g, m, v, descs, n_desc, hyper_dev, h);
    else
        hipLaunchKernelGGL(adamw_stage_kernel<float>, dim3(total_items), dim3(256), 0, (hipStream_t)stream, p, g, m, v, descs, n_desc, hyper_dev, h);
    return check_launch();
}

extern "C" void pa_adamw_hyper(float lr, float beta1, float beta2, float eps, float weight_decay, int step, float* hyper7_host) {
    hyper7_host[0] = lr; hyper7_host[1] = beta1; hyper7_host[2] = beta2; hyper7_host[3] = eps; hyper7_host[4] = weight_decay;
    hyper7_host[5] = 1.f - powf(beta1, (float)step);
    hyper7_host[6] = sqrtf(1.f - powf(beta2, (float)step));
}

extern "C" int pa_wave_augment(const float* x, int B, int64_t ldx, const int32_t* len, const float* amp, const int32_t* shift,
                               const int32_t* partner, const float* lam, float* ws, float* out, int64_t L, void* stream) {
    if (!x || !out || B <= 0 || ldx <= 0 || L <= 0 || (partner && (!lam || !ws))) return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (partner) {
        hipLaunchKernelGGL(wave_mean_kernel, dim3(B), dim3(1024), 0, st, x, ldx, len, amp, ws, L);
        const int rc = check_launch();
        if (rc) return rc;
    }
    dim3 grid((unsigned)std::min<int64_t>(cdiv(L, 256 * 4), 1024), (unsigned)B);
    hipLaunchKernelGGL(wave_mix_kernel, grid, dim3(256), 0, st, x, ldx, len, amp, shift, partner, lam, ws, out, L);
    return check_launch();
}

extern "C" int pa_swa_update(float* avg, const float* p, int64_t n, int num_averaged, void* stream) {
    if (!avg || !p || n <= 0 || num_averaged < 0) return PA_EINVAL;
    const bool vec = (((uintptr_t)avg | (uintptr_t)p) & 15) == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const int blocks = (int)std::min<int64_t>(cdiv(std::max<int64_t>(n4, 1), 256), 8192);
    // the reference divides by (n + 1); a reciprocal multiply differs by <= 1 ulp of the increment
    hipLaunchKernelGGL(swa_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, avg, p, n4, n, 1.0f / (float)(num_averaged + 1));
    return check_launch();
}

extern "C" int pa_sgd(float* p, const float* g, int64_t n, float lr, void* stream) {
    if (!p || !g || n <= 0) return PA_EINVAL;
    const int blocks = (int)std::min<int64_t>(cdiv(n, 256), 8192);
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, n, lr);
    return check_launch();
}
