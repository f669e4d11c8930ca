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

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] -= lr * g[i];
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
    const int blocks = (int)std::min<int64_t>(cdiv(n, 256), 8192);
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sqrtf(bc2));
    return check_launch();
}

extern "C" int pa_sgd(float* p, const float* g, int64_t n, float lr, void* stream) {
    if (!p || !g || n <= 0) return PA_EINVAL;
    const int blocks = (int)std::min<int64_t>(cdiv(n, 256), 8192);
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, n, lr);
    return check_launch();
}
