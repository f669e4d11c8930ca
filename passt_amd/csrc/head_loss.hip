// Classification head and loss.  Reference: PaSST.forward_features :570-574 (final norm, only the
// two prefix tokens are consumed), PaSST.forward :583-585 (mean of the two, head = LayerNorm(1e-5) +
// Linear, :463-464) and the caller's BCE-with-logits mean (ex_audioset.py:184-186).
// Tiny next to the blocks (B rows): plain f32 VALU kernels, one workgroup per batch row.
#include "pa_common.h"

namespace pa {

static constexpr int HEAD_MAXE = 8;   // elements per thread -> D <= 2048

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void head_pre_fwd_kernel(const float* __restrict__ x, int Ntok, int D,
                                                           const float* __restrict__ ng, const float* __restrict__ nb, float eps_n,
                                                           const float* __restrict__ hg, const float* __restrict__ hb, float eps_h,
                                                           float* __restrict__ feat, float* __restrict__ hn, float* __restrict__ stats) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x0 = x + (int64_t)b * Ntok * D;
    const float* x1 = x0 + D;
    float a0[HEAD_MAXE], a1[HEAD_MAXE], f[HEAD_MAXE];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        a0[i] = d < D ? x0[d] : 0.f; a1[i] = d < D ? x1[d] : 0.f;
        s0 += a0[i]; s1 += a1[i];
    }
    const float m0 = block_sum(s0, red) / D, m1 = block_sum(s1, red) / D;
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        if (d < D) { v0 += (a0[i] - m0) * (a0[i] - m0); v1 += (a1[i] - m1) * (a1[i] - m1); }
    }
    const float r0 = rsqrtf(block_sum(v0, red) / D + eps_n), r1 = rsqrtf(block_sum(v1, red) / D + eps_n);
    float sf = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        if (d < D) {
            const float y0 = (a0[i] - m0) * r0 * ng[d] + nb[d], y1 = (a1[i] - m1) * r1 * ng[d] + nb[d];
            f[i] = (y0 + y1) / 2;
            feat[(int64_t)b * D + d] = f[i];
            sf += f[i];
        } else f[i] = 0.f;
    }
    const float mh = block_sum(sf, red) / D;
    float vh = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        if (d < D) vh += (f[i] - mh) * (f[i] - mh);
    }
    const float rh = rsqrtf(block_sum(vh, red) / D + eps_h);
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        if (d < D) hn[(int64_t)b * D + d] = (f[i] - mh) * rh * hg[d] + hb[d];
    }
    if (tid == 0) {
        float* s = stats + (int64_t)b * 6;
        s[0] = m0; s[1] = r0; s[2] = m1; s[3] = r1; s[4] = mh; s[5] = rh;
    }
}

__global__ __launch_bounds__(256) void head_pre_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ dfeat,
                                                           const float* __restrict__ x, const float* __restrict__ feat,
                                                           int Ntok, int D, const float* __restrict__ ng,
                                                           const float* __restrict__ hg, const float* __restrict__ stats,
                                                           float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* s = stats + (int64_t)b * 6;
    const float m0 = s[0], r0 = s[1], m1 = s[2], r1 = s[3], mh = s[4], rh = s[5];
    const float* x0 = x + (int64_t)b * Ntok * D;
    const float* x1 = x0 + D;
    float* pr = part + (int64_t)b * 4 * D;
    float xh[HEAD_MAXE], gy[HEAD_MAXE];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        xh[i] = 0.f; gy[i] = 0.f;
        if (d < D) {
            const float g = dhn[(int64_t)b * D + d];
            xh[i] = (feat[(int64_t)b * D + d] - mh) * rh;
            gy[i] = g * hg[d];
            pr[0 * D + d] = g * xh[i];      // d head.0.weight partial
            pr[1 * D + d] = g;              // d head.0.bias partial
            c1 += gy[i]; c2 += gy[i] * xh[i];
        }
    }
    c1 = block_sum(c1, red) / D; c2 = block_sum(c2, red) / D;
    float dy[HEAD_MAXE], xa[HEAD_MAXE], xb[HEAD_MAXE];
    float p1 = 0.f, p2 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        dy[i] = 0.f; xa[i] = 0.f; xb[i] = 0.f;
        if (d < D) {
            float df = rh * (gy[i] - c1 - xh[i] * c2);
            if (dfeat) df += dfeat[(int64_t)b * D + d];
            dy[i] = 0.5f * df;                                   // d y0 = d y1
            xa[i] = (x0[d] - m0) * r0; xb[i] = (x1[d] - m1) * r1;
            pr[2 * D + d] = dy[i] * (xa[i] + xb[i]);             // d norm.weight partial
            pr[3 * D + d] = 2.f * dy[i];                         // d norm.bias partial
            const float g = dy[i] * ng[d];
            p1 += g; p2 += g * xa[i]; q1 += g; q2 += g * xb[i];
        }
    }
    p1 = block_sum(p1, red) / D; p2 = block_sum(p2, red) / D;
    q1 = p1; q2 = block_sum(q2, red) / D;
    float* dx0 = dx + (int64_t)b * Ntok * D;
#pragma unroll
    for (int i = 0; i < HEAD_MAXE; ++i) {
        const int d = tid + 256 * i;
        if (d < D) {
            const float g = dy[i] * ng[d];
            dx0[d] = r0 * (g - p1 - xa[i] * p2);
            dx0[D + d] = r1 * (g - q1 - xb[i] * q2);
        }
    }
    // every other token row receives no gradient from the head
    const int64_t rest = (int64_t)(Ntok - 2) * D;
    float4* z = (float4*)(dx0 + 2 * D);
    for (int64_t i = tid; i < rest / 4; i += 256) z[i] = make_float4(0, 0, 0, 0);
}

// y[b][c] = x[b] . W[c] + bias[c]; one workgroup per class c, the 4 waves split the batch rows
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int C, int D) {
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* w = W + (int64_t)c * D;
    const float bc = bias ? bias[c] : 0.f;
    for (int b = wave; b < B; b += 4) {
        const float* xr = x + (int64_t)b * D;
        float s0 = 0.f, s1 = 0.f;
        int d = lane;
        for (; d + 64 < D; d += 128) { s0 += xr[d] * w[d]; s1 += xr[d + 64] * w[d + 64]; }
        for (; d < D; d += 64) s0 += xr[d] * w[d];
        const float s = wave_sum(s0 + s1);
        if (lane == 0) y[(int64_t)b * C + c] = s + bc;
    }
}
// the same for D = 64 NV (the model widths): no predicates, the class row in registers, four batch rows in flight per wave
// (round 5: one row at a time is 16 dependent L2 round trips per wave at B = 64; same summation order, bit-identical)
template <int NV>
__global__ __launch_bounds__(256) void linear_fwd_rows4_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ y, int B, int C) {
    constexpr int D = 64 * NV;
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float bc = bias ? bias[c] : 0.f;
    float wr[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) wr[i] = W[(int64_t)c * D + lane + 64 * i];
    auto dot = [&](int b) {
        const float* xr = x + (int64_t)min(b, B - 1) * D + lane;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i + 1 < NV; i += 2) { s0 += xr[64 * i] * wr[i]; s1 += xr[64 * (i + 1)] * wr[i + 1]; }
        if (NV & 1) s0 += xr[64 * (NV - 1)] * wr[NV - 1];
        return s0 + s1;
    };
    for (int b = wave; b < B; b += 16) {
        const float p0 = dot(b), p1 = dot(b + 4), p2 = dot(b + 8), p3 = dot(b + 12);     // rows past B repeat row B - 1, not stored
        const float r0 = wave_sum(p0), r1 = wave_sum(p1), r2 = wave_sum(p2), r3 = wave_sum(p3);
        if (lane == 0) {
            y[(int64_t)b * C + c] = r0 + bc;
            if (b + 4 < B) y[(int64_t)(b + 4) * C + c] = r1 + bc;
            if (b + 8 < B) y[(int64_t)(b + 8) * C + c] = r2 + bc;
            if (b + 12 < B) y[(int64_t)(b + 12) * C + c] = r3 + bc;
        }
    }
}
// dx[b][d] = sum_c dy[b][c] W[c][d]: workgroup = (64 columns d, one row b); the 4 waves split the classes (W rows read
// as 256-byte runs, dy[b][c] is wave-uniform), partial sums meet in LDS
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                            float* __restrict__ dx, int B, int C, int D) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, d = blockIdx.x * 64 + lane;
    const float* dyr = dy + (int64_t)b * C;
    float s = 0.f;
    if (d < D) {
#pragma unroll 4
        for (int c = wave; c < C; c += 4) s += dyr[c] * W[(int64_t)c * D + d];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && d < D) dx[(int64_t)b * D + d] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}
__global__ void linear_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dW,
                                     float* __restrict__ db, int accumulate, int B, int C, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)C * D) {
        const int c = (int)(i / D), d = (int)(i % D);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(int64_t)b * C + c] * x[(int64_t)b * D + d];
        dW[i] = (accumulate ? dW[i] : 0.f) + s;
    } else if (i < (int64_t)C * D + C) {
        const int c = (int)(i - (int64_t)C * D);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(int64_t)b * C + c];
        db[c] = (accumulate ? db[c] : 0.f) + s;
    }
}

// BCE with logits: l = max(z,0) - z*y + log1p(exp(-|z|)); dl/dz = sigmoid(z) - y
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, const float* __restrict__ y, int64_t n,
                                                  float gscale, float* __restrict__ dz, float* __restrict__ partial) {
    __shared__ float red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (i < n) {
        const float zi = z[i], yi = y[i];
        l = fmaxf(zi, 0.f) - zi * yi + log1pf(expf(-fabsf(zi)));
        const float sg = 1.f / (1.f + expf(-zi));
        dz[i] = (sg - yi) * gscale;
    }
    const float s = block_sum(l, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void bce_final_kernel(const float* __restrict__ partial, int nblk, float inv_n, float* __restrict__ loss) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = s * inv_n;
}

// Cross-entropy with mixup targets (ex_esc50.py:159-165): one wave per row.
//   loss_b = lam_b * CE(z_b, y_b) + (1 - lam_b) * CE(z_b, y2_b),  CE(z, y) = logsumexp(z) - z[y]
//   d loss_b / d z = softmax(z_b) - lam_b * onehot(y_b) - (1 - lam_b) * onehot(y2_b)
__global__ __launch_bounds__(256) void ce_mixup_kernel(const float* __restrict__ z, const int32_t* __restrict__ y,
                                                       const int32_t* __restrict__ y2, const float* __restrict__ lam,
                                                       int B, int C, float gscale, float* __restrict__ dz,
                                                       float* __restrict__ row_loss) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* zr = z + (int64_t)b * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, zr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(zr[c] - m);
    s = wave_sum(s);
    const float lse = m + __logf(s);
    const int t1 = y[b], t2 = y2 ? y2[b] : t1;
    const float l1 = lam ? lam[b] : 1.0f;
    if (lane == 0) row_loss[b] = l1 * (lse - zr[t1]) + (1.0f - l1) * (lse - zr[t2]);
    const float inv = 1.0f / s;
    for (int c = lane; c < C; c += 64) {
        float g = __expf(zr[c] - m) * inv;
        if (c == t1) g -= l1;
        if (c == t2) g -= 1.0f - l1;
        dz[(int64_t)b * C + c] = g * gscale;
    }
}

}  // namespace pa

using namespace pa;

extern "C" int pa_head_pre_fwd(const float* x, int B, int Ntok, int D, const float* norm_g, const float* norm_b,
                               float eps_norm, const float* hg, const float* hb, float eps_head, float* feat, float* hn,
                               float* stats, void* stream) {
    if (!x || !norm_g || !norm_b || !hg || !hb || !feat || !hn || !stats || B <= 0 || Ntok < 2) return PA_EINVAL;
    if (D > 256 * HEAD_MAXE) return PA_EUNSUPPORTED;
    hipLaunchKernelGGL(head_pre_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, Ntok, D, norm_g, norm_b,
                       eps_norm, hg, hb, eps_head, feat, hn, stats);
    return check_launch();
}

extern "C" int pa_head_pre_bwd(const float* dhn, const float* dfeat, const float* x, const float* feat, int B, int Ntok,
                               int D, const float* norm_g, const float* hg, const float* stats, float* dx, float* part,
                               void* stream) {
    if (!dhn || !x || !feat || !norm_g || !hg || !stats || !dx || !part || B <= 0 || Ntok < 2) return PA_EINVAL;
    if (D > 256 * HEAD_MAXE || D % 4) return PA_EUNSUPPORTED;
    hipLaunchKernelGGL(head_pre_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dhn, dfeat, x, feat, Ntok, D,
                       norm_g, hg, stats, dx, part);
    return check_launch();
}

extern "C" int pa_linear_f32_fwd(const float* x, const float* W, const float* b, float* y, int B, int C, int D, void* stream) {
    if (!x || !W || !y || B <= 0 || C <= 0 || D <= 0) return PA_EINVAL;
    if (D == 768) hipLaunchKernelGGL(linear_fwd_rows4_kernel<12>, dim3(C), dim3(256), 0, (hipStream_t)stream, x, W, b, y, B, C);
    else if (D == 1024) hipLaunchKernelGGL(linear_fwd_rows4_kernel<16>, dim3(C), dim3(256), 0, (hipStream_t)stream, x, W, b, y, B, C);
    else hipLaunchKernelGGL(linear_fwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, W, b, y, B, C, D);
    return check_launch();
}

extern "C" int pa_linear_f32_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db,
                                 int accumulate, int B, int C, int D, void* stream) {
    if (!dy || !x || !W || !dx || !dW || !db || B <= 0 || C <= 0 || D <= 0) return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3((unsigned)cdiv(D, 64), (unsigned)B), dim3(256), 0, st, dy, W, dx, B, C, D);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3((unsigned)cdiv((int64_t)C * D + C, 256)), dim3(256), 0, st, dy, x, dW, db, accumulate, B, C, D);
    return check_launch();
}

extern "C" int pa_bce_fwd_bwd(const float* logits, const float* target, int B, int C, float grad_scale, float* loss,
                              float* dlogits, float* ws, void* stream) {
    if (!logits || !target || !loss || !dlogits || !ws || B <= 0 || C <= 0) return PA_EINVAL;
    const int64_t n = (int64_t)B * C;
    const int nblk = (int)cdiv(n, 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_kernel, dim3(nblk), dim3(256), 0, st, logits, target, n, grad_scale / (float)n, dlogits, ws);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, st, ws, nblk, 1.0f / (float)n, loss);
    return check_launch();
}

extern "C" int pa_ce_mixup_fwd_bwd(const float* logits, const int32_t* target, const int32_t* target2, const float* lam,
                                   int B, int C, float grad_scale, float* loss, float* dlogits, float* ws, void* stream) {
    if (!logits || !target || !loss || !dlogits || !ws || B <= 0 || C <= 0) return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_mixup_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, st, logits, target, target2, lam, B, C,
                       grad_scale / (float)B, dlogits, ws);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(256), 0, st, ws, B, 1.0f / (float)B, loss);
    return check_launch();
}
