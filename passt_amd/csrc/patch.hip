// Patch embedding around the im2col GEMM: gather of the kept patches, positional table, prefix
// tokens, and the backward reductions.  Reference: PatchEmbed.forward (models/passt.py:318-328) and
// PaSST.forward_features :508-564.  "Gather first": Patchout indices (drawn on the host with the
// reference's own torch CPU RNG calls) select the patches BEFORE the projection, so only
// (F-s_f)(T-s_t)-u of the F*T patches are ever embedded.
#include <algorithm>

#include "pa_common.h"

namespace pa {

// one workgroup (P*P threads) per kept patch: cols[(b*Np+p)][i*P+j] = x[b][f*fs+i][t*ts+j]
template <typename T>
__global__ void patch_gather_kernel(const float* __restrict__ x, int F, int Tt, const int32_t* __restrict__ pf,
                                    const int32_t* __restrict__ pt, int Np, int P, int fs, int ts, T* __restrict__ cols) {
    const int p = blockIdx.x, b = blockIdx.y;
    const int i = threadIdx.x / P, j = threadIdx.x % P;
    const int f = pf[p] * fs + i, t = pt[p] * ts + j;
    const float v = x[((int64_t)b * F + f) * Tt + t];
    cols[((int64_t)b * Np + p) * (P * P) + threadIdx.x] = from_f32<T>(v);
}

__global__ void patch_pos_table_kernel(const float* __restrict__ bias, const float* __restrict__ tpos, int Tpe,
                                       const float* __restrict__ fpos, int Fpe, const int32_t* __restrict__ pf,
                                       const int32_t* __restrict__ pt, int Np, int toff, int D, float* __restrict__ table,
                                       const float* __restrict__ cls, const float* __restrict__ dist,
                                       const float* __restrict__ npe, float* __restrict__ tok, int B, int Ntok) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_table = (int64_t)Np * D, n_tok = (int64_t)B * 2 * D;
    if (i < n_table) {
        const int p = (int)(i / D), d = (int)(i % D);
        table[i] = bias[d] + tpos[(int64_t)d * Tpe + toff + pt[p]] + fpos[(int64_t)d * Fpe + pf[p]];
    } else if (i < n_table + n_tok) {
        const int64_t k = i - n_table;
        const int b = (int)(k / (2 * D)), r = (int)((k / D) % 2), d = (int)(k % D);
        tok[((int64_t)b * Ntok + r) * D + d] = (r == 0 ? cls[d] : dist[d]) + npe[r * D + d];
    }
}

// gsum[n][d] = sum_b dtok[b][n][d]: 16 bytes per thread, four clips in flight (round 5: the scalar one-clip-at-a-time loop ran
// at 2.9 TB/s)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ dtok, int B, int64_t per, float* __restrict__ gsum) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= per) return;
    if (((per | i) & 3) == 0) {
        float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0;
        auto acc = [](float4& s, const float4& v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
        int b = 0;
        for (; b + 4 <= B; b += 4) {
            const float4 v0 = *(const float4*)(dtok + (int64_t)b * per + i), v1 = *(const float4*)(dtok + (int64_t)(b + 1) * per + i);
            const float4 v2 = *(const float4*)(dtok + (int64_t)(b + 2) * per + i), v3 = *(const float4*)(dtok + (int64_t)(b + 3) * per + i);
            acc(s0, v0); acc(s1, v1); acc(s2, v2); acc(s3, v3);
        }
        for (; b < B; ++b) acc(s0, *(const float4*)(dtok + (int64_t)b * per + i));
        acc(s0, s1); acc(s2, s3); acc(s0, s2);
        *(float4*)(gsum + i) = s0;
        return;
    }
    for (int64_t k = i; k < per && k < i + 4; ++k) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dtok[(int64_t)b * per + k];
        gsum[k] = s;
    }
}

// Parameter gradients of the patch stage from gsum[2 + Np][D]: conv bias (all patches), time / frequency position embeddings
// (the patches of one time / frequency slot), cls / dist tokens and their position rows.  Workgroup = one slot x 16 channels
// x 16 patch groups: every thread scans Np / 16 patches for its slot (the test is uniform over the 16 channel lanes) and the
// partial sums meet in LDS.  (Round 5: one thread per (slot, channel) walking all Np patches was 48 us -- 472 dependent
// L2 round trips on three workgroups for the bias, ~40 per thread for a frequency slot.)
__global__ __launch_bounds__(256) void patch_param_grads_kernel(const float* __restrict__ gsum, int D, const int32_t* __restrict__ pf,
                                         const int32_t* __restrict__ pt, int Np, int toff, int Tpe, int Fpe,
                                         float* __restrict__ d_cls, float* __restrict__ d_dist, float* __restrict__ d_npe,
                                         float* __restrict__ d_bias, float* __restrict__ d_tpos, float* __restrict__ d_fpos,
                                         int accumulate) {
    __shared__ float red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int slot = blockIdx.y;                       // 0: bias + prefix tokens; 1 .. Tpe: time; Tpe + 1 .. Tpe + Fpe: frequency
    const int d = blockIdx.x * 16 + cx;
    auto put = [&](float* p, float v) { *p = (accumulate ? *p : 0.f) + v; };
    const int32_t* key = slot == 0 ? nullptr : (slot <= Tpe ? pt : pf);
    const int want = slot == 0 ? 0 : (slot <= Tpe ? slot - 1 - toff : slot - 1 - Tpe);
    float s0 = 0.f, s1 = 0.f;
    if (d < D) {
        int p = ry;
        for (; p + 16 < Np; p += 32) {                 // two patches per step: two loads in flight
            const bool m0 = !key || key[p] == want, m1 = !key || key[p + 16] == want;
            if (m0) s0 += gsum[(int64_t)(2 + p) * D + d];
            if (m1) s1 += gsum[(int64_t)(2 + p + 16) * D + d];
        }
        if (p < Np && (!key || key[p] == want)) s0 += gsum[(int64_t)(2 + p) * D + d];
        // (eight keys requested together + unconditional loads of a clamped row times a 0 / 1 weight: 25 us against 21.5)
    }
    red[ry][cx] = s0 + s1;
    __syncthreads();
    if (ry == 0 && d < D) {
        float s = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) s += red[y][cx];
        if (slot == 0) {
            put(d_bias + d, s);
            put(d_cls + d, gsum[d]);
            put(d_dist + d, gsum[D + d]);
            put(d_npe + d, gsum[d]);
            put(d_npe + D + d, gsum[D + d]);
        } else if (slot <= Tpe) {
            put(d_tpos + (int64_t)d * Tpe + (slot - 1), s);
        } else {
            put(d_fpos + (int64_t)d * Fpe + (slot - 1 - Tpe), s);
        }
    }
}

// dpatch[(b*Np+p)][d] = dtok[b][2+p][d]; 8 elements per thread (D % 8 == 0 fast path: two 16-byte loads, one 16-byte
// bf16 store, one row decomposition per vector), scalar otherwise
template <typename T>
__global__ void patch_rows_kernel(const float* __restrict__ dtok, int Ntok, int D, int Np, T* __restrict__ dpatch, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if ((D & 7) == 0) {
        const int dv = D >> 3;
        const int64_t nv = n >> 3;
        for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
            const int64_t row = v / dv;
            const int d = (int)(v - row * dv) * 8;
            const int64_t b = row / Np, p = row - b * Np;
            float x[8];
            load8<float>(dtok + (b * Ntok + 2 + p) * D + d, x);
            store8<T>(dpatch + row * D + d, x);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t row = i / D;
        const int d = (int)(i % D);
        const int64_t b = row / Np, p = row % Np;
        dpatch[i] = from_f32<T>(dtok[(b * Ntok + 2 + p) * D + d]);
    }
}

}  // namespace pa

using namespace pa;

extern "C" int pa_patch_gather(const float* x, int B, int F, int T, const int32_t* patch_f, const int32_t* patch_t,
                               int Np, int P, int fstride, int tstride, void* cols, int dtype, void* stream) {
    if (!x || !patch_f || !patch_t || !cols || B <= 0 || Np <= 0 || P <= 0) return PA_EINVAL;
    if (P * P > 1024) return PA_EUNSUPPORTED;
    dim3 grid((unsigned)Np, (unsigned)B);
    if (dtype == PA_BF16) hipLaunchKernelGGL(patch_gather_kernel<bf16>, grid, dim3(P * P), 0, (hipStream_t)stream, x, F, T, patch_f, patch_t, Np, P, fstride, tstride, (bf16*)cols);
    else if (dtype == PA_F32) hipLaunchKernelGGL(patch_gather_kernel<float>, grid, dim3(P * P), 0, (hipStream_t)stream, x, F, T, patch_f, patch_t, Np, P, fstride, tstride, (float*)cols);
    else return PA_EINVAL;
    return check_launch();
}

extern "C" int pa_patch_pos_table(const float* bias, const float* time_pos, int Tpe, const float* freq_pos, int Fpe,
                                  const int32_t* patch_f, const int32_t* patch_t, int Np, int toff, int D, float* table,
                                  const float* cls, const float* dist, const float* npe, float* tok, int B, int Ntok,
                                  void* stream) {
    if (!bias || !time_pos || !freq_pos || !patch_f || !patch_t || !table || !cls || !dist || !npe || !tok) return PA_EINVAL;
    const int64_t n = (int64_t)Np * D + (int64_t)B * 2 * D;
    hipLaunchKernelGGL(patch_pos_table_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, bias,
                       time_pos, Tpe, freq_pos, Fpe, patch_f, patch_t, Np, toff, D, table, cls, dist, npe, tok, B, Ntok);
    return check_launch();
}

extern "C" int pa_patch_bwd(const float* dtok, int B, int Ntok, int D, const int32_t* patch_f, const int32_t* patch_t,
                            int Np, int toff, int Tpe, int Fpe, float* gsum, float* d_cls, float* d_dist, float* d_npe,
                            float* d_bias, float* d_time_pos, float* d_freq_pos, int accumulate, void* dpatch, int dtype,
                            void* stream) {
    if (!dtok || !patch_f || !patch_t || !gsum || !d_cls || !d_dist || !d_npe || !d_bias || !d_time_pos || !d_freq_pos || !dpatch)
        return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t per = (int64_t)Ntok * D;
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)cdiv(per, 1024)), dim3(256), 0, st, dtok, B, per, gsum);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(patch_param_grads_kernel, dim3((unsigned)cdiv(D, 16), (unsigned)(1 + Tpe + Fpe)), dim3(256), 0, st, gsum, D, patch_f,
                       patch_t, Np, toff, Tpe, Fpe, d_cls, d_dist, d_npe, d_bias, d_time_pos, d_freq_pos, accumulate);
    rc = check_launch();
    if (rc) return rc;
    const int64_t total = (int64_t)B * Np * D;
    const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 8192);
    if (dtype == PA_BF16) hipLaunchKernelGGL(patch_rows_kernel<bf16>, dim3(blocks), dim3(256), 0, st, dtok, Ntok, D, Np, (bf16*)dpatch, total);
    else if (dtype == PA_F32) hipLaunchKernelGGL(patch_rows_kernel<float>, dim3(blocks), dim3(256), 0, st, dtok, Ntok, D, Np, (float*)dpatch, total);
    else return PA_EINVAL;
    return check_launch();
}
