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

// gsum[n][d] = sum_b dtok[b][n][d]
__global__ void batch_sum_kernel(const float* __restrict__ dtok, int B, int64_t per, float* __restrict__ gsum) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dtok[(int64_t)b * per + i];
    gsum[i] = s;
}

__global__ void patch_param_grads_kernel(const float* __restrict__ gsum, int D, const int32_t* __restrict__ pf,
                                         const int32_t* __restrict__ pt, int Np, int toff, int Tpe, int Fpe,
                                         float* __restrict__ d_cls, float* __restrict__ d_dist, float* __restrict__ d_npe,
                                         float* __restrict__ d_bias, float* __restrict__ d_tpos, float* __restrict__ d_fpos,
                                         int accumulate) {
    // index space: [D bias+prefix] ++ [Tpe x D time] ++ [Fpe x D freq], the channel d fastest: every gsum read is a
    // coalesced row segment and the "does patch p belong to this time / frequency slot" test is wave-uniform
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n0 = D, n1 = (int64_t)D * Tpe, n2 = (int64_t)D * Fpe;
    auto put = [&](float* p, float v) { *p = (accumulate ? *p : 0.f) + v; };
    if (i < n0) {
        const int d = (int)i;
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 8 independent chains: the loop is load-latency bound
        int p = 0;
        for (; p + 8 <= Np; p += 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += gsum[(int64_t)(2 + p + u) * D + d];
        for (; p < Np; ++p) s8[0] += gsum[(int64_t)(2 + p) * D + d];
        const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
        put(d_bias + d, s);
        put(d_cls + d, gsum[d]);
        put(d_dist + d, gsum[D + d]);
        put(d_npe + d, gsum[d]);
        put(d_npe + D + d, gsum[D + d]);
    } else if (i < n0 + n1) {
        const int64_t k = i - n0;
        const int tt = (int)(k / D), d = (int)(k % D);
        float s = 0.f;
        for (int p = 0; p < Np; ++p)
            if (toff + pt[p] == tt) s += gsum[(int64_t)(2 + p) * D + d];
        put(d_tpos + (int64_t)d * Tpe + tt, s);
    } else if (i < n0 + n1 + n2) {
        const int64_t k = i - n0 - n1;
        const int f = (int)(k / D), d = (int)(k % D);
        float s = 0.f;
        for (int p = 0; p < Np; ++p)
            if (pf[p] == f) s += gsum[(int64_t)(2 + p) * D + d];
        put(d_fpos + (int64_t)d * Fpe + f, s);
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
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)cdiv(per, 256)), dim3(256), 0, st, dtok, B, per, gsum);
    int rc = check_launch();
    if (rc) return rc;
    const int64_t n = (int64_t)D * (1 + Tpe + Fpe);
    hipLaunchKernelGGL(patch_param_grads_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, gsum, D, patch_f,
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
