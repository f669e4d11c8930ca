// Fused log-mel front end for gfx950: waveform -> (B, n_mels, T) in ONE kernel.
// Reference: AugmentMelSTFT.forward, models/preprocess.py:57-86
//   :59     pre-emphasis conv1d [-0.97, 1]            -> fused into the LDS loader
//   :60-61  torch.stft(n_fft 1024, hop, hann(win) centred, center=True/reflect) -> LDS-resident
//           radix-8 FFT: one 64-lane wave transforms one frame (1024 real = 512 complex points,
//           8 complex points per lane, three in-register 8-point DFTs, two LDS exchanges)
//   :62     power spectrum                            -> never leaves LDS
//   :71-76  kaldi mel filterbank (dense 128x513 matmul in the reference) -> sparse band product:
//           each FFT bin feeds at most two adjacent triangles (weights u and 1-u)
//   :78     log(mel + 1e-5), :80-82 SpecAugment masks, :84 (x+4.5)/5 -> epilogue
// HBM traffic = read the waveform once (+6% halo) and write the mel tile once.
//
// Workgroup = 4 waves = FR_PER_WG consecutive frames of one clip; the output tile is transposed through LDS so every
// mel row leaves as FR_PER_WG * 4 contiguous bytes.
// r03: 16 frames per workgroup instead of 32.  The kernel is latency bound (dependent butterfly chains, two LDS
// exchanges per frame, ~11 cycles per instruction at two waves per SIMD); 51 KiB of LDS fit THREE workgroups per CU:
// 133 -> 108 us at B = 64 (8 frames: 106 us but 32-byte output rows).  A persistent variant (workgroups walking the tiles,
// geometry / twiddles set up once, next tile's span prefetched into registers) measured slower, 128-130 us: the
// prefetch registers cost the third wave (201 VGPRs) or spill, and static tile ranges balance worse than the dispatcher.
//
// r02 (profiles/r02_mel_variants.txt): the FFT is NOT what the kernel waits for.  With 8 waves and ~100 KiB of LDS per
// workgroup only one workgroup fitted a CU, and each one spent ~3/4 of its life in its un-overlapped prologue (a
// 21-iteration loop of dependent scalar loads staging the waveform span) and epilogue.  Now: 4 waves per workgroup and
// 79 KiB => two workgroups per CU (one stages while the other transforms), the span is staged with 16-byte loads that
// are all in flight together, and the sparse mel product reads ONE precomputed contribution per bin (P*u / P*(1-u)
// stored instead of P) in loops unrolled four-fold with independent accumulators.  (Rejected: LDS float atomics for a
// bin-parallel mel product -- 2.4x slower than the band loops, 402 vs 164 us: same-address ds_add_f32 serialise.)
#include <algorithm>

#include "pa_common.h"

namespace pa {

static constexpr int NFFT = 1024;
static constexpr int NC = 512;            // complex points
#ifndef PA_MEL_FRAMES
#define PA_MEL_FRAMES 16
#endif
static constexpr int FR_DEFAULT = PA_MEL_FRAMES;  // frames per workgroup (template parameter FR_PER_WG of the kernel)
static constexpr int MEL_WAVES = 4;
static constexpr int XROW1 = 68;          // exchange-1 row stride (complex) : conflict-free reads
static constexpr int XROW2 = 72;          // exchange-2 row stride (complex)
static constexpr int WAVE_SCRATCH = 8 * XROW2 * 8;   // 4608 bytes

// complex numbers are 2-vectors: add / sub / scale are ONE packed instruction each (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) --
// written on a {float x, y} struct the same butterflies compile to 131 VALU per two 8-point DFTs + 7 twiddles (a third of them
// v_mov to re-pair operands), written on vectors to 96 (round 5)
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cswap(cf a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cf cmul_negi(cf a) { cf s = cswap(a); s.y = -s.y; return s; }   // a * (-i) = {a.y, -a.x}
__device__ __forceinline__ cf cmul(cf a, cf b) {           // {ax bx - ay by, ax by + ay bx} = a.xx * b + a.yy * {-by, bx}
    const cf axx = {a.x, a.x}, ayy = {a.y, a.y};
    cf t = cswap(b);
    t.x = -t.x;
    return axx * b + ayy * t;
}

// in-place 8-point DFT, natural order in and out:  X[p] = sum_a v[a] exp(-2 pi i a p / 8)
__device__ __forceinline__ void dft8(cf (&v)[8]) {
    const float R = 0.70710678118654752440f;
    const cf s0 = cadd(v[0], v[4]), d0 = csub(v[0], v[4]);
    const cf s1 = cadd(v[1], v[5]), d1 = csub(v[1], v[5]);
    const cf s2 = cadd(v[2], v[6]), d2 = csub(v[2], v[6]);
    const cf s3 = cadd(v[3], v[7]), d3 = csub(v[3], v[7]);
    const cf t0 = cadd(s0, s2), t1 = csub(s0, s2), t2 = cadd(s1, s3), t3 = cmul_negi(csub(s1, s3));
    v[0] = cadd(t0, t2); v[4] = csub(t0, t2); v[2] = cadd(t1, t3); v[6] = csub(t1, t3);
    const cf e0 = d0;
    const cf e1 = (d1 + cmul_negi(d1)) * R;                    // d1 * (1 - i)/sqrt2 = {(x + y) R, (y - x) R}
    const cf e2 = cmul_negi(d2);
    const cf e3 = (cmul_negi(d3) - d3) * R;                    // d3 * (-1 - i)/sqrt2 = {(y - x) R, -(x + y) R}
    const cf u0 = cadd(e0, e2), u1 = csub(e0, e2), u2 = cadd(e1, e3), u3 = cmul_negi(csub(e1, e3));
    v[1] = cadd(u0, u2); v[5] = csub(u0, u2); v[3] = cadd(u1, u3); v[7] = csub(u1, u3);
}

// exp(-2 pi i j / 1024) for j in [0, 1024) from the half table
__device__ __forceinline__ cf tw1024(const float2* __restrict__ tw, int j) {
    const float2 t = tw[j & 511];
    return (j & 512) ? cf{-t.x, -t.y} : cf{t.x, t.y};
}

template <int FR_PER_WG>
__global__ __launch_bounds__(MEL_WAVES * 64) void mel_frontend_kernel(const float* __restrict__ wave, int L,
                                                            const float* __restrict__ window,
                                                            const float* __restrict__ bin_mel,
                                                            const float2* __restrict__ twiddle,
                                                            float* __restrict__ out, const pa_mel_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int span = (FR_PER_WG - 1) * p.hop + NFFT;
    float* sSig = (float*)smem;                                        // [span] pre-emphasised, reflect-padded
    char* sScr = smem + ((span * 4 + 15) & ~15);                       // [8 waves][WAVE_SCRATCH]
    int* sS = (int*)(sScr + MEL_WAVES * WAVE_SCRATCH);                 // [n_mels + 3] first bin with j >= b
    float* sOut = (float*)(sS + 132);                                  // [n_mels][33]
    // the two per-bin tables are only needed until every lane holds its 8 weights and sS is built: they live in the
    // (not yet written) output tile, which keeps the workgroup under 80 KiB = two workgroups per CU
    float* sU = sOut;                                                  // [512] up-slope weight of bin k
    int* sJ = (int*)(sU + NC);                                         // [512] triangle index of bin k

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * FR_PER_WG;
    const int T = p.n_frames;
    const int Ly = L - 1;
    const float* x = wave + (int64_t)b * L;

    // ---- stage the signal span: y[i] = x[i+1] - preemph * x[i], reflect-padded by n_fft/2 ----
    constexpr int NT = MEL_WAVES * 64;
    const int i0 = f0 * p.hop - NFFT / 2;                    // sample index of sSig[0]
    if (i0 >= 0 && i0 + span + 4 <= Ly && ((i0 | L) & 3) == 0 && (span & 3) == 0 && span <= 12 * 4 * NT) {
        // interior tile (no reflection, 16-byte aligned): one 16-byte load + the next sample per 4 outputs, every
        // load of the tile issued before the first use (the loop has a compile-time trip count)
        const float* xs = x + i0;
        const int nv = span >> 2;
        constexpr int MAXIT = 12;                             // 12 x 256 x 4 samples >= span for hop <= 362
        f32x4 a[MAXIT];
        float nx[MAXIT];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int t = tid + it * NT;
            if (it * NT < nv) {                               // uniform; the lane predicate is folded into the address
                const int tc = min(t, nv - 1);
                a[it] = *(const f32x4*)(xs + 4 * tc);
                nx[it] = xs[4 * tc + 4];
            }
        }
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int t = tid + it * NT;
            if (it * NT < nv && t < nv)
                *(f32x4*)(sSig + 4 * t) = f32x4{a[it][1] - p.preemph * a[it][0], a[it][2] - p.preemph * a[it][1],
                                                a[it][3] - p.preemph * a[it][2], nx[it] - p.preemph * a[it][3]};
        }
    } else {
        for (int j = tid; j < span; j += NT) {
            int i = i0 + j;
            if (i < 0) i = -i;
            if (i >= Ly) i = 2 * (Ly - 1) - i;
            i = max(0, min(i, Ly - 1));
            sSig[j] = x[i + 1] - p.preemph * x[i];
        }
    }
    // ---- filterbank geometry for this call's (fmin, fmax): bin k -> triangle j_k, weight u_k ----
    for (int k = tid; k < NC; k += NT) {
        const float t = (bin_mel[k] - p.mel_low) * p.inv_mel_delta;
        const float fl = floorf(t);
        sJ[k] = (int)fmaxf(fminf(fl, 100000.f), -1.f);
        sU[k] = t - fl;
    }
    __syncthreads();
    if (tid <= p.n_mels + 2) {       // sS[b] = #bins with j_k < b  (j_k is non-decreasing in k)
        int lo = 0, hi = NC;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sJ[mid] < tid) lo = mid + 1; else hi = mid;
        }
        sS[tid] = lo;
    }
    __syncthreads();

    // ---- per-lane constants ----
    cf tw1[8], tw2[8], tw3[8];
    float win[16];
    {
        const int m = lane;                  // stage-1 role: m
        const int c = lane >> 3;             // stage-2 role: (c, p)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            tw1[q] = tw1024(twiddle, 2 * m * q);          // W512^(m q)
            tw2[q] = tw1024(twiddle, 16 * c * q);         // W64^(c q)
            tw3[q] = tw1024(twiddle, lane + 64 * q);      // W1024^k, k = lane + 64 q
            win[2 * q] = window[2 * (64 * q + m)];
            win[2 * q + 1] = window[2 * (64 * q + m) + 1];
        }
    }
    cf* scr = (cf*)(sScr + wv * WAVE_SCRATCH);
    float* up = (float*)scr;                 // per-bin contributions overlay the scratch (2 x 512 floats of its 1152)
    float* dn = up + NC;
    float bu[8];                             // up-slope weight of this lane's 8 bins k = lane + 64 s
#pragma unroll
    for (int q = 0; q < 8; ++q) bu[q] = sU[lane + 64 * q];
    __syncthreads();                         // sU / sJ are dead from here on: their LDS is the output tile

    for (int fi = 0; fi < FR_PER_WG / MEL_WAVES; ++fi) {
        const int fl = wv * (FR_PER_WG / MEL_WAVES) + fi;
        const int frame = f0 + fl;
        if (frame >= T) break;               // wave-uniform
        const float* sig = sSig + fl * p.hop;
        cf v[8];
        // stage 1: lane m holds z[64a + m], a = 0..7  (z[n] = y[2n] + i y[2n+1], windowed)
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int n = 64 * a + lane;
            v[a] = cf{sig[2 * n], sig[2 * n + 1]} * cf{win[2 * a], win[2 * a + 1]};
        }
        dft8(v);
#pragma unroll
        for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw1[q]);
        // exchange 1: Y[m][p] -> lane (c, p') reads Y[8b + c][p'], b = 0..7   (layout [p][m], stride 68)
#pragma unroll
        for (int q = 0; q < 8; ++q) scr[q * XROW1 + lane] = v[q];
        {
            const int c = lane >> 3, pp = lane & 7;
#pragma unroll
            for (int bq = 0; bq < 8; ++bq) v[bq] = scr[pp * XROW1 + 8 * bq + c];
        }
        dft8(v);                              // over b -> r
#pragma unroll
        for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw2[q]);
        // exchange 2: U[c][p][r] (lane = c*8+p holds r = 0..7) -> lane' = r*8 + p reads c = 0..7
#pragma unroll
        for (int q = 0; q < 8; ++q) scr[q * XROW2 + lane] = v[q];
        {
            const int r = lane >> 3, pp = lane & 7;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = scr[r * XROW2 + c * 8 + pp];
        }
        dft8(v);                              // over c -> s ; lane' holds Z[lane' + 64 s]
        // untangle the packed real FFT: needs Z[k] and Z[512-k]
#pragma unroll
        for (int s = 0; s < 8; ++s) scr[lane + 64 * s] = v[s];
        float pk[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = lane + 64 * s;
            const cf zk = v[s];
            const cf zc = scr[(NC - k) & (NC - 1)];            // Z[512-k] (Z[512] == Z[0])
            // X = (Z[k] + conj Z[N-k]) / 2 + W^k (Z[k] - conj Z[N-k]) / (2i): the two halves as packed sums of Z[k] and
            // {zc.x, -zc.y}; the 1/2's are applied once, to the power (x 1/4)
            cf zcc = zc;
            zcc.y = -zcc.y;                                    // conj Z[N-k]
            const cf e2 = zk + zcc;                            // 2 e
            const cf o2 = cmul_negi(zk - zcc);                 // 2 o = (Z[k] - conj Z[N-k]) / i
            const cf X2 = e2 + cmul(o2, tw3[s]);               // 2 X
            const cf sq = X2 * X2;
            pk[s] = 0.25f * (sq.x + sq.y);
        }
        // every bin feeds at most two triangles: store its two contributions, up[k] = P u (to triangle j_k) and
        // dn[k] = P (1 - u) (to triangle j_k - 1), so that the band sums below read one value per bin
        // (the Nyquist bin has no column in the kaldi bank: models/preprocess.py:73-74 pads a zero one)
#pragma unroll
        for (int s = 0; s < 8; ++s) {                             // all reads of scr are issued above
            up[lane + 64 * s] = pk[s] * bu[s];
            dn[lane + 64 * s] = pk[s] - pk[s] * bu[s];
        }
        // sparse mel bands: this lane owns bands `lane` and `n_mels-1-lane` (one narrow + one wide): bins [k0, k1) on
        // their up-slope, [k1, k2) on their down-slope; four independent partial sums keep four LDS reads in flight
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int band = h == 0 ? lane : p.n_mels - 1 - lane;
            // h == 0 covers bands 0..63; h == 1 covers the remaining 64..n_mels-1 in reverse
            const bool mine = h == 0 ? band < p.n_mels : band >= 64;
            if (!mine) continue;
            const int k0 = sS[band], k1 = sS[band + 1], k2 = sS[band + 2];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int k = k0;
            for (; k + 3 < k1; k += 4) { a0 += up[k]; a1 += up[k + 1]; a2 += up[k + 2]; a3 += up[k + 3]; }
            for (; k < k1; ++k) a0 += up[k];
            for (; k + 3 < k2; k += 4) { a0 += dn[k]; a1 += dn[k + 1]; a2 += dn[k + 2]; a3 += dn[k + 3]; }
            for (; k < k2; ++k) a1 += dn[k];
            sOut[band * (FR_PER_WG + 1) + fl] = (a0 + a1) + (a2 + a3);
        }
    }
    __syncthreads();
    // ---- epilogue: log, SpecAugment masks, affine; rows of 32 frames = 128 contiguous bytes ----
    for (int idx = tid; idx < p.n_mels * FR_PER_WG; idx += NT) {
        const int mel = idx / FR_PER_WG, fl = idx % FR_PER_WG;
        const int t = f0 + fl;
        if (t >= T) continue;
        float v = __logf(sOut[mel * (FR_PER_WG + 1) + fl] + p.log_eps);
        const bool masked = (mel >= p.fmask_start && mel < p.fmask_end) || (t >= p.tmask_start && t < p.tmask_end);
        if (masked) v = 0.f;
        out[((int64_t)b * p.n_mels + mel) * T + t] = (v + p.out_add) * p.out_scale;
    }
}

}  // namespace pa

using namespace pa;

extern "C" int pa_mel_num_frames(int L, int hop) { return (L < 2 || hop <= 0) ? 0 : 1 + (L - 1) / hop; }

extern "C" int pa_mel_frontend_fwd(const float* wave, int B, int L, const float* window, const float* bin_mel,
                                   const float* twiddle, float* out, const pa_mel_params* p, void* stream) {
    if (!wave || !window || !bin_mel || !twiddle || !out || !p || B <= 0) return PA_EINVAL;
    if (p->n_fft != NFFT || p->n_mels < 4 || p->n_mels > 128 || p->hop <= 0 || p->hop > NFFT) return PA_EUNSUPPORTED;
    if (L - 1 <= NFFT / 2) return PA_EUNSUPPORTED;          // reflect padding needs L-1 > n_fft/2 (torch.stft rule)
    if (p->n_frames != pa_mel_num_frames(L, p->hop)) return PA_EINVAL;
    // 16 frames per workgroup (64-byte output rows, three workgroups per CU).  Measured and not adopted (round 5,
    // profiles/r05_mel_variants.txt): 8 frames per workgroup for grids that leave CUs without their three workgroups (ESC-50 at
    // batch 12: 375 workgroups): 21.5-22.5 us against 20.6-21.3 -- the per-workgroup set-up is amortised over half the frames
    const int fr = FR_DEFAULT;
    const int span = (fr - 1) * p->hop + NFFT;
    const size_t lds = ((span * 4 + 15) & ~15) + MEL_WAVES * WAVE_SCRATCH + 132 * 4 +
                       std::max<size_t>((size_t)p->n_mels * (fr + 1) * 4, 2 * NC * 4);
    if (lds > 160 * 1024) return PA_EUNSUPPORTED;
    static bool attr_set = [] {
        return hipFuncSetAttribute((const void*)mel_frontend_kernel<FR_DEFAULT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    }();
    (void)attr_set;
    dim3 grid((unsigned)cdiv(p->n_frames, fr), (unsigned)B);
    hipLaunchKernelGGL(mel_frontend_kernel<FR_DEFAULT>, grid, dim3(MEL_WAVES * 64), lds, (hipStream_t)stream, wave, L, window, bin_mel,
                       (const float2*)twiddle, out, *p);
    return check_launch();
}
