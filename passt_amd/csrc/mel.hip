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
// r05 (profiles/r05_mel_probe.txt, tools/probe_mel.py = the PA_MEL_PROBE build): time stamps of every wave's phases showed
// that half of a frame was the band sums (one lane per band walking its bins: 26 dependent LDS round trips in data-dependent
// loops).  They are now a lane-local segment recurrence over 8 consecutive bins + ONE segmented DPP scan across the wave
// (tools/emulate_mel_bands.py is that stage on the CPU); the butterflies use one packed FMA for a -+ i b; 107 -> 90-92 us.
// The history below is what led here (the "band loops" of r02 are the ones r05 replaced).
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
#include <cstdlib>

#include "pa_common.h"

namespace pa {

static constexpr int NFFT = 1024;
static constexpr int NC = 512;            // complex points
#ifndef PA_MEL_FRAMES
#define PA_MEL_FRAMES 16
#endif
static constexpr int FR_DEFAULT = PA_MEL_FRAMES;  // frames per workgroup (template parameter FR_PER_WG of the kernel)
static constexpr int MEL_WAVES = 4;
#ifndef PA_MEL_ABL
#define PA_MEL_ABL 0
#endif
#ifndef PA_MEL_LEAN_PERSIST
#define PA_MEL_LEAN_PERSIST 8             // the persistent form carries the tile bookkeeping too: every untangle twiddle from one copy (no scratch:
#endif                                    // a scratch reload in the frame loop would wait for the span DMA in flight -- vmcnt retires in order)
#ifndef PA_MEL_PERSIST_FRAMES
#define PA_MEL_PERSIST_FRAMES 8           // frames per tile of the persistent form (two per wave; 32-byte output rows; 49 KiB: three workgroups per CU)
#endif
#ifndef PA_MEL_LEAN
#define PA_MEL_LEAN 3                     // how many of the untangle twiddles use the one-copy product (register budget: 168)
#endif
static constexpr int XROW1 = 68;          // exchange-1 row stride (complex) : conflict-free reads
static constexpr int XROW2 = 72;          // exchange-2 row stride (complex)
static constexpr int WAVE_SCRATCH = 8 * XROW2 * 8;   // 4608 bytes
static constexpr int SLOT_DUMMY = 130;               // band-stage slots: [0, 128] real, 129 pad, 130 + lane private dummies

// complex numbers are 2-vectors: add / sub / scale are ONE packed instruction each (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32) --
// written on a {float x, y} struct the same butterflies compile to 131 VALU per two 8-point DFTs + 7 twiddles (a third of them
// v_mov to re-pair operands), written on vectors to 96 (round 5)
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cswap(cf a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cf bcast(cf a, int h) { return h ? __builtin_shufflevector(a, a, 1, 1) : __builtin_shufflevector(a, a, 0, 0); }
__device__ __forceinline__ cf cmul_negi(cf a) { cf s = cswap(a); s.y = -s.y; return s; }   // a * (-i) = {a.y, -a.x}
__device__ __forceinline__ cf cmul(cf a, cf b) {           // {ax bx - ay by, ax by + ay bx} = a.xx * b + a.yy * {-by, bx}
    const cf axx = {a.x, a.x}, ayy = {a.y, a.y};
    cf t = cswap(b);
    t.x = -t.x;
    return axx * b + ayy * t;
}

// a -+ i b and the conjugate sums as ONE packed FMA each: the swap of b's halves is an op_sel modifier, the signs are a constant
// pair (exact: the multiplier is +-1).  Written as add(a, negi(b)) the compiler materialises {b.y, -b.x} with v_xor + v_mov first.
__device__ __forceinline__ cf addni(cf a, cf b) { return __builtin_elementwise_fma(cswap(b), cf{1.f, -1.f}, a); }   // a + (-i) b
__device__ __forceinline__ cf subni(cf a, cf b) { return __builtin_elementwise_fma(cswap(b), cf{-1.f, 1.f}, a); }   // a - (-i) b
__device__ __forceinline__ cf addcj(cf a, cf b) { return __builtin_elementwise_fma(b, cf{1.f, -1.f}, a); }          // a + conj b
__device__ __forceinline__ cf subcj(cf a, cf b) { return __builtin_elementwise_fma(b, cf{-1.f, 1.f}, a); }          // a - conj b

// the same product in three instructions but from ONE copy of b: cmul keeps {b.x, b.y} and {-b.y, b.x} of a loop-invariant
// twiddle in registers (two pairs); used for a few twiddles so that the kernel stays at three waves per SIMD
__device__ __forceinline__ cf cmul_lean(cf a, cf b) {
    const cf axx = {a.x, a.x}, ayy = {a.y, a.y};
    return __builtin_elementwise_fma(cswap(ayy * b), cf{-1.f, 1.f}, axx * b);    // {ax bx - ay by, ax by + ay bx}
}

// in-place 8-point DFT, natural order in and out:  X[p] = sum_a v[a] exp(-2 pi i a p / 8)
__device__ __forceinline__ void dft8(cf (&v)[8]) {
    const float R = 0.70710678118654752440f;
    const cf s0 = cadd(v[0], v[4]), d0 = csub(v[0], v[4]);
    const cf s1 = cadd(v[1], v[5]), d1 = csub(v[1], v[5]);
    const cf s2 = cadd(v[2], v[6]), d2 = csub(v[2], v[6]);
    const cf s3 = cadd(v[3], v[7]), d3 = csub(v[3], v[7]);
    const cf t0 = cadd(s0, s2), t1 = csub(s0, s2), t2 = cadd(s1, s3), t3 = csub(s1, s3);      // t3 enters times -i
    v[0] = cadd(t0, t2); v[4] = csub(t0, t2); v[2] = addni(t1, t3); v[6] = subni(t1, t3);
    const cf e1 = addni(d1, d1) * R;                           // d1 * (1 - i)/sqrt2 = {(x + y) R, (y - x) R}
    const cf e3 = subni(d3, d3) * -R;                          // d3 * (-1 - i)/sqrt2 = {(y - x) R, -(x + y) R}
    const cf u0 = addni(d0, d2), u1 = subni(d0, d2), u2 = cadd(e1, e3), u3 = csub(e1, e3);    // u3 enters times -i
    v[1] = cadd(u0, u2); v[5] = csub(u0, u2); v[3] = addni(u1, u3); v[7] = subni(u1, u3);
}

// v_mov_dpp with old = 0 and bound_ctrl: lanes without a source (or outside row_mask) read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xF, true); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ cf dpp_c(cf v) {
    const float x = v[0], y = v[1];       // (__builtin_bit_cast on a vector ELEMENT reads element 0 whichever is named)
    return cf{__int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(x))), __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(y)))};
}

// one step of the segmented scan: every lane fetches (DPP is convergent: all lanes execute it), the predicate only gates the add
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ cf seg_step(cf t, bool take) {
    const cf d = dpp_c<CTRL, ROW_MASK>(t);
    return t + (take ? d : cf{0.f, 0.f});
}

// exp(-2 pi i j / 1024) for j in [0, 1024) from the half table
__device__ __forceinline__ cf tw1024(const float2* __restrict__ tw, int j) {
    const float2 t = tw[j & 511];
    return (j & 512) ? cf{-t.x, -t.y} : cf{t.x, t.y};
}

// PERSIST (round 6): the workgroup walks tiles of FR_PER_WG frames (tile = blockIdx.x, + gridDim.x, ...).  The per-lane constants and
// the filterbank geometry are worked out ONCE per workgroup; the signal span of tile k + 1 is requested by LDS-DMA (global_load_lds: no
// registers, raw samples) into the second span buffer while tile k is transformed, so neither the HBM latency of the span nor the ~40
// table loads per lane sit in front of every 16 frames any more (they were 4.2 us of a 14 us workgroup life, profiles/r05_mel_probe.txt).
// The pre-emphasis then happens where stage 1 reads the samples (the DMA cannot apply it); tiles that touch the reflect padding at the
// clip edges are staged by the lanes themselves, pre-emphasised, as before (`raw` tells stage 1 which form the buffer holds).
template <int FR_PER_WG, bool PERSIST>
__global__ __launch_bounds__(MEL_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void mel_frontend_kernel(const float* __restrict__ wave, int L,
                                                            const float* __restrict__ window,
                                                            const float* __restrict__ bin_mel,
                                                            const float2* __restrict__ twiddle,
                                                            float* __restrict__ out, const pa_mel_params p,
                                                            const int tiles_per_clip, const int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int span = (FR_PER_WG - 1) * p.hop + NFFT;
    // one span buffer holds span (+ 4: the sample behind the last one, for the pre-emphasis) floats, rounded up to whole 1 KiB DMA pieces
    const int span_bytes = PERSIST ? ((span + 4) * 4 + 1023) & ~1023 : ((span * 4 + 15) & ~15);
    float* sSig0 = (float*)smem;                                       // [span] pre-emphasised + reflect-padded, or (PERSIST, interior) raw
    char* sScr = smem + (PERSIST ? 2 : 1) * span_bytes;                // [MEL_WAVES][WAVE_SCRATCH]
    float* sOut = (float*)(sScr + MEL_WAVES * WAVE_SCRATCH);           // [n_mels][FR_PER_WG + 1]
    // the two per-bin tables are only needed until every lane holds its band-stage constants: they live in the
    // (not yet written) output tile; 50 KiB per workgroup at 16 frames and hop 320 = three workgroups per CU
    float* sU = sOut;                                                  // [512] up-slope weight of bin k
    int* sJ = (int*)(sU + NC);                                         // [512] triangle index of bin k

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PA_MEL_PROBE      // probe build (tools/probe_mel.py): s_memrealtime stamps (100 MHz) of the phases of every wave, left in the output tile
    uint32_t stamp[6];       // few and 32-bit: the kernel is at its SGPR / VGPR budget, a bigger probe would change its occupancy
#define MEL_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); stamp[i] = (uint32_t)__builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MEL_STAMP(i) do {} while (0)
#endif
    MEL_STAMP(0);
    const int T = p.n_frames;
    const int Ly = L - 1;
    constexpr int NT = MEL_WAVES * 64;
    static_assert(NC == 2 * NT, "two bins per thread");
    // tile -> (clip, first frame); non-persistent: the grid is (tiles per clip, clips)
    int tile = PERSIST ? (int)blockIdx.x : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    int b = PERSIST ? tile / tiles_per_clip : (int)blockIdx.y;
    int f0 = (PERSIST ? tile - b * tiles_per_clip : (int)blockIdx.x) * FR_PER_WG;
    // a tile whose span needs no reflection and can move as 16-byte pieces (i0: sample index of sSig[0])
    auto interior = [&](int f0_) {
        const int i0 = f0_ * p.hop - NFFT / 2;
        return i0 >= 0 && i0 + span + 4 <= Ly && ((i0 | L) & 3) == 0;
    };
    // PERSIST: request the raw span of tile (b_, f0_) into buffer `buf` by LDS-DMA: 1 KiB pieces dealt round-robin to the waves
    auto stage_dma = [&](int buf, int b_, int f0_) {
        const float* xs = wave + (int64_t)b_ * L + (f0_ * p.hop - NFFT / 2);
        const int nchunk = (span + 4) >> 2;                           // 16-byte chunks covering x[0 .. span] (x[span] feeds y[span - 1])
        // Issued by INLINE ASM, not by __builtin_amdgcn_global_load_lds: with the builtin the compiler, which cannot tell the DMA's LDS
        // destination from the other span buffer, puts s_waitcnt vmcnt(0) in front of the next LDS read -- the first sample read of
        // the frames -- and the prefetch overlaps nothing (measured: 101-105 us against 90-93 for the one-tile form).  The asm form is
        // invisible to that analysis; the landing is ordered by the explicit vmcnt(0) + barrier that closes the tile's frames.  (M0 =
        // LDS base of the piece, saved and restored around the instruction: nothing else in this kernel uses it.)
        const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)(buf * span_bytes);
        for (int k = wv; k * 64 < nchunk; k += MEL_WAVES) {
            const int c = min(k * 64 + lane, nchunk - 1);
            const float* src = xs + 4 * c;
            uint32_t m0_keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_keep) : "s"(__builtin_amdgcn_readfirstlane(dst + (uint32_t)k * 1024u)), "v"(src) : "memory");
        }
    };
    // the lanes stage the span themselves: y[i] = x[i+1] - preemph * x[i], reflect-padded by n_fft/2 (clip edges, odd geometries)
    auto stage_slow = [&](float* sSig, int b_, int f0_) {
        const float* x = wave + (int64_t)b_ * L;
        const int i0 = f0_ * p.hop - NFFT / 2;
        for (int j = tid; j < span; j += NT) {
            int i = i0 + j;
            if (i < 0) i = -i;
            if (i >= Ly) i = 2 * (Ly - 1) - i;
            i = max(0, min(i, Ly - 1));
            sSig[j] = x[i + 1] - p.preemph * x[i];
        }
    };

    // ---- per-lane constants: requested first, so that the (L2-resident) tables arrive while the span is staged ----
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

    // ---- stage the signal span: y[i] = x[i+1] - preemph * x[i], reflect-padded by n_fft/2.  (Measured and not kept, round 5:
    // geometry + band-stage constants worked out while the span loads are in flight, the span stored last -- the prologue
    // shrinks by 0.8 us, the frames of the co-resident workgroups slow down by 0.4, the launch stays at 92 us.) ----
    const float bm0 = bin_mel[tid], bm1 = bin_mel[tid + NT];   // requested here, used behind the span
    if constexpr (PERSIST) {
        if (interior(f0)) stage_dma(0, b, f0);
        else stage_slow(sSig0, b, f0);
    } else {
        const float* x = wave + (int64_t)b * L;
        const int i0 = f0 * p.hop - NFFT / 2;                    // sample index of sSig[0]
        if (interior(f0) && (span & 3) == 0 && span <= 12 * 4 * NT) {
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
                    *(f32x4*)(sSig0 + 4 * t) = f32x4{a[it][1] - p.preemph * a[it][0], a[it][2] - p.preemph * a[it][1],
                                                     a[it][3] - p.preemph * a[it][2], nx[it] - p.preemph * a[it][3]};
            }
        } else {
            stage_slow(sSig0, b, f0);
        }
    }
    MEL_STAMP(1);
    // ---- filterbank geometry for this call's (fmin, fmax): bin k -> triangle j_k, weight u_k ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float t = ((h ? bm1 : bm0) - p.mel_low) * p.inv_mel_delta;
        const float fl = floorf(t);
        sJ[tid + h * NT] = (int)fmaxf(fminf(fl, 100000.f), -1.f);
        sU[tid + h * NT] = t - fl;
    }
    __syncthreads();
    cf* scr = (cf*)(sScr + wv * WAVE_SCRATCH);
    // ---- band-stage constants (tools/emulate_mel_bands.py is this stage on the CPU).  Lane L owns the 8 consecutive bins
    // [8L, 8L + 8); bins with the same triangle index j form a segment whose two sums (up-slope contributions -> band j,
    // down-slope contributions -> band j - 1) are wanted.  keep[i]: bin i continues the segment of bin i - 1;  sa[i]: the slot the
    // segment that ends in front of bin i is flushed to (a private dummy slot when none ends there / it lies outside the bank);
    // sw[s]: the predicate of step s of the wave-wide segmented scan over the lanes' open tails -- geometry only, not data ----
    float* pex = (float*)scr;                                // [512] power spectrum, re-read as 8 consecutive bins per lane
    cf* slot = (cf*)((char*)scr + NC * 4);                   // [n_mels + 1] {U_j, D_j}, one pad, 64 dummies: 1552 B of the 4608
    float un[8];
    cf keep[4];                                              // 1.f / 0.f per bin, two per register pair: the recurrence gates with
    uint32_t sa[4];                                          // ONE packed FMA per bin (the broadcast of a half is an op_sel modifier);
    int j_last;                                              // slot indices two per register (16 bits each)
    bool sw[6];
    {
        const int dummy = SLOT_DUMMY + lane;
        int jprev = lane == 0 ? -1 : sJ[8 * lane - 1];
        int heads = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 8 * lane + i;
            const int jk = sJ[k];
            const bool boundary = k > 0 && jk != jprev;
            keep[i >> 1][i & 1] = boundary ? 0.f : 1.f;
            heads |= boundary;
            const uint32_t a = (boundary && jprev >= 0 && jprev <= p.n_mels) ? jprev : dummy;
            sa[i >> 1] = (i & 1) ? (sa[i >> 1] | (a << 16)) : a;
            un[i] = 0.25f * sU[k];
            jprev = jk;
        }
        j_last = __builtin_amdgcn_readlane(jprev, 63);        // the segment that is still open behind bin 511 (wave-uniform)
        int f = heads;                                       // this lane's tail starts a new segment
        sw[0] = f == 0; f |= dpp_i<0x111, 0xF>(f);           // row_shr:1
        sw[1] = f == 0; f |= dpp_i<0x112, 0xF>(f);           // row_shr:2
        sw[2] = f == 0; f |= dpp_i<0x114, 0xF>(f);           // row_shr:4
        sw[3] = f == 0; f |= dpp_i<0x118, 0xF>(f);           // row_shr:8
        sw[4] = f == 0; f |= dpp_i<0x142, 0xA>(f);           // row_bcast15 into rows 1, 3
        sw[5] = f == 0;                                      // row_bcast31 into rows 2, 3
    }
    if constexpr (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the first tile's span pieces of this wave have landed
    __syncthreads();                         // sU / sJ are dead from here on: their LDS is the output tile
    MEL_STAMP(2);

    for (int it = 0;; ++it) {                 // PERSIST: the tiles of this workgroup; else one pass
    const float* sSig = (const float*)(smem + (PERSIST ? (it & 1) * span_bytes : 0));
    bool raw = false;
    if constexpr (PERSIST) {
        raw = interior(f0);
        // this tile's span has landed (the DMA was waited for -- vmcnt(0) -- in front of the barrier that closed the previous tile's
        // frames, or in the prologue); every wave is through the previous tile's epilogue: its output tile and the other span
        // buffer (read by the tile before) are free
        if (it > 0) __syncthreads();
        const int nt_ = tile + (int)gridDim.x;
        if (nt_ < n_tiles) {
            const int nb = nt_ / tiles_per_clip, nf0 = (nt_ - nb * tiles_per_clip) * FR_PER_WG;
#ifndef PA_MEL_NODMA         // timing ablation: the next tile's span is not requested (stale samples)
            if (interior(nf0)) stage_dma((it + 1) & 1, nb, nf0);
            else stage_slow((float*)(smem + ((it + 1) & 1) * span_bytes), nb, nf0);
#endif
        }
    }
    for (int fi = 0; fi < FR_PER_WG / MEL_WAVES; ++fi) {
        const int fl = wv * (FR_PER_WG / MEL_WAVES) + fi;
        const int frame = f0 + fl;
        if (frame >= T) break;               // wave-uniform
        const float* sig = sSig + fl * p.hop;
        cf v[8];
        // stage 1: lane m holds z[64a + m], a = 0..7  (z[n] = y[2n] + i y[2n+1], windowed)
        if (PERSIST && raw) {                // uniform: the buffer holds raw samples, y[j] = x[j+1] - preemph x[j] is formed here
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const float* s2 = sig + 2 * (64 * a + lane);
                const cf x01 = *(const cf*)s2;
                const float x2 = s2[2];
                const cf y = __builtin_elementwise_fma(x01, cf{-p.preemph, -p.preemph}, cf{x01.y, x2});
                v[a] = y * cf{win[2 * a], win[2 * a + 1]};
            }
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int n = 64 * a + lane;
                v[a] = cf{sig[2 * n], sig[2 * n + 1]} * cf{win[2 * a], win[2 * a + 1]};
            }
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
#if PA_MEL_ABL >= 2      // ablation builds (probe only): 2 = no untangle, no band stage; 1 = no band stage
        {
            cf a = v[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) a += v[s];
            sOut[lane * (FR_PER_WG + 1) + fl] = a.x;
            sOut[(lane + 64) * (FR_PER_WG + 1) + fl] = a.y;
        }
#else
        // untangle the packed real FFT: needs Z[k] and Z[512-k]
#pragma unroll
        for (int s = 0; s < 8; ++s) scr[lane + 64 * s] = v[s];
        float pk[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = lane + 64 * s;
            const cf zk = v[s];
            const cf zc = scr[(NC - k) & (NC - 1)];            // Z[512-k] (Z[512] == Z[0])
            // X = (Z[k] + conj Z[N-k]) / 2 + W^k (Z[k] - conj Z[N-k]) / (2i); the 1/2's are applied once, to the power (x 1/4),
            // and that factor lives in the band weights (un = u / 4: exact)
            const cf e2 = addcj(zk, zc);                       // 2 e
            const cf o2 = cmul_negi(subcj(zk, zc));            // 2 o = (Z[k] - conj Z[N-k]) / i
            const cf X2 = e2 + (s < (PERSIST ? PA_MEL_LEAN_PERSIST : PA_MEL_LEAN) ? cmul_lean(o2, tw3[s]) : cmul(o2, tw3[s]));               // 2 X
            const cf sq = X2 * X2;
            pk[s] = sq.x + sq.y;                               // 4 |X|^2
        }
#if PA_MEL_ABL == 1
        {
            float a = pk[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) a += pk[s];
            sOut[lane * (FR_PER_WG + 1) + fl] = a;
            sOut[(lane + 64) * (FR_PER_WG + 1) + fl] = a * un[0] * keep[0].x * (float)sa[0];
        }
#else
        // ---- sparse mel bands.  Every bin feeds at most two triangles: P u to triangle j_k, P (1 - u) to triangle j_k - 1.
        // The power spectrum changes owner (lane + 64 s -> 8 consecutive bins per lane), every lane runs the segment
        // recurrence acc = (keep ? acc : 0) + {P u, P - P u} over its bins -- once from zero (its open tail), then, after ONE
        // segmented scan of the tails across the wave has produced what the lanes in front of it left open, again from that
        // carry, dropping acc into the slot of every segment that ends.  No data-dependent loop, no LDS round trip per bin
        // (round 5: the lane-per-band loops it replaces were half of the frame time: 26 dependent LDS round trips, probe
        // profiles/r05_mel_probe.txt). ----
#pragma unroll
        for (int s = 0; s < 8; ++s) pex[lane + 64 * s] = pk[s];   // all reads of scr are issued above
        const f32x4 pa = *(const f32x4*)(pex + 8 * lane), pb = *(const f32x4*)(pex + 8 * lane + 4);
        slot[lane] = cf{0.f, 0.f};
        slot[lane + 64] = cf{0.f, 0.f};
        if (lane < 2) slot[128 + lane] = cf{0.f, 0.f};
        cf val[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pw = i < 4 ? pa[i & 3] : pb[i & 3];
            const float u = pw * un[i];                         // pw = 4 P, un = u / 4
            val[i] = cf{u, __builtin_fmaf(pw, 0.25f, -u)};
        }
        cf acc = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_elementwise_fma(acc, bcast(keep[i >> 1], i & 1), val[i]);
        cf t = acc;                                              // inclusive segmented scan of the tails
        t = seg_step<0x111, 0xF>(t, sw[0]);
        t = seg_step<0x112, 0xF>(t, sw[1]);
        t = seg_step<0x114, 0xF>(t, sw[2]);
        t = seg_step<0x118, 0xF>(t, sw[3]);
        t = seg_step<0x142, 0xA>(t, sw[4]);
        t = seg_step<0x143, 0xC>(t, sw[5]);
        acc = dpp_c<0x138, 0xF>(t);                              // wave_shr:1: what the lanes in front left open
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            slot[(i & 1) ? (sa[i >> 1] >> 16) : (sa[i >> 1] & 0xFFFFu)] = acc;
            acc = __builtin_elementwise_fma(acc, bcast(keep[i >> 1], i & 1), val[i]);
        }
        if (lane == 63 && j_last >= 0 && j_last <= p.n_mels) slot[j_last] = acc;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int band = lane + 64 * h;
            if (band < p.n_mels) sOut[band * (FR_PER_WG + 1) + fl] = slot[band].x + slot[band + 1].y;
        }
#endif
#endif
#ifdef PA_MEL_PROBE
        if (fi == 3) MEL_STAMP(3);
#endif
    }
    if constexpr (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's DMA (requested a tile ago) and older stores
    __syncthreads();
    MEL_STAMP(4);
    // ---- epilogue: log, SpecAugment masks, affine; rows of FR_PER_WG frames = 64 contiguous bytes ----
    for (int idx = tid; idx < p.n_mels * FR_PER_WG; idx += NT) {
        const int mel = idx / FR_PER_WG, fl = idx % FR_PER_WG;
        const int t = f0 + fl;
        if (t >= T) continue;
        float v = __logf(sOut[mel * (FR_PER_WG + 1) + fl] + p.log_eps);
        const bool masked = (mel >= p.fmask_start && mel < p.fmask_end) || (t >= p.tmask_start && t < p.tmask_end);
        if (masked) v = 0.f;
#ifdef PA_MEL_NOSTORE        // timing ablation (A/B builds only): no output stores
        if (v != 12345.678f) continue;
#endif
        out[((int64_t)b * p.n_mels + mel) * T + t] = (v + p.out_add) * p.out_scale;
    }
#ifdef PA_MEL_PROBE
    MEL_STAMP(5);
    __syncthreads();
    if (lane == 0) {
        float* o = out + ((int64_t)b * p.n_mels + wv * 16) * T + f0;
        o[0] = (float)(stamp[0] & 0xFFFFFFu);                 // absolute start, 10 ns units, 24 bits
        for (int i = 1; i < 6; ++i) o[(int64_t)i * T] = (float)(stamp[i] - stamp[0]);
    }
#endif
    if constexpr (!PERSIST) break;
    tile += (int)gridDim.x;
    if (tile >= n_tiles) break;
    b = tile / tiles_per_clip;
    f0 = (tile - b * tiles_per_clip) * FR_PER_WG;
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
    // profiles/r05_mel_probe.txt): 8 frames per workgroup -- 125 us against 90 at B = 64, and for grids that leave CUs without
    // their three workgroups (ESC-50 at batch 12: 384 workgroups) 23-29 us against 19: the per-workgroup set-up (4 us of a 14 us
    // life) is amortised over half the frames
    hipStream_t st = (hipStream_t)stream;
    // Round 6, measured and NOT adopted (profiles/r06_mel_persistent.txt): the PERSISTENT form -- resident workgroups walk tiles of 8
    // frames, set-up once per workgroup, the next tile's span requested by LDS-DMA into a second buffer while this one is transformed
    // (mel_frontend_kernel<., true>; needs an even hop and room for two span buffers).  97-104 us against 89-92: and with NEITHER the
    // span requests NOR the output stores it still takes 87 us, as does the one-tile form without its stores -- the launch is bound by
    // the frames themselves at three waves per SIMD (~3 400 cycles per frame and SIMD for ~260 VALU + ~75 LDS instructions: dependent
    // butterfly chains and four LDS round trips per frame), not by the per-workgroup set-up the round-5 model blamed.  PA_MEL_PERSIST=1
    // selects it (A/B, parity-tested).
    static const int persist_env = [] { const char* e = getenv("PA_MEL_PERSIST"); return e ? atoi(e) : 0; }();
    if (persist_env && p->hop % 2 == 0) {
        constexpr int FRP = PA_MEL_PERSIST_FRAMES;
        const int spanp = (FRP - 1) * p->hop + NFFT;
        const size_t span_bytes = ((size_t)(spanp + 4) * 4 + 1023) & ~(size_t)1023;
        const size_t ldsp = 2 * span_bytes + MEL_WAVES * WAVE_SCRATCH + std::max<size_t>((size_t)p->n_mels * (FRP + 1) * 4, 2 * NC * 4);
        if (ldsp <= 160 * 1024) {
            static signed char lds_attr_p[64] = {0};
            static int cus[64] = {0};
            int dev = 0;
            if (lds_attr_on_this_device((const void*)mel_frontend_kernel<FRP, true>, 160 * 1024, lds_attr_p) && hipGetDevice(&dev) == hipSuccess &&
                dev >= 0 && dev < 64) {
                if (cus[dev] == 0) {
                    int n = 0;
                    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
                }
                const int tiles_per_clip = (int)cdiv(p->n_frames, FRP);
                const int64_t n_tiles = (int64_t)tiles_per_clip * B;
                const int per_cu = (int)std::max<size_t>(1, (160 * 1024) / ldsp);
                if (n_tiles < ((int64_t)1 << 31)) {
                    const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cus[dev] * per_cu);
                    hipLaunchKernelGGL((mel_frontend_kernel<FRP, true>), dim3((unsigned)grid), dim3(MEL_WAVES * 64), ldsp, st, wave, L, window, bin_mel,
                                       (const float2*)twiddle, out, *p, tiles_per_clip, (int)n_tiles);
                    return check_launch();
                }
            }
        }
    }
    const int fr = FR_DEFAULT;
    const int span = (fr - 1) * p->hop + NFFT;
    const size_t lds = ((span * 4 + 15) & ~15) + MEL_WAVES * WAVE_SCRATCH +
                       std::max<size_t>((size_t)p->n_mels * (fr + 1) * 4, 2 * NC * 4);
    if (lds > 160 * 1024) return PA_EUNSUPPORTED;
    static signed char lds_attr[64] = {0};
    (void)lds_attr_on_this_device((const void*)mel_frontend_kernel<FR_DEFAULT, false>, 160 * 1024, lds_attr);
    dim3 grid((unsigned)cdiv(p->n_frames, fr), (unsigned)B);
    hipLaunchKernelGGL((mel_frontend_kernel<FR_DEFAULT, false>), grid, dim3(MEL_WAVES * 64), lds, st, wave, L, window, bin_mel,
                       (const float2*)twiddle, out, *p, (int)cdiv(p->n_frames, fr), (int)(cdiv(p->n_frames, fr) * B));
    return check_launch();
}
