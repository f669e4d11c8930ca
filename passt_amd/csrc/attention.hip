// Fused (flash-style) multi-head attention, head_dim 64, forward and backward, for gfx950.
// Reference: Attention.forward, models/passt.py:343-361 without the two Linears:
//   attn = softmax((q @ k^T) * scale) ; x = attn @ v          (:348-358)
// The N x N score matrix never reaches HBM (the reference materialises B*H*N*N*4 bytes per layer).
//
// Work decomposition: one 256-thread workgroup = 4 waves x 32 rows of the "owned" sequence axis
// (queries for fwd / dQ, keys for dK,dV); the other axis streams through LDS in tiles of 64 rows.
// All products use 32x32 MFMA tiles in the *transposed* orientation, so the owned row index is the
// lane (lane&31) and softmax statistics are lane-local; the streamed index runs over accumulator
// registers.  P (or dS) feeds the second product directly from those registers (acc_frag), the
// matching operand is read down LDS columns (ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32).
//
// q/k/v are read in place from the qkv GEMM output [B*N][3*H*64]; o / dqkv are token-major.
#include <algorithm>
#include <type_traits>

#include "pa_mma.h"

namespace pa {

// occupancy targets of the bf16 kernels (waves per SIMD); overridable for A/B builds
#ifndef PA_ATTN_FWD_WAVES
#define PA_ATTN_FWD_WAVES 3
#endif
#ifndef PA_ATTN_DQ_WAVES
#define PA_ATTN_DQ_WAVES 3
#endif
#ifndef PA_ATTN_DKDV_WAVES
#define PA_ATTN_DKDV_WAVES 2
#endif
// 1: forward row sums l from a fifth product against a fragment of ones (matrix pipe); 0: 32 VALU adds per tile
#ifndef PA_ATTN_ONES
#define PA_ATTN_ONES 1
#endif
// timing ablations of the forward kernel (WRONG RESULTS; tools/runs/*): bit 0 no running-max logic, 1 no exponentials,
// 2 no P V product, 3 no Q K^T product, 4 no per-tile staging / barrier (tile 0 is reused)
#ifndef PA_ATTN_ABLATE
#define PA_ATTN_ABLATE 0
#endif
// Wave priorities.  The co-resident waves of a SIMD belong to different workgroups that start together and run the same
// code: with equal priority the arbiter round-robins them, so they pass through the matrix phase (Q K^T, P V) and the
// VALU phase (max, exp, convert) of a tile TOGETHER and the two pipes never overlap (r3 ablations: the phase times add
// up).  1: static, distinct priority per hardware wave slot (HW_ID.wave_id & 3): the arbiter then serves the waves in
// a fixed order, they drift apart by one phase and one wave's matrix phase runs under the others' VALU phases.
#ifndef PA_ATTN_PRIO
#define PA_ATTN_PRIO 0
#endif
// forward kernel: 1 = software-pipelined over key tiles (attn_fwd_pipe_kernel), 0 = plain loop
#ifndef PA_ATTN_PIPE
#define PA_ATTN_PIPE 1
#endif
#ifndef PA_ATTN_PIPE_WAVES
#define PA_ATTN_PIPE_WAVES 2
#endif
// 1: sched_group_barrier interleave of part A (1 MFMA : 4 v_exp : 2 v_cvt_pk)
#ifndef PA_ATTN_SGB
#define PA_ATTN_SGB 1
#endif

static constexpr int HD = 64;       // head dim (all PaSST archs: 768/12, 1024/16, 384/6, 128/2)
static constexpr int TROWS = 64;    // streamed rows per LDS tile
static constexpr float LOG2E = 1.4426950408889634f;
static constexpr float LN2 = 0.6931471805599453f;

template <typename T> struct Tile {
    static constexpr int RB = HD * (int)sizeof(T);            // row bytes: 128 (bf16) / 256 (f32)
    static constexpr int CPR = RB / 16;                        // 16-byte chunks per row
    static constexpr int BYTES = TROWS * RB;                   // 8 KiB / 16 KiB
    static constexpr int EPC = 16 / (int)sizeof(T);            // elements per chunk
    static constexpr int NFRAG = RB / 32;                      // row fragments per row: 4 / 8
};

// global [rows][ld] (row index clamped to nrows-1) -> swizzled LDS tile by LDS-DMA (global_load_lds_dwordx4,
// no VGPR round trip).  A wave-instruction fills 1 KiB lane-linearly, so the XOR swizzle is applied to the
// per-lane SOURCE chunk (guide rule 21).  All 4 waves cooperate: BYTES/4096 instructions per wave.
template <typename T>
__device__ __forceinline__ void stage_tile(char* lds, const T* g, int64_t ld, int row0, int nrows, int wave, int lane) {
    constexpr int PER_WAVE = Tile<T>::BYTES / 4096;             // 2 (bf16) / 4 (f32)
    constexpr int RPI = 1024 / Tile<T>::RB;                     // tile rows per wave-instruction: 8 / 4
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int q = wave * PER_WAVE + i;
        const int row = q * RPI + lane / Tile<T>::CPR;
        const int pc = lane % Tile<T>::CPR;
        const int c = Tile<T>::RB == 128 ? (pc ^ swz_f128(row)) : (pc ^ (row & 15));
        const int gr = min(row0 + row, nrows - 1);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(g + (int64_t)gr * ld + c * Tile<T>::EPC),
            (__attribute__((address_space(3))) void*)(lds + q * 1024), 16, 0, 0);
    }
}
// 64 consecutive floats (index clamped) -> LDS, one 4-byte LDS-DMA per lane, issued by ONE wave
__device__ __forceinline__ void stage_f32x64(float* lds, const float* g, int i0, int n, int lane) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + min(i0 + lane, n - 1)),
                                     (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
}

// row fragment s of tile row `row`: 16 bytes at logical chunk s*2 + (lane>>5)
template <typename T>
__device__ __forceinline__ typename Frag<T>::type row_frag(const char* lds, int row, int s, int lane) {
    return *(const typename Frag<T>::type*)(lds + swz<Tile<T>::RB>(row, s * 2 + (lane >> 5)));
}

// column fragment: the MFMA A operand X^T[d][slot] for d = d0 + (lane&31) where the k-slots are the
// tile rows that acc_frag<T>(., s) of the partner operand owns (see pa_mma.h):
//   bf16: rows rbase + 16s + 4h + {0..3} and + 8 more;  f32: rows rbase + 8s + 4h + {0..3}
template <typename T>
__device__ __forceinline__ typename Frag<T>::type col_frag(const char* lds, int rbase, int s, int d0, int lane);
// bf16: ISSUED ONLY (asm reads, see lds_tr16_asm): the fragment is valid after col_settle<>() on it.  With the
// intrinsic form the compiler drains the LDS-DMA of the NEXT tile (s_waitcnt vmcnt(0)) in front of the first
// column read of every tile, i.e. the prefetch never overlaps anything.
template <>
__device__ __forceinline__ bf16x8 col_frag<bf16>(const char* lds, int rbase, int s, int d0, int lane) {
    const int p = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int r1 = 4 * h + (p >> 2);                           // row inside the 16-row group
    const int d = d0 + g * 16 + (p & 3) * 4;                   // first of this lane's 4 source elements
    const int within = (d & 7) * 2;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    // the swizzle only looks at row bits 1..3, so the (rbase + 16 s) rows are a pure byte offset: it rides in the
    // instruction's immediate and the two per-lane addresses are loop invariant up to the tile base
    const int row_off = (rbase + 16 * s) * 128;
    const bf16x4 lo = lds_tr16_asm_imm(base + swz128(r1, d >> 3) + within, row_off);
    const bf16x4 hi = lds_tr16_asm_imm(base + swz128(r1 + 8, d >> 3) + within, row_off);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
template <>
__device__ __forceinline__ f32x4 col_frag<float>(const char* lds, int rbase, int s, int d0, int lane) {
    const int d = d0 + (lane & 31), h = lane >> 5;
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = rbase + 8 * s + 4 * h + e;
        f[e] = *(const float*)(lds + swz256(row, d >> 2) + (d & 3) * 4);
    }
    return f;
}

// Wait until at most PENDING younger LDS operations are outstanding and tie the fragments to the wait, so no use
// of them can be scheduled above it.  f32 fragments come from plain loads the compiler tracks itself: no-op.
template <int PENDING, typename F> __device__ __forceinline__ void col_settle(F& a, F& b) {
    if constexpr (sizeof(F) == 16 && __is_same(F, bf16x8)) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(PENDING));
}
template <int PENDING, typename F> __device__ __forceinline__ void col_settle(F& a, F& b, F& c, F& d) {
    if constexpr (sizeof(F) == 16 && __is_same(F, bf16x8))
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(PENDING));
}

// Transposed store of two 32x32 accumulator tiles acc[db] (lane = owned row, register = d) as rows of
// 64 contiguous elements: through a per-wave [32][65] f32 LDS slab.
template <typename T>
__device__ __forceinline__ void store_rows_T(float* slab, const f32x16 (&acc)[2], float mul, T* gout,
                                             int64_t ld, int row0, int nvalid_rows, int lane) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(lane & 31) * 65 + db * 32 + acc_row(r, lane)] = acc[db][r] * mul;
    // same-wave LDS RAW is ordered; mul may differ per lane (1/l), applied before the transpose.
    // Rows leave as 16-byte vectors: lane -> row it*8 + lane/8, columns (lane&7)*8 .. +8
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), c0 = (lane & 7) * 8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = slab[row * 65 + c0 + e];
        if (row < nvalid_rows) store8<T>(gout + (int64_t)(row0 + row) * ld + c0, v);
    }
}

static constexpr int SLAB_BYTES = 4 * 32 * 65 * 4;   // 33280

// ------------------------------------------------------------------------------------------------
// Instruction budget (round 3).  At head dim 64 a 32-query x 64-key tile is only 16 MFMAs (512 matrix-pipe cycles per
// SIMD), and a wave issues roughly one instruction per 4 cycles: the round-2 kernels spent 228 VALU instructions per
// tile in the forward (15.8 per MFMA, profiles/r02_attention_pmc.txt) and were issue bound at 0.2 of the MFMA peak.
// What the loops below do instead:
//  * the softmax argument comes out of the matrix pipe: the register operand of the score product (Q in fwd / dQ, K in
//    dK/dV) is pre-multiplied by scale*log2(e) once per workgroup, and the first MFMA of every score chain takes the
//    per-row offset (-running max, -lse) as its C operand, so p = exp2(acc) with no multiply-add per score; the same
//    for dP - delta, so dS = p * acc (one multiply) and `scale` is applied once when the result rows are stored;
//  * lazy running max (guide T13): the row max only moves -- and O, l and the C-operand block are only rescaled --
//    when some row of the wave would exceed 2^RESCALE_LOG2; p <= 64 costs nothing in bf16 (same exponent range as f32);
//  * row sums l come from the matrix pipe too: one more product of the P fragments against a fragment of ones
//    (+4 MFMAs per tile for -32 VALU adds; l then uses the same rounded P as the numerator);
//  * the tail-tile masks live in a separate instance of the loop body (the compiler had if-converted them into 43
//    VALU instructions executed on every tile) and the empty second half of the last key tile is skipped.
// Per tile and wave, forward: 20 MFMAs, 32 v_exp + 16 v_max3 + 16 v_cvt_pk + O(5) other VALU.
// ------------------------------------------------------------------------------------------------
static constexpr float RESCALE_LOG2 = 6.0f;

// ---- probe builds only (-DPA_ATTN_PROBE, tools/probe_attn.py): s_memtime at the phase boundaries of the plain forward
// loop, summed over all waves and tiles into g_attn_probe[phase] (cycles) and g_attn_probe[8] (tiles)
#ifdef PA_ATTN_PROBE
static constexpr int PROBE_WAVES = 16384;
__device__ unsigned long long g_attn_probe[PROBE_WAVES * 8];   // per wave: 5 phase sums, tiles, first and last stamp
#define PA_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PA_STAMP(var) do {} while (0)
#endif

template <typename T> __device__ __forceinline__ typename Frag<T>::type frag_splat(float v) {
    typename Frag<T>::type f;
#pragma unroll
    for (int e = 0; e < Tile<T>::EPC; ++e) f[e] = (T)v;
    return f;
}
template <typename T> __device__ __forceinline__ typename Frag<T>::type frag_scale(const typename Frag<T>::type& a, float c) {
    typename Frag<T>::type f;
#pragma unroll
    for (int e = 0; e < Tile<T>::EPC; ++e) f[e] = (T)((float)a[e] * c);
    return f;
}
__device__ __forceinline__ f32x16 acc_splat(float v) {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = v;
    return a;
}
// 16-byte LDS read the compiler does not track (issued where it is written; the caller settles it with frag_settle)
template <typename F> __device__ __forceinline__ F lds_b128_asm(const char* lds_ptr) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_ptr));
    return __builtin_bit_cast(F, r);
}
// pending must fold to a constant after inlining / unrolling
template <typename F> __device__ __forceinline__ void frag_settle(F& a, F& b, int pending) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(pending));
}
__device__ __forceinline__ void wave_static_prio() {
#if PA_ATTN_PRIO == 1
    const uint32_t slot = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 3;   // wave_id[1:0]
    if (slot == 0) __builtin_amdgcn_s_setprio(3);
    else if (slot == 1) __builtin_amdgcn_s_setprio(2);
    else if (slot == 2) __builtin_amdgcn_s_setprio(1);
#elif PA_ATTN_PRIO == 2
    const uint32_t slot = (blockIdx.x >> 3) & 3;
    if (slot == 0) __builtin_amdgcn_s_setprio(3);
    else if (slot == 1) __builtin_amdgcn_s_setprio(2);
    else if (slot == 2) __builtin_amdgcn_s_setprio(1);
#endif
}
// max / sum of a per-lane value with the lane that holds the other half of the same owned row (lane ^ 32), on the
// VALU (v_permlane32_swap; __shfl_xor goes through ds_bpermute and an lgkmcnt wait)
__device__ __forceinline__ float other_half(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0]: lanes 32..63 now hold the lower half's value; r[1]: lanes 0..31 hold the upper half's value
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup -> (block of 128 queries / keys, sequence x head).  The blocks of one head read the same K / V (resp. Q /
// dO) panels, 121 KB per head: they only share them through an L2 if they run on the same XCD, and the dispatcher
// deals consecutive workgroup ids round-robin over the 8 XCDs.  So the grid is one-dimensional and id L is read as
// XCD x = L & 7, slot s = L >> 3: head = x + 8 * (s / nblk), block = s % nblk -- the nblk blocks of a head are
// consecutive slots of ONE XCD.  (With the (block, head) grid they sat on nblk different XCDs and every block
// re-fetched the panels over the fabric: FETCH_SIZE 393 MB per forward launch for 187 MB of operands,
// profiles/r02_pmc_fetch_run_r04.txt.)
__device__ __forceinline__ bool attn_block(int nblk, int BH, int& blk, int& bh) {
    const int L = blockIdx.x, s = L >> 3;
    const int g = s / nblk;
    blk = s - g * nblk;
    bh = (L & 7) + 8 * g;
    return bh < BH;
}
static inline unsigned attn_grid(int nblk, int BH) { return (unsigned)(nblk * 8 * ((BH + 7) / 8)); }

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_FWD_WAVES : 2))) void attn_fwd_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ o,
                                                       int ldo, float* __restrict__ lse, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    wave_static_prio();
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;      // q of token 0 of this (b,h)
    const int q0 = blk * 128 + wave * 32;
    const int qrow = min(q0 + (lane & 31), N - 1);
    const float sl2 = scale * LOG2E;

    F qf[Tile<T>::NFRAG];                                       // Q * scale * log2(e): scores arrive in log2 units
#pragma unroll
    for (int s = 0; s < Tile<T>::NFRAG; ++s)
        qf[s] = frag_scale<T>(*(const F*)(base + (int64_t)qrow * ldqkv + (s * 2 + (lane >> 5)) * Tile<T>::EPC), sl2);
    const F ones = frag_splat<T>(1.0f);

    f32x16 oacc[2] = {acc_splat(0.f), acc_splat(0.f)};
    f32x16 lacc = acc_splat(0.f);                               // every register: l of this lane's query
    float l_valu = 0.f;                                         // PA_ATTN_ONES == 0: l from VALU adds
    f32x16 negm = acc_splat(0.f);                               // C operand of the score chains: -m_run
    float m_run = 0.f;                                          // reference point of the exponentials, log2 units

    const int ntiles = (N + TROWS - 1) / TROWS;
    // double-buffered K/V tiles: stage kt+1 by LDS-DMA while computing kt; one barrier per tile
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * (2 * Tile<T>::BYTES);
        stage_tile<T>(sb, base + D, ldqkv, kt * TROWS, N, wave, lane);
        stage_tile<T>(sb + Tile<T>::BYTES, base + 2 * D, ldqkv, kt * TROWS, N, wave, lane);
    };
    // one key tile; LAST = the tile that may hold keys >= N (masks, skipped half)
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)pt; (void)ps;
    auto tile = [&](auto last_tag, int kt) {
        constexpr bool LAST = decltype(last_tag)::value;
        PA_STAMP(pt[1]);
        const char* sK = smem + (kt & 1) * (2 * Tile<T>::BYTES);
        const char* sV = sK + Tile<T>::BYTES;
        const bool both = !LAST || kt * TROWS + 32 < N;         // wave-uniform: the second 32 keys exist
        f32x16 s[2];
        // S'^T[key][q] = K (Q sl2)^T - m_run
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) break;
            if (PA_ATTN_ABLATE & 8) { s[kb] = negm; asm volatile("" : "+v"(s[kb])); continue; }
            mma32_c<T>(s[kb], row_frag<T>(sK, kb * 32 + (lane & 31), 0, lane), qf[0], negm);
#pragma unroll
            for (int st = 1; st < Tile<T>::NFRAG; ++st)
                mma32<T>(s[kb], row_frag<T>(sK, kb * 32 + (lane & 31), st, lane), qf[st]);
        }
        PA_STAMP(pt[2]);
        if (LAST && (N & (TROWS - 1))) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both) break;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * TROWS + kb * 32 + acc_row(r, lane) >= N) s[kb][r] = -INFINITY;
            }
        }
        // this lane owns query (lane&31) and 16 of the 32 keys of each key block
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
        if (both) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        }
        mx = fmaxf(mx, other_half(mx));
        PA_STAMP(pt[3]);
        if (!(PA_ATTN_ABLATE & 1) && (kt == 0 || !__all(mx <= RESCALE_LOG2))) {
            // move the reference point: exactly to the row max on the first tile, up to it later
            const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= d;
            }
            m_run += d;
            negm = acc_splat(-m_run);
            if (kt != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
                if (PA_ATTN_ONES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
                } else {
                    l_valu *= alpha;
                }
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = (PA_ATTN_ABLATE & 2) ? s[kb][r] : __builtin_amdgcn_exp2f(s[kb][r]);
        }
        PA_STAMP(pt[4]);
        if (!PA_ATTN_ONES) {
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) psum += s[kb][r];
            }
            l_valu += psum + other_half(psum);
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q], l[q] += 1^T P^T; the V column fragments of step i+1 are in flight
        // under the MFMAs of step i
        if (PA_ATTN_ABLATE & 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { oacc[0][r] += s[0][r]; if (both) oacc[1][r] += s[1][r]; }
        } else {
            constexpr int NSB = AccSteps<T>::N;
            const int ns = both ? 2 * NSB : NSB;
            F vfr[2][2];
            auto issue = [&](int slot, int step) {
                const int kb = step / NSB, st = step % NSB;
#pragma unroll
                for (int db = 0; db < 2; ++db) vfr[slot][db] = col_frag<T>(sV, kb * 32, st, db * 32, lane);
            };
            issue(0, 0);
#pragma unroll
            for (int step = 0; step < 2 * NSB; ++step) {
                if (step >= ns) break;
                const int kb = step / NSB, st = step % NSB;
                if (step + 1 < ns) {
                    issue((step + 1) & 1, step + 1);
                    col_settle<4>(vfr[step & 1][0], vfr[step & 1][1]);
                } else {
                    col_settle<0>(vfr[step & 1][0], vfr[step & 1][1]);
                }
                const F pf = acc_frag<T>(s[kb], st);
#pragma unroll
                for (int db = 0; db < 2; ++db) mma32<T>(oacc[db], vfr[step & 1][db], pf);
                if (PA_ATTN_ONES) mma32<T>(lacc, ones, pf);
            }
        }
        PA_STAMP(pt[5]);
#ifdef PA_ATTN_PROBE
        ps[0] += pt[1] - pt[0]; ps[1] += pt[2] - pt[1]; ps[2] += pt[3] - pt[2]; ps[3] += pt[4] - pt[3]; ps[4] += pt[5] - pt[4]; ps[5] += 1;
#endif
    };
    PA_STAMP(pt[6]);
    stage(0, 0);
    for (int kt = 0; kt < ntiles - 1; ++kt) {
        PA_STAMP(pt[0]);
        if (!(PA_ATTN_ABLATE & 16) || kt == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!(PA_ATTN_ABLATE & 16)) stage((kt + 1) & 1, kt + 1);
        }
        tile(std::false_type{}, (PA_ATTN_ABLATE & 16) ? 0 : kt);
    }
    PA_STAMP(pt[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tile(std::true_type{}, (PA_ATTN_ABLATE & 16) ? 0 : ntiles - 1);
#ifdef PA_ATTN_PROBE
    if (lane == 0 && blockIdx.x * 4 + wave < PROBE_WAVES) {
        for (int i = 0; i < 5; ++i) g_attn_probe[(blockIdx.x * 4 + wave) * 8 + i] = ps[i];
        g_attn_probe[(blockIdx.x * 4 + wave) * 8 + 5] = ps[5] | ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 8) |
                                                        ((unsigned long long)__builtin_amdgcn_s_getreg(20 /*XCC_ID*/ | (0 << 6) | (3 << 11)) << 40);
        g_attn_probe[(blockIdx.x * 4 + wave) * 8 + 6] = pt[6];
        g_attn_probe[(blockIdx.x * 4 + wave) * 8 + 7] = pt[5];
    }
#endif
    __syncthreads();   // tiles are dead; reuse LDS for the transposed store
    // only the first nq queries of every sequence are produced; o / lse are compact (nq rows per sequence)
    if (q0 < nq) {
        const float l_run = PA_ATTN_ONES ? lacc[0] : l_valu;
        if (lane < 32 && q0 + lane < nq) lse[(int64_t)bh * nq + q0 + lane] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;   // natural log units
        store_rows_T<T>((float*)smem + wave * (32 * 65), oacc, __builtin_amdgcn_rcpf(l_run), o + (int64_t)b * nq * ldo + h * HD,
                        ldo, q0, min(32, nq - q0), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// forward, software-pipelined over key tiles (PA_ATTN_PIPE).  The plain loop above runs a tile as three dependent
// phases -- Q K^T (matrix pipe), max / exp / convert (VALU), P V (matrix pipe) -- and the waves that share a SIMD start
// together and run the same code, so they sit in the same phase at the same time: the r3 ablation builds show the phase
// times adding up (matrix-pipe busy 0.39 + VALU busy 0.54 of the kernel time) instead of overlapping.  Here every wave
// carries two tiles: iteration j issues
//   part A:  S'(j+1) = K(j+1) Q^T - m   (8 MFMAs)   beside   P(j) = exp2(S'(j)), bf16 fragments   (32 v_exp + 16 v_cvt_pk)
//   part B:  O += V(j) P(j), l += 1 P(j)  (12 MFMAs) beside  row max of S'(j+1) and the (rare) reference move
// so both pipes have independent work in flight at every point of a single wave's instruction stream.
// LDS: K(t) and V(t) live in slot t & 1; iteration j needs K(j+1) and V(j), and its opening barrier frees the slots of
// K(j) (re-filled with K(j+2)) and V(j-1) (re-filled with V(j+1)): one barrier per tile, every DMA has a full iteration to land.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_PIPE_WAVES : 1))) void attn_fwd_pipe_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ o,
                                                       int ldo, float* __restrict__ lse, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    constexpr int NF = Tile<T>::NFRAG, NSB = AccSteps<T>::N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    wave_static_prio();
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;      // q of token 0 of this (b,h)
    const int q0 = blk * 128 + wave * 32;
    const int qrow = min(q0 + (lane & 31), N - 1);
    const float sl2 = scale * LOG2E;

    F qf[NF];                                                   // Q * scale * log2(e): scores arrive in log2 units
#pragma unroll
    for (int s = 0; s < NF; ++s)
        qf[s] = frag_scale<T>(*(const F*)(base + (int64_t)qrow * ldqkv + (s * 2 + (lane >> 5)) * Tile<T>::EPC), sl2);
    const F ones = frag_splat<T>(1.0f);

    f32x16 oacc[2] = {acc_splat(0.f), acc_splat(0.f)};
    f32x16 lacc = acc_splat(0.f);                               // every register: l of this lane's query
    f32x16 negm = acc_splat(0.f);                               // C operand of the score chains: -m_run
    float m_run = 0.f;                                          // reference point of the exponentials, log2 units
    f32x16 s[2];                                                // S'(j), then P(j)

    const int ntiles = (N + TROWS - 1) / TROWS;
    auto slotK = [&](int t) { return smem + (t & 1) * (2 * Tile<T>::BYTES); };
    auto slotV = [&](int t) { return smem + (t & 1) * (2 * Tile<T>::BYTES) + Tile<T>::BYTES; };
    auto stageK = [&](int t) { stage_tile<T>(slotK(t), base + D, ldqkv, t * TROWS, N, wave, lane); };
    auto stageV = [&](int t) { stage_tile<T>(slotV(t), base + 2 * D, ldqkv, t * TROWS, N, wave, lane); };

    // scores of tile t into sn (C operand = -m_run); the second key block only if it holds a key < N
    auto scores = [&](f32x16 (&sn)[2], int t, bool both) {
        const char* sK = slotK(t);
#pragma unroll
        for (int st = 0; st < NF; ++st)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both) continue;
                const F kf = row_frag<T>(sK, kb * 32 + (lane & 31), st, lane);
                if (st == 0) mma32_c<T>(sn[kb], kf, qf[0], negm);
                else mma32<T>(sn[kb], kf, qf[st]);
            }
    };
    // the same for a full bf16 tile with all eight K fragments requested up front (asm reads: the compiler waits for
    // every ds_read right in front of its MFMA, one LDS round trip per product); the waits are counted per product
    auto scores_issue = [&](F (&kfr)[2 * NF], int t) {
        const char* sK = slotK(t);
#pragma unroll
        for (int st = 0; st < NF; ++st)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                kfr[st * 2 + kb] = lds_b128_asm<F>(sK + swz<Tile<T>::RB>(kb * 32 + (lane & 31), st * 2 + (lane >> 5)));
    };
    auto scores_mma = [&](f32x16 (&sn)[2], F (&kfr)[2 * NF]) {
#pragma unroll
        for (int st = 0; st < NF; ++st) {
            frag_settle(kfr[st * 2], kfr[st * 2 + 1], 2 * NF - 2 - 2 * st);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (st == 0) mma32_c<T>(sn[kb], kfr[kb], qf[0], negm);
                else mma32<T>(sn[kb], kfr[st * 2 + kb], qf[st]);
            }
        }
    };
    // keys >= N of the last tile: -inf (one v_cndmask per score: the condition is uniform per half-wave and register)
    auto mask_tail = [&](f32x16 (&sn)[2], int t, bool both) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * TROWS + kb * 32 + acc_row(r, lane) >= N) sn[kb][r] = -INFINITY;
        }
    };
    auto row_max = [&](const f32x16 (&sn)[2], bool both) {
        float mx = sn[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sn[0][r]);
        if (both) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sn[1][r]);
        }
        return fmaxf(mx, other_half(mx));
    };
    // move the reference point by d: scores of the coming tile, the C block and, except before the first tile, O and l
    auto rebase = [&](f32x16 (&sn)[2], float d, bool both, bool scale_acc) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) sn[kb][r] -= d;
        }
        m_run += d;
        negm = acc_splat(-m_run);
        if (scale_acc) {
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; lacc[r] *= alpha; }
        }
    };

    // ---- prologue: K(0), V(0), K(1) -> LDS; S'(0) and its exact row max
    stageK(0);
    stageV(0);
    if (ntiles > 1) stageK(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool both = TROWS / 2 < N;                                  // of the CURRENT tile (wave-uniform)
    scores(s, 0, both);
    if (ntiles == 1 && (N & (TROWS - 1))) mask_tail(s, 0, both);
    rebase(s, row_max(s, both), both, false);

    // HAS_NEXT: tile j+1 exists; NEXT_LAST: it is the last one (tail mask, maybe only one key block)
    // P V of tile j (fragments pf) and, beside it, the row max of the coming tile's scores
    auto pv_steps = [&](const F (&pf)[2][NSB], int j, bool bothc, F (&vfr)[2][2], bool first_issued) {
        const char* sV = slotV(j);
        const int ns = bothc ? 2 * NSB : NSB;
        auto issue = [&](int slot, int step) {
            const int kb = step / NSB, st = step % NSB;
#pragma unroll
            for (int db = 0; db < 2; ++db) vfr[slot][db] = col_frag<T>(sV, kb * 32, st, db * 32, lane);
        };
        if (!first_issued) issue(0, 0);
#pragma unroll
        for (int step = 0; step < 2 * NSB; ++step) {
            if (step >= ns) break;
            const int kb = step / NSB, st = step % NSB;
            if (step + 1 < ns) {
                issue((step + 1) & 1, step + 1);
                col_settle<4>(vfr[step & 1][0], vfr[step & 1][1]);
            } else {
                col_settle<0>(vfr[step & 1][0], vfr[step & 1][1]);
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) mma32<T>(oacc[db], vfr[step & 1][db], pf[kb][st]);
            mma32<T>(lacc, ones, pf[kb][st]);
        }
    };
    auto exp_block = [&](int kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
    };
    auto next_tile_max = [&](f32x16 (&sn)[2], int j, bool bothn, bool next_last) {
        if (next_last && (N & (TROWS - 1))) mask_tail(sn, j + 1, bothn);
        const float mx = row_max(sn, bothn);
        if (!__all(mx <= RESCALE_LOG2)) rebase(sn, fmaxf(mx, 0.f), bothn, true);
        s[0] = sn[0];
        if (bothn) s[1] = sn[1];
        both = bothn;
    };
    // hot iteration (bf16): tiles j and j+1 are full.  Order inside the wave: barrier, DMA requests, ALL LDS requests of
    // part A, then the first sixteen exponentials while those are in flight, the eight score MFMAs with the other
    // sixteen exponentials between them, then part B.
    auto iter_hot = [&](int j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j + 2 < ntiles) stageK(j + 2);
        stageV(j + 1);
        __builtin_amdgcn_sched_barrier(0);
        F kfr[2 * NF], vfr[2][2];
        scores_issue(kfr, j + 1);
#pragma unroll
        for (int db = 0; db < 2; ++db) vfr[0][db] = col_frag<T>(slotV(j), 0, 0, db * 32, lane);
        __builtin_amdgcn_sched_barrier(0);
        F pf[2][NSB];
        exp_block(0);
#pragma unroll
        for (int st = 0; st < NSB; ++st) pf[0][st] = acc_frag<T>(s[0], st);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sn[2];
        // the V requests of step 0 are younger than the K requests: lgkmcnt counts them too
#pragma unroll
        for (int st = 0; st < NF; ++st) {
            frag_settle(kfr[st * 2], kfr[st * 2 + 1], 2 * NF - 2 - 2 * st + 4);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (st == 0) mma32_c<T>(sn[kb], kfr[kb], qf[0], negm);
                else mma32<T>(sn[kb], kfr[st * 2 + kb], qf[st]);
            }
        }
        exp_block(1);
#pragma unroll
        for (int st = 0; st < NSB; ++st) pf[1][st] = acc_frag<T>(s[1], st);
#if PA_ATTN_SGB
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);      // 2 transcendental
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // 1 VALU (convert)
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        pv_steps(pf, j, true, vfr, true);
        next_tile_max(sn, j, true, false);
    };
    // HAS_NEXT: tile j+1 exists; NEXT_LAST: it is the last one (tail mask, maybe only one key block)
    auto iter = [&](auto has_next_tag, auto next_last_tag, int j) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value, NEXT_LAST = decltype(next_last_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j + 2 < ntiles) stageK(j + 2);
        if (HAS_NEXT) stageV(j + 1);
        const bool bothn = !NEXT_LAST || (j + 1) * TROWS + 32 < N;
        f32x16 sn[2];
        if (HAS_NEXT) scores(sn, j + 1, bothn);
        exp_block(0);
        if (both) exp_block(1);
        F pf[2][NSB], vfr[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < NSB; ++st)
                if (kb == 0 || both) pf[kb][st] = acc_frag<T>(s[kb], st);
        __builtin_amdgcn_sched_barrier(0);
        pv_steps(pf, j, both, vfr, false);
        if (HAS_NEXT) next_tile_max(sn, j, bothn, NEXT_LAST);
    };
    for (int j = 0; j < ntiles - 2; ++j) {
        if constexpr (sizeof(T) == 2) iter_hot(j);
        else iter(std::true_type{}, std::false_type{}, j);
    }
    if (ntiles > 1) iter(std::true_type{}, std::true_type{}, ntiles - 2);
    iter(std::false_type{}, std::false_type{}, ntiles - 1);

    __syncthreads();   // tiles are dead; reuse LDS for the transposed store
    // only the first nq queries of every sequence are produced; o / lse are compact (nq rows per sequence)
    if (q0 < nq) {
        const float l_run = lacc[0];
        if (lane < 32 && q0 + lane < nq) lse[(int64_t)bh * nq + q0 + lane] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;   // natural log units
        store_rows_T<T>((float*)smem + wave * (32 * 65), oacc, __builtin_amdgcn_rcpf(l_run), o + (int64_t)b * nq * ldo + h * HD,
                        ldo, q0, min(32, nq - q0), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 1: dK, dV.  Workgroup owns 128 keys (lane = key); queries stream through LDS.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_DKDV_WAVES : 1))) void attn_bwd_dkdv_kernel(const T* __restrict__ qkv, int ldqkv,
                                                            const T* __restrict__ d_o, int ldo,
                                                            const float* __restrict__ ws, int64_t plane,
                                                            T* __restrict__ dqkv, int lddqkv, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    wave_static_prio();
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;
    const T* dobase = d_o + (int64_t)b * nq * ldo + h * HD;     // d_o / lse / delta: nq rows per sequence
    const int k0 = blk * 128 + wave * 32;
    const int key = k0 + (lane & 31);
    const int krow = min(key, N - 1);
    const bool active = k0 < N;                                 // wave-uniform
    const float sl2 = scale * LOG2E;

    F kf[Tile<T>::NFRAG], vf[Tile<T>::NFRAG];                   // kf = K * scale * log2(e) (used for the scores only)
#pragma unroll
    for (int s = 0; s < Tile<T>::NFRAG; ++s) {
        const int off = (s * 2 + (lane >> 5)) * Tile<T>::EPC;
        kf[s] = frag_scale<T>(*(const F*)(base + D + (int64_t)krow * ldqkv + off), sl2);
        vf[s] = *(const F*)(base + 2 * D + (int64_t)krow * ldqkv + off);
    }
    f32x16 dk[2] = {acc_splat(0.f), acc_splat(0.f)}, dv[2] = {acc_splat(0.f), acc_splat(0.f)};

    const int ntiles = (nq + TROWS - 1) / TROWS;               // only queries < nq carry a gradient
    constexpr int STAGE = 2 * Tile<T>::BYTES + 2 * TROWS * 4;   // Q tile, dO tile, -lse*log2e [64], -delta [64]
    auto stage = [&](int buf, int qt) {
        char* sb = smem + buf * STAGE;
        stage_tile<T>(sb, base, ldqkv, qt * TROWS, N, wave, lane);
        stage_tile<T>(sb + Tile<T>::BYTES, dobase, ldo, qt * TROWS, nq, wave, lane);
        // both per-query scalars come from the dQ kernel's workspace in the form the score chains take as C operand
        if (wave == 0) stage_f32x64((float*)(sb + 2 * Tile<T>::BYTES), ws + plane + (int64_t)bh * nq, qt * TROWS, nq, lane);
        if (wave == 1) stage_f32x64((float*)(sb + 2 * Tile<T>::BYTES) + TROWS, ws + (int64_t)bh * nq, qt * TROWS, nq, lane);
    };
    // one block of 32 queries against this wave's 32 keys
    auto block = [&](auto last_tag, int qt, int qb) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* sQ = smem + (qt & 1) * STAGE;
        const char* sDO = sQ + Tile<T>::BYTES;
        const float* sLse = (const float*)(sQ + 2 * Tile<T>::BYTES);
        const float* sDelta = sLse + TROWS;
        // C operands: accumulator rows 4g..4g+3 are the 4 consecutive queries 8g + 4*(lane>>5) + {0..3}
        f32x16 sa, dpa;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ql = qb * 32 + 8 * g + 4 * (lane >> 5);
            const f32x4 nl = *(const f32x4*)(sLse + ql), dl = *(const f32x4*)(sDelta + ql);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sa[4 * g + e] = nl[e]; dpa[4 * g + e] = dl[e]; }
        }
        // S'[q][key] = Q (K sl2)^T - lse ; dP'[q][key] = dO V^T - delta   (A rows = q, B cols = key = lane)
#pragma unroll
        for (int st = 0; st < Tile<T>::NFRAG; ++st) {
            mma32<T>(sa, row_frag<T>(sQ, qb * 32 + (lane & 31), st, lane), kf[st]);
            mma32<T>(dpa, row_frag<T>(sDO, qb * 32 + (lane & 31), st, lane), vf[st]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(sa[r]);
            sa[r] = p;
            dpa[r] *= p;                                        // dS / scale
        }
        // queries beyond nq exist only in the last tile; lanes whose own key is beyond N only produce their own,
        // never stored, outputs and need no mask
        if (LAST && (nq & (TROWS - 1))) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (qt * TROWS + qb * 32 + acc_row(r, lane) >= nq) { sa[r] = 0.f; dpa[r] = 0.f; }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
        constexpr int NS = AccSteps<T>::N;
        F cf[2][4];
        auto issue = [&](int slot, int st) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                cf[slot][db] = col_frag<T>(sDO, qb * 32, st, db * 32, lane);
                cf[slot][2 + db] = col_frag<T>(sQ, qb * 32, st, db * 32, lane);
            }
        };
        issue(0, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + 1 < NS) {
                issue((st + 1) & 1, st + 1);
                col_settle<8>(cf[st & 1][0], cf[st & 1][1], cf[st & 1][2], cf[st & 1][3]);
            } else {
                col_settle<0>(cf[st & 1][0], cf[st & 1][1], cf[st & 1][2], cf[st & 1][3]);
            }
            const F pf = acc_frag<T>(sa, st), dsf = acc_frag<T>(dpa, st);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                mma32<T>(dv[db], cf[st & 1][db], pf);
                mma32<T>(dk[db], cf[st & 1][2 + db], dsf);
            }
        }
    };
    stage(0, 0);
    for (int qt = 0; qt < ntiles - 1; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage((qt + 1) & 1, qt + 1);
        if (!active) continue;          // all 32 keys of this wave are past N: stage and meet barriers only
        block(std::false_type{}, qt, 0);
        block(std::false_type{}, qt, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
        const int qt = ntiles - 1;
        block(std::true_type{}, qt, 0);
        if (qt * TROWS + 32 < nq) block(std::true_type{}, qt, 1);      // else: the second 32 queries do not exist
    }
    __syncthreads();
    if (k0 < N) {
        float* slab = (float*)smem + wave * (32 * 65);
        T* out = dqkv + (int64_t)b * N * lddqkv + h * HD;
        store_rows_T<T>(slab, dk, scale, out + D, lddqkv, k0, min(32, N - k0), lane);
        store_rows_T<T>(slab, dv, 1.0f, out + 2 * D, lddqkv, k0, min(32, N - k0), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 2: dQ.  Workgroup owns 128 queries (lane = query); keys stream through LDS.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_DQ_WAVES : 1))) void attn_bwd_dq_kernel(const T* __restrict__ qkv, int ldqkv,
                                                          const T* __restrict__ o, const T* __restrict__ d_o, int ldo,
                                                          const float* __restrict__ lse, float* __restrict__ delta, int64_t plane,
                                                          T* __restrict__ dqkv, int lddqkv, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    wave_static_prio();
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;
    const T* dobase = d_o + (int64_t)b * nq * ldo + h * HD;     // d_o / lse / delta: nq rows per sequence
    const int q0 = blk * 128 + wave * 32;
    const int q = q0 + (lane & 31);
    const int qrow = min(q, nq - 1);
    const bool active = q0 < nq;                                // wave-uniform
    const float sl2 = scale * LOG2E;

    F qf[Tile<T>::NFRAG], dof[Tile<T>::NFRAG];                  // qf = Q * scale * log2(e)
    float dlt = 0.f;
    {
        // delta[q] = sum_d dO[q][d] O[q][d]: this lane holds half of row q of dO as fragments already; the same
        // chunks of O are read once here, and the row sum is published for the dK/dV kernel (launched after)
        const T* obase = o + (int64_t)b * nq * ldo + h * HD;
#pragma unroll
        for (int s = 0; s < Tile<T>::NFRAG; ++s) {
            const int off = (s * 2 + (lane >> 5)) * Tile<T>::EPC;
            qf[s] = frag_scale<T>(*(const F*)(base + (int64_t)qrow * ldqkv + off), sl2);
            dof[s] = *(const F*)(dobase + (int64_t)qrow * ldo + off);
            const F of = *(const F*)(obase + (int64_t)qrow * ldo + off);
#pragma unroll
            for (int e = 0; e < Tile<T>::EPC; ++e) dlt = fmaf((float)dof[s][e], (float)of[e], dlt);
        }
        dlt += other_half(dlt);
    }
    const float lse2 = lse[(int64_t)bh * nq + qrow] * LOG2E;
    // workspace for the dK/dV kernel, already in the form its score chains take as C operand: -delta, -lse*log2(e)
    if (lane < 32 && q < nq) {
        delta[(int64_t)bh * nq + q] = -dlt;
        delta[plane + (int64_t)bh * nq + q] = -lse2;
    }
    const f32x16 neglse = acc_splat(-lse2), negdl = acc_splat(-dlt);
    f32x16 dq[2] = {acc_splat(0.f), acc_splat(0.f)};

    const int ntiles = (N + TROWS - 1) / TROWS;
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * (2 * Tile<T>::BYTES);
        stage_tile<T>(sb, base + D, ldqkv, kt * TROWS, N, wave, lane);
        stage_tile<T>(sb + Tile<T>::BYTES, base + 2 * D, ldqkv, kt * TROWS, N, wave, lane);
    };
    // one block of 32 keys against this wave's 32 queries
    auto block = [&](auto last_tag, int kt, int kb) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* sK = smem + (kt & 1) * (2 * Tile<T>::BYTES);
        const char* sV = sK + Tile<T>::BYTES;
        f32x16 sa, dpa;
        // S'^T[key][q] = K (Q sl2)^T - lse ; dP'^T[key][q] = V dO^T - delta
        mma32_c<T>(sa, row_frag<T>(sK, kb * 32 + (lane & 31), 0, lane), qf[0], neglse);
        mma32_c<T>(dpa, row_frag<T>(sV, kb * 32 + (lane & 31), 0, lane), dof[0], negdl);
#pragma unroll
        for (int st = 1; st < Tile<T>::NFRAG; ++st) {
            mma32<T>(sa, row_frag<T>(sK, kb * 32 + (lane & 31), st, lane), qf[st]);
            mma32<T>(dpa, row_frag<T>(sV, kb * 32 + (lane & 31), st, lane), dof[st]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dpa[r] *= __builtin_amdgcn_exp2f(sa[r]);      // dS^T / scale
        if (LAST && (N & (TROWS - 1))) {                        // keys beyond N: last tile only
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * TROWS + kb * 32 + acc_row(r, lane) >= N) dpa[r] = 0.f;
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
        constexpr int NS = AccSteps<T>::N;
        F cf[2][2];
        auto issue = [&](int slot, int st) {
#pragma unroll
            for (int db = 0; db < 2; ++db) cf[slot][db] = col_frag<T>(sK, kb * 32, st, db * 32, lane);
        };
        issue(0, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + 1 < NS) {
                issue((st + 1) & 1, st + 1);
                col_settle<4>(cf[st & 1][0], cf[st & 1][1]);
            } else {
                col_settle<0>(cf[st & 1][0], cf[st & 1][1]);
            }
            const F dsf = acc_frag<T>(dpa, st);
#pragma unroll
            for (int db = 0; db < 2; ++db) mma32<T>(dq[db], cf[st & 1][db], dsf);
        }
    };
    stage(0, 0);
    for (int kt = 0; kt < ntiles - 1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage((kt + 1) & 1, kt + 1);
        if (!active) continue;          // all 32 queries of this wave are past nq
        block(std::false_type{}, kt, 0);
        block(std::false_type{}, kt, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
        const int kt = ntiles - 1;
        block(std::true_type{}, kt, 0);
        if (kt * TROWS + 32 < N) block(std::true_type{}, kt, 1);       // else: the second 32 keys do not exist
    }
    __syncthreads();
    if (q0 < nq)
        store_rows_T<T>((float*)smem + wave * (32 * 65), dq, scale, dqkv + (int64_t)b * N * lddqkv + h * HD,
                        lddqkv, q0, min(32, nq - q0), lane);
}

template <typename T> static size_t fwd_lds() { return std::max<size_t>(4 * Tile<T>::BYTES, SLAB_BYTES); }
template <typename T> static size_t dkdv_lds() { return std::max<size_t>(2 * (2 * Tile<T>::BYTES + 2 * TROWS * 4), SLAB_BYTES); }

template <typename T>
static int attention_fwd_t(const void* qkv, int ldqkv, void* o, int ldo, float* lse, int B, int H, int N, int nq,
                           float scale, hipStream_t st) {
    const int nblk = (int)cdiv(nq, 128);
    if (PA_ATTN_PIPE)
        hipLaunchKernelGGL(attn_fwd_pipe_kernel<T>, dim3(attn_grid(nblk, B * H)), dim3(256), fwd_lds<T>(), st, (const T*)qkv, ldqkv,
                           (T*)o, ldo, lse, H, N, nq, scale, nblk, B * H);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<T>, dim3(attn_grid(nblk, B * H)), dim3(256), fwd_lds<T>(), st, (const T*)qkv, ldqkv, (T*)o,
                           ldo, lse, H, N, nq, scale, nblk, B * H);
    return check_launch();
}

template <typename T>
static int attention_bwd_t(const void* qkv, int ldqkv, const void* o, const void* d_o, int ldo, const float* lse,
                           float* delta, void* dqkv, int lddqkv, int B, int H, int N, int nq, float scale, hipStream_t st) {
    // dQ first: it also fills the workspace the dK/dV kernel consumes, two planes of B*H*nq floats:
    // rowsum(dO * O) * scale and -lse * log2(e)
    const int64_t plane = (int64_t)B * H * nq;
    const int nblkq = (int)cdiv(nq, 128), nblkk = (int)cdiv(N, 128);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3(attn_grid(nblkq, B * H)), dim3(256), fwd_lds<T>(), st, (const T*)qkv, ldqkv,
                       (const T*)o, (const T*)d_o, ldo, lse, delta, plane, (T*)dqkv, lddqkv, H, N, nq, scale, nblkq, B * H);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel<T>, dim3(attn_grid(nblkk, B * H)), dim3(256), dkdv_lds<T>(), st, (const T*)qkv, ldqkv,
                       (const T*)d_o, ldo, delta, plane, (T*)dqkv, lddqkv, H, N, nq, scale, nblkk, B * H);
    return check_launch();
}

}  // namespace pa

using namespace pa;

static bool attn_args_ok(int ld, int dtype) {
    const int es = dtype == PA_BF16 ? 2 : 4;
    return (ld * es) % 16 == 0;
}

#ifdef PA_ATTN_PROBE
// probe library only: read (and clear) the phase-cycle sums of the plain forward kernel
extern "C" int pa_attn_probe_read(unsigned long long* host_out) {   // PROBE_WAVES * 8 values
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pa::g_attn_probe), sizeof(unsigned long long) * pa::PROBE_WAVES * 8);
    return e == hipSuccess ? PA_OK : pa::set_hip_error(e);
}
#endif

extern "C" int pa_attention_fwd(const void* qkv, int ldqkv, void* o, int ldo, float* lse, int B, int H, int N, int nq,
                                float scale, int dtype, void* stream) {
    if (!qkv || !o || !lse || B <= 0 || H <= 0 || N <= 0 || nq <= 0 || nq > N) return PA_EINVAL;
    if (!attn_args_ok(ldqkv, dtype) || !attn_args_ok(ldo, dtype)) return PA_EUNSUPPORTED;
    if (dtype == PA_BF16) return attention_fwd_t<bf16>(qkv, ldqkv, o, ldo, lse, B, H, N, nq, scale, (hipStream_t)stream);
    if (dtype == PA_F32) return attention_fwd_t<float>(qkv, ldqkv, o, ldo, lse, B, H, N, nq, scale, (hipStream_t)stream);
    return PA_EINVAL;
}

extern "C" int pa_attention_bwd(const void* qkv, int ldqkv, const void* o, const void* d_o, int ldo,
                                const float* lse, float* delta, void* dqkv, int lddqkv, int B, int H, int N, int nq,
                                float scale, int dtype, void* stream) {
    if (!qkv || !o || !d_o || !lse || !delta || !dqkv || B <= 0 || H <= 0 || N <= 0 || nq <= 0 || nq > N) return PA_EINVAL;
    if (!attn_args_ok(ldqkv, dtype) || !attn_args_ok(ldo, dtype) || !attn_args_ok(lddqkv, dtype)) return PA_EUNSUPPORTED;
    if (dtype == PA_BF16) return attention_bwd_t<bf16>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, (hipStream_t)stream);
    if (dtype == PA_F32) return attention_bwd_t<float>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, (hipStream_t)stream);
    return PA_EINVAL;
}
