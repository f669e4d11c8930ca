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
// Wave priorities (A/B knob).  1: static, distinct priority per hardware wave slot (HW_ID.wave_id & 3), 2: per workgroup
// id.  Tried in round 3 against the observation that the co-resident waves of a SIMD pass through the matrix phase and
// the VALU phase of a tile together: no gain (fwd 71.4 -> 74.5 us, profiles/r03_attention_experiments.md), default off.
#ifndef PA_ATTN_PRIO
#define PA_ATTN_PRIO 0
#endif
// Packed f32 math on accumulator register pairs (A/B knob, round 4): the row sums of the forward as 16 v_pk_add_f32 instead of
// 32 v_add_f32, the p * dP products of the backward as 8 v_pk_mul_f32 instead of 16 v_mul_f32 (pairs (r, r + 1), r even, are
// 64-bit aligned in an MFMA accumulator block).
#ifndef PA_ATTN_PK
#define PA_ATTN_PK 0
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

// ---- lane-constant address parts, computed once per kernel (round 3) -------------------------------------------
// The tile loops used to spend ~30 VALU instructions per 64-row tile on addresses (64-bit row pointers of the LDS-DMA
// requests with their clamps, swizzled LDS offsets); with the matrix pipe and the VALU of a SIMD running one after the
// other (tests/probes/probe_mfma_valu_overlap.hip) every one of them is kernel time.  What is left in the loops is one
// uniform pointer per tensor and tile (SALU) and one v_add per LDS address register and tile.
template <typename T> struct LaneOff {
    static constexpr int PER_WAVE = Tile<T>::BYTES / 4096;      // LDS-DMA requests per wave and tile: 2 (bf16) / 4 (f32)
    uint32_t rowf[Tile<T>::NFRAG];                              // LDS: row fragment st of tile row (lane & 31)
    uint32_t colf[2][2];                                        // LDS (bf16): the two transposed 8-byte reads of column block db
};
// global byte offsets of this lane's 16-byte chunk in the wave's DMA requests for a tile whose rows 0..row_limit exist
// (rows beyond are clamped to row_limit); ldb = row pitch in bytes
template <typename T>
__device__ __forceinline__ void stage_offsets(uint32_t (&voff)[LaneOff<T>::PER_WAVE], int ldb, int row_limit, int wave, int lane) {
    constexpr int RPI = 1024 / Tile<T>::RB;
#pragma unroll
    for (int i = 0; i < LaneOff<T>::PER_WAVE; ++i) {
        const int row = (wave * LaneOff<T>::PER_WAVE + i) * RPI + lane / Tile<T>::CPR;
        const int pc = lane % Tile<T>::CPR;
        const int c = Tile<T>::RB == 128 ? (pc ^ swz_f128(row)) : (pc ^ (row & 15));
        voff[i] = (uint32_t)min(row, row_limit) * (uint32_t)ldb + (uint32_t)c * 16u;
    }
}
// tile_base: uniform pointer to row 0 of the tile in global memory
template <typename T>
__device__ __forceinline__ void stage_tile_off(char* lds, const char* tile_base, const uint32_t (&voff)[LaneOff<T>::PER_WAVE], int wave) {
#pragma unroll
    for (int i = 0; i < LaneOff<T>::PER_WAVE; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tile_base + voff[i]),
                                         (__attribute__((address_space(3))) void*)(lds + (wave * LaneOff<T>::PER_WAVE + i) * 1024), 16, 0, 0);
}
template <typename T> __device__ __forceinline__ void lane_offsets(LaneOff<T>& lo, int lane) {
    const int row = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < Tile<T>::NFRAG; ++st) lo.rowf[st] = (uint32_t)swz<Tile<T>::RB>(row, st * 2 + h);
    if constexpr (sizeof(T) == 2) {
        const int p = lane & 15, g = (lane >> 4) & 1;
        const int r1 = 4 * h + (p >> 2);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int d = db * 32 + g * 16 + (p & 3) * 4;
            lo.colf[db][0] = (uint32_t)(swz128(r1, d >> 3) + (d & 7) * 2);
            lo.colf[db][1] = (uint32_t)(swz128(r1 + 8, d >> 3) + (d & 7) * 2);
        }
    }
}
// row fragment st of tile row rbase + (lane & 31), rbase a multiple of 32 (the swizzle does not see it): lds = tile base
template <typename T>
__device__ __forceinline__ typename Frag<T>::type row_frag_off(const char* lds, const LaneOff<T>& lo, int rbase, int st) {
    return *(const typename Frag<T>::type*)(lds + rbase * Tile<T>::RB + lo.rowf[st]);
}
// column fragment (see col_frag) from the precomputed lane offsets; bf16: issued only, settle with col_settle<>()
template <typename T>
__device__ __forceinline__ typename Frag<T>::type col_frag_off(const char* lds, const LaneOff<T>& lo, int rbase, int s, int db, int lane) {
    if constexpr (sizeof(T) == 2) {
        const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
        const int row_off = (rbase + 16 * s) * 128;
        const bf16x4 lw = lds_tr16_asm_imm(base + lo.colf[db][0], row_off);
        const bf16x4 hi = lds_tr16_asm_imm(base + lo.colf[db][1], row_off);
        bf16x8 f;
        f[0] = lw[0]; f[1] = lw[1]; f[2] = lw[2]; f[3] = lw[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    } else {
        return col_frag<T>(lds, rbase, s, db * 32, lane);
    }
}

// Wait until at most PENDING younger LDS operations are outstanding and tie the fragments to the wait, so no use
// of them can be scheduled above it.  f32 fragments come from plain loads the compiler tracks itself: no-op.
#ifndef PA_ATTN_DEBUG_WAIT
#define PA_ATTN_DEBUG_WAIT 0
#endif

template <int PENDING, typename F> __device__ __forceinline__ void col_settle(F& a, F& b) {
    if constexpr (sizeof(F) == 16 && __is_same(F, bf16x8)) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(PA_ATTN_DEBUG_WAIT ? 0 : PENDING));
}
template <int PENDING, typename F> __device__ __forceinline__ void col_settle(F& a, F& b, F& c, F& d) {
    if constexpr (sizeof(F) == 16 && __is_same(F, bf16x8))
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(PA_ATTN_DEBUG_WAIT ? 0 : PENDING));
}

// Store two 32x32 accumulator tiles acc[db] (lane = owned row, register = d) as rows of 64 contiguous elements
// straight from the registers: the lanes (q, h = 0 / 1) of a row hold alternating groups of 4
// consecutive columns, 8g + 4h + {0..3}.  f32: one 16-byte store per group.  bf16: a group is 8 bytes; the two lanes of a
// row trade the odd / even groups with v_permlane32_swap (guide T21) and store 16 bytes each.
template <typename T>
__device__ __forceinline__ void store_rows_direct(const f32x16 (&acc)[2], float mul, T* gout, int64_t ld, int row0,
                                                  int nvalid_rows, int lane) {
    const int q = lane & 31, h = lane >> 5;
    T* rowp = gout + (int64_t)(row0 + q) * ld;
    const bool ok = q < nvalid_rows;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul, acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul};
                if (ok) *(f32x4*)(rowp + db * 32 + 8 * g + 4 * h) = v;
            }
        } else {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t w[4];                                   // groups 2gp (w[0..1]) and 2gp + 1 (w[2..3]) as packed bf16 pairs
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x2 pk = {(bf16)(acc[db][8 * gp + 2 * i] * mul), (bf16)(acc[db][8 * gp + 2 * i + 1] * mul)};
                    w[i] = __builtin_bit_cast(uint32_t, pk);
                }
                // lower half keeps its group 2gp and receives the upper half's; upper half keeps 2gp+1 and receives the lower's
                const auto r0 = __builtin_amdgcn_permlane32_swap(w[0], w[2], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(w[1], w[3], false, false);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
                if (ok) *(u32x4*)(rowp + db * 32 + 16 * gp + 8 * h) = v;
            }
        }
    }
}


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

template <typename T> __device__ __forceinline__ typename Frag<T>::type frag_splat(float v) {
    typename Frag<T>::type f;
#pragma unroll
    for (int e = 0; e < Tile<T>::EPC; ++e) f[e] = (T)v;
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
// Work item -> (block of 128 queries / keys, sequence x head).  The blocks of one head read the same K / V (resp. Q /
// dO) panels, 121 KB per head: they only share them through an L2 if they run on the same XCD, and the dispatcher
// deals consecutive workgroup ids round-robin over the 8 XCDs.  So item id L is read as XCD x = L & 7, slot s = L >> 3:
// head = x + 8 * (s / nblk), block = s % nblk -- the nblk blocks of a head are consecutive slots of ONE XCD.  (With a
// (block, head) grid they sat on nblk different XCDs and every block re-fetched the panels over the fabric: FETCH_SIZE
// 393 MB per forward launch for 187 MB of operands, profiles/r02_pmc_fetch_run_r04.txt.)
__device__ __forceinline__ bool attn_item(int L, int nblk, int BH, int& blk, int& bh) {
    const int s = L >> 3;
    const int g = s / nblk;
    blk = s - g * nblk;
    bh = (L & 7) + 8 * g;
    return bh < BH;
}
__device__ __forceinline__ bool attn_block(int nblk, int BH, int& blk, int& bh) { return attn_item(blockIdx.x, nblk, BH, blk, bh); }
static inline unsigned attn_grid(int nblk, int BH) { return (unsigned)(nblk * 8 * ((BH + 7) / 8)); }

// Round 3 (profiles/r03_attention_experiments.md).  The matrix pipe and the VALU of a SIMD do not overlap across waves
// (tests/probes/probe_mfma_valu_overlap.hip: 8 MFMAs + 40 FMAs from two waves take 199 ns against 158 + 56 alone), and
// the kernel's own counters say the same (matrix-busy 0.39 + VALU-busy 0.54 of the time): its time is the SUM of the
// two instruction streams, so every VALU instruction per tile is ~1 ns per wave-tile.  Variants that re-arranged the
// same instructions all measured slower than this plain loop at 3 waves per SIMD: software-pipelined over two key tiles
// (2 waves), persistent with the next item's operands requested ahead, one 32-key block at a time (4 waves, spills),
// static per-slot wave priorities.  So this loop is about instruction COUNT: scores leave the MFMA as "score - reference"
// (C operand), lazy reference moves, row sums as plain adds (a fifth product against ones costs 4 MFMAs = 76 ns per tile,
// 32 adds cost 32), tail masks only in the tail instance, addresses precomputed per lane (LaneOff), rows stored straight
// from the registers, and waves without a query row skip the arithmetic.
template <typename T, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_FWD_WAVES : 2))) void attn_fwd_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ o,
                                                       int ldo, float* __restrict__ lse, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    constexpr int NF = Tile<T>::NFRAG, NSB = AccSteps<T>::N, TB = Tile<T>::BYTES;
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
    const bool active = q0 < nq;                                // wave-uniform: this wave owns at least one stored query
    const float sl2 = scale * LOG2E;
    const int ldb = ldqkv * (int)sizeof(T);

    F qf[NF];                                                   // PRE: Q * scale * log2(e), scores arrive in log2 units
#pragma unroll
    for (int s = 0; s < NF; ++s) qf[s] = *(const F*)(base + (int64_t)qrow * ldqkv + (s * 2 + (lane >> 5)) * Tile<T>::EPC);
    LaneOff<T> lo;
    lane_offsets<T>(lo, lane);
    uint32_t voff[LaneOff<T>::PER_WAVE];
    stage_offsets<T>(voff, ldb, TROWS, wave, lane);             // full tiles: no clamp

    f32x16 oacc[2] = {acc_splat(0.f), acc_splat(0.f)};
    f32x16 negm = acc_splat(0.f);                               // C operand of the score chains: -m_run
    float m_run = 0.f, l_run = 0.f;                             // reference point (log2 units); this lane's HALF of the row sum

    const int ntiles = (N + TROWS - 1) / TROWS;
    // double-buffered K/V tiles: stage kt+1 by LDS-DMA while computing kt; one barrier per tile
    const char* gK = (const char*)(base + D);
    const char* gV = (const char*)(base + 2 * D);
    auto stage = [&](int buf, int kt, const uint32_t (&vo)[LaneOff<T>::PER_WAVE]) {
        char* sb = smem + buf * (2 * TB);
        const int64_t row0 = (int64_t)kt * TROWS * ldb;
        stage_tile_off<T>(sb, gK + row0, vo, wave);
        stage_tile_off<T>(sb + TB, gV + row0, vo, wave);
    };
    // one key tile; LAST = the tile that may hold keys >= N (masks), BOTH = its second 32 keys exist.  Both are compile
    // time: the P V loop below must be straight-line code.  Its column fragments come from asm LDS reads the compiler does
    // not track; with a run-time step count it merged the paths with register copies of fragments whose data was still in
    // flight (placed in front of the counted wait) -- sporadic garbage rows at B = 64, tools/check_lds_asm.py lints for it.
    auto tile = [&](auto last_tag, auto both_tag, int kt) {
        constexpr bool LAST = decltype(last_tag)::value, both = decltype(both_tag)::value;
        const char* sK = smem + (kt & 1) * (2 * TB);
        const char* sV = sK + TB;
        f32x16 s[2];
        // S'^T[key][q] = K (Q sl2)^T - m_run
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) break;
            if constexpr (PRE) mma32_c<T>(s[kb], row_frag_off<T>(sK, lo, kb * 32, 0), qf[0], negm);
            else mma32_first<T>(s[kb], row_frag_off<T>(sK, lo, kb * 32, 0), qf[0]);
#pragma unroll
            for (int st = 1; st < NF; ++st) mma32<T>(s[kb], row_frag_off<T>(sK, lo, kb * 32, st), qf[st]);
            if constexpr (!PRE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = fmaf(s[kb][r], sl2, -m_run);
            }
        }
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
        if (kt == 0 || !__all(mx <= RESCALE_LOG2)) {
            // move the reference point: exactly to the row max on the first tile, up to it later
            const float d = kt == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (kb == 1 && !both) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= d;
            }
            m_run += d;
            if constexpr (PRE) negm = acc_splat(-m_run);
            if (kt != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int r = 0; r < 16; ++r) { oacc[0][r] *= alpha; oacc[1][r] *= alpha; }
                l_run *= alpha;
            }
        }
        float psum = 0.f;
#if PA_ATTN_PK
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) break;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
                s[kb][r + 1] = __builtin_amdgcn_exp2f(s[kb][r + 1]);
                ps2 += f32x2{s[kb][r], s[kb][r + 1]};
            }
        }
        psum = ps2[0] + ps2[1];
#else
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && !both) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
                psum += s[kb][r];
            }
        }
#endif
        l_run += psum;
        // O^T[d][q] += V^T[d][key] P^T[key][q]; the V column fragments of step i+1 are in flight under the MFMAs of step i
        constexpr int ns = both ? 2 * NSB : NSB;
        F vfr[2][2];
        auto issue = [&](int slot, int step) {
            const int kb = step / NSB, st = step % NSB;
#pragma unroll
            for (int db = 0; db < 2; ++db) vfr[slot][db] = col_frag_off<T>(sV, lo, kb * 32, st, db, lane);
        };
        issue(0, 0);
#pragma unroll
        for (int step = 0; step < ns; ++step) {
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
        }
    };
    // the last tile's rows are clamped to the last key (requested once; the only tile if N <= 64)
    auto stage_last = [&](int buf) {
        uint32_t vt[LaneOff<T>::PER_WAVE];
        stage_offsets<T>(vt, ldb, N - 1 - (ntiles - 1) * TROWS, wave, lane);
        stage(buf, ntiles - 1, vt);
    };
    if (ntiles == 1) stage_last(0);
    else stage(0, 0, voff);
    for (int kt = 0; kt < ntiles - 1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < ntiles) stage((kt + 1) & 1, kt + 1, voff);
        else stage_last((kt + 1) & 1);
        if (active) tile(std::false_type{}, std::true_type{}, kt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
        if ((ntiles - 1) * TROWS + 32 < N) tile(std::true_type{}, std::true_type{}, ntiles - 1);
        else tile(std::true_type{}, std::false_type{}, ntiles - 1);
        // only the first nq queries of every sequence are produced; o / lse are compact (nq rows per sequence)
        l_run += other_half(l_run);
        if (lane < 32 && q0 + lane < nq) lse[(int64_t)bh * nq + q0 + lane] = (m_run + __builtin_amdgcn_logf(l_run)) * LN2;   // natural log units
        store_rows_direct<T>(oacc, __builtin_amdgcn_rcpf(l_run), o + (int64_t)b * nq * ldo + h * HD, ldo, q0, min(32, nq - q0), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 1: dK, dV.  Workgroup owns 128 keys (lane = key); queries stream through LDS.
// ------------------------------------------------------------------------------------------------
template <typename T, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_DKDV_WAVES : 1))) void attn_bwd_dkdv_kernel(const T* __restrict__ qkv, int ldqkv,
                                                            const T* __restrict__ d_o, int ldo,
                                                            const float* __restrict__ ws, int64_t plane,
                                                            T* __restrict__ dqkv, int lddqkv, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    constexpr int NF = Tile<T>::NFRAG, NS = AccSteps<T>::N, TB = Tile<T>::BYTES;
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
    const int krow = min(k0 + (lane & 31), N - 1);
    const bool active = k0 < N;                                 // wave-uniform
    const float sl2 = scale * LOG2E;
    const int ldbq = ldqkv * (int)sizeof(T), ldbo = ldo * (int)sizeof(T);

    F kf[NF], vf[NF];
#pragma unroll
    for (int s = 0; s < NF; ++s) {
        const int off = (s * 2 + (lane >> 5)) * Tile<T>::EPC;
        kf[s] = *(const F*)(base + D + (int64_t)krow * ldqkv + off);
        vf[s] = *(const F*)(base + 2 * D + (int64_t)krow * ldqkv + off);
    }
    LaneOff<T> lo;
    lane_offsets<T>(lo, lane);
    uint32_t voq[LaneOff<T>::PER_WAVE], voo[LaneOff<T>::PER_WAVE];
    stage_offsets<T>(voq, ldbq, TROWS, wave, lane);
    stage_offsets<T>(voo, ldbo, TROWS, wave, lane);
    f32x16 dk[2] = {acc_splat(0.f), acc_splat(0.f)}, dv[2] = {acc_splat(0.f), acc_splat(0.f)};

    const int ntiles = (nq + TROWS - 1) / TROWS;               // only queries < nq carry a gradient
    constexpr int STAGE = 2 * TB + 2 * TROWS * 4;               // Q tile, dO tile, -lse*log2e [64], -delta [64]
    auto stage = [&](int buf, int qt, const uint32_t (&vq)[LaneOff<T>::PER_WAVE], const uint32_t (&vo)[LaneOff<T>::PER_WAVE]) {
        char* sb = smem + buf * STAGE;
        stage_tile_off<T>(sb, (const char*)base + (int64_t)qt * TROWS * ldbq, vq, wave);
        stage_tile_off<T>(sb + TB, (const char*)dobase + (int64_t)qt * TROWS * ldbo, vo, wave);
        // both per-query scalars come from the dQ kernel's workspace in the form the score chains take as C operand
        if (wave == 0) stage_f32x64((float*)(sb + 2 * TB), ws + plane + (int64_t)bh * nq, qt * TROWS, nq, lane);
        if (wave == 1) stage_f32x64((float*)(sb + 2 * TB) + TROWS, ws + (int64_t)bh * nq, qt * TROWS, nq, lane);
    };
    auto stage_last = [&](int buf) {
        uint32_t vq[LaneOff<T>::PER_WAVE], vo[LaneOff<T>::PER_WAVE];
        const int lim = nq - 1 - (ntiles - 1) * TROWS;
        stage_offsets<T>(vq, ldbq, lim, wave, lane);
        stage_offsets<T>(vo, ldbo, lim, wave, lane);
        stage(buf, ntiles - 1, vq, vo);
    };
    // one block of 32 queries against this wave's 32 keys
    auto block = [&](auto last_tag, int qt, int qb) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* sQ = smem + (qt & 1) * STAGE;
        const char* sDO = sQ + TB;
        const float* sLse = (const float*)(sQ + 2 * TB);
        const float* sDelta = sLse + TROWS;
        // per-query offsets: accumulator rows 4g..4g+3 are the 4 consecutive queries 8g + 4*(lane>>5) + {0..3}
        f32x16 sa, dpa, nl;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ql = qb * 32 + 8 * g + 4 * (lane >> 5);
            const f32x4 a = *(const f32x4*)(sLse + ql), d = *(const f32x4*)(sDelta + ql);
#pragma unroll
            for (int e = 0; e < 4; ++e) { nl[4 * g + e] = a[e]; dpa[4 * g + e] = d[e]; }
        }
        // S'[q][key] = (Q sl2) K^T - lse ; dP'[q][key] = dO V^T - delta   (A rows = q, B cols = key = lane)
        if constexpr (PRE) {
            sa = nl;
#pragma unroll
            for (int st = 0; st < NF; ++st) {
                mma32<T>(sa, row_frag_off<T>(sQ, lo, qb * 32, st), kf[st]);
                mma32<T>(dpa, row_frag_off<T>(sDO, lo, qb * 32, st), vf[st]);
            }
        } else {
            mma32_first<T>(sa, row_frag_off<T>(sQ, lo, qb * 32, 0), kf[0]);
            mma32<T>(dpa, row_frag_off<T>(sDO, lo, qb * 32, 0), vf[0]);
#pragma unroll
            for (int st = 1; st < NF; ++st) {
                mma32<T>(sa, row_frag_off<T>(sQ, lo, qb * 32, st), kf[st]);
                mma32<T>(dpa, row_frag_off<T>(sDO, lo, qb * 32, st), vf[st]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[r] = fmaf(sa[r], sl2, nl[r]);
        }
#if PA_ATTN_PK
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 pp = {__builtin_amdgcn_exp2f(sa[r]), __builtin_amdgcn_exp2f(sa[r + 1])};
            const f32x2 d2 = f32x2{dpa[r], dpa[r + 1]} * pp;    // dS / scale
            sa[r] = pp[0]; sa[r + 1] = pp[1];
            dpa[r] = d2[0]; dpa[r + 1] = d2[1];
        }
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(sa[r]);
            sa[r] = p;
            dpa[r] *= p;                                        // dS / scale
        }
#endif
        // queries beyond nq exist only in the last tile; lanes whose own key is beyond N only produce their own,
        // never stored, outputs and need no mask
        if (LAST && (nq & (TROWS - 1))) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (qt * TROWS + qb * 32 + acc_row(r, lane) >= nq) { sa[r] = 0.f; dpa[r] = 0.f; }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
        F cf[2][4];
        auto issue = [&](int slot, int st) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                cf[slot][db] = col_frag_off<T>(sDO, lo, qb * 32, st, db, lane);
                cf[slot][2 + db] = col_frag_off<T>(sQ, lo, qb * 32, st, db, lane);
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
    if (ntiles == 1) stage_last(0);
    else stage(0, 0, voq, voo);
    for (int qt = 0; qt < ntiles - 1; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 2 < ntiles) stage((qt + 1) & 1, qt + 1, voq, voo);
        else stage_last((qt + 1) & 1);
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
        T* out = dqkv + (int64_t)b * N * lddqkv + h * HD;
        // PRE: the Q rows in memory are Q * scale * log2(e): dK = dS^T Q * scale = acc * ln 2
        store_rows_direct<T>(dk, PRE ? LN2 : scale, out + D, lddqkv, k0, min(32, N - k0), lane);
        store_rows_direct<T>(dv, 1.0f, out + 2 * D, lddqkv, k0, min(32, N - k0), lane);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 2: dQ.  Workgroup owns 128 queries (lane = query); keys stream through LDS.
// ------------------------------------------------------------------------------------------------
template <typename T, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 2 ? PA_ATTN_DQ_WAVES : 1))) void attn_bwd_dq_kernel(const T* __restrict__ qkv, int ldqkv,
                                                          const T* __restrict__ o, const T* __restrict__ d_o, int ldo,
                                                          const float* __restrict__ lse, float* __restrict__ delta, int64_t plane,
                                                          T* __restrict__ dqkv, int lddqkv, int H, int N, int nq, float scale, int nblk, int BH) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using F = typename Frag<T>::type;
    constexpr int NF = Tile<T>::NFRAG, NS = AccSteps<T>::N, TB = Tile<T>::BYTES;
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
    const int ldb = ldqkv * (int)sizeof(T);

    F qf[NF], dof[NF];
    float dlt = 0.f;
    {
        // delta[q] = sum_d dO[q][d] O[q][d]: this lane holds half of row q of dO as fragments already; the same
        // chunks of O are read once here, and the row sum is published for the dK/dV kernel (launched after)
        const T* obase = o + (int64_t)b * nq * ldo + h * HD;
#pragma unroll
        for (int s = 0; s < NF; ++s) {
            const int off = (s * 2 + (lane >> 5)) * Tile<T>::EPC;
            qf[s] = *(const F*)(base + (int64_t)qrow * ldqkv + off);
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
    // C operands of the two score chains: -lse stays in a block of 16 registers; -delta is splatted per chain (168
    // registers = 3 waves per SIMD do not hold both blocks, and a spill inside the tile loop makes the compiler drain
    // vmcnt -- i.e. wait for the next tile's LDS-DMA -- in front of the reload)
    const f32x16 neglse = acc_splat(-lse2);
    const float ndl = -dlt;
    f32x16 dq[2] = {acc_splat(0.f), acc_splat(0.f)};
    LaneOff<T> lo;
    lane_offsets<T>(lo, lane);
    uint32_t voff[LaneOff<T>::PER_WAVE];
    stage_offsets<T>(voff, ldb, TROWS, wave, lane);

    const int ntiles = (N + TROWS - 1) / TROWS;
    const char* gK = (const char*)(base + D);
    const char* gV = (const char*)(base + 2 * D);
    auto stage = [&](int buf, int kt, const uint32_t (&vo)[LaneOff<T>::PER_WAVE]) {
        char* sb = smem + buf * (2 * TB);
        const int64_t row0 = (int64_t)kt * TROWS * ldb;
        stage_tile_off<T>(sb, gK + row0, vo, wave);
        stage_tile_off<T>(sb + TB, gV + row0, vo, wave);
    };
    auto stage_last = [&](int buf) {
        uint32_t vt[LaneOff<T>::PER_WAVE];
        stage_offsets<T>(vt, ldb, N - 1 - (ntiles - 1) * TROWS, wave, lane);
        stage(buf, ntiles - 1, vt);
    };
    // one block of 32 keys against this wave's 32 queries
    auto block = [&](auto last_tag, int kt, int kb) {
        constexpr bool LAST = decltype(last_tag)::value;
        const char* sK = smem + (kt & 1) * (2 * TB);
        const char* sV = sK + TB;
        f32x16 sa, dpa;
        // S'^T[key][q] = K (Q sl2)^T - lse ; dP'^T[key][q] = V dO^T - delta
        if constexpr (PRE) mma32_c<T>(sa, row_frag_off<T>(sK, lo, kb * 32, 0), qf[0], neglse);
        else mma32_first<T>(sa, row_frag_off<T>(sK, lo, kb * 32, 0), qf[0]);
        {
            float t = ndl;
            asm volatile("" : "+v"(t));                         // keep the splat inside the loop (not 16 hoisted registers)
            dpa = acc_splat(t);
        }
        mma32<T>(dpa, row_frag_off<T>(sV, lo, kb * 32, 0), dof[0]);
#pragma unroll
        for (int st = 1; st < NF; ++st) {
            mma32<T>(sa, row_frag_off<T>(sK, lo, kb * 32, st), qf[st]);
            mma32<T>(dpa, row_frag_off<T>(sV, lo, kb * 32, st), dof[st]);
        }
#if PA_ATTN_PK
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 pp = {__builtin_amdgcn_exp2f(PRE ? sa[r] : fmaf(sa[r], sl2, -lse2)), __builtin_amdgcn_exp2f(PRE ? sa[r + 1] : fmaf(sa[r + 1], sl2, -lse2))};
            const f32x2 d2 = f32x2{dpa[r], dpa[r + 1]} * pp;    // dS^T / scale
            dpa[r] = d2[0]; dpa[r + 1] = d2[1];
        }
#else
#pragma unroll
        for (int r = 0; r < 16; ++r) dpa[r] *= __builtin_amdgcn_exp2f(PRE ? sa[r] : fmaf(sa[r], sl2, -lse2));      // dS^T / scale
#endif
        if (LAST && (N & (TROWS - 1))) {                        // keys beyond N: last tile only
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt * TROWS + kb * 32 + acc_row(r, lane) >= N) dpa[r] = 0.f;
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
        F cf[2][2];
        auto issue = [&](int slot, int st) {
#pragma unroll
            for (int db = 0; db < 2; ++db) cf[slot][db] = col_frag_off<T>(sK, lo, kb * 32, st, db, lane);
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
    if (ntiles == 1) stage_last(0);
    else stage(0, 0, voff);
    for (int kt = 0; kt < ntiles - 1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < ntiles) stage((kt + 1) & 1, kt + 1, voff);
        else stage_last((kt + 1) & 1);
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
        // gradient with respect to the TRUE q in both modes (the upstream linear layer is differentiated as unscaled)
        store_rows_direct<T>(dq, scale, dqkv + (int64_t)b * N * lddqkv + h * HD, lddqkv, q0, min(32, nq - q0), lane);
    }
}

template <typename T> static size_t fwd_lds() { return 4 * Tile<T>::BYTES; }   // K/V double buffer
template <typename T> static size_t dkdv_lds() { return 2 * (2 * Tile<T>::BYTES + 2 * TROWS * 4); }

template <typename T>
static int attention_fwd_t(const void* qkv, int ldqkv, void* o, int ldo, float* lse, int B, int H, int N, int nq,
                           float scale, int flags, hipStream_t st) {
    const int nblk = (int)cdiv(nq, 128);
    const dim3 grid(attn_grid(nblk, B * H)), block(256);
    if (flags & PA_ATTN_Q_PRESCALED)
        hipLaunchKernelGGL((attn_fwd_kernel<T, true>), grid, block, fwd_lds<T>(), st, (const T*)qkv, ldqkv, (T*)o, ldo, lse, H, N, nq, scale, nblk, B * H);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<T, false>), grid, block, fwd_lds<T>(), st, (const T*)qkv, ldqkv, (T*)o, ldo, lse, H, N, nq, scale, nblk, B * H);
    return check_launch();
}

template <typename T, bool PRE>
static int attention_bwd_t(const void* qkv, int ldqkv, const void* o, const void* d_o, int ldo, const float* lse,
                           float* delta, void* dqkv, int lddqkv, int B, int H, int N, int nq, float scale, hipStream_t st) {
    // dQ first: it also fills the workspace the dK/dV kernel consumes, two planes of B*H*nq floats:
    // -rowsum(dO * O) and -lse * log2(e)
    const int64_t plane = (int64_t)B * H * nq;
    const int nblkq = (int)cdiv(nq, 128), nblkk = (int)cdiv(N, 128);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, PRE>), dim3(attn_grid(nblkq, B * H)), dim3(256), fwd_lds<T>(), st, (const T*)qkv, ldqkv,
                       (const T*)o, (const T*)d_o, ldo, lse, delta, plane, (T*)dqkv, lddqkv, H, N, nq, scale, nblkq, B * H);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, PRE>), dim3(attn_grid(nblkk, B * H)), dim3(256), dkdv_lds<T>(), st, (const T*)qkv, ldqkv,
                       (const T*)d_o, ldo, delta, plane, (T*)dqkv, lddqkv, H, N, nq, scale, nblkk, B * H);
    return check_launch();
}

}  // namespace pa

using namespace pa;

static bool attn_args_ok(int ld, int dtype) {
    const int es = dtype == PA_BF16 ? 2 : 4;
    return (ld * es) % 16 == 0;
}

extern "C" int64_t pa_attention_bwd_ws_floats(int B, int H, int nq) { return 2 * (int64_t)B * H * nq; }

extern "C" int pa_attention_fwd(const void* qkv, int ldqkv, void* o, int ldo, float* lse, int B, int H, int N, int nq,
                                float scale, int dtype, int flags, void* stream) {
    if (!qkv || !o || !lse || B <= 0 || H <= 0 || N <= 0 || nq <= 0 || nq > N || (flags & ~PA_ATTN_Q_PRESCALED)) return PA_EINVAL;
    if (!attn_args_ok(ldqkv, dtype) || !attn_args_ok(ldo, dtype)) return PA_EUNSUPPORTED;
    if (dtype == PA_BF16) return attention_fwd_t<bf16>(qkv, ldqkv, o, ldo, lse, B, H, N, nq, scale, flags, (hipStream_t)stream);
    if (dtype == PA_F32) return attention_fwd_t<float>(qkv, ldqkv, o, ldo, lse, B, H, N, nq, scale, flags, (hipStream_t)stream);
    return PA_EINVAL;
}

extern "C" int pa_attention_bwd(const void* qkv, int ldqkv, const void* o, const void* d_o, int ldo,
                                const float* lse, float* delta, void* dqkv, int lddqkv, int B, int H, int N, int nq,
                                float scale, int dtype, int flags, void* stream) {
    if (!qkv || !o || !d_o || !lse || !delta || !dqkv || B <= 0 || H <= 0 || N <= 0 || nq <= 0 || nq > N || (flags & ~PA_ATTN_Q_PRESCALED)) return PA_EINVAL;
    if (!attn_args_ok(ldqkv, dtype) || !attn_args_ok(ldo, dtype) || !attn_args_ok(lddqkv, dtype)) return PA_EUNSUPPORTED;
    const bool pre = flags & PA_ATTN_Q_PRESCALED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PA_BF16)
        return pre ? attention_bwd_t<bf16, true>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st)
                   : attention_bwd_t<bf16, false>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st);
    if (dtype == PA_F32)
        return pre ? attention_bwd_t<float, true>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st)
                   : attention_bwd_t<float, false>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st);
    return PA_EINVAL;
}
