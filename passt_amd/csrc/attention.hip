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
// Cache policy of the result stores (A/B build knob, round 6 sweep profiles/r06_cache_policy.txt): 1 = non-temporal
#ifndef PA_ATTN_FUSED_NT_LD
#define PA_ATTN_FUSED_NT_LD 0      // 2: the single-pass backward stages K and the Q / dO tiles (each read by ONE workgroup) non-temporally
#endif
#ifndef PA_ATTN_NT_KV
#define PA_ATTN_NT_KV 0            // 2: streamed tiles of the forward / two-kernel backward non-temporally
#endif
#ifndef PA_ATTN_NT_ST
#define PA_ATTN_NT_ST 0
#endif
template <typename V> __device__ __forceinline__ void attn_store(V* p, const V& v) {
#if PA_ATTN_NT_ST
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
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
                                         (__attribute__((address_space(3))) void*)(lds + (wave * LaneOff<T>::PER_WAVE + i) * 1024), 16, 0, PA_ATTN_NT_KV);
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
                if (ok) attn_store((f32x4*)(rowp + db * 32 + 8 * g + 4 * h), v);
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
                if (ok) attn_store((u32x4*)(rowp + db * 32 + 16 * gp + 8 * h), v);
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
// the same with an immediate offset (must fold to a constant after inlining / unrolling)
template <typename F> __device__ __forceinline__ F lds_b128_imm(uint32_t lds_addr, int off) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "i"(off));
    return __builtin_bit_cast(F, r);
}
// pending must fold to a constant after inlining / unrolling
template <typename F> __device__ __forceinline__ void frag_settle(F& a, F& b, int pending) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(pending));
}
// one fragment (never pass the SAME variable twice to frag_settle: two tied operands of one asm get a register copy of the
// in-flight value in front of the wait)
template <typename F> __device__ __forceinline__ void frag_settle1(F& a, int pending) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "i"(pending));
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

// ------------------------------------------------------------------------------------------------
// backward, SINGLE PASS (round 5): bf16, q pre-scaled, nq == N <= 512.  One 512-thread workgroup per (sequence, head).
//
// The two-kernel backward above issues 7 tile products for the 5 the algorithm has (S and dP are formed once with lane =
// query for dQ and once with lane = key for dK / dV) and evaluates P = exp2(S - lse) twice.  Here every 32 x 32 block of
// (S, dP) is formed ONCE, in the orientation dK / dV need (lane = key): wave w owns keys [64 w, 64 w + 64) for the whole
// kernel, its dK / dV accumulators (128 registers) and V row fragments (32) stay in registers, the K rows of the head
// stay in LDS (they feed the score product as row fragments and the dQ product as column fragments), and Q / dO stream
// through LDS in double-buffered tiles of 32 queries.  dQ needs dS with lane = query -- the other orientation -- and a
// sum over ALL keys, i.e. over all waves.  Both are solved through LDS at once: every wave drops its bf16 dS block into
// the transposition buffer T[key][query] (4 ds_write_b64 per block; it rounds dS to bf16 for the dK product anyway), and
// after ONE barrier per query tile ("phase 2") each wave owns one 16 x 16 tile of dQ^T[64 d][32 q] and contracts it over
// all keys with v_mfma_f32_16x16x32_bf16, both operands fetched by transposed LDS reads (K^T from the resident K rows,
// dS^T from T): no cross-wave reduction of partial sums, no atomics, deterministic.  T is double buffered, so phase 2 of
// tile t overlaps phase 1 of tile t + 1 on the other wave of the SIMD.
// Per (64 keys x 32 queries): 32 MFMAs of 32x32x16 + 16 of 16x16x32 (= 8 of the former in matrix time) instead of 56, one exp
// pass instead of two, no delta hand-over launch; Q / dO / K / V are read from HBM once per head instead of 4 + 4 times.
// LDS: K 64 KiB + T 2 x 32 KiB + Q/dO stages 2 x 8 KiB + per-query scalars 4 KiB = 148 KiB (one workgroup per CU, two
// waves per SIMD, 256 registers each).  Index math of T and of the phase-2 operand fetches: tools/emulate_attn_bwd_fused.py
// (values and bank conflicts under the MI355X lane-group rules, on the CPU).
// ------------------------------------------------------------------------------------------------
// timing ablations (A/B builds only, results are wrong): 1 = no phase 2, 2 = no exp / dS arithmetic, 4 = no dV / dK products (and no
// column fragments, no T writes), 8 = no score products, 16 = no barrier per tile
#ifndef PA_FUSED_ABL
#define PA_FUSED_ABL 0
#endif


static constexpr int FK = 512;                                   // key rows of the resident K tile (N <= FK)
static constexpr int FT_PLANE = FK * 32;                         // one 16-query plane of T: FK keys x 16 queries x 2 B
static constexpr int F_OFF_T = FK * 128;
static constexpr int F_STAGE = 2 * 32 * 128;                     // Q tile + dO tile, 32 rows each
static constexpr int F_OFF_STAGE = F_OFF_T + 2 * 2 * FT_PLANE;
static constexpr int F_OFF_LSE = F_OFF_STAGE + 2 * F_STAGE;
static constexpr int F_OFF_DELTA = F_OFF_LSE + FK * 4;
static constexpr int F_LDS = F_OFF_DELTA + FK * 4;               // 151 552 B

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// column fragment (see col_frag) from two absolute per-lane LDS addresses and an immediate: issued only, settle with col_settle<>()
__device__ __forceinline__ bf16x8 tr_frag(uint32_t a0, uint32_t a1, int imm) {
    const bf16x4 lw = lds_tr16_asm_imm(a0, imm), hi = lds_tr16_asm_imm(a1, imm);
    bf16x8 f;
    f[0] = lw[0]; f[1] = lw[1]; f[2] = lw[2]; f[3] = lw[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// W16 (round 6): the same kernel with SIXTEEN waves of 32 keys (1024 threads, four waves per SIMD, <= 128
// registers) instead of eight waves of 64 keys: twice the independent chains per SIMD in front of the same LDS round trips.  One
// key block per wave (dK / dV accumulators 64 registers); the LDS image is unchanged; the Q / dO stage requests and phase 2 (eight
// 16 x 16 tiles of dQ^T) are done by waves 0-7 -- two per SIMD (a workgroup's waves go round the SIMDs in fours).
template <bool W16>
__global__ __launch_bounds__(W16 ? 1024 : 512) __attribute__((amdgpu_waves_per_eu(W16 ? 4 : 2, W16 ? 4 : 2))) void attn_bwd_fused_kernel(
    const bf16* __restrict__ qkv, int ldqkv, const bf16* __restrict__ o, const bf16* __restrict__ d_o, int ldo,
    const float* __restrict__ lse, bf16* __restrict__ dqkv, int lddqkv, int H, int N, float scale) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    using T = bf16;
    using F = bf16x8;
    constexpr int NF = 4, NS = 2;
    constexpr int NKB = W16 ? 1 : 2;                             // 32-key blocks per wave
    constexpr int NT = W16 ? 1024 : 512;                         // threads
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;       // q of token 0 of this (b, h)
    const T* dobase = d_o + (int64_t)b * N * ldo + h * HD;
    const T* obase = o + (int64_t)b * N * ldo + h * HD;
    const int ldbq = ldqkv * 2, ldbo = ldo * 2;
    char* sK = smem;
    char* sT = smem + F_OFF_T;
    char* sStage = smem + F_OFF_STAGE;
    float* sLse = (float*)(smem + F_OFF_LSE);
    float* sDelta = (float*)(smem + F_OFF_DELTA);
    const int hf = lane >> 5, l31 = lane & 31;
    const int nt = (N + 31) >> 5;                                // query tiles == key steps of phase 2
    constexpr int nk4 = FK / 32;                                 // phase 2 always walks all FK key rows (16 steps): rows up to there must be finite

    // ---- prologue.  K rows -> sK (8 rows per 1 KiB request, all FK rows, rows >= N as copies of row N - 1, zeroed below); T starts
    // all-zero: phase 2 reads T rows no wave ever writes (keys in [32 nt, FK)) and multiplies them by the zero rows of K, so they
    // must be finite.  (Measured and dropped, profiles/r05_attention_single_pass.txt: V rows through an LDS-DMA into the T region
    // instead of the strided fragment loads below: +-0; per-query scalars from eight lanes per row + three cross-lane adds: +9 us.)
    {
        const char* gK = (const char*)(base + D);
#pragma unroll
        for (int i = 0; i < 4 * NKB; ++i) {
            const int rq = wave * (4 * NKB) + i;
            const int row = rq * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_f128(row);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gK + (int64_t)min(row, N - 1) * ldbq + c * 16),
                                             (__attribute__((address_space(3))) void*)(sK + rq * 1024), 16, 0, PA_ATTN_FUSED_NT_LD);
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 4 * FT_PLANE / (NT * 16); ++i) *(u32x4*)(sT + (i * NT + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    // Q / dO tile t -> stage buffer: 8 requests of 8 rows, one per wave (waves 0-3: Q, 4-7: dO; W16: waves 8-15 request nothing)
    const int st_tensor = (wave >> 2) & 1, st_piece = wave & 3;
    const int st_row = st_piece * 8 + (lane >> 3);
    const int st_chunk = ((lane & 7) ^ swz_f128(st_row)) * 16;
    const char* st_src = st_tensor ? (const char*)dobase : (const char*)base;
    const int st_ldb = st_tensor ? ldbo : ldbq;
    auto stage = [&](int t) __attribute__((always_inline)) {
        if (W16 && wave >= 8) return;
        const int grow = min(t * 32 + st_row, N - 1);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st_src + (int64_t)grow * st_ldb + st_chunk),
                                         (__attribute__((address_space(3))) void*)(sStage + (t & 1) * F_STAGE + st_tensor * 4096 + st_piece * 1024), 16, 0, PA_ATTN_FUSED_NT_LD);
    };
    stage(0);
    const int kbase0 = wave * (32 * NKB);
    F vf[NKB][NF];                                               // V row fragments of this wave's keys (B operand of the dP products)
    // W16 has no 16 registers to keep them in: they are fetched again for every tile (four 16-byte loads per lane out of the L2,
    // requested at the top of the block, used behind the score chain and the exponentials) from this per-lane row pointer
    const T* vrow = base + 2 * D + (int64_t)min(kbase0 + l31, N - 1) * ldqkv + hf * 8;
    if constexpr (!W16) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int krow = min(kbase0 + kb * 32 + l31, N - 1);
#pragma unroll
            for (int s = 0; s < NF; ++s) vf[kb][s] = *(const F*)(base + 2 * D + (int64_t)krow * ldqkv + (s * 2 + hf) * 8);
        }
    }
    // per-query scalars in the form the score chains take as C operand: -lse * log2(e), -delta = -rowsum(dO * O).  Wave w owns
    // queries [32 NKB w, 32 NKB (w + 1)).  Queries past N (last tile; their Q / dO rows are copies of row N - 1) get the C operand -inf:
    // P = exp2(-inf) = 0 and dS = P * finite = 0 -- they contribute nothing to dK / dV and no tile needs a mask.
    {
        // two lanes per row (what the dQ kernel of the two-kernel backward does)
#pragma unroll
        for (int r = 0; r < NKB; ++r) {
            const int q = wave * (32 * NKB) + r * 32 + l31;
            if (wave * (32 * NKB) + r * 32 < nt * 32) {          // wave-uniform
                const int qrow = min(q, N - 1);
                float dlt = 0.f;
#pragma unroll
                for (int s = 0; s < NF; ++s) {
                    const int off = (s * 2 + hf) * 8;
                    const F df = *(const F*)(dobase + (int64_t)qrow * ldo + off);
                    const F of = *(const F*)(obase + (int64_t)qrow * ldo + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dlt = fmaf((float)df[e], (float)of[e], dlt);
                }
                dlt += other_half(dlt);
                if (lane < 32) {
                    sDelta[q] = -dlt;
                    sLse[q] = q < N ? -lse[(int64_t)bh * N + qrow] * LOG2E : -INFINITY;
                }
            }
        }
    }
    // ---- per-lane ABSOLUTE LDS addresses, set up once: every access of the tile loop is one of these registers plus an
    // immediate (the stage / T buffer of a tile is a compile-time constant: the tile loop is unrolled by two)
    typedef __attribute__((address_space(3))) char lchar;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lchar*)smem;
    LaneOff<T> lo;
    lane_offsets<T>(lo, lane);
    uint32_t aS[NF], aC[2][2];
#pragma unroll
    for (int st = 0; st < NF; ++st) aS[st] = lds0 + F_OFF_STAGE + lo.rowf[st];    // row fragment st of stage row (lane & 31): Q (+ BUF * F_STAGE), dO (+ 4096)
    // the same fragment of this wave's K row (key block 1: + 4096) is aS[st] + kdelta: four v_add per block instead of four registers
    const int kdelta = kbase0 * 128 - F_OFF_STAGE;
#pragma unroll
    for (int db = 0; db < 2; ++db) { aC[db][0] = lds0 + F_OFF_STAGE + lo.colf[db][0]; aC[db][1] = lds0 + F_OFF_STAGE + lo.colf[db][1]; }
    // T writes of phase 1: row = key, 8-byte slot (4 queries) h + 2 e, XOR-swizzled by key bits 2..3 (key block 1 = + 32 keys =
    // + 1024 bytes, same swizzle; the second slot is the first ^ 2: byte offset ^ 16, T is 32-byte aligned inside the 1 KiB-aligned
    // dynamic LDS).  Lanes whose key is past N write their (finite) dS column like everybody else: phase 2 multiplies it by the
    // ZERO rows the prologue put behind key N - 1 of the K tile.
    const uint32_t aT0 = lds0 + (uint32_t)(F_OFF_T + (kbase0 + l31) * 32 + ((hf ^ (((kbase0 + l31) >> 2) & 3)) << 3));
    // phase 2: this wave's tile of dQ^T is (d16, q16); lane = (g4, p): piece row r = p >> 2, column group cg = p & 3
    const int d16 = wave & 3, q16 = wave >> 2;
    auto ld128 = [](uint32_t addr) { return *(const __attribute__((address_space(3))) F*)(size_t)addr; };
    auto ld4f = [](uint32_t addr) { return *(const __attribute__((address_space(3))) f32x4*)(size_t)addr; };
    auto st64 = [](uint32_t addr, uint32_t w0, uint32_t w1) { *(__attribute__((address_space(3))) u32x2_t*)(size_t)addr = u32x2_t{w0, w1}; };
    f32x16 dk[NKB][2], dv[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int db = 0; db < 2; ++db) { dk[kb][db] = acc_splat(0.f); dv[kb][db] = acc_splat(0.f); }

    // one block: 32 queries of tile t (stage / T buffer BUF) against key block kb of this wave
    auto block = [&](auto kb_tag, auto buf_tag, int t) __attribute__((always_inline)) {
        constexpr int kb = decltype(kb_tag)::value, BUF = decltype(buf_tag)::value;
        constexpr int OQ = BUF * F_STAGE, ODO = OQ + 4096;
        f32x16 sa, dpa;
        // per-query scalars of accumulator rows 8 g + 4 h + {0..3}: + g * 32 (recomputed per block: the kernel has no register to spare)
        uint32_t hsel = (uint32_t)lane;
        asm volatile("" : "+v"(hsel));
        const uint32_t sc = lds0 + F_OFF_LSE + ((hsel >> 5) << 4) + (uint32_t)t * 128u;
        // S'[q][key] = (Q sl2) K^T - lse      (A rows = q, B cols = key = lane); C operand = -lse * log2(e)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = ld4f(sc + g * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) sa[4 * g + e] = a[e];
        }
        F pfp[NS];                                               // W16: P as bf16 fragments (sa is dead after the exponentials)
        F cf[4];
        auto issue_cf = [&](int st) __attribute__((always_inline)) {
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                cf[db] = tr_frag(aC[db][0], aC[db][1], ODO + 16 * st * 128);
                cf[2 + db] = tr_frag(aC[db][0], aC[db][1], OQ + 16 * st * 128);
            }
        };
        if constexpr (W16) {
            // The sixteen-wave form has 128 registers: 64 of accumulators, 16 of V fragments, ~10 of addresses.  So: fragments of
            // the score chain two k-steps deep (16 registers, not 32); the exponentials right behind it and P packed to bf16 at once
            // (what the dV product takes anyway: 8 registers instead of 16 across the dP chain; dS = bf16(P) * dP', one more
            // rounding of P than the eight-wave form); dO fragments two deep; column fragments requested after the arithmetic.
#pragma unroll
            for (int st = 0; st < NF; ++st) vf[0][st] = *(const F*)(vrow + st * 16);
            if (!(PA_FUSED_ABL & 8)) {
                F qf[2], kf[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    qf[st] = lds_b128_imm<F>(aS[st], OQ);
                    kf[st] = lds_b128_imm<F>(aS[st] + (uint32_t)kdelta, kb * 4096);
                }
#pragma unroll
                for (int st = 0; st < NF; ++st) {
                    frag_settle(qf[st & 1], kf[st & 1], st + 1 < NF ? 2 : 0);
                    mma32<T>(sa, qf[st & 1], kf[st & 1]);
                    if (st + 2 < NF) {
                        qf[st & 1] = lds_b128_imm<F>(aS[st + 2], OQ);
                        kf[st & 1] = lds_b128_imm<F>(aS[st + 2] + (uint32_t)kdelta, kb * 4096);
                    }
                }
            }
            if (!(PA_FUSED_ABL & 2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sa[r] = __builtin_amdgcn_exp2f(sa[r]);
            }
#pragma unroll
            for (int st = 0; st < NS; ++st) pfp[st] = acc_frag<T>(sa, st);
            // pin the order (volatile asm statements keep theirs, and the fragment loads of the dP chain are such): left alone the
            // scheduler sinks the exponentials and the packing to the products, keeping 16 + 16 accumulator registers live through the
            // dP chain and spilling a dK / dV block around every product group
            asm volatile("" : "+v"(pfp[0]), "+v"(pfp[1]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 d = ld4f(sc + FK * 4 + g * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) dpa[4 * g + e] = d[e];
            }
            if (!(PA_FUSED_ABL & 8)) {
                F df[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) df[st] = lds_b128_imm<F>(aS[st], ODO);
#pragma unroll
                for (int st = 0; st < NF; ++st) {
                    frag_settle1(df[st & 1], st + 1 < NF ? 1 : 0);
                    mma32<T>(dpa, df[st & 1], vf[kb][st]);
                    if (st + 2 < NF) df[st & 1] = lds_b128_imm<F>(aS[st + 2], ODO);
                }
            }
            // the next tile's Q / dO rows are requested HERE, behind the only vector-memory loads of the block (the V fragments above:
            // the compiler's wait for them is then exact, nothing younger is in flight) and in front of half a tile of work
            if (t + 1 < nt) stage(t + 1);
            if (!(PA_FUSED_ABL & 2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dpa[r] *= (float)pfp[r >> 3][r & 7];       // dS / scale
            }
            if (PA_FUSED_ABL & 4) {
                asm volatile("" :: "v"(pfp[0]), "v"(pfp[1]), "v"(dpa));
                return;
            }
            issue_cf(0);
        } else {
        // the fragment reads are asm (issued where they are written, settled by counted waits): left to itself the compiler, at
        // the register limit, either hoists all twelve loads and spills the V fragments or serialises load -> wait -> MFMA
        if (!(PA_FUSED_ABL & 8)) {
            F qf[NF], kf[NF];
#pragma unroll
            for (int st = 0; st < NF; ++st) {
                qf[st] = lds_b128_imm<F>(aS[st], OQ);
                kf[st] = lds_b128_imm<F>(aS[st] + (uint32_t)kdelta, kb * 4096);
            }
#pragma unroll
            for (int st = 0; st < NF; ++st) {
                frag_settle(qf[st], kf[st], 2 * (NF - 1 - st));
                mma32<T>(sa, qf[st], kf[st]);
            }
        }
        // dP'[q][key] = dO V^T - delta: requested behind the score chain, whose MFMAs cover the latency
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 d = ld4f(sc + FK * 4 + g * 32);
#pragma unroll
            for (int e = 0; e < 4; ++e) dpa[4 * g + e] = d[e];
        }
        // (column fragments single-buffered: 16 registers, not 32 -- the kernel sits at the 256-register limit of two waves per
        // SIMD, and a spilled accumulator inside the tile loop drains the Q / dO prefetch in front of its reload.  The first
        // set is requested in front of the dP chain / the exponentials, which cover its latency)
        if (!(PA_FUSED_ABL & 8)) {
            F df[NF];
#pragma unroll
            for (int st = 0; st < NF; ++st) df[st] = lds_b128_imm<F>(aS[st], ODO);
            if (!(PA_FUSED_ABL & 4)) issue_cf(0);
#pragma unroll
            for (int st = 0; st < NF; ++st) {
                frag_settle1(df[st], NF - 1 - st + ((PA_FUSED_ABL & 4) ? 0 : 8));
                mma32<T>(dpa, df[st], vf[kb][st]);
            }
        } else if (!(PA_FUSED_ABL & 4)) {
            issue_cf(0);
        }
        if (!(PA_FUSED_ABL & 2)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sa[r] = __builtin_amdgcn_exp2f(sa[r]);
                dpa[r] *= sa[r];                                // dS / scale
            }
        }
        if (PA_FUSED_ABL & 4) {
            asm volatile("" :: "v"(sa), "v"(dpa));
            return;
        }
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key] ; T[key][q] = bf16(dS)
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st > 0) issue_cf(st);
            F pf;
            if constexpr (W16) pf = pfp[st];
            else pf = acc_frag<T>(sa, st);
            const F dsf = acc_frag<T>(dpa, st);
            col_settle<0>(cf[0], cf[1], cf[2], cf[3]);
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                mma32<T>(dv[kb][db], cf[db], pf);
                mma32<T>(dk[kb][db], cf[2 + db], dsf);
            }
            // queries 16 st + 4 h + {0..3} (slot h) and + 8 (slot h + 2 = slot ^ 2 after the key swizzle: byte offset ^ 16)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 w = __builtin_bit_cast(u32x4, dsf);
            const int TOFF = BUF * (2 * FT_PLANE) + st * FT_PLANE + kb * 1024;
            // lanes whose key is past N keep their hands off T (its rows behind key N - 1 stay the prologue's zeros): their score
            // is exp2(-lse * log2 e) -- the K row is zero --, which overflows for a query with lse < -88 (short sequences, strongly
            // negative logits), and inf or NaN in T times the zero K rows of phase 2 would be NaN in dQ of VALID queries.  In their
            // own dK / dV columns (never stored) anything may stand.  Nothing asm-issued is in flight here (settled above).
            uint32_t tl = (uint32_t)lane;
            asm volatile("" : "+v"(tl));                         // formed here (two VALU), not carried through the block in a register
            if (kbase0 + kb * 32 + (int)(tl & 31u) < N) {
                st64(aT0 + TOFF, w[0], w[1]);
                st64((aT0 ^ 16u) + TOFF, w[2], w[3]);
            }
        }
    };
    // phase 2 of tile t: dQ^T tile (d16, q16) = sum over keys, 4 NCH steps of 32 keys.  Straight-line: the transposed reads are
    // asm the compiler cannot see, so nothing in flight may cross a control-flow merge; the four reads of step k + 3 are issued
    // in front of the MFMA of step k and settled in issue order.  Steps past the last key multiply finite T rows
    // (zeros from the prologue, or dS columns of lanes past N) by the zero rows behind key N - 1 of the K tile.
    auto phase2 = [&](auto buf_tag, auto nch_tag, int t) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value, NCH = decltype(nch_tag)::value;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // the two per-lane base addresses are recomputed per tile (a dozen VALU) instead of living in registers across phase 1
        uint32_t p2K, p2T;
        {
            uint32_t ln = (uint32_t)lane;
            asm volatile("" : "+v"(ln));
            const int g4 = ln >> 4, p = ln & 15, r = p >> 2, cg = p & 3;
            const int kl = 4 * g4 + r;                           // key inside a 32-key step: kl (first read), 16 + kl (second)
            const int d = d16 * 16 + cg * 4;
            p2K = lds0 + (uint32_t)(swz128(kl, d >> 3) + (d & 7) * 2);
            p2T = lds0 + (uint32_t)(F_OFF_T + q16 * FT_PLANE + kl * 32 + ((cg ^ g4) << 3));
        }
        constexpr int NSTEP = 4 * NCH, AHEAD = 3;               // lgkmcnt counts to 15: three steps (12 reads) in flight
        bf16x4 ka[4][2], ta[4][2];                              // ring of four steps
        auto issue = [&](int k) {
            ka[k & 3][0] = lds_tr16_asm_imm(p2K, k * 4096);
            ka[k & 3][1] = lds_tr16_asm_imm(p2K, k * 4096 + 2048);
            ta[k & 3][0] = lds_tr16_asm_imm(p2T, BUF * (2 * FT_PLANE) + k * 1024);
            ta[k & 3][1] = lds_tr16_asm_imm(p2T, BUF * (2 * FT_PLANE) + k * 1024 + 512);
        };
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) issue(k);
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            if (k + AHEAD < NSTEP) issue(k + AHEAD);
            bf16x4 &k0 = ka[k & 3][0], &k1 = ka[k & 3][1], &t0 = ta[k & 3][0], &t1 = ta[k & 3][1];
            const int younger = (NSTEP - 1 - k) < AHEAD ? (NSTEP - 1 - k) : AHEAD;      // steps issued after this one
            if (younger == 3) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(k0), "+v"(k1), "+v"(t0), "+v"(t1) : "n"(PA_ATTN_DEBUG_WAIT ? 0 : 12));
            else if (younger == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(k0), "+v"(k1), "+v"(t0), "+v"(t1) : "n"(PA_ATTN_DEBUG_WAIT ? 0 : 8));
            else if (younger == 1) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(k0), "+v"(k1), "+v"(t0), "+v"(t1) : "n"(PA_ATTN_DEBUG_WAIT ? 0 : 4));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k0), "+v"(k1), "+v"(t0), "+v"(t1));
            const F a = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
            const F bq = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, acc, 0, 0, 0);
        }
        // lane holds dQ^T[d = 16 d16 + 4 g4 + {0..3}][q = 16 q16 + p]: 8 bytes of row q; gradient w.r.t. the TRUE q
        const int q = t * 32 + q16 * 16 + (lane & 15);
        if (q < N) {
            const bf16x2 lo2 = {(bf16)(acc[0] * scale), (bf16)(acc[1] * scale)}, hi2 = {(bf16)(acc[2] * scale), (bf16)(acc[3] * scale)};
            attn_store((u32x2_t*)(dqkv + ((int64_t)b * N + q) * lddqkv + h * HD + d16 * 16 + 4 * (lane >> 4)),
                       u32x2_t{__builtin_bit_cast(uint32_t, lo2), __builtin_bit_cast(uint32_t, hi2)});
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // K rows behind the last key := 0 (so that the dS columns of lanes past N, and the T rows nobody writes, contribute nothing
    // to dQ); the scores of those lanes become exp2(-lse): finite
    for (int i = tid; i < (nk4 * 32 - N) * 8; i += NT) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        *(u32x4*)(sK + (N + (i >> 3)) * 128 + (i & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using NCH = std::integral_constant<int, 4>;      // phase 2 walks the whole K tile (16 steps of 32 keys) whatever N is
    auto phase1 = [&](auto buf_tag, int t) __attribute__((always_inline)) {
        if (kbase0 < N) block(std::integral_constant<int, 0>{}, buf_tag, t);
        else if (W16 && t + 1 < nt) stage(t + 1);                // (W16 requests the next tile from inside the block)
        if constexpr (NKB == 2) {
            if (kbase0 + 32 < N) block(std::integral_constant<int, 1>{}, buf_tag, t);
        }
    };
    auto sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(PA_FUSED_ABL & 16)) __syncthreads();
    };
    // (Measured and dropped, profiles/r05_attention_single_pass.txt: the two waves of a SIMD running phase 2 / phase 1 of an
    // inter-barrier interval in OPPOSITE orders -- two loop nests chosen once per wave.  The second nest costs the allocator four
    // spilled registers per tile, and a reload inside the tile loop waits for the Q / dO prefetch in flight: 270-277 us against 183.)
    auto tile = [&](auto buf_tag, int t) __attribute__((always_inline)) {
        if (!W16 && t + 1 < nt) stage(t + 1);
        phase1(buf_tag, t);
        sync();
        if (!(PA_FUSED_ABL & 1) && (!W16 || wave < 8)) phase2(buf_tag, NCH{}, t);
    };
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        tile(B0{}, t);
        tile(B1{}, t + 1);
    }
    if (t < nt) tile(B0{}, t);
    T* out = dqkv + (int64_t)b * N * lddqkv + h * HD;
    int ln = lane;
    asm volatile("" : "+v"(ln));                                 // the row pointers are formed here, not kept in registers across the tile loop
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int k0 = kbase0 + kb * 32;
        if (k0 < N) {
            // the Q rows in memory are Q * scale * log2(e): dK = dS^T Q * scale = acc * ln 2
            store_rows_direct<T>(dk[kb], LN2, out + D, lddqkv, k0, min(32, N - k0), ln);
            store_rows_direct<T>(dv[kb], 1.0f, out + 2 * D, lddqkv, k0, min(32, N - k0), ln);
        }
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

// 1 = attribute set on this device, -1 = refused (per device, decided on the device's first call; a benign race between two
// host threads setting the same attribute twice)
static bool fused_lds_ok() {
    static signed char state[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[dev] == 0) {
        const hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        const hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS);
        if (e != hipSuccess || e2 != hipSuccess) (void)hipGetLastError();
        state[dev] = (e == hipSuccess && e2 == hipSuccess) ? 1 : -1;
    }
    return state[dev] > 0;
}

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
    if (!qkv || !o || !d_o || !lse || !delta || !dqkv || B <= 0 || H <= 0 || N <= 0 || nq <= 0 || nq > N ||
        (flags & ~(PA_ATTN_Q_PRESCALED | PA_ATTN_BWD_TWO_PASS | PA_ATTN_BWD_SINGLE_PASS | PA_ATTN_BWD_SINGLE_PASS_W16)))
        return PA_EINVAL;
    if (!attn_args_ok(ldqkv, dtype) || !attn_args_ok(ldo, dtype) || !attn_args_ok(lddqkv, dtype)) return PA_EUNSUPPORTED;
    const bool pre = flags & PA_ATTN_Q_PRESCALED;
    hipStream_t st = (hipStream_t)stream;
    // single pass where it applies and fills the chip: one workgroup per (sequence, head) takes a whole CU (148 KiB of LDS), so B * H
    // below two rounds of 256 CUs leaves the two-kernel form (3-4 x as many, smaller workgroups) ahead -- ESC-50 at batch 12: 144
    const bool can_fuse = dtype == PA_BF16 && pre && nq == N && N <= FK;
    if (can_fuse && !(flags & PA_ATTN_BWD_TWO_PASS) && ((flags & (PA_ATTN_BWD_SINGLE_PASS | PA_ATTN_BWD_SINGLE_PASS_W16)) || (int64_t)B * H >= 512)) {
        // the 148 KiB of dynamic LDS need the attribute on EVERY device this process drives (it is per device: keyed on
        // hipGetDevice(), not set once for whichever device made the first call); a device that refuses it runs the two-kernel form
        if (fused_lds_ok()) {
            if (flags & PA_ATTN_BWD_SINGLE_PASS_W16)
                hipLaunchKernelGGL(attn_bwd_fused_kernel<true>, dim3((unsigned)(B * H)), dim3(1024), F_LDS, st, (const bf16*)qkv, ldqkv, (const bf16*)o,
                                   (const bf16*)d_o, ldo, lse, (bf16*)dqkv, lddqkv, H, N, scale);
            else
                hipLaunchKernelGGL(attn_bwd_fused_kernel<false>, dim3((unsigned)(B * H)), dim3(512), F_LDS, st, (const bf16*)qkv, ldqkv, (const bf16*)o,
                                   (const bf16*)d_o, ldo, lse, (bf16*)dqkv, lddqkv, H, N, scale);
            return check_launch();
        }
    }
    if (dtype == PA_BF16)
        return pre ? attention_bwd_t<bf16, true>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st)
                   : attention_bwd_t<bf16, false>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st);
    if (dtype == PA_F32)
        return pre ? attention_bwd_t<float, true>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st)
                   : attention_bwd_t<float, false>(qkv, ldqkv, o, d_o, ldo, lse, delta, dqkv, lddqkv, B, H, N, nq, scale, st);
    return PA_EINVAL;
}
