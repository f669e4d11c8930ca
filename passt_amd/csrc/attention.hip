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
// 1 = packed f32 math (v_pk_fma_f32 ...) around the exponentials.  Measured slower than scalar (bwd 239 vs 234 us, run
// 60): the register-pair constraint costs ~30 v_mov per tile and packed VALU issues badly next to MFMAs.
#ifndef PA_ATTN_PK
#define PA_ATTN_PK 0
#endif
#ifndef PA_ATTN_FWD_WAVES
#define PA_ATTN_FWD_WAVES 4
#endif
#ifndef PA_ATTN_DQ_WAVES
#define PA_ATTN_DQ_WAVES 3
#endif
#ifndef PA_ATTN_DKDV_WAVES
#define PA_ATTN_DKDV_WAVES 2
#endif

static constexpr int HD = 64;       // head dim (all PaSST archs: 768/12, 1024/16, 384/6, 128/2)
static constexpr int TROWS = 64;    // streamed rows per LDS tile
static constexpr float LOG2E = 1.4426950408889634f;
static constexpr float LN2 = 0.6931471805599453f;

// The softmax / dS element loops are VALU bound (the exponential is a quarter-rate instruction and there is one
// per score).  Two-scores-per-instruction forms (PA_ATTN_PK) are kept for reference only.
__device__ __forceinline__ f32x2 exp2_2(f32x2 a) { return f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])}; }

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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;      // q of token 0 of this (b,h)
    const int q0 = blk * 128 + wave * 32;
    const int qrow = min(q0 + (lane & 31), N - 1);

    typename Frag<T>::type qf[Tile<T>::NFRAG];
#pragma unroll
    for (int s = 0; s < Tile<T>::NFRAG; ++s)
        qf[s] = *(const typename Frag<T>::type*)(base + (int64_t)qrow * ldqkv + (s * 2 + (lane >> 5)) * Tile<T>::EPC);

    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = scale * LOG2E;

    const int ntiles = (N + TROWS - 1) / TROWS;
    // double-buffered K/V tiles: stage kt+1 by LDS-DMA while computing kt; one barrier per tile
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * (2 * Tile<T>::BYTES);
        stage_tile<T>(sb, base + D, ldqkv, kt * TROWS, N, wave, lane);
        stage_tile<T>(sb + Tile<T>::BYTES, base + 2 * D, ldqkv, kt * TROWS, N, wave, lane);
    };
    stage(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < ntiles) stage((kt + 1) & 1, kt + 1);
        const char* sK = smem + (kt & 1) * (2 * Tile<T>::BYTES);
        const char* sV = sK + Tile<T>::BYTES;
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
            for (int st = 0; st < Tile<T>::NFRAG; ++st)
                mma32<T>(s[kb], row_frag<T>(sK, kb * 32 + (lane & 31), st, lane), qf[st]);
        }
        // online softmax; this lane owns query (lane&31) and 16 of the 32 keys of each key block.
        // Keys beyond N exist only in the last tile: the mask is a wave-uniform branch.
        if (kt == ntiles - 1 && (N & (TROWS - 1))) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * TROWS + kb * 32 + acc_row(r, lane) >= N) s[kb][r] = -INFINITY;
        }
        float mloc = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[kb][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc * sl2);         // running max in log2 units (sl2 > 0)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#if !PA_ATTN_PK
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sl2, -m_new));
                s[kb][r] = p;
                psum += p;
            }
#else
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 p = exp2_2(pk_fma(f32x2{s[kb][r], s[kb][r + 1]}, pk_splat(sl2), pk_splat(-m_new)));
                s[kb][r] = p[0];
                s[kb][r + 1] = p[1];
                ps2 += p;
            }
        float psum = ps2[0] + ps2[1];
#endif
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {                          // no row max moved: skip the O rescale
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        }
        // O^T[d][q] += V^T[d][key] P^T[key][q]; the V column fragments of step i+1 are in flight under the
        // MFMAs of step i
        {
            constexpr int NS = 2 * AccSteps<T>::N;
            typename Frag<T>::type vfr[2][2];
            auto issue = [&](int slot, int step) {
                const int kb = step / AccSteps<T>::N, st = step % AccSteps<T>::N;
#pragma unroll
                for (int db = 0; db < 2; ++db) vfr[slot][db] = col_frag<T>(sV, kb * 32, st, db * 32, lane);
            };
            issue(0, 0);
#pragma unroll
            for (int step = 0; step < NS; ++step) {
                const int kb = step / AccSteps<T>::N, st = step % AccSteps<T>::N;
                if (step + 1 < NS) {
                    issue((step + 1) & 1, step + 1);
                    col_settle<4>(vfr[step & 1][0], vfr[step & 1][1]);
                } else {
                    col_settle<0>(vfr[step & 1][0], vfr[step & 1][1]);
                }
                const typename Frag<T>::type pf = acc_frag<T>(s[kb], st);
#pragma unroll
                for (int db = 0; db < 2; ++db) mma32<T>(oacc[db], vfr[step & 1][db], pf);
            }
        }
    }
    __syncthreads();   // tiles are dead; reuse LDS for the transposed store
    // only the first nq queries of every sequence are produced; o / lse are compact (nq rows per sequence)
    if (q0 < nq) {
        if (lane < 32 && q0 + lane < nq) lse[(int64_t)bh * nq + q0 + lane] = m_run * LN2 + __logf(l_run);   // natural log units
        store_rows_T<T>((float*)smem + wave * (32 * 65), oacc, 1.0f / l_run, o + (int64_t)b * nq * ldo + h * HD,
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;
    const T* dobase = d_o + (int64_t)b * nq * ldo + h * HD;     // d_o / lse / delta: nq rows per sequence
    const int k0 = blk * 128 + wave * 32;
    const int key = k0 + (lane & 31);
    const int krow = min(key, N - 1);
    const bool active = k0 < N;                                 // wave-uniform

    typename Frag<T>::type kf[Tile<T>::NFRAG], vf[Tile<T>::NFRAG];
#pragma unroll
    for (int s = 0; s < Tile<T>::NFRAG; ++s) {
        const int off = (s * 2 + (lane >> 5)) * Tile<T>::EPC;
        kf[s] = *(const typename Frag<T>::type*)(base + D + (int64_t)krow * ldqkv + off);
        vf[s] = *(const typename Frag<T>::type*)(base + 2 * D + (int64_t)krow * ldqkv + off);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }
    const float sl2 = scale * LOG2E;

    const int ntiles = (nq + TROWS - 1) / TROWS;               // only queries < nq carry a gradient
    constexpr int STAGE = 2 * Tile<T>::BYTES + 2 * TROWS * 4;   // Q tile, dO tile, lse[64], delta[64]
    auto stage = [&](int buf, int qt) {
        char* sb = smem + buf * STAGE;
        stage_tile<T>(sb, base, ldqkv, qt * TROWS, N, wave, lane);
        stage_tile<T>(sb + Tile<T>::BYTES, dobase, ldo, qt * TROWS, nq, wave, lane);
        // both per-query scalars come pre-scaled from the dQ kernel's workspace: -lse*log2(e) and delta*scale
        if (wave == 0) stage_f32x64((float*)(sb + 2 * Tile<T>::BYTES), ws + plane + (int64_t)bh * nq, qt * TROWS, nq, lane);
        if (wave == 1) stage_f32x64((float*)(sb + 2 * Tile<T>::BYTES) + TROWS, ws + (int64_t)bh * nq, qt * TROWS, nq, lane);
    };
    stage(0, 0);
    for (int qt = 0; qt < ntiles; ++qt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (qt + 1 < ntiles) stage((qt + 1) & 1, qt + 1);
        const char* sQ = smem + (qt & 1) * STAGE;
        const char* sDO = sQ + Tile<T>::BYTES;
        const float* sLse = (const float*)(sQ + 2 * Tile<T>::BYTES);
        const float* sDelta = sLse + TROWS;
        if (!active) continue;          // all 32 keys of this wave are past N: stage and meet barriers only
        // (r02 experiment: software-pipelining the two 32-query blocks of a tile inside the wave -- products(0) | products(1) +
        // softmax(0) | dV/dK(0) + softmax(1) | dV/dK(1), regions fenced with sched_barrier -- needs sa/dpa of both blocks live:
        // 256 VGPRs + 64-80 B of scratch at 2 waves per SIMD, and measured 8 % SLOWER for the whole backward, 246 vs 227 us.)
        const bool half_tile = qt == ntiles - 1 && qt * TROWS + 32 >= nq;     // second 32 queries of the tile do not exist
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            if (qb == 1 && half_tile) continue;
            f32x16 sa, dpa;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dpa[r] = 0.f; }
            // S[q][key] = Q K^T ; dP[q][key] = dO V^T   (A rows = q, B cols = key = lane)
#pragma unroll
            for (int st = 0; st < Tile<T>::NFRAG; ++st) {
                mma32<T>(sa, row_frag<T>(sQ, qb * 32 + (lane & 31), st, lane), kf[st]);
                mma32<T>(dpa, row_frag<T>(sDO, qb * 32 + (lane & 31), st, lane), vf[st]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {                       // accumulator rows 4g..4g+3 are 4 consecutive queries
                const int ql = qb * 32 + 8 * g + 4 * (lane >> 5);
                const f32x4 nl = *(const f32x4*)(sLse + ql), dl = *(const f32x4*)(sDelta + ql);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, nl[e]));
                    sa[r] = p;
                    dpa[r] = p * fmaf(dpa[r], scale, -dl[e]);
                }
            }
            // queries beyond nq exist only in the last tile (uniform branch); lanes whose own key is beyond
            // N only produce their own, never stored, outputs and need no mask
            if (qt == ntiles - 1 && (nq & (TROWS - 1))) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (qt * TROWS + qb * 32 + acc_row(r, lane) >= nq) { sa[r] = 0.f; dpa[r] = 0.f; }
            }
            // dV^T[d][key] += dO^T[d][q] P[q][key] ; dK^T[d][key] += Q^T[d][q] dS[q][key]
            {
                constexpr int NS = AccSteps<T>::N;
                typename Frag<T>::type cf[2][4];
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
                    const typename Frag<T>::type pf = acc_frag<T>(sa, st), dsf = acc_frag<T>(dpa, st);
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        mma32<T>(dv[db], cf[st & 1][db], pf);
                        mma32<T>(dk[db], cf[st & 1][2 + db], dsf);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (k0 < N) {
        float* slab = (float*)smem + wave * (32 * 65);
        T* out = dqkv + (int64_t)b * N * lddqkv + h * HD;
        store_rows_T<T>(slab, dk, 1.0f, out + D, lddqkv, k0, min(32, N - k0), lane);
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk, bh;
    if (!attn_block(nblk, BH, blk, bh)) return;
    const int b = bh / H, h = bh % H;
    const int D = H * HD;
    const T* base = qkv + (int64_t)b * N * ldqkv + h * HD;
    const T* dobase = d_o + (int64_t)b * nq * ldo + h * HD;     // d_o / lse / delta: nq rows per sequence
    const int q0 = blk * 128 + wave * 32;
    const int q = q0 + (lane & 31);
    const int qrow = min(q, nq - 1);
    const bool active = q0 < nq;                                // wave-uniform

    typename Frag<T>::type qf[Tile<T>::NFRAG], dof[Tile<T>::NFRAG];
#pragma unroll
    for (int s = 0; s < Tile<T>::NFRAG; ++s) {
        const int off = (s * 2 + (lane >> 5)) * Tile<T>::EPC;
        qf[s] = *(const typename Frag<T>::type*)(base + (int64_t)qrow * ldqkv + off);
        dof[s] = *(const typename Frag<T>::type*)(dobase + (int64_t)qrow * ldo + off);
    }
    const float lse2 = lse[(int64_t)bh * nq + qrow] * LOG2E;
    // delta[q] = sum_d dO[q][d] O[q][d]: this lane holds half of row q of dO as fragments already; the same
    // chunks of O are read once here, and the row sum is published for the dK/dV kernel (launched after)
    float dlt = 0.f;
    {
        const T* obase = o + (int64_t)b * nq * ldo + h * HD;
#pragma unroll
        for (int s = 0; s < Tile<T>::NFRAG; ++s) {
            const typename Frag<T>::type of =
                *(const typename Frag<T>::type*)(obase + (int64_t)qrow * ldo + (s * 2 + (lane >> 5)) * Tile<T>::EPC);
#pragma unroll
            for (int e = 0; e < Tile<T>::EPC; ++e) dlt = fmaf((float)dof[s][e], (float)of[e], dlt);
        }
        dlt += __shfl_xor(dlt, 32, 64);
        // workspace for the dK/dV kernel, already in the form its inner loop consumes: delta*scale, -lse*log2(e)
        if (lane < 32 && q < nq) {
            delta[(int64_t)bh * nq + q] = dlt * scale;
            delta[plane + (int64_t)bh * nq + q] = -lse2;
        }
    }
    f32x16 dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
    const float sl2 = scale * LOG2E;

    const int ntiles = (N + TROWS - 1) / TROWS;
    auto stage = [&](int buf, int kt) {
        char* sb = smem + buf * (2 * Tile<T>::BYTES);
        stage_tile<T>(sb, base + D, ldqkv, kt * TROWS, N, wave, lane);
        stage_tile<T>(sb + Tile<T>::BYTES, base + 2 * D, ldqkv, kt * TROWS, N, wave, lane);
    };
    stage(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < ntiles) stage((kt + 1) & 1, kt + 1);
        const char* sK = smem + (kt & 1) * (2 * Tile<T>::BYTES);
        const char* sV = sK + Tile<T>::BYTES;
        if (!active) continue;          // all 32 queries of this wave are past nq
        const bool half_tile = kt == ntiles - 1 && kt * TROWS + 32 >= N;      // second 32 keys of the tile do not exist
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (kb == 1 && half_tile) continue;
            f32x16 sa, dpa;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dpa[r] = 0.f; }
            // S^T[key][q] = K Q^T ; dP^T[key][q] = V dO^T
#pragma unroll
            for (int st = 0; st < Tile<T>::NFRAG; ++st) {
                mma32<T>(sa, row_frag<T>(sK, kb * 32 + (lane & 31), st, lane), qf[st]);
                mma32<T>(dpa, row_frag<T>(sV, kb * 32 + (lane & 31), st, lane), dof[st]);
            }
#pragma unroll
#if !PA_ATTN_PK
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sa[r], sl2, -lse2));
                dpa[r] = p * fmaf(dpa[r], scale, -dlt * scale);
            }
#else
            for (int r = 0; r < 16; r += 2) {
                const f32x2 p = exp2_2(pk_fma(f32x2{sa[r], sa[r + 1]}, pk_splat(sl2), pk_splat(-lse2)));
                const f32x2 ds = p * pk_fma(f32x2{dpa[r], dpa[r + 1]}, pk_splat(scale), pk_splat(-dlt * scale));   // dS^T
                dpa[r] = ds[0]; dpa[r + 1] = ds[1];
            }
#endif
            if (kt == ntiles - 1 && (N & (TROWS - 1))) {       // keys beyond N: last tile only
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * TROWS + kb * 32 + acc_row(r, lane) >= N) dpa[r] = 0.f;
            }
            // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
            {
                constexpr int NS = AccSteps<T>::N;
                typename Frag<T>::type cf[2][2];
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
                    const typename Frag<T>::type dsf = acc_frag<T>(dpa, st);
#pragma unroll
                    for (int db = 0; db < 2; ++db) mma32<T>(dq[db], cf[st & 1][db], dsf);
                }
            }
        }
    }
    __syncthreads();
    if (q0 < nq)
        store_rows_T<T>((float*)smem + wave * (32 * 65), dq, 1.0f, dqkv + (int64_t)b * N * lddqkv + h * HD,
                        lddqkv, q0, min(32, nq - q0), lane);
}

template <typename T> static size_t fwd_lds() { return std::max<size_t>(4 * Tile<T>::BYTES, SLAB_BYTES); }
template <typename T> static size_t dkdv_lds() { return std::max<size_t>(2 * (2 * Tile<T>::BYTES + 2 * TROWS * 4), SLAB_BYTES); }

template <typename T>
static int attention_fwd_t(const void* qkv, int ldqkv, void* o, int ldo, float* lse, int B, int H, int N, int nq,
                           float scale, hipStream_t st) {
    const int nblk = (int)cdiv(nq, 128);
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
