// MFMA GEMM  C[M][N] = A[M][K] * B[N][K]^T  with fused epilogues, plus the small data-movement
// kernels around it (convert, transpose, split-K reduce, row/column sums).
//
// Replaces the reference's nn.Linear calls (models/passt.py:285,288,345,359), the patch-embed
// conv as an im2col GEMM (:323) and their autograd gradients.  See include/passt_amd.h.
//
// Tiling (gfx950): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave =
// 2x2 MFMA 32x32 tiles, 64 f32 accumulators per lane); K consumed 128 BYTES per step (64 bf16 /
// 32 f32) so the same LDS image and read pattern serve both dtypes.  Operand tiles are staged
// HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered (2 x 32 KiB); the
// LDS image is lane-linear, so the bank-conflict XOR swizzle is applied on the per-lane SOURCE
// address and again on the ds_read_b128 address (guide rule 21).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "pa_mma.h"

namespace pa {

#ifndef PA_NT_GROUP_M
#define PA_NT_GROUP_M 4      // row-tiles walked per column-tile by consecutive work items (L2 patch shape)
#endif
#ifndef PA_NT_SUBSTEPS
#define PA_NT_SUBSTEPS 1
#endif
static constexpr int BM = 128, BN = 128, KB = 128;  // KB: K bytes per step
static constexpr int TILE_BYTES = BM * KB;            // 16 KiB per operand tile
static constexpr int GEMM_LDS = 2 * 2 * TILE_BYTES;   // 64 KiB (TN kernel)

// ---- probe build only (-DPA_PROBE -> libpasst_amd_probe.so, tools/probe_epilogue.py): s_memtime stamps of the
// role-split kernel's item timeline and switches that remove parts of the epilogue (pa_gemm_args.reserved:
// bit 0 = no global stores, bit 1 = no epilogue at all).  The product library is built without it.
#ifdef PA_PROBE
__device__ unsigned long long* g_probe_buf = nullptr;
static constexpr int PROBE_SLOTS = 512;
// stamps go to LDS (past G::LDS) so that they add no global store to the vmcnt queue being measured; the kernel copies
// them out once at its end
#define PA_PROBE_STAMP(cond, idx)                                                                             \
    do {                                                                                                      \
        if ((cond) && (wave & 3) == 0 && lane == 0 && (idx) < PROBE_SLOTS)                                    \
            ((unsigned long long*)(smem + G::LDS))[(wave >> 2) * PROBE_SLOTS + (idx)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define PA_PROBE_FLAG(a, bit) (((a).reserved >> (bit)) & 1)
// the same for the weight-gradient kernel (stamps behind its 144 KiB of stages)
#define PA_PROBE_STAMP_TN(cond, idx)                                                                          \
    do {                                                                                                      \
        if ((cond) && (wave & 3) == 0 && lane == 0 && (idx) < PROBE_SLOTS)                                    \
            ((unsigned long long*)(smem + TN_LDS))[(wave >> 2) * PROBE_SLOTS + (idx)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PA_PROBE_STAMP(cond, idx) do {} while (0)
#define PA_PROBE_STAMP_TN(cond, idx) do {} while (0)
#define PA_PROBE_FLAG(a, bit) 0
#endif

// ---- Cache policy of the step's GEMM memory instructions, per role (round 6, profiles/r06_cache_policy.txt) ----------------
// The `aux` immediate of the buffer / LDS-DMA instructions: 1 = sc0, 2 = nt (non-temporal), 16 = sc1.  Every output line and
// every epilogue operand row of a GEMM is touched exactly once by that GEMM; weights are re-read by every row band, activation
// tiles by N / 256 column tiles, and the NEXT kernel wants this kernel's outputs.  With the default policy the single-use
// lines compete for the caches with the re-used ones; marked non-temporal they do not: 1.5 - 2.4 % of the whole training step
// (config #2, same box, ABBA: 22.39 -> 21.98, 21.78 -> 21.46, 21.98 -> 21.48 ms; ESC-50 - 2.9 %, config #4 - 1.6 %), e.g.
// fc2 / proj + residual 104.5 -> 95.0 us, fc1 + GELU 170 -> 157 us.  The bytes crossing the L2's fabric side do NOT change
// (FETCH_SIZE / WRITE_SIZE: 439 against 440 MB per launch): the effect is behind them, in the 256 MiB Infinity Cache / HBM.
// What the sweep settled, role by role:
//   nt  fc1's blocked pre-activation (written in the forward, next read in the backward)                      PA_AUX_ST_PRE
//   nt  the bf16 outputs of the STORE / GELU / GELU' epilogues (full 128-byte lines per 8 lanes)               PA_AUX_ST_OUT
//   nt  the epilogue operand rows read once (residual rows, pre-activation blocks)                            PA_AUX_LD_AUX
//   nt  the A operand of the residual GEMMs (N = D: a tile is read by D / 256 column tiles only)              PA_AUX_DMA_A_RESID
//   nt  the split-K slabs when the finishing reduction reads them (each once)                                 PA_NT_LD_SLAB
//   default: the f32 residual-stream output (nt: the LayerNorm behind it slows down, +0.5 %), the A operand of every other GEMM
//       (nt: fc1 + GELU 170 -> 188 us), the weights (nt: +4 % on the step), the weight-gradient operands and slab stores
//       (+- 0), sc1 instead of nt on the outputs (half the gain), nt on the A operand of plain-store GEMMs with N = D (+0.5 %).
// -DPA_NO_CACHE_POLICY builds the library with the default policy everywhere (A/B: tools/build_variant.sh).
#ifdef PA_NO_CACHE_POLICY
#define PA_CP(x) 0
#else
#define PA_CP(x) x
#endif
#ifndef PA_AUX_ST_PRE
#define PA_AUX_ST_PRE PA_CP(2)
#endif
#ifndef PA_AUX_ST_OUT
#define PA_AUX_ST_OUT PA_CP(2)
#endif
#ifndef PA_AUX_ST_ACT
#define PA_AUX_ST_ACT PA_AUX_ST_OUT      // ... fc1's activation only (the fc2 GEMM behind it reads it non-temporally)
#endif
#ifndef PA_AUX_ST_DPRE
#define PA_AUX_ST_DPRE PA_AUX_ST_OUT     // ... the GELU' epilogue's output only
#endif
#ifndef PA_AUX_ST_STORE
#define PA_AUX_ST_STORE PA_AUX_ST_OUT    // ... the plain-store epilogue only
#endif
#ifndef PA_AUX_ST_RES
#define PA_AUX_ST_RES 0
#endif
#ifndef PA_AUX_LD_AUX
#define PA_AUX_LD_AUX PA_CP(2)
#endif
#ifndef PA_AUX_LD_PRE
#define PA_AUX_LD_PRE PA_AUX_LD_AUX      // ... the GELU' epilogue's pre-activation only
#endif
#ifndef PA_AUX_LD_RES
#define PA_AUX_LD_RES PA_AUX_LD_AUX      // ... the residual rows only
#endif
#ifndef PA_AUX_DMA_A
#define PA_AUX_DMA_A 0         // LDS-DMA of the A operand (activations) of the role-split NT kernel
#endif
#ifndef PA_AUX_DMA_A_RESID
#define PA_AUX_DMA_A_RESID PA_CP(2)      // ... in the residual GEMMs
#endif
#ifndef PA_NT_A_STORE_MAXN
#define PA_NT_A_STORE_MAXN 0   // plain-store GEMMs with N <= this read their A operand non-temporally too (input gradients: N = D)
#endif
#ifndef PA_NT_LD_SLAB
#define PA_NT_LD_SLAB PA_CP(1) // the finishing reduction reads the slabs (each once) non-temporally
#endif
#ifndef PA_NT_STORE_MAXN
#define PA_NT_STORE_MAXN 0     // > 0: the plain-store epilogue uses PA_AUX_ST_OUT only when N <= this, the default policy above
#endif
#ifndef PA_NT_ST_SLAB
#define PA_NT_ST_SLAB 0        // 1: split-K partial slabs (weight gradients, small-M NT GEMMs) leave as non-temporal stores
#endif
#ifndef PA_AUX_DMA_B
#define PA_AUX_DMA_B 0         // LDS-DMA of the B operand (weights)
#endif
#ifndef PA_AUX_DMA_TN
#define PA_AUX_DMA_TN 0        // LDS-DMA of both operands of the weight-gradient kernel
#endif
#ifndef PA_AUX_DMA_TN_A
#define PA_AUX_DMA_TN_A PA_AUX_DMA_TN      // ... of dY only
#endif
#ifndef PA_AUX_DMA_TN_B
#define PA_AUX_DMA_TN_B PA_AUX_DMA_TN      // ... of X only
#endif

// Shared epilogue: the TM x 2 MFMA accumulators of this wave (TM*32 x 64 outputs at rows m0 + wr*TM*32,
// columns n0 + wc*64) -> per-wave LDS slab [32][68] -> 8-wide row vectors with the fused epilogue math.
//
// The epilogue of a tile is latency bound, not bandwidth bound (profiles/r01_epilogue_experiment.json: it
// costs the same with 60 resident workgroups as with 252), and vmcnt counts stores as well as loads and
// retires in order, so a load issued after a store waits for that store's write acknowledge.  Hence:
//   * the bias row is requested by the caller BEFORE the K loop (load_bias8) -- its latency is never exposed;
//   * auxiliary rows (residual / pre-activation) are requested AUX_DEPTH passes ahead as raw 16-byte
//     vectors (decoded only when used): with AUX_DEPTH == TM every load of the tile is in flight before the
//     first store is issued;
//   * each pass issues its stores back to back with no wait in between.
static constexpr int SLAB_BYTES = 32 * 68 * 4;     // 8704 per wave

// PA_EPI_DGELU with colsum_out: the epilogue left one row of column sums per 32*TM-row wave tile in colsum_ws
// ([rows][N]); this adds them up (defined next to colsum_f32_kernel)
static int finish_gemm_colsum(const pa_gemm_args& a, int rows, hipStream_t st);

template <int EPI, int TM> __host__ __device__ constexpr int aux_depth() {
    return EPI == PA_EPI_DGELU ? TM : (EPI == PA_EPI_RESID ? (TM <= 3 ? TM : 2) : 0);
}

template <int EPI>
__device__ __forceinline__ void load_bias8(const pa_gemm_args& a, int n0, int wc, int lane, float (&bias8)[8]) {
    const int ncol = n0 + wc * 64 + (lane & 7) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if constexpr (EPI != PA_EPI_PARTIAL && EPI != PA_EPI_DGELU) {
        if (a.bias && ncol < a.N) {
            const f32x4 lo = *(const f32x4*)(a.bias + ncol), hi = *(const f32x4*)(a.bias + ncol + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bias8[e] = lo[e]; bias8[4 + e] = hi[e]; }
        }
    }
}

template <typename T, int EPI, int TM = 2>
__device__ __forceinline__ void gemm_epilogue(const pa_gemm_args& a, f32x16 (&acc)[TM][2], float* slab, int m0,
                                              int n0, int split, int wr, int wc, int lane, const float (&bias8)[8],
                                              int colsum_row = 0) {
    const int erow = lane >> 3, ecol = (lane & 7) * 8;
    const int ncol = n0 + wc * 64 + ecol;
    const bool colok = ncol < a.N;               // N % 8 == 0: the lane's 8-vector is all in or all out
    // PA_EPI_STORE: (acc + bias) * colscale for the columns below colscale_n (a multiple of 64)
    float cs = 1.f, bias8s[8];
    if constexpr (EPI == PA_EPI_STORE) {
        if (ncol < a.colscale_n) cs = a.colscale;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8s[e] = bias8[e] * cs;
    constexpr int P = aux_depth<EPI, TM>();
    constexpr int NV = (EPI == PA_EPI_RESID || sizeof(T) == 4) ? 2 : 1;     // 16-byte vectors per 8 aux elements
    f32x4 xr[P > 0 ? P : 1][4][NV];
    auto pass_row = [&](int i, int it) {
        const int m = m0 + wr * (TM * 32) + i * 32 + it * 8 + erow;
        return (m < a.M && colok && !PA_PROBE_FLAG(a, 0)) ? m : -1;
    };
    auto load_aux = [&](int slot, int i) {
        if constexpr (P > 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = pass_row(i, it);
#pragma unroll
                for (int k = 0; k < NV; ++k) xr[slot][it][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m >= 0) {
                    const char* src;
                    if constexpr (EPI == PA_EPI_RESID) {
                        const int64_t rrow = a.row_mod > 0 ? m % a.row_mod : m;
                        src = (const char*)(a.resid + rrow * a.ldr + ncol);
                    } else {
                        src = (const char*)((const T*)a.aux + (int64_t)m * a.ldaux + ncol);
                    }
#pragma unroll
                    for (int k = 0; k < NV; ++k) xr[slot][it][k] = *(const f32x4*)(src + 16 * k);
                }
            }
        }
    };
    auto aux_val = [&](int slot, int it, float (&x)[8]) {
        if constexpr (NV == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = xr[slot][it][0][e]; x[4 + e] = xr[slot][it][1][e]; }
        } else {
            const bf16x8 b = __builtin_bit_cast(bf16x8, xr[slot][it][0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)b[e];
        }
    };
#pragma unroll
    for (int i = 0; i < P && i < TM; ++i) load_aux(i, i);
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // PA_EPI_DGELU: column sums of the outputs
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[acc_row(r, lane) * 68 + j * 32 + (lane & 31)] = acc[i][j][r];
        // same-wave LDS RAW: the LDS queue is in order per wave; no barrier needed
        float v[4][8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + erow;
            const f32x4 lo = *(const f32x4*)(slab + row * 68 + ecol);
            const f32x4 hi = *(const f32x4*)(slab + row * 68 + ecol + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[it][e] = fmaf(lo[e], cs, bias8s[e]); v[it][4 + e] = fmaf(hi[e], cs, bias8s[4 + e]); }
        }
        if constexpr (P > 0) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float x[8];
                aux_val(i % P, it, x);
                if constexpr (EPI == PA_EPI_RESID) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] += x[e];
                } else {
                    mul_gelu_grad8<T>(v[it], x);
                    if (pass_row(i, it) >= 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) csum[e] += v[it][e];
                    }
                }
            }
            if (i + P < TM) load_aux(i % P, i + P);    // the slot was consumed just above
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = pass_row(i, it);
            if (m < 0) continue;
            if constexpr (EPI == PA_EPI_STORE || EPI == PA_EPI_DGELU) {
                store8<T>((T*)a.out_lp + (int64_t)m * a.ldolp + ncol, v[it]);
            } else if constexpr (EPI == PA_EPI_GELU) {
                float g[8];
                gelu8<T>(v[it], g);                  // of the f32 value (the reference applies GELU before rounding too)
                store8<T>((T*)a.out_lp + (int64_t)m * a.ldolp + ncol, v[it]);
                store8<T>((T*)a.out_lp2 + (int64_t)m * a.ldolp2 + ncol, g);
            } else if constexpr (EPI == PA_EPI_RESID) {
                int64_t orow = m;
                if (a.row_mod > 0) orow = (int64_t)(m / a.row_mod) * a.out_batch_rows + a.out_row_off + m % a.row_mod;
                store8<float>(a.out_f32 + orow * a.ldo32 + ncol, v[it]);
            } else {  // PA_EPI_PARTIAL
                store8<float>(a.out_f32 + ((int64_t)split * a.M + m) * a.ldo32 + ncol, v[it]);
            }
        }
    }
    if constexpr (EPI == PA_EPI_DGELU) {
        if (a.colsum_out) {     // rows of this wave's tile: lanes with equal (lane & 7) hold the same 8 columns
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                csum[e] += __shfl_xor(csum[e], 8, 64);
                csum[e] += __shfl_xor(csum[e], 16, 64);
                csum[e] += __shfl_xor(csum[e], 32, 64);
            }
            if (lane < 8 && colok) store8<float>(a.colsum_ws + (int64_t)colsum_row * a.N + ncol, csum);
        }
    }
}

// f32-output epilogues (residual add, split-K partial) straight from the MFMA accumulator layout: register r of
// a 32x32 block holds row acc_row(r) and column lane&31, so one dword per lane is already a 128-byte run per
// row -- no LDS transposition at all.  Row bases are uniform (SGPR) + a fixed per-lane offset.  Residual rows
// are requested two 32-row passes ahead of their use and before the stores of the pass in between (vmcnt
// retires in order: a load queued behind a store waits for that store's acknowledge).
template <int EPI, int TM>
__device__ __forceinline__ void gemm_epilogue_f32_direct(const pa_gemm_args& a, f32x16 (&acc)[TM][2], const float* bias_row,
                                                         int m0, int n0, int split, int wr, int wc, int lane) {
    static_assert(EPI == PA_EPI_RESID || EPI == PA_EPI_PARTIAL, "f32 outputs only");
    constexpr bool RES = EPI == PA_EPI_RESID;
    const int half = lane >> 5, nl = lane & 31;
    const int mb = m0 + wr * (TM * 32);             // uniform: first row of this wave's tile
    const int nb = n0 + wc * 64 + nl;               // this lane's column in block j = 0 (j = 1: + 32)
    float b2[2] = {0.f, 0.f};
    if constexpr (RES) {
        if (bias_row) { b2[0] = bias_row[wc * 64 + nl]; b2[1] = bias_row[wc * 64 + 32 + nl]; }
    }
    float* outp = RES ? a.out_f32 : a.out_f32 + (int64_t)split * a.M * a.ldo32;
    const bool full = mb + TM * 32 <= a.M && n0 + wc * 64 + 64 <= a.N && !(RES && a.row_mod > 0);
#ifdef PA_PROBE
    if (PA_PROBE_FLAG(a, 0)) {          // no global traffic: keep the accumulators live, touch nothing
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
        return;
    }
#endif
    if (full) {
        // one uniform base per tile (SGPRs) + 32-bit per-lane offsets; the row term of the offset is uniform and
        // added on the fly, so nothing per row has to stay live
        const char* rbase = RES ? (const char*)a.resid + (int64_t)mb * a.ldr * 4 : nullptr;
        char* obase = (char*)outp + (int64_t)mb * a.ldo32 * 4;
        const uint32_t vo = (uint32_t)(4 * half * a.ldo32 + nb) * 4u, ldo4 = (uint32_t)a.ldo32 * 4u;
        const uint32_t vr = RES ? (uint32_t)(4 * half * a.ldr + nb) * 4u : 0u, ldr4 = RES ? (uint32_t)a.ldr * 4u : 0u;
        constexpr int U = 2 * TM, DEPTH = 3;          // units of one 32x32 block (16 rows x 1 dword per lane)
        float x[DEPTH][16];
        auto load_unit = [&](int slot, int u) {
            if constexpr (RES) {
                const int i = u >> 1, j = u & 1;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    x[slot][r] = *(const float*)(rbase + (vr + (uint32_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldr4 + j * 128));
            }
        };
#pragma unroll
        for (int u = 0; u < DEPTH && u < U; ++u) load_unit(u, u);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u >> 1, j = u & 1;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + b2[j] + (RES ? x[u % DEPTH][r] : 0.f);
            if (u + DEPTH < U) load_unit(u % DEPTH, u + DEPTH);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* dstp = (float*)(obase + (vo + (uint32_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldo4 + j * 128));
                if (PA_NT_ST_SLAB && !RES) __builtin_nontemporal_store(v[r], dstp);
                else *dstp = v[r];
            }
        }
    } else {   // edge tiles and the row-remapped (patch embedding) form: per-element checks
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + i * 32 + acc_row(r, lane);
                if (m >= a.M) continue;
                int64_t rrow = m, orow = m;
                if constexpr (RES) {
                    if (a.row_mod > 0) {
                        rrow = m % a.row_mod;
                        orow = (int64_t)(m / a.row_mod) * a.out_batch_rows + a.out_row_off + rrow;
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = nb + j * 32;
                    if (n >= a.N) continue;
                    float v = acc[i][j][r] + b2[j];
                    if constexpr (RES) v += a.resid[rrow * a.ldr + n];
                    outp[orow * a.ldo32 + n] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Epilogue v2 (r02) of the role-split kernels.  The first version was bound by its own instruction stream, not by the
// stores (profiles/r02_epilogue_probe.json: 5.4k cycles STORE / 15.3k GELU per item whatever the other CUs do): LDS write
// issue (ds_write_b32 moves 64 B/clk per CU), exec-mask branches around every 8 rows, 64-bit address arithmetic per
// access, dword-per-lane accesses for the f32 residual (1 536 vector-memory instructions of 256 bytes per 192x256 tile).
// This version is straight-line code:
//  * global accesses are BUFFER instructions: one descriptor per operand whose base is the wave tile's origin and whose
//    size is the bytes from there to the end of the matrix; the offset is a 32-bit VGPR (lane part + row term, one
//    v_add).  Rows past M and lanes whose 8 (4) columns are past N fall outside the descriptor: the hardware drops
//    those stores and returns 0 for those loads -- no predicate, no branch, edge tiles run the same code.
//  * bf16 outputs (STORE / GELU / DGELU): all math is done in the ACCUMULATOR layout, where registers r, r+1 (r even) of
//    a 32x32 block are two consecutive rows of one column: bias / GELU / GELU' work on such pairs and
//    v_cvt_pk_bf16_f32 packs the pair into ONE dword, written with one ds_write_b32 into a "row-pair" slab:
//        dword(p, c) = {row 2p, row 2p+1} of column c   at byte  p*256 + (((c>>2) ^ ((p>>1)&1)) << 4) + (c&3)*4
//    (16 pairs x 64 columns = 4 KiB per 32x64 pass: HALF the LDS write instructions; the XOR makes the ds_read_b128
//    below conflict free).  A lane then reads the 8 dwords of (pair p, columns 8g..8g+7), splits them with 8 v_perm_b32
//    into row 2p and row 2p+1 (8 bf16 = 16 bytes each) and stores two 16-byte vectors: 8 lanes cover a 128-byte row
//    segment.  PA_EPI_DGELU reads its pre-activation rows the same way in reverse: 16-byte row vectors -> interleaved
//    pair dwords -> slab -> ds_read_b32 in the accumulator layout; the column sums (fc1.bias gradient) are lane-local.
//  * f32 outputs (RESID): [32][64] f32 slab, no padding and no swizzle needed (writes: 32 lanes = 32 consecutive
//    columns of a row; reads: one 16-byte slot per lane, 16 lanes = one row, 4 rows per instruction); the residual rows
//    and the outputs move as 16-byte vectors, 4x fewer vector-memory instructions than the dword-per-lane form.
// ------------------------------------------------------------------------------------------------
#ifndef PA_EPILOGUE_V2
#define PA_EPILOGUE_V2 1
#endif
#ifndef PA_V2_DEPTH_X
#define PA_V2_DEPTH_X 2        // DGELU: 32-row passes of pre-activation rows requested ahead of their use (16 registers each)
#endif
#ifndef PA_V2_DEPTH_R
#define PA_V2_DEPTH_R 1        // RESID: 32-row passes of residual rows requested ahead (32 registers each; 2 spills at TM = 4)
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(int)))) int rsrc_bits_t;
__device__ __forceinline__ uint32_t perm_lo16(uint32_t hi_src, uint32_t lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x05040100u); }
__device__ __forceinline__ uint32_t perm_hi16(uint32_t hi_src, uint32_t lo_src) { return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u); }
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) { return __builtin_bit_cast(uint32_t, bf16x2{(bf16)lo, (bf16)hi}); }

// descriptor over [origin, origin + bytes): uniform arguments only (the tile origin comes from SGPRs)
__device__ __forceinline__ auto tile_rsrc(const void* origin, int64_t bytes) {
    const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(origin), 0, n, 0x00020000);
}
static constexpr uint32_t V2_OOB = 0x80000000u;     // a lane offset no descriptor of < 2 GiB contains

// rows of the blocked pre-activation buffer: whole multiples of 768 = lcm(256, 192, 128), so every 32-row pass of every
// tile height (TM = 4 / 3 / 2) has its block inside the buffer -- rounding to 256 left the last passes of a 192-row tile
// outside it for e.g. M = 474 * 7 (the block base rides in soffset, which the descriptor's range check does
// not cover)
__host__ __device__ __forceinline__ int64_t blocked_pre_rows(int M) { return ((int64_t)M + 767) / 768 * 768; }

// Auxiliary rows of an item's epilogue (DGELU: pre-activation, RESID: residual): descriptor + the row vectors requested
// ahead of their use.  The kernel calls issue() DURING the last K-tile of the item (the first rows come from HBM, not from
// a cache: requested at the start of the epilogue their ~2-3k-cycle latency was exposed once per item), the epilogue
// consumes the slots and refills them.  issue() is unconditional and always emits V2Aux::N loads: the K loop's counted
// vmcnt wait relies on that number (rows / columns outside the matrix fall outside the descriptor and read as 0).
// BLK (PA_GEMM_BLOCKED_PRE, GELU / DGELU only): the pre-activation tensor lives in the library's BLOCKED layout instead
// of row-major: one 4 KiB block per 32-row x 64-column pass of a wave tile, holding the pass in the ACCUMULATOR layout
// (block (R, C) at ((R * N/64) + C) * 4096 bytes; inside: [q = 0..3][lane][4 dwords], dword = rows {2k, 2k+1} of column
// 32 j + (lane & 31) for j = q >> 1, k = 4 (q & 1) + dword).  fc1's epilogue stores it and the dgrad-fc2 epilogue loads it
// with four 1 KiB-contiguous 16-byte-per-lane accesses per pass and NO LDS transposition on either side (nobody else
// reads the pre-activation; both GEMMs have the same M x N and pass geometry whatever their tile height).
template <int EPI, int TM, bool BLK = false> struct V2Aux {
    static constexpr bool X = EPI == PA_EPI_DGELU, R = EPI == PA_EPI_RESID;
    uint32_t blk_base = 0, blk_pitch = 0;                 // BLK: byte offset of pass 0's block, bytes between passes
    static constexpr int PASSES = X ? (PA_V2_DEPTH_X < TM ? PA_V2_DEPTH_X : TM) : 0;                    // DGELU: 32-row passes in flight
    static constexpr int HALVES = R ? (PA_V2_DEPTH_R < TM ? 2 * PA_V2_DEPTH_R : 2 * TM) : 0;        // RESID: 16-row half passes
    static constexpr int N = X ? PASSES * 4 : HALVES * 4;                                           // loads per wave in issue()
    u32x4 v[N > 0 ? N : 1];
    decltype(tile_rsrc(nullptr, 0)) rs;
    uint32_t vofs, ld;

    __device__ __forceinline__ void issue(const pa_gemm_args& a, int m0, int n0, int wr, int wc, int lane) {
        if constexpr (N > 0) {
            const int mb = m0 + wr * (TM * 32), nb = n0 + wc * 64;
            const bool exists = mb < a.M && nb < a.N;
            if constexpr (X && BLK) {
                const int cb = a.N >> 6;                                   // 64-column blocks per row of blocks
                blk_pitch = (uint32_t)cb * 4096u;
                blk_base = ((uint32_t)(mb >> 5) * (uint32_t)cb + (uint32_t)(nb >> 6)) * 4096u;
                rs = tile_rsrc(a.aux, exists ? (int64_t)blocked_pre_rows(a.M) * a.N * 2 : 0);
                vofs = (uint32_t)lane * 16u;
                ld = 0;
#pragma unroll
                for (int i = 0; i < PASSES; ++i) load_pass(i, i);
            } else if constexpr (X) {
                ld = (uint32_t)a.ldaux * 2u;
                rs = tile_rsrc((const char*)a.aux + ((int64_t)mb * a.ldaux + nb) * 2, exists ? ((int64_t)(a.M - mb - 1) * a.ldaux + (a.N - nb)) * 2 : 0);
                const int g = lane & 7;
                vofs = nb + g * 8 < a.N ? (uint32_t)(2 * (lane >> 3)) * ld + (uint32_t)g * 16u : V2_OOB;
#pragma unroll
                for (int i = 0; i < PASSES; ++i) load_pass(i, i);
            } else {
                ld = (uint32_t)a.ldr * 4u;
                rs = tile_rsrc((const char*)a.resid + ((int64_t)mb * a.ldr + nb) * 4, exists ? ((int64_t)(a.M - mb - 1) * a.ldr + (a.N - nb)) * 4 : 0);
                const int er = lane >> 4, sl = lane & 15;
                vofs = nb + sl * 4 < a.N ? (uint32_t)er * ld + (uint32_t)sl * 16u : V2_OOB;
#pragma unroll
                for (int hp = 0; hp < HALVES; ++hp) load_half(hp, hp);
            }
        }
    }
    // DGELU: rows 2p, 2p+1 of task t (p = lane>>3 + 8t) of pass i  ->  v[slot*4 + t*2 + q]
    __device__ __forceinline__ void load_pass(int slot, int i) {
        if constexpr (BLK) {       // blocked layout: the four 1 KiB quarters of pass i's block -> v[slot*4 + q]
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[slot * 4 + q] = __builtin_amdgcn_raw_buffer_load_b128(rs, vofs + (uint32_t)q * 1024u, blk_base + (uint32_t)i * blk_pitch, PA_AUX_LD_PRE);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    v[slot * 4 + t * 2 + q] = __builtin_amdgcn_raw_buffer_load_b128(rs, vofs + (uint32_t)(i * 32 + 16 * t + q) * ld, 0, PA_AUX_LD_PRE);
        }
    }
    // RESID: rows it*4 + (lane>>4) of half pass hp  ->  v[slot*4 + it]
    __device__ __forceinline__ void load_half(int slot, int hp) {
#pragma unroll
        for (int it = 0; it < 4; ++it) v[slot * 4 + it] = __builtin_amdgcn_raw_buffer_load_b128(rs, vofs + (uint32_t)(hp * 16 + it * 4) * ld, 0, PA_AUX_LD_RES);
    }
};

template <int EPI, int TM, bool BLK = false>
__device__ __forceinline__ void gemm_epilogue_v2_bf16(const pa_gemm_args& a, f32x16 (&acc)[TM][2], char* slab, const float* bias_row,
                                                      int m0, int n0, int wr, int wc, int lane, int colsum_row, V2Aux<EPI, TM, BLK>& aux) {
    static_assert(EPI == PA_EPI_STORE || EPI == PA_EPI_GELU || EPI == PA_EPI_DGELU, "bf16 outputs");
    const int h = lane >> 5, c = lane & 31;
    const int mb = m0 + wr * (TM * 32), nb = n0 + wc * 64;          // uniform: origin of this wave's tile
    if (mb >= a.M || nb >= a.N) {                                  // uniform: nothing of this wave tile exists
        if constexpr (EPI == PA_EPI_DGELU) {                       // ... but its row of column-sum partials is summed later
            if (a.colsum_out && nb < a.N && lane < 32) {
                float* cw = a.colsum_ws + (int64_t)colsum_row * a.N + nb + c;
                if (nb + c < a.N) cw[0] = 0.f;
                if (nb + 32 + c < a.N) cw[32] = 0.f;
            }
        }
        return;
    }
    // slab addresses: writes in the accumulator layout (pair k of block j: + j*128 + ((k&1) + 4*(k>>1))*256),
    // reads of task t = 0,1 (pair p = (lane>>3) + 8t, column group g = lane&7: + t*2048)
    char* const wbase = slab + 2 * h * 256 + (((c >> 2) ^ h) << 4) + (c & 3) * 4;
    const int g = lane & 7, fsw = (lane >> 4) & 1;
    char* const r0 = slab + (lane >> 3) * 256 + (((2 * g) ^ fsw) << 4);
    char* const r1 = slab + (lane >> 3) * 256 + (((2 * g + 1) ^ fsw) << 4);
    float b2[2] = {0.f, 0.f};
    if constexpr (EPI != PA_EPI_DGELU) {
        if (bias_row) { b2[0] = bias_row[wc * 64 + c]; b2[1] = bias_row[wc * 64 + 32 + c]; }
    }
    // PA_EPI_STORE: (acc + bias) * colscale for the columns below colscale_n (a multiple of 64: uniform per wave tile);
    // the scale rides in the multiply-add that used to be the bias add
    float cs = 1.f;
    if constexpr (EPI == PA_EPI_STORE) {
        if (nb < a.colscale_n) { cs = a.colscale; b2[0] *= cs; b2[1] *= cs; }
    }
    const bool colok = nb + g * 8 < a.N;                              // N % 8 == 0: all 8 columns of the lane in or out
    const uint32_t ld2 = (uint32_t)a.ldolp * 2u;
    const auto ors = tile_rsrc((const char*)a.out_lp + ((int64_t)mb * a.ldolp + nb) * 2, ((int64_t)(a.M - mb - 1) * a.ldolp + (a.N - nb)) * 2);
    const uint32_t vo = colok ? (uint32_t)(2 * (lane >> 3)) * ld2 + (uint32_t)g * 16u : V2_OOB;
    uint32_t ld2b = 0, vo2 = 0;
    auto ors2 = ors;
    if constexpr (EPI == PA_EPI_GELU) {
        ld2b = (uint32_t)a.ldolp2 * 2u;
        ors2 = tile_rsrc((const char*)a.out_lp2 + ((int64_t)mb * a.ldolp2 + nb) * 2, ((int64_t)(a.M - mb - 1) * a.ldolp2 + (a.N - nb)) * 2);
        vo2 = colok ? (uint32_t)(2 * (lane >> 3)) * ld2b + (uint32_t)g * 16u : V2_OOB;
    }
    constexpr int XD = V2Aux<EPI, TM, BLK>::PASSES > 0 ? V2Aux<EPI, TM, BLK>::PASSES : 1;
    float csum[2] = {0.f, 0.f};
    // BLK + GELU: the pre-activation goes out in the blocked layout (see V2Aux): descriptor over the whole buffer, block of
    // pass i at bbase + i * bpitch, lane part lane * 16
    auto brs = ors;
    uint32_t bbase = 0, bpitch = 0;
    if constexpr (BLK && EPI == PA_EPI_GELU) {
        const int cb = a.N >> 6;
        bpitch = (uint32_t)cb * 4096u;
        bbase = ((uint32_t)(mb >> 5) * (uint32_t)cb + (uint32_t)(nb >> 6)) * 4096u;
        brs = tile_rsrc(a.out_lp, (int64_t)blocked_pre_rows(a.M) * a.N * 2);
    }
    // (r02: running the LDS round trip of pass i under the polynomial math of pass i+1 -- software pipelining by hand --
    // measured no faster, 15.3k vs 14.0k cycles per 256x256 GELU tile, and costs ~50 registers: the passes stay sequential)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_sched_barrier(0);       // keep the passes apart: hoisting every pass's math to the top spills
        u32x4 xblk[4];                           // BLK + DGELU: the pass's pre-activations, already in the accumulator layout
        if constexpr (EPI == PA_EPI_DGELU && BLK) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xblk[q] = aux.v[(i % XD) * 4 + q];
            __builtin_amdgcn_sched_barrier(0);
            if (i + XD < TM) aux.load_pass(i % XD, i + XD);
        } else if constexpr (EPI == PA_EPI_DGELU) {
            // pre-activation rows -> pair dwords -> slab (two 16-byte writes per task) -> accumulator layout
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 x0 = aux.v[(i % XD) * 4 + t * 2], x1 = aux.v[(i % XD) * 4 + t * 2 + 1];   // rows 2p, 2p+1; dword q = columns 8g+2q, 8g+2q+1
                const u32x4 e0 = {perm_lo16(x1[0], x0[0]), perm_hi16(x1[0], x0[0]), perm_lo16(x1[1], x0[1]), perm_hi16(x1[1], x0[1])};
                const u32x4 e1 = {perm_lo16(x1[2], x0[2]), perm_hi16(x1[2], x0[2]), perm_lo16(x1[3], x0[3]), perm_hi16(x1[3], x0[3])};
                *(u32x4*)(r0 + t * 2048) = e0;
                *(u32x4*)(r1 + t * 2048) = e1;
            }
            __builtin_amdgcn_sched_barrier(0);   // the refill below must not be hoisted above the reads of its slot
            if (i + XD < TM) aux.load_pass(i % XD, i + XD);
        }
        const int rows_left = a.M - mb - i * 32;       // rows of this pass that exist (uniform): only the column sums need it
        uint32_t pk[2][8], pk2[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f32x2 v = {acc[i][j][2 * k] + b2[j], acc[i][j][2 * k + 1] + b2[j]};
                if constexpr (EPI == PA_EPI_STORE) v = f32x2{fmaf(acc[i][j][2 * k], cs, b2[j]), fmaf(acc[i][j][2 * k + 1], cs, b2[j])};
                if constexpr (EPI == PA_EPI_DGELU) {
                    uint32_t xw;
                    if constexpr (BLK) xw = xblk[j * 2 + (k >> 2)][k & 3];
                    else xw = *(const uint32_t*)(wbase + j * 128 + ((k & 1) + 4 * (k >> 1)) * 256);
                    const f32x2 x = {__builtin_bit_cast(float, xw << 16), __builtin_bit_cast(float, xw & 0xffff0000u)};
                    v = v * (PA_PROBE_FLAG(a, 2) ? x : gelu_grad_fast2(x));
                    csum[j] += v[0] + v[1];
                    if (rows_left < 32) {        // last row tile of the matrix (uniform branch): rows >= M hold duplicates of row M-1
                        // (subtracting what was just added keeps the hot path free of selects: masking BEFORE the add costs
                        // the TM = 4 kernel 212 B of scratch and 75 us per launch)
                        const int row = 2 * ((k & 1) + 4 * (k >> 1) + 2 * h);
                        csum[j] -= (row < rows_left ? 0.f : v[0]) + (row + 1 < rows_left ? 0.f : v[1]);
                    }
                }
                pk[j][k] = cvt_pk_bf16(v[0], v[1]);
                if constexpr (EPI == PA_EPI_GELU) {
                    const f32x2 gl = PA_PROBE_FLAG(a, 2) ? v : gelu_fast2(v);     // of the f32 value (the reference applies GELU before rounding too)
                    pk2[j][k] = cvt_pk_bf16(gl[0], gl[1]);
                }
            }
        if constexpr (BLK && EPI == PA_EPI_GELU) {       // pre-activation: straight from the registers, 4 x 1 KiB contiguous
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 d = {pk[q >> 1][4 * (q & 1)], pk[q >> 1][4 * (q & 1) + 1], pk[q >> 1][4 * (q & 1) + 2], pk[q >> 1][4 * (q & 1) + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(d, brs, (PA_PROBE_FLAG(a, 0) ? V2_OOB : (uint32_t)lane * 16u) + (uint32_t)q * 1024u,
                                                       bbase + (uint32_t)i * bpitch, PA_AUX_ST_PRE);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (!(BLK && EPI == PA_EPI_GELU)) *(uint32_t*)(wbase + j * 128 + ((k & 1) + 4 * (k >> 1)) * 256) = pk[j][k];
                if constexpr (EPI == PA_EPI_GELU) *(uint32_t*)(wbase + 4096 + j * 128 + ((k & 1) + 4 * (k >> 1)) * 256) = pk2[j][k];
            }
        // same-wave LDS accesses execute in order: no barrier between the writes and the reads
#pragma unroll
        for (int o = (BLK && EPI == PA_EPI_GELU) ? 1 : 0; o < (EPI == PA_EPI_GELU ? 2 : 1); ++o)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 d0 = *(const u32x4*)(r0 + o * 4096 + t * 2048), d1 = *(const u32x4*)(r1 + o * 4096 + t * 2048);
                const u32x4 lo = {perm_lo16(d0[1], d0[0]), perm_lo16(d0[3], d0[2]), perm_lo16(d1[1], d1[0]), perm_lo16(d1[3], d1[2])};
                const u32x4 hi = {perm_hi16(d0[1], d0[0]), perm_hi16(d0[3], d0[2]), perm_hi16(d1[1], d1[0]), perm_hi16(d1[3], d1[2])};
                const uint32_t ldb = o ? ld2b : ld2;
                const uint32_t off = (PA_PROBE_FLAG(a, 0) ? V2_OOB : (o ? vo2 : vo)) + (uint32_t)(i * 32 + 16 * t) * ldb;
                if (EPI == PA_EPI_STORE && PA_NT_STORE_MAXN > 0 && a.N > PA_NT_STORE_MAXN) {       // uniform
                    __builtin_amdgcn_raw_buffer_store_b128(lo, ors, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, ors, off + ldb, 0, 0);
                } else {
                    constexpr int AUX_ST = EPI == PA_EPI_GELU ? PA_AUX_ST_ACT : (EPI == PA_EPI_DGELU ? PA_AUX_ST_DPRE : PA_AUX_ST_STORE);
                    __builtin_amdgcn_raw_buffer_store_b128(lo, o ? ors2 : ors, off, 0, AUX_ST);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, o ? ors2 : ors, off + ldb, 0, AUX_ST);
                }
            }
    }
    if constexpr (EPI == PA_EPI_DGELU) {
        if (a.colsum_out) {       // this lane holds the sums of 16 of the 32 rows of each block: the other half is lane ^ 32
#pragma unroll
            for (int j = 0; j < 2; ++j) csum[j] += __shfl_xor(csum[j], 32, 64);
            if (lane < 32) {
                float* cw = a.colsum_ws + (int64_t)colsum_row * a.N + nb + c;
                if (nb + c < a.N) cw[0] = csum[0];
                if (nb + 32 + c < a.N) cw[32] = csum[1];
            }
        }
    }
}

// f32 residual epilogue through LDS: out[m][n] = acc + bias[n] + resid[m][n]
template <int TM>
__device__ __forceinline__ void gemm_epilogue_v2_resid(const pa_gemm_args& a, f32x16 (&acc)[TM][2], char* slab, const float* bias_row,
                                                       int m0, int n0, int wr, int wc, int lane, V2Aux<PA_EPI_RESID, TM>& aux) {
    const int h = lane >> 5, c = lane & 31;
    const int mb = m0 + wr * (TM * 32), nb = n0 + wc * 64;
    if (mb >= a.M || nb >= a.N) return;
    char* const wbase = slab + 4 * h * 256 + c * 4;                    // + ((r&3) + 8*(r>>2))*256 + j*128
    const int er = lane >> 4, sl = lane & 15;
    const char* const rdbase = slab + er * 256 + sl * 16;              // + it*1024
    float b2[2] = {0.f, 0.f};
    if (bias_row) { b2[0] = bias_row[wc * 64 + c]; b2[1] = bias_row[wc * 64 + 32 + c]; }
    const bool colok = nb + sl * 4 < a.N;                              // ld % 4 == 0 and N % 8 == 0: the lane's 4 columns are all in or out
    const uint32_t ldo4 = (uint32_t)a.ldo32 * 4u;
    const auto ors = tile_rsrc((const char*)a.out_f32 + ((int64_t)mb * a.ldo32 + nb) * 4, ((int64_t)(a.M - mb - 1) * a.ldo32 + (a.N - nb)) * 4);
    const uint32_t vo = colok ? (uint32_t)er * ldo4 + (uint32_t)sl * 16u : V2_OOB;
    // residual rows: DEPTH half passes (16 rows = 4 vectors per lane) were requested during the last K-tile (V2Aux)
    constexpr int NH = 2 * TM, DEPTH = V2Aux<PA_EPI_RESID, TM>::HALVES;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_sched_barrier(0);       // keep the passes apart (register pressure)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) *(float*)(wbase + ((r & 3) + 8 * (r >> 2)) * 256 + j * 128) = acc[i][j][r] + b2[j];
        // same-wave LDS accesses execute in order: no barrier between the writes and the reads
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int hp = 2 * i + hh;
            f32x4 o[4];
#pragma unroll
            for (int it = 0; it < 4; ++it)
                o[it] = *(const f32x4*)(rdbase + (hh * 4 + it) * 1024) + __builtin_bit_cast(f32x4, aux.v[(hp % DEPTH) * 4 + it]);
            __builtin_amdgcn_sched_barrier(0);   // the refill must not be hoisted above the adds (it would get fresh registers)
            if (hp + DEPTH < NH) aux.load_half(hp % DEPTH, hp + DEPTH);
#pragma unroll
            for (int it = 0; it < 4; ++it)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[it]), ors, vo + (uint32_t)(hp * 16 + it * 4) * ldo4, 0, PA_AUX_ST_RES);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Epilogue v3 (r04, opt-in): no LDS at all.  The role-split kernels can issue their MFMAs with the operands swapped
// (template TR: acc = B-fragment x A-fragment), which leaves the TRANSPOSED tile in the accumulators: the LANE is now the
// token row (m = lane & 31) and the registers run over the output columns, n = 8g + 4h + e for register 4g + e of the lane
// half h = lane >> 5 -- four CONSECUTIVE columns per register group.  Same products, same k order inside the MFMA: the
// values are bit-identical to the v2 orientation (tests/test_gpu_kernels.py::test_gemm_epilogue_v3_equals_v2).  What it buys:
//  * f32 outputs (RESID): a register group IS a 16-byte row vector: residual rows are loaded and results stored straight
//    from / to the registers (v2: 32 ds_write_b32 + 8 ds_read_b128 per 32x64 pass);
//  * bf16 outputs (STORE / GELU / DGELU): the two lanes of a row hold alternating 4-column groups; one v_permlane32_swap
//    per packed dword pair gives each lane 8 consecutive columns = one 16-byte store (the form attention.hip uses for its
//    output rows); the pre-activation rows of DGELU come in the same way, mirrored (16-byte load, swap, unpack).  v2 spent
//    16 ds_write_b32 + 4 ds_read_b128 + 16 v_perm per pass and output tensor there, and the LDS write issue (64 B/clk per
//    CU) was what its transposition cost (profiles/r02_epilogue_probe.json: 24 of the 68 us of the fc1 + GELU epilogue);
//  * the pre-activation needs no private blocked layout any more (row-major both ways, no LDS either side).
// Not in this orientation: the column sums of the DGELU output (lane-local when the lane is the column).  The fc1.bias
// gradient can come out of the weight-gradient launch instead (a ninth MFMA per phase against ones, as qkv.bias does:
// PASST_AMD_BIAS_FROM_WGRAD=1); a DGELU call with colsum_out keeps the v2 kernel.
// MEASURED SLOWER THAN v2 and therefore opt-in (see epilogue_v3_enabled): every store instruction covers 32 rows x 32 bytes
// (two lanes per row) -- four partial-line requests per row where v2 sends one full line.
// ------------------------------------------------------------------------------------------------
template <int EPI, int TM> struct V3Aux {
    static constexpr bool X = EPI == PA_EPI_DGELU, R = EPI == PA_EPI_RESID;
    static constexpr int PASSES = X ? (PA_V2_DEPTH_X < TM ? PA_V2_DEPTH_X : TM) : 0;              // DGELU: 32-row passes in flight
    static constexpr int HALVES = R ? (PA_V2_DEPTH_R < TM ? 2 * PA_V2_DEPTH_R : 2 * TM) : 0;  // RESID: 32-row x 32-column blocks in flight
    static constexpr int N = X ? PASSES * 4 : HALVES * 4;        // loads per wave in issue(): the K loop's counted vmcnt wait relies on it
    u32x4 v[N > 0 ? N : 1];
    decltype(tile_rsrc(nullptr, 0)) rs;
    uint32_t vofs[4], ld;       // DGELU: lane offset of chunk (j, gp) = vofs[2j + gp]; RESID: of group g = vofs[g] (block j: + 128 bytes)

    __device__ __forceinline__ void issue(const pa_gemm_args& a, int m0, int n0, int wr, int wc, int lane) {
        if constexpr (N > 0) {
            const int mb = m0 + wr * (TM * 32), nb = n0 + wc * 64;
            const bool exists = mb < a.M && nb < a.N;
            const int q = lane & 31, h = lane >> 5;
            if constexpr (X) {
                ld = (uint32_t)a.ldaux * 2u;
                rs = tile_rsrc((const char*)a.aux + ((int64_t)mb * a.ldaux + nb) * 2, exists ? ((int64_t)(a.M - mb - 1) * a.ldaux + (a.N - nb)) * 2 : 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = (c >> 1) * 32 + (c & 1) * 16 + 8 * h;
                    vofs[c] = nb + col < a.N ? (uint32_t)q * ld + (uint32_t)col * 2u : V2_OOB;
                }
#pragma unroll
                for (int i = 0; i < PASSES; ++i) load_pass(i, i);
            } else {
                ld = (uint32_t)a.ldr * 4u;
                rs = tile_rsrc((const char*)a.resid + ((int64_t)mb * a.ldr + nb) * 4, exists ? ((int64_t)(a.M - mb - 1) * a.ldr + (a.N - nb)) * 4 : 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) vofs[g] = (uint32_t)q * ld + (uint32_t)(8 * g + 4 * h) * 4u;
#pragma unroll
                for (int hp = 0; hp < HALVES; ++hp) load_half(hp, hp, a.N - nb);
            }
        }
    }
    // DGELU: the 64 pre-activations of this lane's row in pass i, as four 16-byte chunks (j, gp)
    __device__ __forceinline__ void load_pass(int slot, int i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[slot * 4 + c] = __builtin_amdgcn_raw_buffer_load_b128(rs, vofs[c] + (uint32_t)(i * 32) * ld, 0, 0);
    }
    // RESID: block hp = 2 i + j: the lane's four 4-column groups of row (lane & 31) of pass i
    __device__ __forceinline__ void load_half(int slot, int hp, int ncols) {
        const int j = hp & 1, h = threadIdx.x >> 5 & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t o = j * 32 + 8 * g + 4 * h < ncols ? vofs[g] + (uint32_t)((hp >> 1) * 32) * ld + (uint32_t)j * 128u : V2_OOB;
            v[slot * 4 + g] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
        }
    }
};

__device__ __forceinline__ uint32_t swap32_lo(uint32_t a, uint32_t b, uint32_t& other) {
    // v_permlane32_swap: the upper 32 lanes of a trade places with the lower 32 lanes of b
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    other = r[1];
    return r[0];
}

template <int EPI, int TM>
__device__ __forceinline__ void gemm_epilogue_v3_bf16(const pa_gemm_args& a, f32x16 (&acc)[TM][2], const float* bias_row,
                                                      int m0, int n0, int wr, int wc, int lane, V3Aux<EPI, TM>& aux) {
    static_assert(EPI == PA_EPI_STORE || EPI == PA_EPI_GELU || EPI == PA_EPI_DGELU, "bf16 outputs");
    const int h = lane >> 5, q = lane & 31;
    const int mb = m0 + wr * (TM * 32), nb = n0 + wc * 64;          // uniform: origin of this wave's tile
    if (mb >= a.M || nb >= a.N) return;
    const int ncols = a.N - nb;
    // lane offsets of the 16-byte chunk (j, gp): row q, columns 32 j + 16 gp + 8 h .. + 7 (N % 8 == 0: all in or all out)
    const uint32_t ld2 = (uint32_t)a.ldolp * 2u;
    const auto ors = tile_rsrc((const char*)a.out_lp + ((int64_t)mb * a.ldolp + nb) * 2, ((int64_t)(a.M - mb - 1) * a.ldolp + ncols) * 2);
    uint32_t vo[4], vo2[4];
    uint32_t ld2b = 0;
    auto ors2 = ors;
    if constexpr (EPI == PA_EPI_GELU) {
        ld2b = (uint32_t)a.ldolp2 * 2u;
        ors2 = tile_rsrc((const char*)a.out_lp2 + ((int64_t)mb * a.ldolp2 + nb) * 2, ((int64_t)(a.M - mb - 1) * a.ldolp2 + ncols) * 2);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int col = (c >> 1) * 32 + (c & 1) * 16 + 8 * h;
        const bool ok = col < ncols && !PA_PROBE_FLAG(a, 0);
        vo[c] = ok ? (uint32_t)q * ld2 + (uint32_t)col * 2u : V2_OOB;
        vo2[c] = ok ? (uint32_t)q * ld2b + (uint32_t)col * 2u : V2_OOB;
    }
    // bias of this lane's columns: register 4g + e of block j <-> column 32 j + 8 g + 4 h + e (LDS broadcast reads)
    f32x4 b4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            b4[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI != PA_EPI_DGELU) {
                if (bias_row) b4[j][g] = *(const f32x4*)(bias_row + wc * 64 + j * 32 + 8 * g + 4 * h);
            }
        }
    float cs = 1.f;
    if constexpr (EPI == PA_EPI_STORE) {
        if (nb < a.colscale_n) {
            cs = a.colscale;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[j][g] *= cs;
        }
    }
    constexpr int XD = V3Aux<EPI, TM>::PASSES > 0 ? V3Aux<EPI, TM>::PASSES : 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_sched_barrier(0);       // keep the passes apart (register pressure)
        // DGELU: this row's pre-activations, redistributed to the accumulator's column groups: chunk (j, gp) holds columns
        // 16 gp + 8 h + {0..7} as dwords x0..x3; group 2 gp of this lane = {lower: x0 x1, upper: lower's x2 x3}, group 2 gp + 1
        // = {lower: upper's x0 x1, upper: x2 x3}
        uint32_t xg[2][4][2];
        if constexpr (EPI == PA_EPI_DGELU) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 x = aux.v[(i % XD) * 4 + c];
                const int j = c >> 1, gp = c & 1;
                xg[j][2 * gp][0] = swap32_lo(x[0], x[2], xg[j][2 * gp + 1][0]);
                xg[j][2 * gp][1] = swap32_lo(x[1], x[3], xg[j][2 * gp + 1][1]);
            }
            __builtin_amdgcn_sched_barrier(0);   // the refill below must not be hoisted above the reads of its slot
            if (i + XD < TM) aux.load_pass(i % XD, i + XD);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t w[4][2], w2[4][2];          // group g: packed {col 0, 1}, {col 2, 3} of the group
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int r = 4 * g + 2 * e2;
                    f32x2 v = {acc[i][j][r] + b4[j][g][2 * e2], acc[i][j][r + 1] + b4[j][g][2 * e2 + 1]};
                    if constexpr (EPI == PA_EPI_STORE) v = f32x2{fmaf(acc[i][j][r], cs, b4[j][g][2 * e2]), fmaf(acc[i][j][r + 1], cs, b4[j][g][2 * e2 + 1])};
                    if constexpr (EPI == PA_EPI_DGELU) {
                        const uint32_t xw = xg[j][g][e2];
                        const f32x2 x = {__builtin_bit_cast(float, xw << 16), __builtin_bit_cast(float, xw & 0xffff0000u)};
                        v = v * (PA_PROBE_FLAG(a, 2) ? x : gelu_grad_fast2(x));
                    }
                    w[g][e2] = cvt_pk_bf16(v[0], v[1]);
                    if constexpr (EPI == PA_EPI_GELU) {
                        const f32x2 gl = PA_PROBE_FLAG(a, 2) ? v : gelu_fast2(v);     // of the f32 value (the reference applies GELU before rounding too)
                        w2[g][e2] = cvt_pk_bf16(gl[0], gl[1]);
                    }
                }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // lower half keeps its group 2 gp and receives the upper half's; upper half keeps 2 gp + 1 and receives the lower's
                uint32_t o0, o1;
                const uint32_t k0 = swap32_lo(w[2 * gp][0], w[2 * gp + 1][0], o0), k1 = swap32_lo(w[2 * gp][1], w[2 * gp + 1][1], o1);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{k0, k1, o0, o1}, ors, vo[2 * j + gp] + (uint32_t)(i * 32) * ld2, 0, 0);
                if constexpr (EPI == PA_EPI_GELU) {
                    const uint32_t g0 = swap32_lo(w2[2 * gp][0], w2[2 * gp + 1][0], o0), g1 = swap32_lo(w2[2 * gp][1], w2[2 * gp + 1][1], o1);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{g0, g1, o0, o1}, ors2, vo2[2 * j + gp] + (uint32_t)(i * 32) * ld2b, 0, 0);
                }
            }
        }
    }
}

// out[m][n] = acc + bias[n] + resid[m][n], f32, straight from / to the registers
template <int TM>
__device__ __forceinline__ void gemm_epilogue_v3_resid(const pa_gemm_args& a, f32x16 (&acc)[TM][2], const float* bias_row,
                                                       int m0, int n0, int wr, int wc, int lane, V3Aux<PA_EPI_RESID, TM>& aux) {
    const int h = lane >> 5, q = lane & 31;
    const int mb = m0 + wr * (TM * 32), nb = n0 + wc * 64;
    if (mb >= a.M || nb >= a.N) return;
    const int ncols = a.N - nb;
    const uint32_t ldo4 = (uint32_t)a.ldo32 * 4u;
    const auto ors = tile_rsrc((const char*)a.out_f32 + ((int64_t)mb * a.ldo32 + nb) * 4, ((int64_t)(a.M - mb - 1) * a.ldo32 + ncols) * 4);
    uint32_t vo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) vo[g] = (uint32_t)q * ldo4 + (uint32_t)(8 * g + 4 * h) * 4u;
    constexpr int NH = 2 * TM, DEPTH = V3Aux<PA_EPI_RESID, TM>::HALVES;
#pragma unroll
    for (int hp = 0; hp < NH; ++hp) {
        const int i = hp >> 1, j = hp & 1;
        __builtin_amdgcn_sched_barrier(0);       // keep the blocks apart (register pressure: 256-row tiles spilled otherwise)
        f32x4 o[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // the bias of this lane's four columns: an LDS broadcast read per group, not 32 registers held across the epilogue
            const f32x4 b = bias_row ? *(const f32x4*)(bias_row + wc * 64 + j * 32 + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 r = __builtin_bit_cast(f32x4, aux.v[(hp % DEPTH) * 4 + g]);
            o[g] = f32x4{acc[i][j][4 * g] + b[0], acc[i][j][4 * g + 1] + b[1], acc[i][j][4 * g + 2] + b[2], acc[i][j][4 * g + 3] + b[3]} + r;
        }
        __builtin_amdgcn_sched_barrier(0);       // the refill must not be hoisted above the adds (it would get fresh registers)
        if (hp + DEPTH < NH) aux.load_half(hp % DEPTH, hp + DEPTH, ncols);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t off = (j * 32 + 8 * g + 4 * h < ncols && !PA_PROBE_FLAG(a, 0)) ? vo[g] + (uint32_t)(i * 32) * ldo4 + (uint32_t)j * 128u : V2_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[g]), ors, off, 0, 0);
        }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `k * PER` of this wave's global->LDS copies are still in flight (k = 0..3)
template <int PER> __device__ __forceinline__ void wait_vm_groups(int k) {
    switch (k) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<PER>(); break;
        case 2: wait_vmcnt<2 * PER>(); break;
        default: wait_vmcnt<3 * PER>(); break;
    }
}

// L2-friendly tile order: consecutive workgroup ids (which the XCD remap keeps on one XCD) walk down GROUP_M
// row-tiles of one column-tile before moving to the next column, so the ~64 co-resident workgroups of an
// XCD touch a GROUP_M x 8 patch of tiles (A and B panels both reused from the 4 MiB L2).
__device__ __forceinline__ void tile_coords(int pid, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int g = pid / per_group;
    const int first = g * group_m;
    const int gm = min(tiles_m - first, group_m);
    const int r = pid - g * per_group;
    tm = first + r % gm;
    tn = r / gm;
}

// Workgroup tile = (WM*TM*32) x (WN*64): WM x WN waves, each owning TM x 2 MFMA 32x32 accumulators
// (wave tile TM*32 rows x 64 columns); STAGES-deep LDS ring.  Larger wave tiles cut LDS bytes per MFMA
// (fragments per MFMA: (TM+2)/(2 TM) = 1.0 at TM=2, 0.75 at TM=4).
// KBT = K bytes per ring stage: 128 (default) or 64.  With 64-byte rows the XOR term is (row >> 2) & 3: the four rows
// of a 16-lane ds_read_b128 group that share a 64-byte bank quarter get four different 16-byte slots.  The 64-byte
// form exists for the 128x256 / 4-wave / 3-stage variant (72 KiB: TWO workgroups per CU, so one workgroup's epilogue
// overlaps the other's K loop).
template <int KBT> __device__ __forceinline__ int swz_rows(int row) {
    if constexpr (KBT == 128) return swz_f128(row);
    else return (row >> 2) & 3;
}

// BLK (r06): the epilogue is the role-split kernels' v2 epilogue (accumulator-layout math, packed row-pair slab, buffer
// instructions) with the MLP pre-activation in the library's blocked layout -- what makes the blocked pre-activation reachable
// from the tiles that fit TWO workgroups per CU (tunes 3 / 9: one workgroup's transposition + store burst under the other's K loop)
// EPV (r06): 0 = the first-generation epilogue; 1 = epilogue v2 (row-major outputs); 2 = epilogue v2 with the blocked pre-activation
template <typename T, int EPI, int WM, int WN, int TM, int STAGES, int KBT = KB, int EPV = 0>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu((KBT == 64 || EPV) ? 2 : 1))) void gemm_nt_kernel(const pa_gemm_args a, const int tiles_m, const int tiles_n,
                                                               const int nwg, const int ksteps_per_split) {
    constexpr int TBM = WM * TM * 32, TBN = WN * 64;
    constexpr int A_BYTES = TBM * KBT, B_BYTES = TBN * KBT, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int CPR = KBT / 16, NSUB = KBT / 32;       // 16-byte chunks per row, 16-wide k-substeps per stage
    constexpr int NW = WM * WN;
    constexpr int A_PER = (A_BYTES / 1024) / NW;      // A copies per wave per stage
    constexpr int B_PER = (B_BYTES / 1024) / NW;      // B copies per wave per stage
    constexpr int PER = A_PER + B_PER;
    static_assert(A_PER * NW * 1024 == A_BYTES && B_PER * NW * 1024 == B_BYTES, "tile must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;

    int tm, tn;
    tile_coords(xcd_swizzle(blockIdx.x, nwg), tiles_m, tiles_n, 1024 / TBM, tm, tn);
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int ksteps_total = (int)((int64_t)a.K * sizeof(T) / KBT);
    const int ks_begin = blockIdx.y * ksteps_per_split;
    const int ks_end = min(ksteps_total, ks_begin + ksteps_per_split);
    const int nsteps = ks_end - ks_begin;

    // per-lane source addresses: wave-instruction i of this wave fills LDS bytes [(wave*PERX+i)*1024, +1024)
    // of the A (resp. B) tile: lane -> physical chunk q -> row q>>3, logical chunk (q&7) ^ swz_f128(row).
    const char* srcA[A_PER];
    const char* srcB[B_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int q = (wave * A_PER + i) * 64 + lane;
        const int row = q / CPR;
        const int c = (q % CPR) ^ swz_rows<KBT>(row);
        const int gm = min(m0 + row, a.M - 1);
        srcA[i] = (const char*)a.A + ((int64_t)gm * a.lda) * sizeof(T) + c * 16 + (int64_t)ks_begin * KBT;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int q = (wave * B_PER + i) * 64 + lane;
        const int row = q / CPR;
        const int c = (q % CPR) ^ swz_rows<KBT>(row);
        const int gn = min(n0 + row, a.N - 1);
        srcB[i] = (const char*)a.B + ((int64_t)gn * a.ldb) * sizeof(T) + c * 16 + (int64_t)ks_begin * KBT;
    }
    auto stage = [&](int buf, int step) {
        char* sA = smem + buf * STAGE_BYTES + wave * (A_PER * 1024);
        char* sB = smem + buf * STAGE_BYTES + A_BYTES + wave * (B_PER * 1024);
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(srcA[i] + (int64_t)step * KBT),
                (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(srcB[i] + (int64_t)step * KBT),
                (__attribute__((address_space(3))) void*)(sB + i * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // read offsets: row = base + i*32 + (lane&31); swz_f128(row) == swz_f128(lane) for every i
    const int rsw = swz_rows<KBT>(lane & 31);
    const int offA = (wr * (TM * 32) + (lane & 31)) * KBT;
    const int offB = (wc * 64 + (lane & 31)) * KBT;
    const int half = lane >> 5;

    // prologue: fill STAGES-1 ring slots
#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
        if (p < nsteps) stage(p, p);
    float bias8[8];
    load_bias8<EPI>(a, n0, wc, lane, bias8);
    int buf = 0;
    for (int t = 0; t < nsteps; ++t) {
        // tile t must have landed; up to min(STAGES-2, nsteps-1-t) younger tiles may stay in flight
        wait_vm_groups<PER>(min(STAGES - 2, nsteps - 1 - t));
        __builtin_amdgcn_s_barrier();     // raw barrier: keeps the younger copies in flight (no vmcnt(0))
        // everyone has finished reading ring slot (t-1)%STAGES during iteration t-1: refill it with tile t+STAGES-1
        if (t + STAGES - 1 < nsteps) {
            int nb = buf + STAGES - 1;
            if (nb >= STAGES) nb -= STAGES;
            stage(nb, t + STAGES - 1);
        }
        const char* sA = smem + buf * STAGE_BYTES;
        const char* sB = sA + A_BYTES;
        // fragment double buffer: the ds_read_b128s of k-substep ks+1 are issued before the MFMAs of ks, so
        // LDS latency overlaps the matrix pipe inside one wave (not only across the waves of a SIMD)
        typename Frag<T>::type fa[2][TM], fb[2][2];
        auto read_frags = [&](int slot, int ks) {
            const int coff = ((ks * 2 + half) ^ rsw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[slot][i] = *(const typename Frag<T>::type*)(sA + offA + i * 32 * KBT + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[slot][j] = *(const typename Frag<T>::type*)(sB + offB + j * 32 * KBT + coff);
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < NSUB; ++ks) {
            if (ks + 1 < NSUB) read_frags((ks + 1) & 1, ks + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fa[ks & 1][i], fb[ks & 1][j]);
        }
        if (++buf == STAGES) buf = 0;
    }
    if constexpr (EPV != 0) {
        constexpr bool BLK = EPV == 2;
        static_assert(sizeof(T) == 2 && EPI != PA_EPI_PARTIAL, "epilogue v2: bf16 operands, fused epilogues");
        static_assert(!BLK || EPI == PA_EPI_GELU || EPI == PA_EPI_DGELU, "blocked pre-activation: GELU / GELU' only");
        constexpr bool BLKX = BLK && (EPI == PA_EPI_GELU || EPI == PA_EPI_DGELU);
        V2Aux<EPI, TM, BLKX> aux;
        aux.issue(a, m0, n0, wr, wc, lane);          // GELU' / residual: the first auxiliary rows are in flight across the barrier
        __syncthreads();                             // every wave is done reading the operand tiles; LDS is reused by the slabs
        // (the bias row is read from global memory: N % 64 == 0 -- launch guard -- and the wave tile's early-out keep wc * 64 + c inside it)
        const float* brow = (EPI != PA_EPI_DGELU && a.bias) ? a.bias + n0 : nullptr;
        if constexpr (EPI == PA_EPI_RESID) gemm_epilogue_v2_resid<TM>(a, acc, smem + wave * 8192, brow, m0, n0, wr, wc, lane, aux);
        else gemm_epilogue_v2_bf16<EPI, TM, BLKX>(a, acc, smem + wave * 8192, brow, m0, n0, wr, wc, lane, tm * WM + wr, aux);
    } else {
        __syncthreads();  // every wave is done reading the operand tiles; LDS is reused by the slabs
        gemm_epilogue<T, EPI, TM>(a, acc, (float*)(smem + wave * SLAB_BYTES), m0, n0, blockIdx.y, wr, wc, lane, bias8, tm * WM + wr);
    }
}

template <typename T, int EPI, int WM, int WN, int TM, int STAGES, int KBT = KB, int EPV = 0>
static int launch_gemm_v(const pa_gemm_args& a, hipStream_t st) {
    constexpr int TBM = WM * TM * 32, TBN = WN * 64;
    constexpr int LDS = STAGES * (TBM + TBN) * KBT;
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static_assert(LDS >= WM * WN * 32 * 68 * 4, "epilogue slabs must fit");
    if constexpr (EPV != 0) {     // buffer-descriptor epilogue: every row within 2 GiB of the first one, whole 64-column wave tiles,
        const int64_t lim = (int64_t)1 << 31;                      // no row remap (the patch embedding keeps the first-generation epilogue)
        if ((int64_t)a.M * a.ldolp * 2 >= lim || (int64_t)a.M * a.ldolp2 * 2 >= lim || (int64_t)a.M * a.ldaux * 2 >= lim ||
            (int64_t)a.M * a.ldr * 4 >= lim || (int64_t)a.M * a.ldo32 * 4 >= lim || a.N % 64 || (EPI == PA_EPI_RESID && a.row_mod > 0))
            return EPV == 2 ? PA_EUNSUPPORTED : launch_gemm_v<T, EPI, WM, WN, TM, STAGES, KBT, 0>(a, st);
    }
    const int tiles_m = (int)cdiv(a.M, TBM), tiles_n = (int)cdiv(a.N, TBN);
    const int nwg = tiles_m * tiles_n;
    const int ksteps = (int)((int64_t)a.K * sizeof(T) / KBT);
    const int splits = EPI == PA_EPI_PARTIAL ? a.split_k : 1;
    const int per = (int)cdiv(ksteps, splits);
    static signed char lds_attr[64] = {0};
    (void)lds_attr_on_this_device((const void*)gemm_nt_kernel<T, EPI, WM, WN, TM, STAGES, KBT, EPV>, LDS, lds_attr);
    hipLaunchKernelGGL((gemm_nt_kernel<T, EPI, WM, WN, TM, STAGES, KBT, EPV>), dim3(nwg, splits), dim3(WM * WN * 64), LDS, st, a,
                       tiles_m, tiles_n, nwg, per);
    const int rc = check_launch();
    if (rc == PA_OK && EPI == PA_EPI_DGELU && a.colsum_out) return finish_gemm_colsum(a, tiles_m * WM, st);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Role-split ("staggered") 256x256 kernel.  8 waves = 2 groups x 4; group g owns rows g*128..+127, each
// wave a 128x64 output tile (4x2 MFMA 32x32 accumulators).  A workgroup's waves w and w+4 share a SIMD, so
// every SIMD hosts one wave of each group.  The K loop is cut into segments separated by s_barrier:
//     L (load):  6 ds_read_b128 (fragments of ONE k-substep) + a share of the next K-tile's LDS-DMA
//     M (math):  8 MFMA 32x32x16 under s_setprio 1
// and group 1 runs ONE barrier behind group 0, so in every barrier interval one wave of each SIMD is in M
// while its partner is in L: the matrix pipe and the LDS/VMEM paths are used concurrently instead of
// alternately (guide 5.5 T3-T5, MI355X_MICROARCH "Two waves per SIMD").
// LDS: two K-tile buffers of 64 KiB.  The DMA of tile t+1 is issued in the first two L segments of tile t
// (>= 5 barrier intervals before its first use); every wave drains its own DMA (vmcnt 0) just before the
// barrier that ends tile t's last interval, so the buffer is complete when either group starts reading it.
// ------------------------------------------------------------------------------------------------
// The kernel is PERSISTENT: min(#work items, 256) workgroups (one per CU), each walking work items
// (output tile x K split) round by round.  The first K-tile of the NEXT item is DMA'd during the last K-tile
// of the current one, so the K loop runs without a prologue bubble across items, and the epilogue's slabs
// live in the K-tile buffer that was read last (+ a few KiB past the ring), never in the one already
// holding the next item's data.
// A3 (r02): the A operand (activations) gets THREE ring slots and is requested TWO K-tiles ahead, the B operand (weights,
// L2 resident) keeps two slots / one tile.  In the training step the activation panels come from HBM (the 140-186 MB
// operands of the K = 2304 / 3072 input-gradient GEMMs do not survive in the 256 MB Infinity Cache between producer and
// consumer), and one K-tile of lead (~1.1-1.4 us) is less than a loaded HBM round trip: those GEMMs ran 21-23 % slower in
// the step than alone on cache-resident data (profiles/r02_instep_vs_isolated.json).  Needs TM <= 3 (LDS) and >= 2 K-tiles.
template <int TM, bool A3> struct StaggerGeom {
    static constexpr int TBM = 64 * TM, TBN = 256;
    static constexpr int A_BYTES = TBM * KB, B_BYTES = TBN * KB, STAGE_BYTES = A_BYTES + B_BYTES;   // 48/56/64 KiB
    static constexpr int NA = A3 ? 3 : 2;
    static constexpr int RING = NA * A_BYTES + 2 * B_BYTES;
    // ring layout.  !A3: [A0 B0][A1 B1] (a K-tile's operands adjacent: the epilogue slabs take one whole stage);
    //               A3 : [A0 A1 A2][B0 B1]
    __host__ __device__ static constexpr int a_off(int s) { return A3 ? s * A_BYTES : s * STAGE_BYTES; }
    __host__ __device__ static constexpr int b_off(int s) { return A3 ? 3 * A_BYTES + s * B_BYTES : s * STAGE_BYTES + A_BYTES; }
    // epilogue slabs live in the ring slots consumed last.  !A3: SLAB_BYTES each, SLABS_IN_STAGE of them in that stage;
    // A3: 8 KiB each (what epilogue v2 needs), SLABS_A in the free A slot, 4 in the free B slot; the rest past the ring
    static constexpr int SLAB = A3 ? 8192 : SLAB_BYTES;
    static constexpr int SLABS_IN_STAGE = STAGE_BYTES / SLAB_BYTES;                                 // 5 / 6 / 7
    static constexpr int SLABS_A = A_BYTES / 8192, SLABS_B = B_BYTES / 8192;
    static constexpr int EXTRA = A3 ? 8 - SLABS_A - SLABS_B : 8 - SLABS_IN_STAGE;
    static constexpr int BIAS_OFF = RING + EXTRA * SLAB;    // 256 floats: the tile's bias row
    static constexpr int TAB_OFF = BIAS_OFF + 1024;         // {m0, n0, split, -} of this workgroup's item in every round
    static constexpr int MAX_ROUNDS = A3 ? 128 : 512;
    static constexpr int LDS = TAB_OFF + MAX_ROUNDS * 16;
};

// TR (r04): MFMA operands swapped -> transposed accumulators (lane = token row) and the LDS-free epilogue v3 (see there)
template <typename T, int EPI, int TM, bool A3 = false, bool BLK = false, bool TR = false>
__global__ __launch_bounds__(512) void gemm_nt_stagger_kernel(const pa_gemm_args a, const int tiles_m, const int tiles_n,
                                                              const int nwg, const int ksteps_per_split, const int total) {
    using G = StaggerGeom<TM, A3>;
    static_assert(!TR || (PA_EPILOGUE_V2 && sizeof(T) == 2 && EPI != PA_EPI_PARTIAL && !BLK), "v3: bf16 operands, fused epilogues, row-major pre-activation");
    constexpr int WN = 4;
    constexpr int TBM = G::TBM, TBN = G::TBN;         // TM = 2/3/4 -> 128/192/256-row tiles (tile quantisation)
    constexpr int A_PER = TM;                         // A copies per wave per stage: TBM*128/1024/8
    constexpr int NA = G::NA, PDA = NA - 1;           // A ring slots; K-tiles of lead of the A requests (B: always 1)
    constexpr bool USE_V2 = PA_EPILOGUE_V2 && sizeof(T) == 2 && EPI != PA_EPI_PARTIAL;
    static_assert(!A3 || (TM <= 3 && (USE_V2 || EPI == PA_EPI_PARTIAL)), "A3: LDS budget / 8 KiB slabs");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;      // wr = group
    const int ksteps_total = (int)((int64_t)a.K * sizeof(T) / KB);
    const int nres = gridDim.x, bid = blockIdx.x;

    // Work items of this workgroup: round r -> logical id r*nres + xcd_swizzle(bid, n_r) (within a round
    // consecutive ids share an XCD and walk an L2-friendly patch of tiles).  The id -> (tile, split) mapping
    // needs integer divisions, which have no scalar instruction: one lane per round does them once, up
    // front, into an LDS table; the K loop only reads the table.
    const int full_rounds = total / nres;
    const int my_rounds = full_rounds + (bid < total - full_rounds * nres ? 1 : 0);
    int4* tab = (int4*)(smem + G::TAB_OFF);
    for (int r = tid; r < my_rounds; r += 512) {
        const int n = min(total - r * nres, nres);
        const int item = r * nres + xcd_swizzle(bid, n);
        int tm, tn;
        const int sp = item / nwg;
        tile_coords(item - sp * nwg, tiles_m, tiles_n, PA_NT_GROUP_M, tm, tn);
        tab[r] = make_int4(tm * TBM, tn * TBN, sp, 0);
    }
#ifdef PA_PROBE
    for (int i = tid; i < 2 * PROBE_SLOTS; i += 512) ((unsigned long long*)(smem + G::LDS))[i] = 0ull;
#endif
    __syncthreads();
    // DMA cursors, one per operand (A runs PDA tiles ahead of the MFMAs, B one: near an item's end they point at
    // different items): a uniform 64-bit base (SGPRs: tile origin + K offset) plus per-lane 32-bit offsets (row within
    // the tile, clamped at the matrix edge, and the swizzled 16-byte chunk) -- a handful of VALU per item, none per K-tile.
    const char* baseA;
    const char* baseB;
    uint32_t voffA[A_PER], voffB[4];
    auto item_of = [&](int r, int& m0, int& n0, int& split) {
        const int4 e = tab[r];
        m0 = __builtin_amdgcn_readfirstlane(e.x);
        n0 = __builtin_amdgcn_readfirstlane(e.y);
        split = __builtin_amdgcn_readfirstlane(e.z);
        const int ks_begin = split * ksteps_per_split;
        return min(ksteps_total, ks_begin + ksteps_per_split) - ks_begin;     // K-tiles of the item (>= 1, see launch)
    };
    auto point_A = [&](int r) {
        int m0, n0, split;
        item_of(r, m0, n0, split);
        baseA = (const char*)a.A + (int64_t)m0 * a.lda * sizeof(T) + (int64_t)split * ksteps_per_split * KB;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int q = (wave * A_PER + i) * 64 + lane;
            const int row = q >> 3;
            const int c = (q & 7) ^ swz_f128(row);
            voffA[i] = (uint32_t)min(row, a.M - 1 - m0) * (uint32_t)(a.lda * (int)sizeof(T)) + c * 16;
        }
    };
    auto point_B = [&](int r) {
        int m0, n0, split;
        item_of(r, m0, n0, split);
        baseB = (const char*)a.B + (int64_t)n0 * a.ldb * sizeof(T) + (int64_t)split * ksteps_per_split * KB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = (wave * 4 + i) * 64 + lane;
            const int row = q >> 3;
            const int c = (q & 7) ^ swz_f128(row);
            voffB[i] = (uint32_t)min(row, a.N - 1 - n0) * (uint32_t)(a.ldb * (int)sizeof(T)) + c * 16;
        }
    };
    auto dmaA = [&](int slot, int step) {
        char* sA = smem + G::a_off(slot) + wave * (A_PER * 1024);
        const char* sb = baseA + (int64_t)step * KB;
        if (EPI == PA_EPI_STORE && PA_NT_A_STORE_MAXN > 0 && a.N <= PA_NT_A_STORE_MAXN) {       // uniform
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voffA[i]),
                                                 (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, 2);
            return;
        }
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voffA[i]),
                                             (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, EPI == PA_EPI_RESID ? PA_AUX_DMA_A_RESID : PA_AUX_DMA_A);
    };
    auto dmaB = [&](int slot, int step) {
        char* sB = smem + G::b_off(slot) + wave * 4096;
        const char* sb = baseB + (int64_t)step * KB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voffB[i]),
                                             (__attribute__((address_space(3))) void*)(sB + i * 1024), 16, 0, PA_AUX_DMA_B);
    };

    const int rsw = swz_f128(lane);
    const int offA = (wr * (TM * 32) + (lane & 31)) * 128;
    const int offB = (wc * 64 + (lane & 31)) * 128;
    const int half = lane >> 5;

    int round = 0;
    int m0, n0, split;
    int nsteps = item_of(0, m0, n0, split);                // grid <= total: every workgroup owns an item in round 0
    point_A(0);
    point_B(0);
    dmaA(0, 0);
    if constexpr (A3) dmaA(1, 1);                          // launch guard: every item has >= 2 K-tiles
    dmaB(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // K-tile 0 (and 1 of A) complete for everyone
    int ia = 0, ib = 0;                                    // ring slots of the K-tile about to be consumed

    for (;;) {
        const bool have_next = round + 1 < my_rounds;
        // the tile's bias row goes to LDS (4 bytes per lane, one DMA by each wave of group 0): it is read back in
        // the epilogue, so it costs no registers during the K loop and its latency is never exposed
        constexpr bool HAS_BIAS = EPI != PA_EPI_PARTIAL && EPI != PA_EPI_DGELU;
        if constexpr (HAS_BIAS) {
            if (a.bias && wr == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.bias + min(n0 + wc * 64 + lane, a.N - 1)),
                                                 (__attribute__((address_space(3))) void*)(smem + G::BIAS_OFF + wc * 256), 4, 0, 0);
        }
        // the accumulators are NOT cleared: the first k-substep of an item issues its MFMAs with C = 0 (mma32_first), which
        // saves TM x 32 v_mov per wave and item in the seam between two items
        f32x16 acc[TM][2];
        constexpr int AUX_EPI = USE_V2 ? EPI : PA_EPI_STORE;
        std::conditional_t<TR, V3Aux<AUX_EPI, TM>, V2Aux<AUX_EPI, TM, BLK>> aux;
        constexpr int AUXN = decltype(aux)::N;
        static_assert(V3Aux<AUX_EPI, TM>::N == V2Aux<AUX_EPI, TM, false>::N, "both epilogues leave the same number of loads in flight");
        const int cur_m0 = m0, cur_n0 = n0, cur_split = split;
        if (wr == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind
        PA_PROBE_STAMP(round < 24, round * 16);

        for (int t = 0; t < nsteps; ++t) {
            const char* sA = smem + G::a_off(ia);
            const char* sB = smem + G::b_off(ib);
            const bool last = t + 1 == nsteps;
            // requests made during this tile: B of tile t+1, A of tile t+PDA -- of this item, or the first tiles of the next
            const bool fetchB = !last || have_next;
            if (last && have_next) point_B(round + 1);
            const int stepB = last ? 0 : t + 1;
            const int ta = t + PDA;
            const bool fetchA = ta < nsteps || have_next;
            if (ta == nsteps && have_next) point_A(round + 1);
            const int stepA = ta < nsteps ? ta : ta - nsteps;
            int slotA = ia + PDA;
            if (slotA >= NA) slotA -= NA;
            // loads this wave may leave in flight at the end of the tile (vmcnt retires in issue order, so the youngest
            // ones are named by their count): with A3 the A request of tile t+2, and in the item's last tile the
            // epilogue's auxiliary rows.  Issue order within the tile: B, then A, then aux.
            const int tail = (A3 && fetchA ? A_PER : 0) + (USE_V2 && AUXN > 0 && last ? AUXN : 0);
            auto wait_tile = [&]() {
                if (tail == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (tail == A_PER) wait_vmcnt<A_PER>();
                else if (tail == AUXN) wait_vmcnt<(AUXN > 0 ? AUXN : 1)>();
                else wait_vmcnt<A_PER + AUXN>();
            };
            // PA_NT_SUBSTEPS 16-wide k-substeps per L / M segment pair (1: 8 barriers per K-tile, 2: 4)
            constexpr int SUB = PA_NT_SUBSTEPS, NPH = 4 / SUB;
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                // ---------------- L segment ----------------
                typename Frag<T>::type fa[SUB][TM], fb[SUB][2];
#pragma unroll
                for (int u = 0; u < SUB; ++u) {
                    const int coff = (((ph * SUB + u) * 2 + half) ^ rsw) << 4;
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[u][i] = *(const typename Frag<T>::type*)(sA + offA + i * 32 * 128 + coff);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[u][j] = *(const typename Frag<T>::type*)(sB + offB + j * 32 * 128 + coff);
                }
                // the next K-tile(s) are requested at least one full segment pair before they are waited for
                if constexpr (A3) {
                    if (ph == 0 && fetchB) dmaB(ib ^ 1, stepB);
                    if (ph == (NPH == 4 ? 1 : 0) && fetchA) dmaA(slotA, stepA);
                } else {
                    if (ph == 0 && fetchA) dmaA(slotA, stepA);
                    if (ph == (NPH == 4 ? 1 : 0) && fetchB) dmaB(ib ^ 1, stepB);
                }
                // last K-tile of the item: request the first auxiliary rows of its epilogue now, AFTER this tile's LDS-DMA:
                // they land under the remaining MFMAs
                if constexpr (USE_V2 && AUXN > 0) {
                    if (last && ph == NPH / 2) aux.issue(a, cur_m0, cur_n0, wr, wc, lane);
                }
                if (ph == NPH - 1 && wr == 1) wait_tile();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- M segment ----------------
                __builtin_amdgcn_s_setprio(1);
                // (r02, measured and rejected: hitting the closing barrier 1..3 MFMAs EARLY, so that the barrier's ~100-cycle
                // resolution overlaps the tail of the M segment, is 3..7 % SLOWER -- the tail MFMAs then share the SIMD's matrix
                // pipe with the other group's first ones: profiles/r02_kloop_experiments.json)
                // TR: the weight fragment is the MFMA's A operand: the accumulator holds the transposed block (lane = token)
                if (ph == 0 && t == 0) {              // uniform: first substep of the item starts the accumulation chains
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) mma32_first<T>(acc[i][j], TR ? fb[0][j] : fa[0][i], TR ? fa[0][i] : fb[0][j]);
#pragma unroll
                    for (int u = 1; u < SUB; ++u)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], TR ? fb[u][j] : fa[u][i], TR ? fa[u][i] : fb[u][j]);
                } else {
#pragma unroll
                    for (int u = 0; u < SUB; ++u)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], TR ? fb[u][j] : fa[u][i], TR ? fa[u][i] : fb[u][j]);
                }
                __builtin_amdgcn_s_setprio(0);
                if (ph == NPH - 1 && wr == 0) wait_tile();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            if (++ia == NA) ia = 0;
            ib ^= 1;
            PA_PROBE_STAMP(round < 24 && t < 13, round * 16 + 1 + t);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();         // re-align the two groups: every fragment read is done
        // slabs go into the ring slots that were read last (ia / ib now name the ones holding the next item's tile 0)
        const int freeA = ia == 0 ? NA - 1 : ia - 1, freeB = ib ^ 1;
        char* slab_c;
        if constexpr (A3) {
            slab_c = wave < G::SLABS_A ? smem + G::a_off(freeA) + wave * 8192
                     : wave < G::SLABS_A + G::SLABS_B ? smem + G::b_off(freeB) + (wave - G::SLABS_A) * 8192
                                                      : smem + G::RING + (wave - G::SLABS_A - G::SLABS_B) * 8192;
        } else {
            slab_c = wave < G::SLABS_IN_STAGE ? smem + G::a_off(freeA) + wave * SLAB_BYTES
                                              : smem + G::RING + (wave - G::SLABS_IN_STAGE) * SLAB_BYTES;
        }
        float* slab = (float*)slab_c;
        if (PA_PROBE_FLAG(a, 1)) {             // probe: no epilogue at all (accumulators kept live)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
        } else if constexpr (USE_V2) {
            // straight-line epilogue; matrix edges are handled by the buffer descriptors (launch_gemm_stagger keeps the
            // row-remapped patch-embedding form and matrices >= 2 GiB away from this kernel)
            const float* brow = HAS_BIAS && a.bias ? (const float*)(smem + G::BIAS_OFF) : nullptr;
            if constexpr (TR) {
                (void)slab;
                if constexpr (EPI == PA_EPI_RESID) gemm_epilogue_v3_resid<TM>(a, acc, brow, cur_m0, cur_n0, wr, wc, lane, aux);
                else gemm_epilogue_v3_bf16<EPI, TM>(a, acc, brow, cur_m0, cur_n0, wr, wc, lane, aux);
            } else if constexpr (EPI == PA_EPI_RESID) gemm_epilogue_v2_resid<TM>(a, acc, (char*)slab, brow, cur_m0, cur_n0, wr, wc, lane, aux);
            else gemm_epilogue_v2_bf16<EPI, TM, BLK>(a, acc, (char*)slab, brow, cur_m0, cur_n0, wr, wc, lane, (cur_m0 / TBM) * 2 + wr, aux);
        } else if constexpr (EPI == PA_EPI_RESID || EPI == PA_EPI_PARTIAL) {
            (void)slab;
            gemm_epilogue_f32_direct<EPI, TM>(a, acc, HAS_BIAS && a.bias ? (const float*)(smem + G::BIAS_OFF) : nullptr, cur_m0,
                                              cur_n0, cur_split, wr, wc, lane);
        } else {
            float bias8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
            if constexpr (HAS_BIAS) {
                if (a.bias) {
                    const float* brow = (const float*)(smem + G::BIAS_OFF) + wc * 64 + (lane & 7) * 8;
                    const f32x4 lo = *(const f32x4*)brow, hi = *(const f32x4*)(brow + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bias8[e] = lo[e]; bias8[4 + e] = hi[e]; }
                }
            }
            gemm_epilogue<T, EPI, TM>(a, acc, slab, cur_m0, cur_n0, cur_split, wr, wc, lane, bias8, (cur_m0 / TBM) * 2 + wr);
        }
        PA_PROBE_STAMP(round < 24, round * 16 + 15);
#ifdef PA_PROBE
        if (!have_next && g_probe_buf && blockIdx.x < 8) {
            __syncthreads();
            for (int i = tid; i < 2 * PROBE_SLOTS; i += 512)
                g_probe_buf[blockIdx.x * 2 * PROBE_SLOTS + i] = ((const unsigned long long*)(smem + G::LDS))[i];
        }
#endif
        if (!have_next) break;
        // (re)derive the DMA cursors of the item just started: cheaper than carrying them through the epilogue
        ++round;
        nsteps = item_of(round, m0, n0, split);
        point_A(round);
        point_B(round);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // slabs are dead: the slots may be refilled by the DMA
    }
}

// The LDS-free epilogues are OPT-IN (PA_EPILOGUE_V3=1 in the environment, or PA_GEMM_EPILOGUE_V3 on a call): bit-identical to
// v2 and 25-40 % fewer epilogue instructions, but measured SLOWER in the step on MI355X (profiles/r04_epilogue_v3_ab.txt:
// store 94.4 -> 99.0 us, gelu 172.6 -> 185.8, resid 106.0 -> 123.6, step 22.84 -> 23.76 ms): with two lanes per row a store
// instruction covers 32 rows x 32 bytes, i.e. 32 partial-line write requests where v2's transposed rows give 8 full 128-byte
// lines -- the L1 -> L2 request rate, not the instruction stream, is what the v2 transposition buys.
static bool epilogue_v3_enabled() {
    static const bool on = [] { const char* e = getenv("PA_EPILOGUE_V3"); return e && atoi(e) != 0; }();
    return on;
}

template <typename T, int EPI, int TM, bool A3 = false, bool BLK = false, bool TR = false>
static int launch_gemm_stagger(const pa_gemm_args& a, hipStream_t st) {
    using G = StaggerGeom<TM, A3>;
    // the LDS-free epilogue (transposed accumulators) wherever it applies: bf16 operands, a fused epilogue, row-major
    // pre-activation, no column sums asked of the DGELU epilogue (they are lane-local only in the v2 orientation)
    // (the 256-row RESID tile keeps v2: 128 accumulators + 32 residual registers in flight spill in the v3 form)
    if constexpr (!TR && !BLK && PA_EPILOGUE_V2 && sizeof(T) == 2 && EPI != PA_EPI_PARTIAL && !(EPI == PA_EPI_RESID && TM == 4)) {
        if ((epilogue_v3_enabled() || (a.reserved & PA_GEMM_EPILOGUE_V3)) && !(EPI == PA_EPI_DGELU && a.colsum_out) &&
            !(EPI == PA_EPI_RESID && a.row_mod > 0))
            return launch_gemm_stagger<T, EPI, TM, A3, false, true>(a, st);
    }
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    const int tiles_m = (int)cdiv(a.M, 64 * TM), tiles_n = (int)cdiv(a.N, 256);
    const int nwg = tiles_m * tiles_n;
    const int ksteps = (int)((int64_t)a.K * sizeof(T) / KB);
    const int splits = EPI == PA_EPI_PARTIAL ? a.split_k : 1;
    const int per = (int)cdiv(ksteps, splits);
    const int total = nwg * splits;
    // an empty K split, or more rounds than the item table holds: the generic kernel handles it.  So does it handle what
    // the buffer-descriptor epilogue (v2) does not cover: the row-remapped residual form (patch embedding) and
    // matrices whose rows do not all lie within 2 GiB of the first one
    const int64_t lim = (int64_t)1 << 31;
    const bool big = (int64_t)a.M * a.ldolp * 2 >= lim || (int64_t)a.M * a.ldolp2 * 2 >= lim || (int64_t)a.M * a.ldaux * 2 >= lim ||
                     (int64_t)a.M * a.ldr * 4 >= lim || (int64_t)a.M * a.ldo32 * 4 >= lim;
    if (ksteps < 1 || (int64_t)(splits - 1) * per >= ksteps ||
        (PA_EPILOGUE_V2 && sizeof(T) == 2 && EPI != PA_EPI_PARTIAL && (big || (EPI == PA_EPI_RESID && a.row_mod > 0))))
        return BLK ? PA_EUNSUPPORTED : launch_gemm_v<T, EPI, 2, 2, 2, 2>(a, st);     // (the generic kernel has no blocked form)
    if constexpr (A3) {      // two K-tiles of lead need two K-tiles in every item; the short item table bounds the rounds
        if (ksteps - (splits - 1) * per < 2 || per < 2 || cdiv(total, 256) > G::MAX_ROUNDS) return launch_gemm_stagger<T, EPI, TM, false, BLK, TR>(a, st);
    } else {
        if (cdiv(total, 256) > G::MAX_ROUNDS) return BLK ? PA_EUNSUPPORTED : launch_gemm_v<T, EPI, 2, 2, 2, 2>(a, st);
    }
#ifdef PA_PROBE
    constexpr int LDS_BYTES = G::LDS + 2 * PROBE_SLOTS * 8;
#else
    constexpr int LDS_BYTES = G::LDS;
#endif
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget (probe build)");
    static signed char lds_attr[64] = {0};
    (void)lds_attr_on_this_device((const void*)gemm_nt_stagger_kernel<T, EPI, TM, A3, BLK, TR>, LDS_BYTES, lds_attr);
    // PA_GEMM_NO_PERSIST (a.reserved) / PA_NT_PERSISTENT=0: one work item per workgroup, handed out by the hardware as CUs free
    // up, instead of 256 resident workgroups walking their item lists.  The persistent form assumes it owns every CU: when
    // another kernel (an RCCL all-reduce on the communication stream) holds R of them, R of its workgroups only start after the
    // others have finished ALL their items and the launch takes twice as long; one item per workgroup degrades by R / 256.
    static const bool persist_env = [] { const char* e = getenv("PA_NT_PERSISTENT"); return !e || atoi(e) != 0; }();
    const bool persistent = persist_env && !(a.reserved & PA_GEMM_NO_PERSIST);
    hipLaunchKernelGGL((gemm_nt_stagger_kernel<T, EPI, TM, A3, BLK, TR>), dim3(persistent ? std::min(total, 256) : total), dim3(512), LDS_BYTES, st, a,
                       tiles_m, tiles_n, nwg, per, total);
    const int rc = check_launch();
    if (rc == PA_OK && EPI == PA_EPI_DGELU && a.colsum_out) return finish_gemm_colsum(a, tiles_m * 2, st);
    return rc;
}

// Variant choice when pa_gemm_args.tune == 0: minimise  rounds x tile_time  with
//   rounds    = ceil(#tiles / resident workgroups)     (tile quantisation on 256 CUs)
//   tile_time = tile area / steady-state rate of the schedule (relative rates measured on MI355X at the
//               passt_s shapes, profiles/r01_*: the role-split kernels reach ~1.25-1.4 PF/s on long K but pay
//               an un-overlapped prologue+epilogue per tile, which at K = 768 cancels their advantage)
static int pick_nt_variant(int M, int N, int K) {
    struct Cand { int id, bm, bn, slots; float rate_short, rate_long; };
    // The rates are RELATIVE weights tuned on the training step (rounds 1-2), not isolated throughputs.  Round 6 re-measured every
    // schedule ALONE at M = 30 336 (profiles/r06_nt_variants.txt: K <= 1024 -> {1: 700, 8: 800, 7: 950, 6: 930} TF/s, K > 1024 ->
    // {1: 870, 8: 950, 7: 1075, 6: 900}, 7 / 8 in their A3 form) and tried that table (PA_NT_RATES=r06): it moves the MLP-epilogue GEMMs
    // of config #2 and the long-K GEMMs of config #4 from the 256-row to the 192-row tile and the small GEMMs of ESC-50 from the
    // generic to the role-split kernel, and every configuration got SLOWER in the step -- config #2 22.31 -> 22.45 ms, config #4
    // 66.2 -> 67.4, ESC-50 5.83 -> 5.92 (ABBA, 60-step lines, profiles/r06_pick_table_ab.txt): alone, operands come out of the caches
    // and the epilogue's stores meet an idle memory system; in the step they do not.  The step-tuned table stays.
    static const Cand r06[] = {
        {1, 128, 128, 512, 700.f, 870.f},
        {8, 128, 256, 256, 800.f, 950.f},
        {7, 192, 256, 256, 950.f, 1075.f},
        {6, 256, 256, 256, 930.f, 900.f},
    };
    static const Cand r01[] = {
        {1, 128, 128, 512, 750.f, 930.f},
        {8, 128, 256, 256, 650.f, 1000.f},
        {7, 192, 256, 256, 780.f, 1250.f},
        {6, 256, 256, 256, 800.f, 1150.f},
    };
    static const bool isolated_table = [] { const char* e = getenv("PA_NT_RATES"); return e && !strcmp(e, "r06"); }();
    const Cand (&cands)[4] = isolated_table ? r06 : r01;
    int best = 1;
    double best_cost = 1e30;
    for (const Cand& c : cands) {
        const int64_t tiles = cdiv(M, c.bm) * cdiv(N, c.bn);
        const double rounds = (double)cdiv(tiles, c.slots);
        const double rate = (K <= 1024 ? c.rate_short : c.rate_long) / c.slots;   // per resident workgroup
        const double cost = rounds * (double)c.bm * c.bn / rate;
        if (cost < best_cost) { best_cost = cost; best = c.id; }
    }
    return best;
}

// a.tune selects the tile/pipeline variant (0 = heuristic above); see include/passt_amd.h
template <typename T, int EPI>
static int launch_gemm(const pa_gemm_args& a, hipStream_t st) {
    if constexpr (sizeof(T) == 4) return launch_gemm_v<T, EPI, 2, 2, 2, 2>(a, st);   // parity mode: one variant
    else {
        int v = a.tune ? a.tune : pick_nt_variant(a.M, a.N, a.K);
        // the 192- / 128-row role-split tiles run with the A operand two K-tiles ahead (A3) unless PA_NT_A3=0 (A/B knob)
        static const bool a3 = [] { const char* e = getenv("PA_NT_A3"); return !e || atoi(e) != 0; }();
        if (!a.tune && a3 && EPI != PA_EPI_PARTIAL && (v == 7 || v == 8)) v += 10;
        if constexpr (EPI == PA_EPI_GELU || EPI == PA_EPI_DGELU) {
            if (a.reserved & PA_GEMM_BLOCKED_PRE) {      // blocked pre-activation: the kernels with epilogue v2 (pa_gemm_blocked_pre_ok)
                if (a.N % 64 || blocked_pre_rows(a.M) * a.N * 2 >= ((int64_t)1 << 31)) return PA_EUNSUPPORTED;
                // r06: the 192 x 128 tile that fits TWO workgroups per CU -- one workgroup's transposition + store
                // burst under the other's K loop -- has the blocked pre-activation too (tune 3 / 9, epilogue v2).  In the step at config
                // #2 (M = 30 336, N = 3072, K = 768; ABBA, 60-step lines, profiles/r06_gemm_variants_epi13.txt) it was 1-2 % FASTER than
                // the 256-row role-split tile on one box (fc1 + GELU 165.2 -> 163.4 us, GELU' 177.6 -> 173.6, step -0.2 %) and 2 %
                // SLOWER on another (170.5 -> 173.6, 183.3 -> 187.8, step +0.4 %); slower at ESC-50's M = 4 236 (33 -> 39.5 us) and
                // for fc1 + GELU at config #4's K = 1024: a wash, so it is opt-in (PA_NT_MLP_2WG=1), not the default.
                static const bool mlp_2wg = [] { const char* e = getenv("PA_NT_MLP_2WG"); return e && atoi(e) != 0; }();
                if (!a.tune && mlp_2wg && a.M >= 16384 && a.N >= 1024 && a.K <= (EPI == PA_EPI_DGELU ? 1024 : 768)) v = 3;
                switch (v) {
                    case 6: return launch_gemm_stagger<T, EPI, 4, false, true>(a, st);
                    case 7: return launch_gemm_stagger<T, EPI, 3, false, true>(a, st);
                    case 8: return launch_gemm_stagger<T, EPI, 2, false, true>(a, st);
                    case 17: return launch_gemm_stagger<T, EPI, 3, true, true>(a, st);
                    case 18: return launch_gemm_stagger<T, EPI, 2, true, true>(a, st);
                    // r06: the two-workgroups-per-CU tiles with the v2 epilogue (explicit tune only; profiles/r06_gemm_variants_epi13.txt)
                    case 3: return launch_gemm_v<T, EPI, 2, 2, 3, 2, KB, 2>(a, st);       // 192x128, 4 waves, 80 KiB
                    case 9: return launch_gemm_v<T, EPI, 1, 4, 4, 3, 64, 2>(a, st);       // 128x256, 4 waves, 72 KiB
                }
                return PA_EUNSUPPORTED;
            }
        }
        switch (v) {
            case 1: return launch_gemm_v<T, EPI, 2, 2, 2, 2>(a, st);   // 128x128, 4 waves (64x64 each), 2-stage
            case 2: return launch_gemm_v<T, EPI, 2, 4, 4, 2>(a, st);   // 256x256, 8 waves (128x64 each), lockstep
            case 3: return launch_gemm_v<T, EPI, 2, 2, 3, 2>(a, st);   // 192x128, 4 waves (96x64 each), 80 KiB: 2 workgroups / CU
            case 9: return launch_gemm_v<T, EPI, 1, 4, 4, 3, 64>(a, st);   // 128x256, 4 waves (128x64 each), 3 x 24 KiB stages: 2 workgroups / CU
            // r06: the two-workgroups-per-CU tiles with epilogue v2 (row-major outputs)
            case 13: if constexpr (EPI != PA_EPI_PARTIAL) return launch_gemm_v<T, EPI, 2, 2, 3, 2, KB, 1>(a, st); else return PA_EINVAL;
            case 19: if constexpr (EPI != PA_EPI_PARTIAL) return launch_gemm_v<T, EPI, 1, 4, 4, 3, 64, 1>(a, st); else return PA_EINVAL;
            case 6: return launch_gemm_stagger<T, EPI, 4>(a, st);      // 256x256 role-split schedule (8 waves)
            case 7: return launch_gemm_stagger<T, EPI, 3>(a, st);      // 192x256 role-split
            case 8: return launch_gemm_stagger<T, EPI, 2>(a, st);      // 128x256 role-split
            case 17: return launch_gemm_stagger<T, EPI, 3, true>(a, st);   // 192x256 role-split, A two K-tiles ahead (3 A slots)
            case 18: return launch_gemm_stagger<T, EPI, 2, true>(a, st);   // 128x256 role-split, A two K-tiles ahead
        }
        return PA_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------
// TN GEMM for weight gradients:  C[N][K] = sum_m A[m][N]^T B[m][K]  (A = dY, B = X, both token-major,
// read IN PLACE -- no transposed copies).  The reduction index m runs down LDS tile ROWS, so both MFMA
// operands are column fragments: ds_read_b64_tr_b16 (bf16) / ds_read_b32 (f32).  Tiles are
// [MROWS tokens][128 columns], 16 KiB per operand per stage, global_load_lds double buffered; bf16 rows
// (256 B = one full bank line) are XOR-swizzled by (row&3) so the 4 rows of a transpose read hit 4
// different 64-byte bank quarters.  Split-K over tokens writes f32 partial slabs (EPI_PARTIAL).
// ------------------------------------------------------------------------------------------------
template <typename T> struct TNGeom {
    static constexpr int COLS = 128;
    static constexpr int ROWB = COLS * (int)sizeof(T);     // 256 / 512
    static constexpr int MROWS = TILE_BYTES / ROWB;        // 64 / 32 tokens per stage
    static constexpr int RPI = 1024 / ROWB;                // tile rows filled per wave-instruction: 4 / 2
    static constexpr int LPR = ROWB / 16;                  // lanes (16-byte chunks) per row: 16 / 32
    static constexpr int EPC = 16 / (int)sizeof(T);
    static constexpr int FSTEP = Frag<T>::K;               // tokens per fragment step: 16 / 8
};

template <typename T> __device__ __forceinline__ int tn_swz(int row, int c) {
    if constexpr (sizeof(T) == 2) return row * 256 + ((c ^ ((row & 3) << 2)) << 4);
    else return row * 512 + (c << 4);
}

// column fragment for tokens [ms*FSTEP, +FSTEP) at columns cbase + (lane&31)
template <typename T>
__device__ __forceinline__ typename Frag<T>::type tn_frag(const char* tile, int ms, int cbase, int lane);
template <>
__device__ __forceinline__ bf16x8 tn_frag<bf16>(const char* tile, int ms, int cbase, int lane) {
    const int p = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int r1 = ms * 16 + h * 8 + (p >> 2);
    const int col = cbase + g * 16 + (p & 3) * 4;
    const int within = (col & 7) * 2;
    const bf16x4 lo = lds_tr16(tile + tn_swz<bf16>(r1, col >> 3) + within);
    const bf16x4 hi = lds_tr16(tile + tn_swz<bf16>(r1 + 4, col >> 3) + within);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
template <>
__device__ __forceinline__ f32x4 tn_frag<float>(const char* tile, int ms, int cbase, int lane) {
    const int col = cbase + (lane & 31), h = lane >> 5;
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = *(const float*)(tile + (ms * 8 + h * 4 + e) * 512 + col * 4);
    return f;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const pa_gemm_args a, const int tiles_n, const int nwg,
                                                      const int steps_per_split) {
    using G = TNGeom<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int bid = xcd_swizzle(blockIdx.x, nwg);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;   // output rows (A columns) / output cols (B columns)
    const int Mtok = a.K;
    const int steps_total = (Mtok + G::MROWS - 1) / G::MROWS;
    const int st_begin = blockIdx.y * steps_per_split;
    const int st_end = min(steps_total, st_begin + steps_per_split);
    const int nsteps = st_end - st_begin;

    // per-lane sources of this wave's 4 + 4 copies per stage
    const char* srcA[4];
    const char* srcB[4];
    int rowin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave * 4 + i;
        const int row = q * G::RPI + lane / G::LPR;
        const int pc = lane % G::LPR;
        const int c = sizeof(T) == 2 ? (pc ^ ((row & 3) << 2)) : pc;
        rowin[i] = row;
        // column chunks beyond the matrix are clamped (their outputs are never stored)
        const int ca = min(m0 + c * G::EPC, a.M - G::EPC), cb = min(n0 + c * G::EPC, a.N - G::EPC);
        srcA[i] = (const char*)a.A + (int64_t)ca * sizeof(T);
        srcB[i] = (const char*)a.B + (int64_t)cb * sizeof(T);
    }
    auto stage = [&](int buf, int step) {
        char* sA = smem + buf * (2 * TILE_BYTES) + wave * 4096;
        char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t tok = min((int64_t)(st_begin + step) * G::MROWS + rowin[i], (int64_t)Mtok - 1);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(srcA[i] + tok * a.lda * sizeof(T)),
                (__attribute__((address_space(3))) void*)(sA + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(srcB[i] + tok * a.ldb * sizeof(T)),
                (__attribute__((address_space(3))) void*)(sB + i * 1024), 16, 0, 0);
        }
    };
    // rows of the LAST stage beyond Mtok were filled with a clamped (duplicate) token: zero them, each wave
    // the rows it staged itself, after its copies landed and before the barrier
    auto zero_tail = [&](int buf, int step) {
        const int valid = Mtok - (st_begin + step) * G::MROWS;      // valid rows in this stage
        if (valid >= G::MROWS) return;
        char* sA = smem + buf * (2 * TILE_BYTES);
        char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (rowin[i] >= valid) {
                const int off = (wave * 4 + i) * 1024 + lane * 16;
                *(uint4*)(sA + off) = make_uint4(0, 0, 0, 0);
                *(uint4*)(sB + off) = make_uint4(0, 0, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nsteps > 0) stage(0, 0);
    for (int t = 0; t < nsteps; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        zero_tail(t & 1, t);
        __syncthreads();
        if (t + 1 < nsteps) stage((t + 1) & 1, t + 1);
        const char* sA = smem + (t & 1) * (2 * TILE_BYTES);
        const char* sB = sA + TILE_BYTES;
#pragma unroll
        for (int ms = 0; ms < G::MROWS / G::FSTEP; ++ms) {
            typename Frag<T>::type fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = tn_frag<T>(sA, ms, wr * 64 + i * 32, lane);
                fb[i] = tn_frag<T>(sB, ms, wc * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma32<T>(acc[i][j], fa[i], fb[j]);
        }
    }
    __syncthreads();
    const float nobias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    gemm_epilogue<T, PA_EPI_PARTIAL, 2>(a, acc, (float*)(smem + wave * SLAB_BYTES), m0, n0, blockIdx.y, wr, wc, lane, nobias);
}

// ------------------------------------------------------------------------------------------------
// Role-split TN kernel (bf16): 256 x 256 output tile of dW, 8 waves = 2 groups x 4, wave tile 128 x 64, the
// token axis streams through two 64 KiB LDS stages ([64 tokens][256 columns] of dY and of X).  Same
// L / M segment structure and one-barrier stagger as gemm_nt_stagger_kernel; fragments are transpose reads.
// The long reduction (tokens / split_k) amortises prologue and epilogue, so this kernel runs at the
// steady-state rate of the schedule.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn2_swz(int row, int c) { return row * 512 + ((c ^ ((row & 3) << 2)) << 4); }

__device__ __forceinline__ bf16x8 tn2_frag(const char* tile, int ms, int cbase, int lane) {
    const int p = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
    const int r1 = ms * 16 + h * 8 + (p >> 2);
    const int col = cbase + g * 16 + (p & 3) * 4;
    const int within = (col & 7) * 2;
    const bf16x4 lo = lds_tr16(tile + tn2_swz(r1, col >> 3) + within);
    const bf16x4 hi = lds_tr16(tile + tn2_swz(r1 + 4, col >> 3) + within);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// one work item: output tile `tile_id` (already XCD-remapped) of problem `a`, K slice `split`
// r02: THREE stages of 48 tokens (144 KiB) instead of two of 64, every stage requested TWO stages ahead: both operands
// are activations that come from HBM in the training step, and one stage of lead (~1.4 us) is less than a loaded HBM
// round trip (the same effect cost the K = 2304 / 3072 NT GEMMs 15 %: profiles/r02_instep_vs_isolated.json).
static constexpr int TN_ROWS = PA_TN_STEP_ROWS;          // tokens per stage (host code sizes the K slices in these units)
#ifndef PA_TN_FRAG_PIPE
// Three schedules of the same work were measured in the step, same box, same call (profiles/r02_tn_loop_variants.json):
//   0: role split, fragment reads in the L segment                                   355.9 us per block launch, 763 k cycles
//   1: role split, fragment reads issued inside the previous M segment               360.9 us, 771 k
//   2: lockstep, software pipelined, counted lgkmcnt waits, 2 barriers per stage     368.0 us, 801 k   (all at 2.03 GHz)
// although the probe timeline (tools/probe_tn.py) shows variant 0 spending as long in L as in M.  What fills L is apparently not the
// twelve LDS reads (~300 cycles of loaded latency, hidden in 1 and 2) but the ISSUE of the LDS-DMA pieces (100-185 cycles
// each inside a busy phase, MI355X_MICROARCH.md; two per wave and phase): the role split keeps it out of the instruction
// stream of the wave that is issuing MFMAs, a lockstep loop puts it back in.
#define PA_TN_FRAG_PIPE 0
#endif
#ifndef PA_TN_LOCKSTEP_OFFSET
#define PA_TN_LOCKSTEP_OFFSET 0   // variant 2 only: s_sleep units (64 cycles) group 1 waits after every stage barrier (A/B:
                                  // 0 / 2 / 4 -> 359.8 / 371.6 / 376.6 us against 353.2 for the role split, run r10)
#endif
#ifndef PA_TN_STAGES
#define PA_TN_STAGES 3         // (A/B: -DPA_TN_STEP_ROWS=64 -DPA_TN_STAGES=2 is the r01 pipeline)
#endif
static constexpr int TN_STAGES = PA_TN_STAGES;
static constexpr int TN_OP_BYTES = TN_ROWS * 512, TN_STAGE_BYTES = 2 * TN_OP_BYTES;     // 24 KiB per operand, 48 KiB per stage
static constexpr int TN_LDS = TN_STAGES * TN_STAGE_BYTES;                               // 144 KiB
static_assert(TN_ROWS % 16 == 0 && TN_ROWS * 512 % (8 * 1024) == 0, "whole 16-token phases, whole 1 KiB pieces per wave");

// CSUM (tiles of X-column block 0 of a problem with colsum_ws set): every wave also sums ONE of its four dY fragments
// over the tokens with a ninth MFMA per phase against a fragment of ones -- wave (wr, wc) owns dY columns
// wr*128 + wc*32 + [0,32), so the eight waves cover the tile's 256 dY columns once -- and the slice's column sums of dY
// (= the partial bias gradient) leave with the partial slab.  Replaces a separate pass over dY (qkv.bias: 186 MB read
// per block) by +1/8 MFMA work in 1/12 of the work items.
// (CSUM is a wave-uniform run-time flag, not a template parameter: a second copy of the item would double a kernel that
// is already ~50 KB of code, and the instruction cache is 64 KB per two CUs.)
// One work item: tile `tile_id` of problem `a` over the token steps [st_begin, st_begin + nsteps); the f32 partial tile goes
// to out[m * ldo + n] (m, n = coordinates in the whole gradient matrix), the column sums of dY (CSUM) to cs_out[m].
__device__ __forceinline__ void gemm_tn_stagger_range(const pa_gemm_args& a, const int tiles_n, const int tile_id, const int st_begin,
                                                      const int nsteps, float* out, const int ldo, float* cs_out) {
    const bool CSUM = cs_out != nullptr && tile_id % tiles_n == 0;
    constexpr int TM = 4, WN = 4, MROWS = TN_ROWS, NPH = MROWS / 16;
    constexpr int OP_BYTES = TN_OP_BYTES, STAGE_BYTES = TN_STAGE_BYTES;
    constexpr int PER = OP_BYTES / 1024 / 8;               // LDS-DMA pieces per wave, operand and stage: 3
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int bid = tile_id;
    const int m0 = (bid / tiles_n) * 256, n0 = (bid % tiles_n) * 256;     // dY columns / X columns of this tile
    const int Mtok = a.K;

    // this wave's PER + PER LDS-DMA pieces per stage: piece q = wave*PER+i covers tile rows 2q, 2q+1 (512 B each).
    // Source address = uniform base of the stage (SGPRs, advanced by MROWS token rows per stage) + a per-lane
    // 32-bit offset fixed for the whole kernel: no 64-bit vector arithmetic in the K loop.
    uint32_t voffA[PER], voffB[PER];
    int rowin[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int q = wave * PER + i;
        const int row = q * 2 + (lane >> 5);
        const int pc = lane & 31;
        const int c = pc ^ ((row & 3) << 2);
        rowin[i] = row;
        voffA[i] = (uint32_t)row * (uint32_t)a.lda * 2u + (uint32_t)min(m0 + c * 8, a.M - 8) * 2u;
        voffB[i] = (uint32_t)row * (uint32_t)a.ldb * 2u + (uint32_t)min(n0 + c * 8, a.N - 8) * 2u;
    }
    const char* baseA = (const char*)a.A + (int64_t)st_begin * MROWS * a.lda * 2;
    const char* baseB = (const char*)a.B + (int64_t)st_begin * MROWS * a.ldb * 2;
    auto dma = [&](const bool opB, const char* base, int ld, const uint32_t (&voff)[PER], char* dst, int step) __attribute__((always_inline)) {
        const char* sb = base + (int64_t)step * MROWS * ld * 2;              // uniform
        const int valid = Mtok - (st_begin + step) * MROWS;                  // token rows that exist in this stage
        if (valid >= MROWS) {
            if (opB) {         // (a constant at both call sites: folded after inlining)
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voff[i]),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, PA_AUX_DMA_TN_B);
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + voff[i]),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, PA_AUX_DMA_TN_A);
            }
        } else {   // last stage of the last split: rows past the end re-read the last token (zeroed by zero_tail)
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const uint32_t back = (uint32_t)max(rowin[i] - (valid - 1), 0) * (uint32_t)ld * 2u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + (voff[i] - back)),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
    };
    auto dmaA = [&](int buf, int step) { dma(false, baseA, a.lda, voffA, smem + buf * STAGE_BYTES + wave * (PER * 1024), step); };
    auto dmaB = [&](int buf, int step) { dma(true, baseB, a.ldb, voffB, smem + buf * STAGE_BYTES + OP_BYTES + wave * (PER * 1024), step); };
    // token rows beyond Mtok (last stage only) were filled from a clamped row: zero what THIS wave staged,
    // after its DMA landed and before the barrier that publishes the stage
    auto zero_tail = [&](int buf, int step) {
        const int valid = Mtok - (st_begin + step) * MROWS;
        if (valid >= MROWS) return;
        char* sA = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (rowin[i] >= valid) {
                const int off = (wave * PER + i) * 1024 + lane * 16;
                *(uint4*)(sA + off) = make_uint4(0, 0, 0, 0);
                *(uint4*)(sA + OP_BYTES + off) = make_uint4(0, 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // partial stage only: the zeros are in LDS before the barrier
    };

    f32x16 acc[TM][2];                 // started by the first token step's MFMAs (C = 0), cleared only for an empty slice
    if (nsteps <= 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    f32x16 accb;                       // CSUM only
    bf16x8 ones;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    if (nsteps > 0) { dmaA(0, 0); dmaB(0, 0); }
    if (TN_STAGES > 2 && nsteps > 1) { dmaA(1, 1); dmaB(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nsteps > 0) zero_tail(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if PA_TN_FRAG_PIPE != 2
    if (wr == 1) __builtin_amdgcn_s_barrier();             // group 1 runs one barrier behind
#endif

    // per-lane fragment addresses inside a stage (see tn2_frag: the +4-row partner is +2048 bytes, a 16-token
    // phase +8192, and the swizzle term only depends on the lane)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t offA[TM], offB[2], offCS;
    {
        const int p = lane & 15, g = (lane >> 4) & 1, h = lane >> 5;
        const int r1 = h * 8 + (p >> 2);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int col = wr * 128 + i * 32 + g * 16 + (p & 3) * 4;
            offA[i] = lds0 + tn2_swz(r1, col >> 3) + (col & 7) * 2;
        }
        {
            const int col = wr * 128 + wc * 32 + g * 16 + (p & 3) * 4;
            offCS = lds0 + tn2_swz(r1, col >> 3) + (col & 7) * 2;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = wc * 64 + j * 32 + g * 16 + (p & 3) * 4;
            offB[j] = lds0 + OP_BYTES + tn2_swz(r1, col >> 3) + (col & 7) * 2;
        }
    }
    auto join = [](bf16x4 lo, bf16x4 hi) {
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

#if PA_TN_FRAG_PIPE == 2
    // Software-pipelined lockstep loop (r02).  The probe timeline of the role-split loop (tools/probe_tn.py) showed the L
    // segment as long as the M segment: a loaded ds_read_b64_tr_b16 takes ~300 cycles to return and L ended with a wait for
    // all twelve, so every barrier interval carried that latency; issuing the reads inside the previous M segment only
    // moved the wait.  Here all eight waves run the same schedule:
    //   * the fragments of phase p+1 are requested inside phase p, each one right after the last MFMA that reads its old
    //     value has been issued (operands are read at issue, LDS data returns >= 64 cycles later): no second register set;
    //   * MFMAs go column-major -- (0,0) (1,0) (2,0) (3,0) | fb0' | (0,1) fa0' (1,1) fa1' (2,1) fa2' (3,1) fa3' fb1' -- and
    //     every MFMA waits only for ITS operands with a counted s_waitcnt lgkmcnt (LDS returns in issue order; the twelve
    //     reads outstanding at a phase start are fb0 fa0 fa1 fa2 fa3 fb1): lgkmcnt(8) (6) (4) (2) | (2);
    //   * no barrier per phase: B1 at the end of the second-to-last phase publishes stage t+1 (every wave waited for its
    //     own DMA pieces first) because the last phase requests fragments of stage t+1; B2 at the end of the stage: every
    //     wave has consumed all its reads of stage t's slot, so the DMA of stage t+3, requested in phases 0 / 1 of stage
    //     t+1, may overwrite it.
    // Two barriers per stage instead of six, no LDS drain anywhere, the two waves of a SIMD share the matrix pipe freely.
    static_assert(TN_STAGES == 3, "two stages of DMA lead");
    constexpr int LEAD = TN_STAGES - 1;
    int slot = 0;
    bf16x8 fa[TM], fb[2];
    auto wait_lgkm = [](auto nc) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(decltype(nc)::value) : "memory"); };
    fb[0] = join(lds_tr16_asm<0>(offB[0]), lds_tr16_asm<2048>(offB[0]));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = join(lds_tr16_asm<0>(offA[i]), lds_tr16_asm<2048>(offA[i]));
    fb[1] = join(lds_tr16_asm<0>(offB[1]), lds_tr16_asm<2048>(offB[1]));
    for (int t = 0; t < nsteps; ++t) {
        const uint32_t sb = slot * STAGE_BYTES;
        const bool more = t + 1 < nsteps, more2 = t + LEAD < nsteps;
        const int slot1 = slot == TN_STAGES - 1 ? 0 : slot + 1;
        const int slot2 = slot1 == TN_STAGES - 1 ? 0 : slot1 + 1;
        auto wait_stage = [&]() {      // stage t+1 landed; the 2*PER pieces of stage t+2 (younger) may stay in flight
            if (more2) wait_vmcnt<2 * PER>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (more) zero_tail(slot1, t + 1);
        };
        auto phase = [&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            constexpr int NOFF = ph == NPH - 1 ? 0 : (ph + 1) * 8192;       // next phase: same stage, or phase 0 of stage t+1
            const uint32_t nsb = ph == NPH - 1 ? (uint32_t)slot1 * STAGE_BYTES : sb;
            if (more2) {
                if (ph == 0) dmaA(slot2, t + LEAD);
                if (ph == 1) dmaB(slot2, t + LEAD);
            }
            bf16x8 fcs = ones;       // CSUM items (1 in 12): own fragment of dY row-block wc, and a full LDS drain (simple)
            if (CSUM) {
                fcs = join(lds_tr16_asm<ph * 8192>(offCS + sb), lds_tr16_asm<ph * 8192 + 2048>(offCS + sb));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            auto msteps = [&](auto firstc) {
                constexpr bool FIRST = decltype(firstc)::value;
                auto mm = [&](int i, int j) {
                    if constexpr (FIRST) mma32_first<bf16>(acc[i][j], fa[i], fb[j]); else mma32<bf16>(acc[i][j], fa[i], fb[j]);
                };
                wait_lgkm(std::integral_constant<int, 8>{}); __builtin_amdgcn_sched_barrier(0);
                mm(0, 0); __builtin_amdgcn_sched_barrier(0);
                wait_lgkm(std::integral_constant<int, 6>{}); __builtin_amdgcn_sched_barrier(0);
                mm(1, 0); __builtin_amdgcn_sched_barrier(0);
                wait_lgkm(std::integral_constant<int, 4>{}); __builtin_amdgcn_sched_barrier(0);
                mm(2, 0); __builtin_amdgcn_sched_barrier(0);
                wait_lgkm(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
                mm(3, 0); __builtin_amdgcn_sched_barrier(0);
                fb[0] = join(lds_tr16_asm<NOFF>(offB[0] + nsb), lds_tr16_asm<NOFF + 2048>(offB[0] + nsb));
                wait_lgkm(std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    mm(i, 1); __builtin_amdgcn_sched_barrier(0);
                    fa[i] = join(lds_tr16_asm<NOFF>(offA[i] + nsb), lds_tr16_asm<NOFF + 2048>(offA[i] + nsb));
                    __builtin_amdgcn_sched_barrier(0);
                }
                fb[1] = join(lds_tr16_asm<NOFF>(offB[1] + nsb), lds_tr16_asm<NOFF + 2048>(offB[1] + nsb));
            };
            if (ph == 0 && t == 0) msteps(std::true_type{}); else msteps(std::false_type{});
            if (CSUM) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accb) : "v"(fcs), "v"(ones));
            if (ph == NPH - 2) {           // B1: publish stage t+1 before anyone requests its fragments
                wait_stage();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            if (ph == NPH - 1) {           // B2: every wave is through with its reads of stage t's slot
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
#if PA_TN_LOCKSTEP_OFFSET
                if (wr == 1) __builtin_amdgcn_s_sleep(PA_TN_LOCKSTEP_OFFSET);     // A/B: keep the two waves of a SIMD half a phase apart
#endif
            }
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 3);
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        if constexpr (NPH == 4) phase(std::integral_constant<int, 3>{});
        static_assert(NPH == 3 || NPH == 4, "three or four 16-token phases per stage");
        slot = slot1;
    }
#elif PA_TN_FRAG_PIPE == 1
    // Fragment pipeline (r02): the 12 transpose reads of a phase are issued INSIDE the previous M segment, each fragment
    // right after the last MFMA that reads its old value has been issued (operands are read at issue; the LDS data
    // returns >= 64 cycles later), so they cost no registers and the L segment shrinks to DMA requests + the wait.
    // ds_read_b64_tr_b16 only reaches its rate with several waves issuing (MI355X_MICROARCH.md, LDS): with the reads in
    // the L segment only one wave per SIMD was issuing them and L (~12 reads) was longer than M (8 MFMAs).
    // Consequence: the last phase of a stage reads the NEXT stage, so that stage is waited for / published one barrier
    // earlier than before (group 0: in its last L segment; group 1, one barrier behind: after its second-to-last M).
    static_assert(TN_STAGES == 3, "the fragment pipeline assumes two stages of lead");
    constexpr int LEAD = TN_STAGES - 1;
    int slot = 0;
    bf16x8 fa[TM], fb[2];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = join(lds_tr16_asm<0>(offA[i]), lds_tr16_asm<2048>(offA[i]));
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = join(lds_tr16_asm<0>(offB[j]), lds_tr16_asm<2048>(offB[j]));
    for (int t = 0; t < nsteps; ++t) {
        const uint32_t sb = slot * STAGE_BYTES;
        const bool more = t + 1 < nsteps, more2 = t + LEAD < nsteps;
        const int slot1 = slot == TN_STAGES - 1 ? 0 : slot + 1;
        const int slot2 = slot1 == TN_STAGES - 1 ? 0 : slot1 + 1;
        // stage t+1 (requested during stage t-1) must have landed; the 2*PER pieces of stage t+2, the youngest in this
        // wave's queue, may stay in flight (vmcnt retires in issue order)
        auto wait_stage = [&]() {
            if (more2) wait_vmcnt<2 * PER>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (more) zero_tail(slot1, t + 1);
        };
        auto phase = [&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            constexpr int NOFF = ph == NPH - 1 ? 0 : (ph + 1) * 8192;       // next phase: same stage, or phase 0 of stage t+1
            const uint32_t nsb = ph == NPH - 1 ? (uint32_t)slot1 * STAGE_BYTES : sb;
            // ---------------- L segment: DMA requests, (CSUM: own fragment), wait for the fragments ----------------
            bf16x8 fcs = ones;       // CSUM: this wave's own copy of dY row-block wc (fa[wc] would be a run-time register index)
            if (CSUM) fcs = join(lds_tr16_asm<ph * 8192>(offCS + sb), lds_tr16_asm<ph * 8192 + 2048>(offCS + sb));
            if (more2) {
                if (ph == 0) dmaA(slot2, t + LEAD);
                if (ph == 1) dmaB(slot2, t + LEAD);
            }
            if (ph == NPH - 1 && wr == 0) wait_stage();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 1);
            // ---------------- M segment: MFMAs, each fragment reloaded as soon as it is free ----------------
            __builtin_amdgcn_s_setprio(1);
            auto msteps = [&](auto firstc) {
                constexpr bool FIRST = decltype(firstc)::value;
                auto mm = [&](int i, int j) {
                    if constexpr (FIRST) mma32_first<bf16>(acc[i][j], fa[i], fb[j]); else mma32<bf16>(acc[i][j], fa[i], fb[j]);
                };
#pragma unroll
                for (int i = 0; i < TM - 1; ++i) {
                    mm(i, 0);
                    mm(i, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    fa[i] = join(lds_tr16_asm<NOFF>(offA[i] + nsb), lds_tr16_asm<NOFF + 2048>(offA[i] + nsb));
                    __builtin_amdgcn_sched_barrier(0);
                }
                mm(TM - 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                fb[0] = join(lds_tr16_asm<NOFF>(offB[0] + nsb), lds_tr16_asm<NOFF + 2048>(offB[0] + nsb));
                __builtin_amdgcn_sched_barrier(0);
                mm(TM - 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                fa[TM - 1] = join(lds_tr16_asm<NOFF>(offA[TM - 1] + nsb), lds_tr16_asm<NOFF + 2048>(offA[TM - 1] + nsb));
                fb[1] = join(lds_tr16_asm<NOFF>(offB[1] + nsb), lds_tr16_asm<NOFF + 2048>(offB[1] + nsb));
            };
            if (ph == 0 && t == 0) msteps(std::true_type{}); else msteps(std::false_type{});
            if (CSUM) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accb) : "v"(fcs), "v"(ones));
            __builtin_amdgcn_s_setprio(0);
            if (ph == NPH - 2 && wr == 1) wait_stage();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 3);
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        if constexpr (NPH == 4) phase(std::integral_constant<int, 3>{});
        static_assert(NPH == 3 || NPH == 4, "three or four 16-token phases per stage");
        slot = slot1;
    }
#else
    constexpr int LEAD = TN_STAGES - 1; // stages of lead of the requests (2; 1 in the two-stage A/B build)
    int slot = 0;                       // ring slot of stage t;  stage t+LEAD goes into the slot stage t-1 just left
    for (int t = 0; t < nsteps; ++t) {
        const uint32_t sb = slot * STAGE_BYTES;
        const bool more = t + 1 < nsteps, more2 = t + LEAD < nsteps;
        const int slot1 = slot == TN_STAGES - 1 ? 0 : slot + 1;
        const int slot2 = LEAD == 1 ? slot1 : (slot1 == TN_STAGES - 1 ? 0 : slot1 + 1);
        // end of the stage: stage t+1 (requested during stage t-1) must have landed; the 2*PER pieces of stage t+2, the
        // youngest in this wave's queue, may stay in flight (vmcnt retires in issue order)
        auto wait_stage = [&]() {
            if (LEAD > 1 && more2) wait_vmcnt<2 * PER>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (more) zero_tail(slot1, t + 1);
        };
        auto phase = [&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            // ---------------- L segment: fragments of 16 tokens ----------------
            bf16x8 fa[TM], fb[2];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = join(lds_tr16_asm<ph * 8192>(offA[i] + sb), lds_tr16_asm<ph * 8192 + 2048>(offA[i] + sb));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[j] = join(lds_tr16_asm<ph * 8192>(offB[j] + sb), lds_tr16_asm<ph * 8192 + 2048>(offB[j] + sb));
            bf16x8 fcs = ones;       // CSUM: this wave's own copy of dY row-block wc (fa[wc] would be a run-time register index)
            if (CSUM) fcs = join(lds_tr16_asm<ph * 8192>(offCS + sb), lds_tr16_asm<ph * 8192 + 2048>(offCS + sb));
            if (more2) {
                if (ph == 0) dmaA(slot2, t + LEAD);
                if (ph == 1) dmaB(slot2, t + LEAD);
            }
            if (ph == NPH - 1 && wr == 1) wait_stage();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 1);
            // ---------------- M segment ----------------
            __builtin_amdgcn_s_setprio(1);
            if (ph == 0 && t == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma32_first<bf16>(acc[i][j], fa[i], fb[j]);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) mma32<bf16>(acc[i][j], fa[i], fb[j]);
            }
            // CSUM is wave-uniform: a scalar branch around one MFMA.  Inline asm with the accumulator tied: through the
            // builtin the register allocator gave the conditional MFMA a fresh destination and moved 16 registers back
            // behind MFMA-latency s_nops inside the barrier-paced segment (+20 % on the whole launch).  The s_nop is the
            // VALU-write -> MFMA-read wait the compiler inserts for its own MFMAs (it re-materialises `ones` with v_movs
            // right in front of the asm and cannot know the asm is an MFMA).
            if (CSUM) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accb) : "v"(fcs), "v"(ones));
            __builtin_amdgcn_s_setprio(0);
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 2);
            if (ph == NPH - 1 && wr == 0) {
                wait_stage();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            PA_PROBE_STAMP_TN(t >= 4 && t < 12, ((t - 4) * NPH + ph) * 4 + 3);
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        if constexpr (NPH == 4) phase(std::integral_constant<int, 3>{});
        static_assert(NPH == 3 || NPH == 4, "three or four 16-token phases per stage");
        slot = slot1;
    }
#endif
#if PA_TN_FRAG_PIPE != 2
    if (wr == 0) __builtin_amdgcn_s_barrier();
#endif
#ifdef PA_PROBE
    __syncthreads();
    if (g_probe_buf && blockIdx.x < 8)
        for (int i = tid; i < 2 * PROBE_SLOTS; i += 512)
            g_probe_buf[(size_t)blockIdx.x * 2 * PROBE_SLOTS + i] = ((unsigned long long*)(smem + TN_LDS))[i];
#endif
    {
        pa_gemm_args e = a;
        e.out_f32 = out;
        e.ldo32 = ldo;
        gemm_epilogue_f32_direct<PA_EPI_PARTIAL, TM>(e, acc, nullptr, m0, n0, 0, wr, wc, lane);
    }
    if (CSUM) {                        // every column of the 32x32 block holds the same sums: lanes 0 and 32 write them
        if ((lane & 31) == 0) {
            float* dst = cs_out;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 128 + wc * 32 + acc_row(r, lane);
                if (m < a.M) dst[m] = accb[r];
            }
        }
    }
}

// split-K form: slice `split` of `steps_per_split` token steps, partial slab out_f32[split][M][ldo32]
__device__ __forceinline__ void gemm_tn_stagger_item(const pa_gemm_args& a, const int tiles_n, const int tile_id, const int split,
                                                     const int steps_per_split) {
    const int steps_total = (a.K + TN_ROWS - 1) / TN_ROWS;
    const int st_begin = split * steps_per_split;
    const int nsteps = min(steps_total, st_begin + steps_per_split) - st_begin;
    gemm_tn_stagger_range(a, tiles_n, tile_id, st_begin, nsteps, a.out_f32 + (int64_t)split * a.M * a.ldo32, a.ldo32,
                          a.colsum_ws ? a.colsum_ws + (int64_t)split * a.M : nullptr);
}

__global__ __launch_bounds__(512) void gemm_tn_stagger_kernel(const pa_gemm_args a, const int tiles_n, const int nwg,
                                                              const int steps_per_split) {
    gemm_tn_stagger_item(a, tiles_n, xcd_swizzle(blockIdx.x, nwg), blockIdx.y, steps_per_split);
}

// Several weight-gradient problems in one launch (pa_gemm_tn_batched): workgroup -> (problem, tile, K slice).  The
// hardware hands work items to CUs as they free up, so problems of different length pack without a launch boundary
// (drain + prologue) between them.
struct TnBatch {
    pa_gemm_args a[PA_TN_BATCH_MAX];
    int32_t first[PA_TN_BATCH_MAX + 1];      // first work item of problem p; first[n] = total
    int32_t tiles_n[PA_TN_BATCH_MAX], nwg[PA_TN_BATCH_MAX], per[PA_TN_BATCH_MAX];
    int32_t tfirst[PA_TN_BATCH_MAX + 1];     // first tile of problem p in the slice-major order (order = 1)
    int32_t n, order;
};
__global__ __launch_bounds__(512) void gemm_tn_stagger_batched_kernel(const TnBatch b) {
    // Item order (b.order = 1, the default whenever all problems use the same slice count): slice-major over the whole
    // batch, every XCD owning a contiguous run of that order (xcd_swizzle over all items), so an XCD's ~32 resident
    // workgroups work on one token slice and the tiles that share a dY or an X panel find its stages in their own L2:
    // L2->fabric reads 1 984 -> 962 MB per launch for 746 MB of operands (rocprofv3 FETCH_SIZE, run r04c), launch
    // 336 -> 335 us in the step.  (With the two-stage kernel of round 1 the same order measured 3-4 % slower and was
    // rejected; with three stages in flight it is no longer behind.)  b.order = 0 (tune = 2 on the first problem, A/B
    // only): problem-major, the tiles of one (problem, slice) spread over the XCDs.
    int p = 0, t, split;
    if (b.order == 1) {
        const int logical = xcd_swizzle(blockIdx.x, b.first[b.n]);
        const int tiles_all = b.tfirst[b.n];
        split = logical / tiles_all;
        t = logical - split * tiles_all;
        while (p + 1 < b.n && t >= b.tfirst[p + 1]) ++p;
        t -= b.tfirst[p];
    } else {
        while (p + 1 < b.n && (int)blockIdx.x >= b.first[p + 1]) ++p;
        const int item = blockIdx.x - b.first[p];
        split = item / b.nwg[p];
        t = xcd_swizzle(item - split * b.nwg[p], b.nwg[p]);
    }
    gemm_tn_stagger_item(b.a[p], b.tiles_n[p], t, split, b.per[p]);
}

// (r03: a stream-K form of the batched launch -- ONE sequence of tiles x token steps cut into 256 equal contiguous runs,
// one per CU, 364 partial tiles instead of 756, no round quantisation -- measured 504 us + 23 us fix-up against 382 + 36:
// in tile-major runs the CUs working at the same time sit at different token offsets, nobody shares a dY / X panel stage and
// every tile streams both panels from HBM, 3.35 GB per launch instead of 0.96.  The slice-major split-K order is what keeps
// this kernel off the HBM roofline; profiles/r03_streamk_experiment.txt.)
static int launch_gemm_tn_stagger(const pa_gemm_args& a, hipStream_t st) {
#ifdef PA_PROBE
    constexpr int LDS = TN_LDS + 2 * PROBE_SLOTS * 8;
#else
    constexpr int LDS = TN_LDS;
#endif
    const int tiles_m = (int)cdiv(a.M, 256), tiles_n = (int)cdiv(a.N, 256);
    const int nwg = tiles_m * tiles_n;
    const int steps = (int)cdiv(a.K, TN_ROWS);
    const int per = (int)cdiv(steps, a.split_k);
    static signed char lds_attr[64] = {0};
    (void)lds_attr_on_this_device((const void*)gemm_tn_stagger_kernel, LDS, lds_attr);
    hipLaunchKernelGGL(gemm_tn_stagger_kernel, dim3(nwg, a.split_k), dim3(512), LDS, st, a, tiles_n, nwg, per);
    return check_launch();
}

template <typename T>
static int launch_gemm_tn(const pa_gemm_args& a, hipStream_t st) {
    const int tiles_m = (int)cdiv(a.M, BM), tiles_n = (int)cdiv(a.N, BN);
    const int nwg = tiles_m * tiles_n;
    const int steps = (int)cdiv(a.K, TNGeom<T>::MROWS);
    const int per = (int)cdiv(steps, a.split_k);
    static signed char lds_attr[64] = {0};
    (void)lds_attr_on_this_device((const void*)gemm_tn_kernel<T>, GEMM_LDS, lds_attr);
    hipLaunchKernelGGL((gemm_tn_kernel<T>), dim3(nwg, a.split_k), dim3(256), GEMM_LDS, st, a, tiles_n, nwg, per);
    return check_launch();
}

// ---- column sums of a [R][C] matrix (bias gradients): stage 1 = per-(row block, 64-column tile) partials
template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const T* __restrict__ in, int R, int C, int ld,
                                                            float* __restrict__ part, int rows_per_block) {
    __shared__ float red[32][65];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;           // 8 threads x 8 columns, 32 rows per pass
    const int c0 = blockIdx.x * 64 + tx * 8;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(R, r_begin + rows_per_block);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    for (int r = r_begin + ty; r < r_end; r += 32) {
        float v[8];
        if (c0 < C) {     // C % 8 == 0
            load8<T>(in + (int64_t)r * ld + c0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 32; ++y) t += red[y][threadIdx.x];
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < C) part[(int64_t)blockIdx.y * C + c] = t;
    }
}

template <typename T>
static int dispatch_gemm(const pa_gemm_args& a, hipStream_t st) {
    switch (a.epilogue) {
        case PA_EPI_STORE: return a.out_lp ? launch_gemm<T, PA_EPI_STORE>(a, st) : PA_EINVAL;
        case PA_EPI_GELU: return (a.out_lp && a.out_lp2) ? launch_gemm<T, PA_EPI_GELU>(a, st) : PA_EINVAL;
        case PA_EPI_RESID: return (a.out_f32 && a.resid) ? launch_gemm<T, PA_EPI_RESID>(a, st) : PA_EINVAL;
        case PA_EPI_DGELU: return (a.out_lp && a.aux) ? launch_gemm<T, PA_EPI_DGELU>(a, st) : PA_EINVAL;
        case PA_EPI_PARTIAL: return (a.out_f32 && a.split_k >= 1) ? launch_gemm<T, PA_EPI_PARTIAL>(a, st) : PA_EINVAL;
    }
    return PA_EINVAL;
}

// ---- elementwise / data movement -----------------------------------------------------------
template <typename T>
__global__ void convert_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *(const float4*)(in + i);
        out[i] = from_f32<T>(v.x); out[i + 1] = from_f32<T>(v.y);
        out[i + 2] = from_f32<T>(v.z); out[i + 3] = from_f32<T>(v.w);
    }
    if (i < n) for (int64_t k = i; k < n && k < i + 4; ++k) out[k] = from_f32<T>(in[k]);
}

template <typename T>
__global__ void convert_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    const bool vec = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    for (; i < n; i += stride) {
        if (vec && i + 7 < n) {
            float v[8];
            load8<T>(in + i, v);
            store8<float>(out + i, v);
        } else {
            for (int64_t k = i; k < n && k < i + 8; ++k) out[k] = to_f32<T>(in[k]);
        }
    }
}

// 64x64 tile transpose through LDS; out[c][r] = in[r][c], zero fill for r in [R, ldo).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, int R, int C, int ldi,
                                                        TO* __restrict__ out, int ldo) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? to_f32<TI>(in[(int64_t)r * ldi + c]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < ldo) out[(int64_t)c * ldo + r] = from_f32<TO>(tile[tx][i]);
    }
}

// All weight copies of a step in one launch (pa_stage_weights): block = one 64x64 tile of one entry.
template <typename TO>
__global__ __launch_bounds__(256) void stage_weights_kernel(const pa_stage_desc* __restrict__ descs, int n_desc) {
    __shared__ float tile[64][65];
    __shared__ int s_entry;
    const int bid = blockIdx.x;
    if (threadIdx.x < 64) {        // one wave scans the (short) table: the last entry with tile_begin <= bid
        int found = -1;
        for (int e = threadIdx.x; e < n_desc; e += 64)
            if (descs[e].tile_begin <= bid) found = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) found = max(found, __shfl_xor(found, o, 64));
        if (threadIdx.x == 0) s_entry = found;
    }
    __syncthreads();
    const pa_stage_desc d = descs[s_entry];
    const int tiles_c = (d.cols + 63) >> 6;
    const int t = bid - d.tile_begin;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    TO* dst = (TO*)d.dst;
    TO* dst_t = (TO*)d.dst_t;
    if (((d.rows | d.cols) & 3) == 0 && ((uintptr_t)d.src & 15) == 0) {
        // fast path (every PaSST weight): 16-byte loads, 4-element stores, 128+ contiguous bytes per 16 lanes
        const int lr = threadIdx.x >> 4, l4 = (threadIdx.x & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + lr + 16 * i, c = c0 + l4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < d.rows && c < d.cols) {
                v = *(const f32x4*)(d.src + (int64_t)r * d.cols + c);
                if (dst) {
                    if constexpr (sizeof(TO) == 2) *(bf16x4*)(dst + (int64_t)r * d.cols + c) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    else *(f32x4*)(dst + (int64_t)r * d.cols + c) = v;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[lr + 16 * i][l4 + k] = v[k];
        }
        if (!dst_t) return;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + lr + 16 * i, r = r0 + l4;
            if (c < d.cols && r < d.rows) {
                const f32x4 v = {tile[l4][lr + 16 * i], tile[l4 + 1][lr + 16 * i], tile[l4 + 2][lr + 16 * i], tile[l4 + 3][lr + 16 * i]};
                if constexpr (sizeof(TO) == 2) *(bf16x4*)(dst_t + (int64_t)c * d.rows + r) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                else *(f32x4*)(dst_t + (int64_t)c * d.rows + r) = v;
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        const bool ok = r < d.rows && c < d.cols;
        const float v = ok ? d.src[(int64_t)r * d.cols + c] : 0.f;
        tile[i][tx] = v;
        if (ok && dst) dst[(int64_t)r * d.cols + c] = from_f32<TO>(v);
    }
    if (!dst_t) return;
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < d.cols && r < d.rows) dst_t[(int64_t)c * d.rows + r] = from_f32<TO>(tile[tx][i]);
    }
}

// 16-byte accesses over n4 = n/4 vectors (0 when a pointer or the slab pitch is not 16-byte aligned) + scalar tail
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int splits, int64_t n4, int64_t n,
                                                              float* __restrict__ out, int accumulate) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n4; i += stride) {
        f32x4 s = accumulate ? ((const f32x4*)out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < splits; ++z) s += *(const f32x4*)(part + (int64_t)z * n + i * 4);
        ((f32x4*)out)[i] = s;
    }
    for (int64_t i = n4 * 4 + t0; i < n; i += stride) {
        float s = accumulate ? out[i] : 0.f;
        for (int z = 0; z < splits; ++z) s += part[(int64_t)z * n + i];
        out[i] = s;
    }
}

// one wave per row
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* __restrict__ in, int R, int C, int ld,
                                                     float* __restrict__ out, int accumulate) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += to_f32<T>(in[(int64_t)row * ld + c]);
    s = wave_sum(s);
    if (lane == 0) out[row] = (accumulate ? out[row] : 0.f) + s;
}

// out[c] (+)= sum_r in[r][c]; block = 64 columns x 4 row groups, 4 independent accumulators per thread so
// several loads are in flight (the loop is latency-, not bandwidth-bound)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ in, int R, int C, int ld,
                                                         float* __restrict__ out, int accumulate) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int r = ry;
        for (; r + 12 < R; r += 16) {
            s0 += in[(int64_t)r * ld + c];
            s1 += in[(int64_t)(r + 4) * ld + c];
            s2 += in[(int64_t)(r + 8) * ld + c];
            s3 += in[(int64_t)(r + 12) * ld + c];
        }
        for (; r < R; r += 4) s0 += in[(int64_t)r * ld + c];
    }
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && c < C) {
        const float s = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
        out[c] = (accumulate ? out[c] : 0.f) + s;
    }
}

// out[i][:] = in[idx[i]][:]   (row gather; one workgroup per output row, any element size via bytes)
__global__ void gather_rows_kernel(const char* __restrict__ in, const int32_t* __restrict__ idx, int64_t row_bytes,
                                   char* __restrict__ out, int scatter) {
    const int i = blockIdx.x;
    const char* src = scatter ? in + (int64_t)i * row_bytes : in + (int64_t)idx[i] * row_bytes;
    char* dst = scatter ? out + (int64_t)idx[i] * row_bytes : out + (int64_t)i * row_bytes;
    if ((row_bytes & 15) == 0) {
        for (int64_t o = (int64_t)threadIdx.x * 16; o < row_bytes; o += (int64_t)blockDim.x * 16) *(uint4*)(dst + o) = *(const uint4*)(src + o);
    } else {
        for (int64_t o = (int64_t)threadIdx.x * 4; o < row_bytes; o += (int64_t)blockDim.x * 4) *(uint32_t*)(dst + o) = *(const uint32_t*)(src + o);
    }
}

// Finish of a split-K NT GEMM (pa_gemm_nt_splitk): out = epilogue(sum_z ws[z][m][n]) for the two epilogues whose output is
// a plain function of the accumulator -- STORE: (acc + bias) * colscale -> bf16, RESID: acc + bias + resid -> f32.
// One 16-byte vector of 4 columns per thread and step; N % 8 == 0.
template <int EPI>
__global__ void nt_splitk_finish_kernel(pa_gemm_args a, const float* __restrict__ ws, int splits) {
    const int nv = a.N >> 2;
    const int64_t total = (int64_t)a.M * nv, slab = (int64_t)a.M * a.N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nv), n = (int)(i - (int64_t)m * nv) * 4;
        const float* src = ws + (int64_t)m * a.N + n;
        f32x4 v = *(const f32x4*)src;
        for (int z = 1; z < splits; ++z) v += *(const f32x4*)(src + z * slab);
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) b = *(const f32x4*)(a.bias + n);
        if constexpr (EPI == PA_EPI_STORE) {
            const float cs = n < a.colscale_n ? a.colscale : 1.f;
            uint32_t o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e)        // the same multiply-add as the fused epilogue: acc * cs + bias * cs
                o[e] = cvt_pk_bf16(fmaf(v[2 * e], cs, b[2 * e] * cs), fmaf(v[2 * e + 1], cs, b[2 * e + 1] * cs));
            *(uint2*)((bf16*)a.out_lp + (int64_t)m * a.ldolp + n) = uint2{o[0], o[1]};
        } else {
            const f32x4 r = *(const f32x4*)(a.resid + (int64_t)m * a.ldr + n);
            *(f32x4*)(a.out_f32 + (int64_t)m * a.ldo32 + n) = (v + b) + r;
        }
    }
}

}  // namespace pa

using namespace pa;

extern "C" int pa_gemm_nt(const pa_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || a->M <= 0 || a->N <= 0 || a->K <= 0) return PA_EINVAL;
    const size_t es = a->dtype == PA_BF16 ? 2 : 4;
    if ((a->K * es) % KB) return PA_EUNSUPPORTED;
    if ((a->lda * es) % 16 || (a->ldb * es) % 16) return PA_EUNSUPPORTED;
    if (a->epilogue != PA_EPI_PARTIAL && a->split_k > 1) return PA_EINVAL;
    if (a->N % 8) return PA_EUNSUPPORTED;
    // the epilogue moves 8-element vectors: every output / auxiliary leading dimension must keep
    // them 16-byte aligned
    if (a->out_lp && a->ldolp % 8) return PA_EUNSUPPORTED;
    if (a->out_lp2 && a->ldolp2 % 8) return PA_EUNSUPPORTED;
    if (a->out_f32 && a->ldo32 % 4) return PA_EUNSUPPORTED;
    if (a->resid && a->ldr % 4) return PA_EUNSUPPORTED;
    if (a->aux && a->ldaux % 8) return PA_EUNSUPPORTED;
    if (a->colsum_out && (a->epilogue != PA_EPI_DGELU || !a->colsum_ws)) return PA_EINVAL;
    if (a->colscale_n && (a->epilogue != PA_EPI_STORE || a->colscale_n < 0 || a->colscale_n % 64)) return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (a->dtype == PA_BF16) return dispatch_gemm<bf16>(*a, st);
    if (a->dtype == PA_F32) return dispatch_gemm<float>(*a, st);
    return PA_EINVAL;
}

// ---- split-K for NT problems that leave most of the chip idle -------------------------------------------------------
// A [M][768] output of 128x256 tiles is 3 tiles per 128 rows: ESC-50 fine-tuning at batch 12 (M = 4236) has 102 tiles for
// 256 CUs, the prefix-only tail of the last block (M = 2 B) has 3 -- and K = 2304 / 3072 of serial work in each.  The plan
// cuts K into `splits` slices so that tiles x splits fills the CUs (role-split kernel, PA_EPI_PARTIAL slabs in the caller's
// workspace) and one elementwise kernel applies the epilogue.  Only STORE / RESID (without the patch-embed row remap), bf16.
extern "C" int pa_gemm_nt_splitk_plan(int M, int N, int K, int epilogue, int dtype) {
    if (dtype != PA_BF16 || (epilogue != PA_EPI_STORE && epilogue != PA_EPI_RESID)) return 1;
    if (M <= 0 || N <= 0 || K <= 0 || (K * 2) % KB || N % 8) return 1;
    static const int off = [] { const char* e = getenv("PA_NT_SPLITK"); return e && atoi(e) == 0 ? 1 : 0; }();
    if (off) return 1;
    const int v = pick_nt_variant(M, N, K);
    if (v != 6 && v != 7 && v != 8) return 1;
    const int tm = v == 6 ? 4 : (v == 7 ? 3 : 2);
    const int64_t tiles = cdiv(M, 64 * tm) * cdiv(N, 256);
    const int ksteps = K * 2 / KB;
    if (tiles > 128 || ksteps < 16) return 1;
    // at least 6 K-tiles per slice (prologue and slab write stay a small part of an item), at most 16 slices
    int s = (int)std::min<int64_t>(std::min<int64_t>(256 / tiles, ksteps / 6), 16);
    // every slice must hold at least one K-tile: with ceil-divided slices the last ones can come out empty (97 K-tiles in 16
    // slices of 7: slices 14 and 15 are empty), and the role-split kernel then hands the whole problem to the generic one
    while (s >= 2 && (int64_t)(s - 1) * cdiv(ksteps, s) >= ksteps) --s;
    return s >= 2 ? s : 1;
}

extern "C" int64_t pa_gemm_nt_splitk_ws_floats(int M, int N, int K, int epilogue, int dtype) {
    const int s = pa_gemm_nt_splitk_plan(M, N, K, epilogue, dtype);
    return s > 1 ? (int64_t)s * M * N : 0;
}

extern "C" int pa_gemm_nt_splitk(const pa_gemm_args* a, float* ws, int64_t ws_floats, void* stream) {
    if (!a) return PA_EINVAL;
    const int s = (a->row_mod > 0 || a->split_k > 1 || a->tune) ? 1 : pa_gemm_nt_splitk_plan(a->M, a->N, a->K, a->epilogue, a->dtype);
    if (s <= 1 || !ws || ws_floats < (int64_t)s * a->M * a->N) return pa_gemm_nt(a, stream);
    if (a->epilogue == PA_EPI_STORE ? !a->out_lp || a->ldolp % 8 : (!a->out_f32 || !a->resid || a->ldo32 % 4 || a->ldr % 4)) return PA_EINVAL;
    if (a->colscale_n && (a->epilogue != PA_EPI_STORE || a->colscale_n < 0 || a->colscale_n % 64)) return PA_EINVAL;
    pa_gemm_args p = *a;
    p.epilogue = PA_EPI_PARTIAL;
    p.split_k = s;
    p.out_f32 = ws;
    p.ldo32 = a->N;
    p.bias = nullptr;
    p.resid = nullptr;
    p.colscale_n = 0;
    const int v = pick_nt_variant(a->M, a->N, a->K);
    p.tune = v == 6 ? 6 : v + 10;            // the 192- / 128-row tiles with A two K-tiles ahead where a slice allows it
    int rc = pa_gemm_nt(&p, stream);
    // drop-in for pa_gemm_nt: a shape the partial form does not cover runs unsplit instead of failing
    if (rc == PA_EUNSUPPORTED || rc == PA_EINVAL) return pa_gemm_nt(a, stream);
    if (rc != PA_OK) return rc;
    const int64_t vecs = (int64_t)a->M * (a->N / 4);
    const int blocks = (int)std::min<int64_t>(cdiv(vecs, 256), 2048);
    if (a->epilogue == PA_EPI_STORE)
        hipLaunchKernelGGL(nt_splitk_finish_kernel<PA_EPI_STORE>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a, ws, s);
    else
        hipLaunchKernelGGL(nt_splitk_finish_kernel<PA_EPI_RESID>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a, ws, s);
    return check_launch();
}

extern "C" int pa_convert_f32(const float* in, void* out, int64_t n, int dtype, void* stream) {
    if (!in || !out || n < 0) return PA_EINVAL;
    if (n == 0) return PA_OK;
    const int blocks = (int)std::min<int64_t>(cdiv(n, 1024), 4096);
    if (dtype == PA_BF16) hipLaunchKernelGGL(convert_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, n);
    else if (dtype == PA_F32) hipLaunchKernelGGL(convert_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (float*)out, n);
    else return PA_EINVAL;
    return check_launch();
}

extern "C" int pa_convert_to_f32(const void* in, int dtype, float* out, int64_t n, void* stream) {
    if (!in || !out || n < 0) return PA_EINVAL;
    if (n == 0) return PA_OK;
    const int blocks = (int)std::min<int64_t>(cdiv(n, 2048), 4096);
    if (dtype == PA_BF16) hipLaunchKernelGGL(convert_to_f32_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, out, n);
    else if (dtype == PA_F32) hipLaunchKernelGGL(convert_to_f32_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)in, out, n);
    else return PA_EINVAL;
    return check_launch();
}

extern "C" int pa_transpose(const void* in, int in_dtype, int R, int C, int ldi, void* out,
                            int out_dtype, int ldo, void* stream) {
    if (!in || !out || R <= 0 || C <= 0 || ldo < R || ldi < C) return PA_EINVAL;
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)cdiv(ldo, 64));
    hipStream_t st = (hipStream_t)stream;
#define PA_TR(TI, TO) hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(256), 0, st, (const TI*)in, R, C, ldi, (TO*)out, ldo)
    if (in_dtype == PA_F32 && out_dtype == PA_F32) PA_TR(float, float);
    else if (in_dtype == PA_F32 && out_dtype == PA_BF16) PA_TR(float, bf16);
    else if (in_dtype == PA_BF16 && out_dtype == PA_BF16) PA_TR(bf16, bf16);
    else if (in_dtype == PA_BF16 && out_dtype == PA_F32) PA_TR(bf16, float);
    else return PA_EINVAL;
#undef PA_TR
    return check_launch();
}

extern "C" int pa_stage_weights(const pa_stage_desc* descs, int n_desc, int total_tiles, int dtype, void* stream) {
    if (!descs || n_desc <= 0 || total_tiles <= 0) return PA_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PA_BF16) hipLaunchKernelGGL(stage_weights_kernel<bf16>, dim3(total_tiles), dim3(256), 0, st, descs, n_desc);
    else if (dtype == PA_F32) hipLaunchKernelGGL(stage_weights_kernel<float>, dim3(total_tiles), dim3(256), 0, st, descs, n_desc);
    else return PA_EINVAL;
    return check_launch();
}

extern "C" int pa_reduce_partials(const float* partial, int splits, int64_t n, float* out,
                                  int accumulate, void* stream) {
    if (!partial || !out || splits < 1 || n <= 0) return PA_EINVAL;
    const bool vec = (((uintptr_t)partial | (uintptr_t)out) & 15) == 0 && n % 4 == 0;
    const int64_t n4 = vec ? n / 4 : 0;
    const int blocks = (int)std::min<int64_t>(cdiv(std::max<int64_t>(n4, n - 4 * n4), 256), 4096);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, partial, splits, n4, n, out, accumulate);
    return check_launch();
}

extern "C" int pa_rowsum(const void* in, int dtype, int R, int C, int ld, float* out, int accumulate,
                         void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return PA_EINVAL;
    dim3 grid((unsigned)cdiv(R, 4));
    if (dtype == PA_BF16) hipLaunchKernelGGL(rowsum_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)in, R, C, ld, out, accumulate);
    else if (dtype == PA_F32) hipLaunchKernelGGL(rowsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, R, C, ld, out, accumulate);
    else return PA_EINVAL;
    return check_launch();
}

static thread_local int t_last_colsum_rows = 0;
extern "C" int pa_gemm_last_colsum_rows(void) { return t_last_colsum_rows; }

static int pa::finish_gemm_colsum(const pa_gemm_args& a, int rows, hipStream_t st) {
    t_last_colsum_rows = rows;
    if (a.reserved & PA_GEMM_COLSUM_DEFER) return PA_OK;       // the caller reduces the rows (pa_reduce_partials_batched)
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)cdiv(a.N, 64)), dim3(256), 0, st, a.colsum_ws, rows, a.N, a.N, a.colsum_out,
                       a.colsum_accumulate);
    return check_launch();
}

// rows of colsum_ws: two wave-tile rows per workgroup tile, tiles of >= 128 rows in every variant
extern "C" int64_t pa_gemm_colsum_ws_floats(int M, int N) { return (M <= 0 || N <= 0) ? 0 : 2 * cdiv(M, 128) * (int64_t)N; }

extern "C" int pa_colsum_f32(const float* in, int R, int C, int ld, float* out, int accumulate,
                             void* stream) {
    if (!in || !out || R <= 0 || C <= 0) return PA_EINVAL;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, in, R, C, ld, out, accumulate);
    return check_launch();
}

extern "C" int64_t pa_gemm_blocked_pre_elems(int M, int N) { return blocked_pre_rows(M) * N; }
// 1 when pa_gemm_nt with tune = 0 runs the bf16 M x N x K GELU / DGELU GEMMs on a kernel that has the blocked form
extern "C" int pa_gemm_blocked_pre_ok(int M, int N, int K) {
    // the LDS-free epilogues (v3, opt-in) write and read the pre-activation row-major without any transposition: the
    // private blocked layout only exists for the v2 kernels
    if (epilogue_v3_enabled()) return 0;
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 64 || blocked_pre_rows(M) * N * 2 >= ((int64_t)1 << 31)) return 0;
    const int v = pick_nt_variant(M, N, K);
    if (v != 6 && v != 7 && v != 8) return 0;
    // the same conditions under which launch_gemm_stagger keeps the role-split kernel (no split-K here)
    const int tm = v == 6 ? 4 : (v == 7 ? 3 : 2);
    const int64_t tiles = cdiv(M, 64 * tm) * cdiv(N, 256);
    return K / 64 >= 2 && cdiv(tiles, 256) <= 128;
}

extern "C" int pa_gemm_tn_step_rows(void) { return TN_ROWS; }

extern "C" int pa_gemm_tn(const pa_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->out_f32 || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->split_k < 1) return PA_EINVAL;
    if (a->epilogue != PA_EPI_PARTIAL) return PA_EUNSUPPORTED;
    const size_t es = a->dtype == PA_BF16 ? 2 : 4;
    if ((a->lda * es) % 16 || (a->ldb * es) % 16 || a->ldo32 % 4) return PA_EUNSUPPORTED;
    if (a->M < (int)(16 / es) || a->N < (int)(16 / es) || a->N % 8 || a->M % (int)(16 / es)) return PA_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (a->colsum_ws && (a->dtype != PA_BF16 || a->tune == 1)) return PA_EUNSUPPORTED;    // only the role-split kernel sums dY
    if (a->dtype == PA_BF16) return a->tune == 1 ? launch_gemm_tn<bf16>(*a, st) : launch_gemm_tn_stagger(*a, st);
    if (a->dtype == PA_F32) return launch_gemm_tn<float>(*a, st);
    return PA_EINVAL;
}

extern "C" int pa_gemm_tn_batched(const pa_gemm_args* a, int n, void* stream) {
    if (!a || n < 1 || n > PA_TN_BATCH_MAX) return PA_EINVAL;
    TnBatch batch;
    TnBatch* pb = &batch;
    int total = 0, tiles = 0;
    bool same_split = true;
    for (int p = 0; p < n; ++p) {
        const pa_gemm_args& x = a[p];
        if (!x.A || !x.B || !x.out_f32 || x.M <= 0 || x.N <= 0 || x.K <= 0 || x.split_k < 1) return PA_EINVAL;
        if (x.dtype != PA_BF16 || x.epilogue != PA_EPI_PARTIAL) return PA_EUNSUPPORTED;
        if ((x.lda * 2) % 16 || (x.ldb * 2) % 16 || x.ldo32 % 4 || x.M < 8 || x.N < 8 || x.N % 8 || x.M % 8) return PA_EUNSUPPORTED;
        pb->a[p] = x;
        pb->tiles_n[p] = (int)cdiv(x.N, 256);
        pb->nwg[p] = (int)cdiv(x.M, 256) * pb->tiles_n[p];
        pb->per[p] = (int)cdiv(cdiv(x.K, TN_ROWS), x.split_k);
        pb->first[p] = total;
        pb->tfirst[p] = tiles;
        total += pb->nwg[p] * x.split_k;
        tiles += pb->nwg[p];
        same_split = same_split && x.split_k == a[0].split_k;
    }
    pb->first[n] = total;
    pb->tfirst[n] = tiles;
    pb->n = n;
    pb->order = (a[0].tune != 2 && same_split) ? 1 : 0;
#ifdef PA_PROBE
    constexpr int LDS = TN_LDS + 2 * PROBE_SLOTS * 8;
#else
    constexpr int LDS = TN_LDS;
#endif
    static signed char lds_attr[64] = {0};
    (void)lds_attr_on_this_device((const void*)gemm_tn_stagger_batched_kernel, LDS, lds_attr);
    hipLaunchKernelGGL(gemm_tn_stagger_batched_kernel, dim3(total), dim3(512), LDS, (hipStream_t)stream, batch);
    return check_launch();
}

namespace pa {
struct ReduceBatch {
    pa_reduce_desc d[PA_REDUCE_BATCH_MAX];
    int32_t n;
};
// blockIdx.y = problem.  SLABS: same arithmetic (and summation order) as reduce_partials_kernel.  ROWS: the arithmetic of
// ln_bwd_reduce_kernel (layernorm.hip): 16 columns x 16 row groups per workgroup, four loads in flight, then a 16-way tree in LDS.
__global__ __launch_bounds__(256) void reduce_partials_batched_kernel(const ReduceBatch b) {
    const pa_reduce_desc& d = b.d[blockIdx.y];
    if (d.mode == PA_REDUCE_ROWS) {
        __shared__ float red[16][17];
        const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
        for (int64_t c0 = (int64_t)blockIdx.x * 16; c0 < d.n; c0 += (int64_t)gridDim.x * 16) {     // workgroup-uniform
            const int64_t k = c0 + cx;
            float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            if (k < d.n) {
                int r = ry;
                for (; r + 48 < d.splits; r += 64) {
                    s += d.partial[(int64_t)r * d.pitch + k];
                    s1 += d.partial[(int64_t)(r + 16) * d.pitch + k];
                    s2 += d.partial[(int64_t)(r + 32) * d.pitch + k];
                    s3 += d.partial[(int64_t)(r + 48) * d.pitch + k];
                }
                for (; r < d.splits; r += 16) s += d.partial[(int64_t)r * d.pitch + k];
            }
            red[ry][cx] = (s + s1) + (s2 + s3);
            __syncthreads();
            if (ry == 0 && k < d.n) {
                float t = 0.f;
#pragma unroll
                for (int y = 0; y < 16; ++y) t += red[y][cx];
                d.out[k] = (d.accumulate ? d.out[k] : 0.f) + t;
            }
            __syncthreads();
        }
        return;
    }
    const int64_t n = d.n, n4 = (n % 4 == 0 && (((uintptr_t)d.partial | (uintptr_t)d.out) & 15) == 0) ? n / 4 : 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n4; i += stride) {
        f32x4 s = d.accumulate ? ((const f32x4*)d.out)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < d.splits; ++z) {
#if PA_NT_LD_SLAB
            s += __builtin_nontemporal_load((const f32x4*)(d.partial + (int64_t)z * n + i * 4));
#else
            s += *(const f32x4*)(d.partial + (int64_t)z * n + i * 4);
#endif
        }
        ((f32x4*)d.out)[i] = s;
    }
    for (int64_t i = n4 * 4 + t0; i < n; i += stride) {
        float s = d.accumulate ? d.out[i] : 0.f;
        for (int z = 0; z < d.splits; ++z) s += d.partial[(int64_t)z * n + i];
        d.out[i] = s;
    }
}
}  // namespace pa

extern "C" int pa_reduce_partials_batched(const pa_reduce_desc* d, int n, void* stream) {
    if (!d || n < 1 || n > PA_REDUCE_BATCH_MAX) return PA_EINVAL;
    ReduceBatch b;
    int64_t blocks = 1;
    for (int p = 0; p < n; ++p) {
        if (!d[p].partial || !d[p].out || d[p].splits < 1 || d[p].n <= 0) return PA_EINVAL;
        if (d[p].mode != PA_REDUCE_SLABS && d[p].mode != PA_REDUCE_ROWS) return PA_EINVAL;
        if (d[p].mode == PA_REDUCE_ROWS && d[p].pitch < d[p].n) return PA_EINVAL;
        b.d[p] = d[p];
        blocks = std::max<int64_t>(blocks, d[p].mode == PA_REDUCE_ROWS ? cdiv(d[p].n, 16) : cdiv(cdiv(d[p].n, 4), 256));
    }
    b.n = n;
    hipLaunchKernelGGL(reduce_partials_batched_kernel, dim3((unsigned)std::min<int64_t>(blocks, 2048), n), dim3(256), 0, (hipStream_t)stream, b);
    return check_launch();
}

extern "C" int64_t pa_colsum_ws_floats(int R, int C) { return (int64_t)std::min<int64_t>(32, cdiv(R, 256)) * C; }

extern "C" int pa_colsum(const void* in, int dtype, int R, int C, int ld, float* out, int accumulate, float* ws,
                         void* stream) {
    if (!in || !out || !ws || R <= 0 || C <= 0) return PA_EINVAL;
    const size_t es = dtype == PA_BF16 ? 2 : 4;
    if ((ld * es) % 16 || C % 8) return PA_EUNSUPPORTED;
    const int rblocks = (int)std::min<int64_t>(32, cdiv(R, 256));
    const int rpb = (int)cdiv(R, rblocks);
    dim3 grid((unsigned)cdiv(C, 64), (unsigned)rblocks);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PA_BF16) hipLaunchKernelGGL(colsum_stage1_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)in, R, C, ld, ws, rpb);
    else if (dtype == PA_F32) hipLaunchKernelGGL(colsum_stage1_kernel<float>, grid, dim3(256), 0, st, (const float*)in, R, C, ld, ws, rpb);
    else return PA_EINVAL;
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((unsigned)cdiv(C, 64)), dim3(256), 0, st, ws, rblocks, C, C, out, accumulate);
    return check_launch();
}

extern "C" int pa_gather_rows(const void* in, const int32_t* idx, int n_idx, int64_t row_bytes, void* out, void* stream) {
    if (!in || !idx || !out || n_idx <= 0 || row_bytes <= 0 || (row_bytes & 3)) return PA_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_idx), dim3(row_bytes >= 2048 ? 128 : 64), 0, (hipStream_t)stream,
                       (const char*)in, idx, row_bytes, (char*)out, 0);
    return check_launch();
}

extern "C" int pa_scatter_rows(const void* in, const int32_t* idx, int n_idx, int64_t row_bytes, void* out, void* stream) {
    if (!in || !idx || !out || n_idx <= 0 || row_bytes <= 0 || (row_bytes & 3)) return PA_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n_idx), dim3(row_bytes >= 2048 ? 128 : 64), 0, (hipStream_t)stream,
                       (const char*)in, idx, row_bytes, (char*)out, 1);
    return check_launch();
}

namespace pa {
// strided zero fill with 16-byte stores (hipMemset2DAsync's fill kernel ran at 0.7 TB/s on the [M][768] bf16 third of dqkv)
__global__ __launch_bounds__(256) void zero2d_kernel(char* __restrict__ p, int64_t pitch, int64_t w16, int64_t rows) {
    const int64_t n = rows * w16, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t r = i / w16, c = i - r * w16;
        *(f32x4*)(p + r * pitch + c * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
}  // namespace pa

extern "C" int pa_zero2d(void* ptr, int64_t pitch_bytes, int64_t width_bytes, int64_t rows, void* stream) {
    if (!ptr || pitch_bytes < width_bytes || width_bytes <= 0 || rows <= 0) return PA_EINVAL;
    if (pitch_bytes != width_bytes && ((uintptr_t)ptr | pitch_bytes | width_bytes) % 16 == 0) {
        const int64_t w16 = width_bytes / 16;
        const int blocks = (int)std::min<int64_t>(cdiv(rows * w16, 256), 8192);
        hipLaunchKernelGGL(zero2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (char*)ptr, pitch_bytes, w16, rows);
        return check_launch();
    }
    hipError_t e = pitch_bytes == width_bytes
                       ? hipMemsetAsync(ptr, 0, (size_t)(width_bytes * rows), (hipStream_t)stream)
                       : hipMemset2DAsync(ptr, (size_t)pitch_bytes, 0, (size_t)width_bytes, (size_t)rows, (hipStream_t)stream);
    return e == hipSuccess ? PA_OK : set_hip_error(e);
}

#ifdef PA_PROBE
// probe library only: where the role-split kernel writes its s_memtime stamps (NULL = off)
extern "C" int pa_probe_set_buffer(void* dev_buf) {
    unsigned long long* p = (unsigned long long*)dev_buf;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(pa::g_probe_buf), &p, sizeof(p));
    return e == hipSuccess ? PA_OK : pa::set_hip_error(e);
}
#endif
