// Is a WHOLE-ROW tile family viable for the D = 768 outputs?  (a 128 x 768 tile holds complete rows of the
// residual stream, so LayerNorm -- forward: statistics + normalised bf16 row; backward: the two row reductions -- could
// live in the epilogue of proj / fc2 and of the qkv / fc1 input-gradient GEMMs instead of in separate HBM passes.)
// This probe measures only what decides that: the K loop of such a tile.
//   C[M][768] = A[M][K] * B[768][K]^T, bf16 in, f32 out, one 128 x 768 tile per workgroup (M / 128 workgroups: 237 for
//   64 x 474 tokens = one round on 256 CUs), 8 waves = 2 row groups x 4 column groups, wave tile 64 x 192 (2 x 6 MFMA
//   32x32 blocks, 192 accumulator registers), K consumed 32 per stage (64-byte LDS rows: A 8 KiB + B 48 KiB per stage,
//   two stages), LDS-DMA staging, the role-split schedule of gemm.hip (group 1 one barrier behind group 0; L = 8 fragment
//   reads + DMA requests, M = 12 MFMAs per 16-wide k-substep).
// 64-byte rows: chunk c (16 bytes) of row r lives in slot c ^ ((r >> 2) & 3): the 16 lanes of a ds_read_b128 group (16
// consecutive rows, one logical chunk) then hit 16 different 16-byte bank slots.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../passt_amd/csrc -I../../include probe_wide_gemm.hip -o probe_wide_gemm
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "pa_mma.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace pa;

static constexpr int TM_ROWS = 128, TN_COLS = 768, KSTEP = 32, ROWB = KSTEP * 2;        // 64-byte LDS rows
static constexpr int A_BYTES = TM_ROWS * ROWB, B_BYTES = TN_COLS * ROWB, STAGE = A_BYTES + B_BYTES;   // 8 + 48 KiB
static constexpr int LDS_BYTES = 2 * STAGE;

__device__ __forceinline__ int slot64(int row, int c) { return row * ROWB + ((c ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(512) void wide_gemm_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, float* __restrict__ C,
                                                        int M, int K, int store) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, wc = wave & 3;                   // row group, column group
    const int m0 = blockIdx.x * TM_ROWS;
    const int nk = K / KSTEP;

    // DMA: the stage image is [A rows | B rows] in 1 KiB pieces of 16 rows; lane l of a piece -> row (l >> 2), slot l & 3, which
    // holds global chunk slot ^ f(row).  Wave w stages A piece w and B pieces 6w .. 6w+5: f(row) = (row >> 2) & 3 does not
    // depend on the piece, so ONE lane offset per operand serves all pieces (the piece term is uniform: SGPR arithmetic).
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ (lrow >> 2);
    const uint32_t voffA = (uint32_t)min(wave * 16 + lrow, M - 1 - m0) * (uint32_t)K * 2u + (uint32_t)lchunk * 16u;
    const uint32_t voffB = (uint32_t)lrow * (uint32_t)K * 2u + (uint32_t)lchunk * 16u;
    const char* gA = (const char*)(A + (int64_t)m0 * K);
    const char* gB = (const char*)(B + (int64_t)wave * 96 * K);
    auto dma = [&](int stage, int t) {
        char* dA = smem + stage * STAGE + wave * 1024;
        char* dB = smem + stage * STAGE + A_BYTES + wave * 6144;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + (int64_t)t * ROWB + voffA),
                                         (__attribute__((address_space(3))) void*)dA, 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + (int64_t)i * 16 * K * 2 + (int64_t)t * ROWB + voffB),
                                             (__attribute__((address_space(3))) void*)(dB + i * 1024), 16, 0, 0);
    };

    f32x16 acc[2][6];
    const int half = lane >> 5, r31 = lane & 31;
    // rows 32 apart share f(row): one lane address per operand and k-substep, the block term is an immediate
    const int f = (r31 >> 2) & 3;
    int adrA[2], adrB[2];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int sw = ((ph * 2 + half) ^ f) << 4;
        adrA[ph] = (g * 64 + r31) * ROWB + sw;
        adrB[ph] = A_BYTES + (wc * 192 + r31) * ROWB + sw;
    }

    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind

    for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * STAGE;
        const bool more = t + 1 < nk;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            bf16x8 fa[2], fb[6];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(st + adrA[ph] + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < 6; ++j) fb[j] = *(const bf16x8*)(st + adrB[ph] + j * 32 * ROWB);
            if (ph == 0 && more) dma((t + 1) & 1, t + 1);
            if (ph == 1 && g == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
            if (t == 0 && ph == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) mma32_first<bf16>(acc[i][j], fa[i], fb[j]);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) mma32<bf16>(acc[i][j], fa[i], fb[j]);
            }
            __builtin_amdgcn_s_setprio(0);
            if (ph == 1 && g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (g == 0) __builtin_amdgcn_s_barrier();

    // epilogue: plain f32 stores from the accumulator layout (register r: row acc_row(r), column lane & 31)
    if (store) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + g * 64 + i * 32 + acc_row(r, lane);
                if (m < M) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) C[(int64_t)m * TN_COLS + wc * 192 + j * 32 + r31] = acc[i][j][r];
                }
            }
    } else {          // K-loop only: keep the accumulators live
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
    }
}

static float urand(uint32_t& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main() {
    const int M = 64 * 474;
    CK(hipFuncSetAttribute((const void*)wide_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (int K : {768, 2304, 3072}) {
        std::vector<bf16> hA((size_t)M * K), hB((size_t)TN_COLS * K);
        uint32_t s = 12345u + K;
        for (auto& v : hA) v = (bf16)urand(s);
        for (auto& v : hB) v = (bf16)urand(s);
        bf16 *dA, *dB;
        float* dC;
        CK(hipMalloc(&dA, hA.size() * 2));
        CK(hipMalloc(&dB, hB.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * TN_COLS * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        const int grid = (M + TM_ROWS - 1) / TM_ROWS;
        hipLaunchKernelGGL(wide_gemm_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, K, 1);
        CK(hipDeviceSynchronize());
        // check a few hundred entries against the host
        std::vector<float> hC((size_t)M * TN_COLS);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0.0, maxref = 0.0;
        for (int k = 0; k < 400; ++k) {
            const int m = (int)(((uint64_t)k * 7919u * 131u) % M), n = (int)(((uint64_t)k * 104729u) % TN_COLS);
            double ref = 0.0;
            for (int x = 0; x < K; ++x) ref += (double)(float)hA[(size_t)m * K + x] * (double)(float)hB[(size_t)n * K + x];
            maxerr = fmax(maxerr, fabs(ref - hC[(size_t)m * TN_COLS + n]));
            maxref = fmax(maxref, fabs(ref));
        }
        // last row too (edge tile)
        for (int mode = 1; mode >= 0; --mode) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(wide_gemm_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, K, mode);
            CK(hipEventRecord(e0));
            const int iters = 20;
            for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(wide_gemm_kernel, dim3(grid), dim3(512), LDS_BYTES, 0, dA, dB, dC, M, K, mode);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / iters, tf = 2.0 * M * TN_COLS * K / (us * 1e-6) / 1e12;
            printf("M=%d N=768 K=%d  %s: %.1f us  %.1f TF/s   (max abs err %.3g of %.3g)\n", M, K, mode ? "K loop + f32 store" : "K loop only       ",
                   us, tf, maxerr, maxref);
        }
        CK(hipFree(dA));
        CK(hipFree(dB));
        CK(hipFree(dC));
    }
    return 0;
}
