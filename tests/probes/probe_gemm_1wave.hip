// What does the K loop reach with ONE wave per SIMD?  (the role-split kernels alternate two waves per SIMD
// between an LDS/DMA segment and an MFMA segment with an s_barrier at every hand-over: 73-79 % matrix-pipe duty inside the
// loop.  The classic alternative gives each of FOUR waves a 128 x 128 (or 96 x 128) share of a 256 x 256 (192 x 256) tile --
// accumulators in AGPRs, 512 registers per lane at one wave per SIMD -- and lets the wave itself issue its fragment reads and
// LDS-DMA requests in the shadow of its own MFMAs: one barrier per 64-wide K-tile instead of eight, 0.5 fragment reads per
// MFMA instead of 0.75-0.83.)  This probe measures that loop alone:
//   C[M][N] = A[M][K] * B[N][K]^T, bf16 in, f32 out, one (64*TM2) x 256 tile per 256-thread workgroup, TM2 = 4 or 3,
//   wave (wr, wc) of 2 x 2 owns (32*TM2) x 128 = TM2 x 4 MFMA 32x32 blocks; 128-byte LDS rows with gemm.hip's swizzle;
//   two 64 KiB (48 + ... ) stages; fragments double buffered in registers; the K-tile barrier sits in the MIDDLE of the last
//   k-substep's MFMAs so that its latency and the first reads of the next tile are covered by the other half.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../passt_amd/csrc -I../../include probe_gemm_1wave.hip -o probe_gemm_1wave
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "pa_mma.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace pa;

template <int TM2> struct Geo {
    static constexpr int TBM = 64 * TM2, TBN = 256, ROWB = 128;
    static constexpr int A_BYTES = TBM * ROWB, B_BYTES = TBN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int A_PER = A_BYTES / 1024 / 4, B_PER = B_BYTES / 1024 / 4;      // 1 KiB pieces per wave: 8 (6) + 8
    static constexpr int LDS = 2 * STAGE;
};

// 16-byte LDS read the compiler does not track: issued where it is written, settled by settle() (a counted s_waitcnt that
// names the fragments as operands, so no use of them can be scheduled above it).  With plain loads the compiler waits with
// lgkmcnt(0) in front of every MFMA group, i.e. also for the reads it has just issued for the NEXT substep.
__device__ __forceinline__ bf16x8 lds_read_asm(uint32_t addr, int imm) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(imm));
    return __builtin_bit_cast(bf16x8, r);
}
template <int TM2> __device__ __forceinline__ void settle(bf16x8 (&fa)[TM2], bf16x8 (&fb)[4], int pending) {
    if constexpr (TM2 == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) : "i"(pending));
    else
        asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]) : "i"(pending));
}

template <int TM2, bool ASMRD>
__global__ __launch_bounds__(256) void gemm_1wave_kernel(const bf16* __restrict__ A, const bf16* __restrict__ B, float* __restrict__ C,
                                                         int M, int N, int K, int tiles_n, int store) {
    using G = Geo<TM2>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * G::TBM, n0 = tn * G::TBN;
    const int nk = K / 64;

    // DMA: piece q = 8 rows of 128 bytes; lane l -> row q*8 + (l >> 3), slot l & 7 holds global chunk slot ^ swz_f128(row)
    uint32_t voffA[G::A_PER], voffB[G::B_PER];
#pragma unroll
    for (int i = 0; i < G::A_PER; ++i) {
        const int row = (wave * G::A_PER + i) * 8 + (lane >> 3);
        voffA[i] = (uint32_t)min(row, M - 1 - m0) * (uint32_t)K * 2u + (uint32_t)(((lane & 7) ^ swz_f128(row)) * 16);
    }
#pragma unroll
    for (int i = 0; i < G::B_PER; ++i) {
        const int row = (wave * G::B_PER + i) * 8 + (lane >> 3);
        voffB[i] = (uint32_t)min(row, N - 1 - n0) * (uint32_t)K * 2u + (uint32_t)(((lane & 7) ^ swz_f128(row)) * 16);
    }
    const char* gA = (const char*)(A + (int64_t)m0 * K);
    const char* gB = (const char*)(B + (int64_t)n0 * K);
    auto dma = [&](int stage, int t) {
        char* dA = smem + stage * G::STAGE + wave * (G::A_PER * 1024);
        char* dB = smem + stage * G::STAGE + G::A_BYTES + wave * (G::B_PER * 1024);
#pragma unroll
        for (int i = 0; i < G::A_PER; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + (int64_t)t * 128 + voffA[i]),
                                             (__attribute__((address_space(3))) void*)(dA + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < G::B_PER; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + (int64_t)t * 128 + voffB[i]),
                                             (__attribute__((address_space(3))) void*)(dB + i * 1024), 16, 0, 0);
    };

    const int half = lane >> 5, r31 = lane & 31, rsw = swz_f128(r31);      // rows 32 apart share the swizzle term
    const int adrA = (wr * (32 * TM2) + r31) * 128, adrB = G::A_BYTES + (wc * 128 + r31) * 128;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto read_frags = [&](const char* st, int s, bf16x8 (&fa)[TM2], bf16x8 (&fb)[4]) {
        const int coff = ((s * 2 + half) ^ rsw) << 4;
        if constexpr (ASMRD) {
            const uint32_t base = lds0 + (uint32_t)(st - smem) + (uint32_t)coff;
            const uint32_t pa = base + (uint32_t)adrA, pb = base + (uint32_t)adrB;
#pragma unroll
            for (int i = 0; i < TM2; ++i) fa[i] = lds_read_asm(pa, i * 32 * 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = lds_read_asm(pb, j * 32 * 128);
        } else {
#pragma unroll
            for (int i = 0; i < TM2; ++i) fa[i] = *(const bf16x8*)(st + adrA + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(st + adrB + j * 32 * 128 + coff);
        }
    };
    constexpr int NRD = TM2 + 4;             // reads per substep
    auto ready = [&](bf16x8 (&fa)[TM2], bf16x8 (&fb)[4], int pending) {        // fragments usable; `pending` younger reads may fly
        if constexpr (ASMRD) {
            __builtin_amdgcn_sched_barrier(0);
            settle<TM2>(fa, fb, pending);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f32x16 acc[TM2][4];
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nk > 1) dma(1, 1);
    bf16x8 fa0[TM2], fb0[4], fa1[TM2], fb1[4];
    read_frags(smem, 0, fa0, fb0);

    auto mfmas = [&](bf16x8 (&fa)[TM2], bf16x8 (&fb)[4], int j_lo, int j_hi) {
#pragma unroll
        for (int j = j_lo; j < j_hi; ++j)
#pragma unroll
            for (int i = 0; i < TM2; ++i) mma32<bf16>(acc[i][j], fa[i], fb[j]);
    };

    for (int t = 0; t < nk; ++t) {
        const char* st = smem + (t & 1) * G::STAGE;
        const char* stn = smem + ((t + 1) & 1) * G::STAGE;
        // substep 0: reads of substep 1 in flight under the MFMAs of substep 0, and so on
        read_frags(st, 1, fa1, fb1);
        ready(fa0, fb0, NRD);
        mfmas(fa0, fb0, 0, 4);
        read_frags(st, 2, fa0, fb0);
        ready(fa1, fb1, NRD);
        mfmas(fa1, fb1, 0, 4);
        read_frags(st, 3, fa1, fb1);
        ready(fa0, fb0, NRD);
        mfmas(fa0, fb0, 0, 4);
        // substep 3: first half, then the K-tile hand-over (everybody is done reading stage t; tile t+1 has landed),
        // the first reads of tile t+1 and the request for tile t+2, then the second half
        ready(fa1, fb1, 0);
        mfmas(fa1, fb1, 0, 2);
        if (t + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // own DMA pieces of tile t+1 landed; own reads of tile t done
            __builtin_amdgcn_s_barrier();
            read_frags(stn, 0, fa0, fb0);
            if (t + 2 < nk) dma(t & 1, t + 2);
        }
        if constexpr (ASMRD) __builtin_amdgcn_sched_barrier(0);
        mfmas(fa1, fb1, 2, 4);
    }

    if (store) {
#pragma unroll
        for (int i = 0; i < TM2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * (32 * TM2) + i * 32 + acc_row(r, lane);
                if (m < M) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = n0 + wc * 128 + j * 32 + r31;
                        if (n < N) C[(int64_t)m * N + n] = acc[i][j][r];
                    }
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
    }
}

static float urand(uint32_t& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

template <int TM2, bool ASMRD>
static void run(int M, int N, int K, const bf16* dA, const bf16* dB, float* dC, const std::vector<bf16>& hA, const std::vector<bf16>& hB) {
    using G = Geo<TM2>;
    CK(hipFuncSetAttribute((const void*)(gemm_1wave_kernel<TM2, ASMRD>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
    const int tiles_m = (M + G::TBM - 1) / G::TBM, tiles_n = (N + 255) / 256, grid = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_1wave_kernel<TM2, ASMRD>), dim3(grid), dim3(256), G::LDS, 0, dA, dB, dC, M, N, K, tiles_n, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0.0, maxref = 0.0;
    for (int k = 0; k < 300; ++k) {
        const int m = k == 0 ? M - 1 : (int)(((uint64_t)k * 7919u * 131u) % M), n = (int)(((uint64_t)k * 104729u) % N);
        double ref = 0.0;
        for (int x = 0; x < K; ++x) ref += (double)(float)hA[(size_t)m * K + x] * (double)(float)hB[(size_t)n * K + x];
        maxerr = fmax(maxerr, fabs(ref - hC[(size_t)m * N + n]));
        maxref = fmax(maxref, fabs(ref));
    }
    for (int mode = 1; mode >= 0; --mode) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gemm_1wave_kernel<TM2, ASMRD>), dim3(grid), dim3(256), G::LDS, 0, dA, dB, dC, M, N, K, tiles_n, mode);
        CK(hipEventRecord(e0));
        const int iters = 20;
        for (int w = 0; w < iters; ++w) hipLaunchKernelGGL((gemm_1wave_kernel<TM2, ASMRD>), dim3(grid), dim3(256), G::LDS, 0, dA, dB, dC, M, N, K, tiles_n, mode);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
        printf("tile %dx256 %s  M=%d N=%d K=%d  tiles=%d (%.2f rounds)  %s: %.1f us  %.1f TF/s   (max abs err %.3g of %.3g)\n", G::TBM,
               ASMRD ? "counted waits " : "compiler waits", M, N, K, grid,
               grid / 256.0, mode ? "K loop + f32 store" : "K loop only       ", us, tf, maxerr, maxref);
    }
}

int main() {
    const int M = 64 * 474;
    const int shapes[][2] = {{768, 3072}, {768, 2304}, {3072, 768}, {2304, 768}, {768, 768}};
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        std::vector<bf16> hA((size_t)M * K), hB((size_t)N * K);
        uint32_t s = 999u + N + K;
        for (auto& v : hA) v = (bf16)urand(s);
        for (auto& v : hB) v = (bf16)urand(s);
        bf16 *dA, *dB;
        float* dC;
        CK(hipMalloc(&dA, hA.size() * 2));
        CK(hipMalloc(&dB, hB.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        run<4, false>(M, N, K, dA, dB, dC, hA, hB);
        run<4, true>(M, N, K, dA, dB, dC, hA, hB);
        run<3, false>(M, N, K, dA, dB, dC, hA, hB);
        run<3, true>(M, N, K, dA, dB, dC, hA, hB);
        CK(hipFree(dA));
        CK(hipFree(dB));
        CK(hipFree(dC));
    }
    return 0;
}
