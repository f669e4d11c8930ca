// Do the matrix pipe and the VALU of ONE SIMD run at the same time when the two instruction streams belong to DIFFERENT
// waves -- and when they are interleaved inside one wave?  (Round 3: the attention kernels' SQ counters show
// matrix-busy + VALU-busy = 0.93 of the kernel time with three waves per SIMD, i.e. no overlap at all.)
// A 512-thread workgroup puts waves w and w + 4 on the same SIMD.  Roles per wave half:
//   M = a loop of independent v_mfma_f32_32x32x16_bf16 (8 accumulators), V = a loop of independent v_fma_f32,
//   X = one wave doing both, interleaved 1 MFMA : 5 FMA.
//   hipcc --offload-arch=gfx950 -O2 probe_mfma_valu_overlap.hip -o probe_mfma_valu_overlap && ./probe_mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// role: 0 idle, 1 MFMA loop (8 per iteration), 2 VALU loop (40 v_fma per iteration), 3 both interleaved (8 MFMA + 40 FMA),
//       4 VALU loop with 40 v_exp per iteration
template <int ROLE_A, int ROLE_B>
__global__ __launch_bounds__(512) void k(const bf16x8* __restrict__ ops, float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    if (role == 0) return;
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = ops[(i * 64 + lane) & 1023];
    for (int j = 0; j < 2; ++j) b[j] = ops[((4 + j) * 64 + lane) & 1023];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[40];
    for (int i = 0; i < 40; ++i) v[i] = (float)(lane + i) * 1e-3f;
    const float c0 = out[1], c1 = out[2];
    if (role == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 40; ++i) v[i] = __builtin_fmaf(v[i], c0, c1);
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 40; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 5; ++j) v[i * 5 + j] = __builtin_fmaf(v[i * 5 + j], c0, c1);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 40; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;       // keep the loops alive
}

// ---- round 4: the same question in CYCLES -----------------------------------------------------------------------------
// The wall-clock table above cannot tell issue contention from clock droop (the chip is power limited: MFMA-heavy code runs
// at ~1.6 GHz, VALU-only code near 2.4), and its V wave ran the same number of iterations as the M wave, i.e. it was gone
// after a third of the M wave's lifetime.  Here the partner wave keeps issuing VALU work for as long as the M wave runs
// (twice the iteration count), the M wave brackets its loop with s_memtime (shader-clock counter) and s_memrealtime (constant
// 100 MHz): cycles per iteration say whether the VALU stream takes issue slots from the MFMA stream, and the ratio of the two
// counters is the clock each combination ran at.  ROLE_B: 0 none, 2 v_fma, 4 v_exp, 5 v_pk_fma (packed).
template <int ROLE_B>
__global__ __launch_bounds__(512) void kc(const bf16x8* __restrict__ ops, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float c0 = out[1], c1 = out[2];
    float s = 0.f;
    if (wave < 4) {
        bf16x8 a[4], b[2];
        for (int i = 0; i < 4; ++i) a[i] = ops[(i * 64 + lane) & 1023];
        for (int j = 0; j < 2; ++j) b[j] = ops[((4 + j) * 64 + lane) & 1023];
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();     // constant 100 MHz
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();         // shader clock
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        asm volatile("s_nop 0" :: "v"(s));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
    } else {
        if (ROLE_B == 0) return;
        float v[40];
        for (int i = 0; i < 40; ++i) v[i] = (float)(lane + i) * 1e-3f;
        // the M waves run `iters` iterations of >= 256 cycles; 40 VALU are <= 160 cycles, so 2 * iters VALU iterations
        // outlast them
        for (int it = 0; it < 2 * iters; ++it) {
            if (ROLE_B == 2) {
#pragma unroll
                for (int i = 0; i < 40; ++i) v[i] = __builtin_fmaf(v[i], c0, c1);
            } else if (ROLE_B == 4) {
#pragma unroll
                for (int i = 0; i < 40; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
            } else {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int i = 0; i < 40; i += 2) {
                    f32x2 x = {v[i], v[i + 1]};
                    const f32x2 m = {c0, c0}, a2 = {c1, c1};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(a2));
                    v[i] = x[0]; v[i + 1] = x[1];
                }
            }
        }
        for (int i = 0; i < 40; ++i) s += v[i];
    }
    if (s == 12345.678f) out[0] = s;
}

// ---- and the converse: how fast does the VALU wave run while its SIMD partner issues MFMAs back to back? ----------------
// V waves run `iters` iterations of 40 v_fma and clock themselves; the M partner runs 4x as many MFMA iterations (it outlasts
// them).  PRIO_M / PRIO_V: s_setprio of the two roles.  If the arbiter served the VALU stream in the issue slots an 8-pass
// MFMA leaves free (28 of 32 cycles), V would run at its solo rate whatever the priorities.
// GAP: what the M wave puts between two MFMAs.  0 nothing (the next MFMA waits at the head of the wave); 1..16: s_nop GAP-1
// (GAP idle cycles with a SALU-class instruction at the head); 17: two s_nop (28 cycles); 20: s_sleep 0; 30: a ds_read_b128
// of LDS garbage (an LDS-class instruction at the head, as between the MFMAs of a real K loop)
template <int GAP> __device__ __forceinline__ void mfma_gap(const char* lds, int lane) {
    if constexpr (GAP >= 1 && GAP <= 16) asm volatile("s_nop %0" :: "n"(GAP - 1));
    else if constexpr (GAP == 17) asm volatile("s_nop 13\n\ts_nop 13");
    else if constexpr (GAP == 20) asm volatile("s_sleep 0");
    else if constexpr (GAP == 30) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 r;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((uint32_t)(lane * 16)));
    }
}
template <int WITH_M, int PRIO_M, int PRIO_V, int GAP = 0>
__global__ __launch_bounds__(512) void kv(const bf16x8* __restrict__ ops, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float c0 = out[1], c1 = out[2];
    float s = 0.f;
    if (wave < 4) {
        if (!WITH_M) return;
        __builtin_amdgcn_s_setprio(PRIO_M);
        bf16x8 a[4], b[2];
        for (int i = 0; i < 4; ++i) a[i] = ops[(i * 64 + lane) & 1023];
        for (int j = 0; j < 2; ++j) b[j] = ops[((4 + j) * 64 + lane) & 1023];
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 4 * iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
                mfma_gap<GAP>(nullptr, lane);
            }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        asm volatile("s_nop 0" :: "v"(s));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    } else {
        __builtin_amdgcn_s_setprio(PRIO_V);
        float v[40];
        for (int i = 0; i < 40; ++i) v[i] = (float)(lane + i) * 1e-3f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 40; ++i) v[i] = __builtin_fmaf(v[i], c0, c1);
        for (int i = 0; i < 40; ++i) s += v[i];
        asm volatile("s_nop 0" :: "v"(s));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) cyc[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    }
    if (s == 12345.678f) out[0] = s;
}

template <int WITH_M, int PRIO_M, int PRIO_V, int GAP = 0> static void run_v(const char* name, int nwg, const bf16x8* dops, float* dout, unsigned long long* dcyc) {
    const int iters = 10000;
    CK(hipMemset(dcyc, 0, 4096 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((kv<WITH_M, PRIO_M, PRIO_V, GAP>), dim3(nwg), dim3(512), 64, 0, dops, dout, dcyc, iters);
    CK(hipDeviceSynchronize());
    static unsigned long long h[4096];
    CK(hipMemcpy(h, dcyc, sizeof(unsigned long long) * nwg * 16, hipMemcpyDeviceToHost));
    double m = 0, v = 0;
    for (int g = 0; g < nwg; ++g)
        for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[(g * 8 + w) * 2];
    printf("%-58s %4d workgroups: V wave %6.1f ticks per 40 v_fma (%.2f each)", name, nwg, v / (nwg * 4) / iters, v / (nwg * 4) / iters / 40);
    if (WITH_M) printf(",  M wave %6.1f ticks per 8 MFMA", m / (nwg * 4) / (4.0 * iters));
    printf("\n");
}

template <int B> static void run_cycles(const char* name, int nwg, const bf16x8* dops, float* dout, unsigned long long* dcyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    double cyc_best = 0, real_best = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kc<B>), dim3(nwg), dim3(512), 0, 0, dops, dout, dcyc, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        static unsigned long long h[2048];
        CK(hipMemcpy(h, dcyc, sizeof(unsigned long long) * nwg * 8, hipMemcpyDeviceToHost));
        double sum = 0, rsum = 0;
        for (int i = 0; i < nwg * 4; ++i) { sum += (double)h[2 * i]; rsum += (double)h[2 * i + 1]; }
        if (ms < best) { best = ms; cyc_best = sum / (nwg * 4) / iters; real_best = rsum / (nwg * 4) / iters; }
    }
    // s_memrealtime ticks at 100 MHz: ns = ticks * 10; clock = shader ticks / ns
    printf("%-24s %4d workgroups: %7.1f shader-clock ticks per 8 MFMA (%5.2f per MFMA), %6.1f ns per 8 MFMA => %.2f GHz; launch %7.3f ms\n",
           name, nwg, cyc_best, cyc_best / 8, real_best * 10.0, cyc_best / (real_best * 10.0), best);
}

template <int A, int B> static float run(const char* name, const bf16x8* dops, float* dout) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, dops, dout, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms   (%.1f ns per iteration of 8 MFMA and/or 40 VALU)\n", name, best, best * 1e6 / iters);
    return best;
}

int main() {
    bf16x8* dops;
    float* dout;
    CK(hipMalloc(&dops, 1024 * sizeof(bf16x8)));
    CK(hipMalloc(&dout, 64));
    unsigned short h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = 0x3c00 + (rand() & 0x3ff);      // bf16 around 0.01
    CK(hipMemcpy(dops, h, sizeof(h), hipMemcpyHostToDevice));
    float hv[16] = {0, 0.999f, 1e-3f};
    CK(hipMemcpy(dout, hv, 64, hipMemcpyHostToDevice));
    run<1, 0>("M alone (one wave per SIMD)", dops, dout);
    run<2, 0>("V(fma) alone", dops, dout);
    run<4, 0>("V(exp) alone", dops, dout);
    run<1, 1>("M + M on one SIMD", dops, dout);
    run<2, 2>("V + V on one SIMD", dops, dout);
    run<1, 2>("M + V(fma) on one SIMD (two waves)", dops, dout);
    run<1, 4>("M + V(exp) on one SIMD (two waves)", dops, dout);
    run<3, 0>("X: one wave, 1 MFMA : 5 FMA interleaved", dops, dout);
    run<3, 3>("X + X on one SIMD", dops, dout);
    unsigned long long* dcyc;
    CK(hipMalloc(&dcyc, 4096 * sizeof(unsigned long long)));
    printf("\n-- cycles (s_memtime) of the MFMA wave while a partner wave on the same SIMD issues VALU work for its whole lifetime --\n");
    for (int nwg : {8, 256}) {
        run_cycles<0>("M alone", nwg, dops, dout, dcyc);
        run_cycles<2>("M + V(v_fma_f32)", nwg, dops, dout, dcyc);
        run_cycles<5>("M + V(v_pk_fma_f32)", nwg, dops, dout, dcyc);
        run_cycles<4>("M + V(v_exp_f32)", nwg, dops, dout, dcyc);
    }
    printf("\n-- cycles of the VALU wave while its partner (older wave slot) issues MFMAs back to back --\n");
    for (int nwg : {8, 256}) {
        run_v<0, 0, 0>("V alone", nwg, dops, dout, dcyc);
        run_v<1, 0, 0>("V next to M, equal priority", nwg, dops, dout, dcyc);
        run_v<1, 0, 3>("V next to M, s_setprio 3 on the V wave", nwg, dops, dout, dcyc);
        run_v<1, 3, 0>("V next to M, s_setprio 3 on the M wave", nwg, dops, dout, dcyc);
    }
    printf("\n-- the same with something that is not a VALU-class instruction between the M wave's MFMAs (256 workgroups) --\n");
    run_v<1, 0, 0, 4>("M: mfma; s_nop 3 (4 cycles)", 256, dops, dout, dcyc);
    run_v<1, 0, 0, 8>("M: mfma; s_nop 7 (8 cycles)", 256, dops, dout, dcyc);
    run_v<1, 0, 0, 16>("M: mfma; s_nop 15 (16 cycles)", 256, dops, dout, dcyc);
    run_v<1, 0, 0, 17>("M: mfma; s_nop 13; s_nop 13 (28 cycles)", 256, dops, dout, dcyc);
    run_v<1, 0, 0, 20>("M: mfma; s_sleep 0", 256, dops, dout, dcyc);
    run_v<1, 0, 0, 30>("M: mfma; ds_read_b128 + lgkmcnt(0)", 256, dops, dout, dcyc);
    run_v<1, 0, 3, 16>("M: mfma; s_nop 15, s_setprio 3 on the V wave", 256, dops, dout, dcyc);
    run_v<1, 3, 0, 16>("M: mfma; s_nop 15, s_setprio 3 on the M wave", 256, dops, dout, dcyc);
    return 0;
}
