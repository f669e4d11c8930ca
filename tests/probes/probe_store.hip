// Micro-probes behind the GEMM epilogue design; standalone, no library needed.
//   hipcc --offload-arch=gfx950 -O2 probe_store.hip -o probe_store && ./probe_store
//
// 1. store burst: every wave of a 512-thread workgroup issues NST x 1 KiB row-vector stores (the epilogue's
//    pattern: 8 lanes x 16 B = one 128-byte row segment, rows at a 6 KiB pitch) and waits vmcnt(0).  Cycles per
//    workgroup vs #workgroups resident (8 ... 256) and vs the cache policy of the store (default / nt / sc1 / sc0 sc1)
//    -> is the drain bound per CU or chip wide, and does a policy change it?
// 2. vmcnt order: a wave issues an LDS-DMA load from a COLD line (HBM miss) and then a store to a HOT line, waits
//    vmcnt(1) and reads the LDS bytes.  If loads and stores retired out of order, the cheap store would satisfy
//    the count while the DMA is still in flight and the wave would see the sentinel.  Counted waits with stores
//    younger than the DMA in the queue rely on this.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(char* p, f32x4 v) {
    if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// out: [rows][pitch] bytes; workgroup w owns rows [w*256, +256) x 512-byte column strip (w / tiles_m) -- like a
// 256x256 bf16 output tile; wave = 32 rows x 64 cols per pass ... here simply NST passes of 8 rows x 128 bytes.
template <int POLICY>
__global__ __launch_bounds__(512) void store_burst(char* out, int64_t pitch, int nst, int reps, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
    unsigned long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        // a fresh tile every repetition (never the same lines twice in a row)
        char* tile = out + ((int64_t)(blockIdx.x + (int64_t)r * gridDim.x) * 256) * pitch;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < nst; ++i) {
            const int row = wr * 128 + i * 8 + (lane >> 3);
            store16<POLICY>(tile + (int64_t)row * pitch + wc * 128 + (lane & 7) * 16, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        tot += t1 - t0;
    }
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = tot / reps;
}

__global__ __launch_bounds__(64) void vmcnt_order(const char* cold, int64_t cold_stride, char* hot, int iters, unsigned* stale,
                                                  unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[1024];
    const int lane = threadIdx.x;
    unsigned bad = 0;
    unsigned long long tot = 0;
    for (int it = 0; it < iters; ++it) {
        *(f32x4*)(lds + lane * 16) = f32x4{-1.f, -1.f, -1.f, -1.f};       // sentinel
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const char* src = cold + ((int64_t)blockIdx.x * iters + it) * cold_stride + lane * 16;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
        store16<0>(hot + (int64_t)blockIdx.x * 1024 + lane * 16, f32x4{1.f, 1.f, 1.f, 1.f});
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        const f32x4 got = *(const f32x4*)(lds + lane * 16);
        if (got[0] == -1.f) ++bad;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tot += t1 - t0;
    }
    atomicAdd(stale, bad);
    if (lane == 0) cyc[blockIdx.x] = tot / iters;
}

template <int POLICY> static void run_burst(const char* name, char* out, int64_t pitch, unsigned long long* dcyc) {
    for (int nst : {4, 12, 32}) {
        printf("store_burst policy=%-8s nst=%2d (KiB per CU = %3d):", name, nst, nst * 8);
        for (int grid : {8, 64, 256}) {
            const int reps = 8;
            hipLaunchKernelGGL(store_burst<POLICY>, dim3(grid), dim3(512), 0, 0, out, pitch, nst, reps, dcyc);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(grid * 8);
            CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long mx = 0, sum = 0;
            for (auto c : h) { mx = c > mx ? c : mx; sum += c; }
            printf("  grid %3d: mean %6llu max %6llu cyc (%.1f B/cyc/CU)", grid, sum / h.size(), mx, nst * 8192.0 / (sum / h.size()));
        }
        printf("\n");
    }
}

int main() {
    const int64_t pitch = 6144;                         // bytes: an [M][3072] bf16 matrix
    const int64_t rows = 256LL * 256 * 8;               // grid 256 x reps 8 tiles of 256 rows
    char* out;
    CK(hipMalloc(&out, (rows + 512) * pitch));
    unsigned long long* dcyc;
    CK(hipMalloc(&dcyc, 256 * 8 * 8));
    run_burst<0>("default", out, pitch, dcyc);
    run_burst<1>("nt", out, pitch, dcyc);
    run_burst<2>("sc1", out, pitch, dcyc);
    run_burst<3>("sc0 sc1", out, pitch, dcyc);

    // vmcnt order
    const int grid = 256, iters = 64;
    const int64_t cold_stride = 1 << 20;               // 1 MiB apart: 256*64 = 16 Gi?  too much -> 64 KiB apart over 1 GiB
    const int64_t stride = 65536;
    char* cold;
    CK(hipMalloc(&cold, (int64_t)grid * iters * stride));
    CK(hipMemset(cold, 0x3f, (int64_t)grid * iters * stride));          // 0x3f3f3f3f = 0.747 (not the sentinel)
    // flush caches: stream a large buffer
    CK(hipMemset(out, 0, rows * pitch));
    char* hot;
    CK(hipMalloc(&hot, grid * 1024));
    CK(hipMemset(hot, 0, grid * 1024));
    unsigned* stale;
    CK(hipMalloc(&stale, 4));
    CK(hipMemset(stale, 0, 4));
    (void)cold_stride;
    hipLaunchKernelGGL(vmcnt_order, dim3(grid), dim3(64), 0, 0, cold, stride, hot, iters, stale, dcyc);
    CK(hipDeviceSynchronize());
    unsigned hs = 0;
    CK(hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), dcyc, grid * 8, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (auto c : h) sum += c;
    printf("vmcnt_order: %u stale LDS reads of %d (0 = loads and stores retire in issue order); mean wait %llu cyc\n", hs,
           grid * iters * 64, sum / grid);
    return 0;
}
