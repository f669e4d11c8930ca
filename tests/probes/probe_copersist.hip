// What does a kernel that holds R CUs (an RCCL all-reduce on the communication stream) cost the role-split GEMMs, launched
// persistent (256 resident workgroups walking their item lists) or one item per workgroup (PA_GEMM_NO_PERSIST)?
// No multi-GPU box has been available, so DESIGN.md 6 made this choice "by construction"; this probe measures it on ONE GPU:
// an occupier kernel of R workgroups x 1024 threads with 64 KiB of LDS each (one per CU) spins on a second, high-priority
// stream for longer than the GEMM while the GEMM of a passt_s block (fc2: M = 30336, N = 768, K = 3072, RESID epilogue) runs
// on the first, through the library's C ABI (pa_gemm_nt).
//   hipcc --offload-arch=gfx950 -O2 -I../../include probe_copersist.hip -o probe_copersist -L../../passt_amd -lpasst_amd
//   LD_LIBRARY_PATH=../../passt_amd ./probe_copersist
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "passt_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void occupy(unsigned long long ticks_100mhz, int* sink) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) __builtin_amdgcn_s_sleep(8);
    if (sink && lds[threadIdx.x] == 77) *sink = 1;
}

int main(int argc, char** argv) {
    // argv[1]: 0 = both streams at the default priority, 1 (default) = occupier on a high-priority stream, GEMM at the lowest
    const int prio_mode = argc > 1 ? atoi(argv[1]) : 1;
    // argv[2], argv[3]: threads and LDS bytes of an occupier workgroup (default 1024 threads, 64 KiB: cannot share a CU with a GEMM
    // workgroup; 64 threads and 0 bytes can)
    const int occ_threads = argc > 2 ? atoi(argv[2]) : 1024, occ_lds = argc > 3 ? atoi(argv[3]) : 64 * 1024;
    const int M = 30336, N = 768, K = 3072;
    uint16_t *A, *W;
    float *bias, *resid, *out;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&resid, (size_t)M * N * 4)); CK(hipMalloc(&out, (size_t)M * N * 4));
    std::vector<uint16_t> h((size_t)M * K);
    srand(1);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x7ff) + ((rand() & 1) << 15);          // bf16 around +-0.01
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(resid, 0, (size_t)M * N * 4));
    hipStream_t sa, sb;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    if (prio_mode == 1) {
        CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, lo));
        CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    } else if (prio_mode == 2) {         // what bench.py sets up: compute at the default priority, communication high
        CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    } else {
        CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    }
    printf("stream priorities: mode %d (range lowest %d .. highest %d)\n", prio_mode, lo, hi);
    CK(hipFuncSetAttribute((const void*)occupy, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    pa_gemm_args a = {};
    a.dtype = PA_BF16; a.epilogue = PA_EPI_RESID; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.A = A; a.B = W; a.bias = bias;
    a.resid = resid; a.ldr = N; a.out_f32 = out; a.ldo32 = N; a.split_k = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("fc2-shaped GEMM (M %d, N %d, K %d, RESID epilogue), us per launch, best of 5; occupier: R workgroups of %d threads + %d bytes LDS\n", M, N, K,
           occ_threads, occ_lds);
    printf("%6s %14s %14s\n", "R", "persistent", "one item/wg");
    for (int R : {0, 8, 16, 32, 64}) {
        float best[2] = {1e9f, 1e9f};
        for (int mode = 0; mode < 2; ++mode) {
            a.reserved = mode ? PA_GEMM_NO_PERSIST : 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipDeviceSynchronize());
                if (R) hipLaunchKernelGGL(occupy, dim3(R), dim3(occ_threads), occ_lds, sb, 100ull * 600, (int*)nullptr);      // 600 us
                // give the occupier time to become resident before the GEMM is enqueued
                for (volatile int spin = 0; spin < 400000; ++spin) {}
                CK(hipEventRecord(e0, sa));
                if (pa_gemm_nt(&a, sa) != PA_OK) { printf("pa_gemm_nt failed\n"); return 1; }
                CK(hipEventRecord(e1, sa));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best[mode]) best[mode] = ms;
            }
        }
        printf("%6d %14.1f %14.1f\n", R, best[0] * 1e3f, best[1] * 1e3f);
    }
    return 0;
}
