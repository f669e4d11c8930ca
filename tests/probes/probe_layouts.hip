// GPU micro-probe (diagnostic, not part of the product): verifies on real gfx950 hardware the three
// layout facts the kernels rely on -- MFMA 32x32x16 bf16 / 32x32x2 f32 operand+accumulator maps and
// the ds_read_b64_tr_b16 transpose-read lane mapping.  Build: hipcc --offload-arch=gfx950 probe_layouts.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// C[32][32] = A[32][K] * B[32][K]^T with the lane mapping the kernels assume
__global__ void k_mfma_bf16(const float* A, const float* B, float* C) {   // K = 16
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (bf16)A[(l & 31) * 16 + (l >> 5) * 8 + j]; b[j] = (bf16)B[(l & 31) * 16 + (l >> 5) * 8 + j]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k_mfma_f32(const float* A, const float* B, float* C) {    // K = 8 as 4 steps of 2
    const int l = threadIdx.x;
    f32x16 acc = {0};
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 8 + (l >> 5) * 4 + e], B[(l & 31) * 8 + (l >> 5) * 4 + e], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// tile[16 rows][64 cols] bf16 row-major (128-byte rows). Lane l: group g=l>>4, p=l&15 passes the address of
// tile[4g + (p>>2)][ (p&3)*4 ] and should receive tile[4g + 0..3][p].
__global__ void k_tr16(const float* in, float* out) {
    __shared__ __attribute__((aligned(16))) bf16 tile[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) tile[i] = (bf16)in[i];
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, p = l & 15;
    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4;
    v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4*)(tile + (4 * g + (p >> 2)) * 64 + (p & 3) * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)r[j];
}
int main() {
    int fails = 0;
    float *dA, *dB, *dC; hipMalloc(&dA, 4096 * 4); hipMalloc(&dB, 4096 * 4); hipMalloc(&dC, 4096 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        const int K = mode == 0 ? 16 : 8;
        std::vector<float> A(32 * K), B(32 * K), C(1024), R(1024);
        for (int i = 0; i < 32 * K; ++i) { A[i] = (float)((i * 7 + 3) % 13 - 6); B[i] = (float)((i * 5 + 1) % 11 - 5); }
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < K; ++k) s += A[m * K + k] * B[n * K + k]; R[m * 32 + n] = s; }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        if (mode == 0) hipLaunchKernelGGL(k_mfma_bf16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        else hipLaunchKernelGGL(k_mfma_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 1024; ++i) if (std::fabs(C[i] - R[i]) > 1e-3) ++bad;
        printf("probe mfma %s: %s (%d mismatches)\n", mode == 0 ? "32x32x16 bf16" : "32x32x2 f32", bad ? "FAIL" : "PASS", bad);
        fails += bad != 0;
    }
    {
        std::vector<float> T(16 * 64), O(256);
        for (int i = 0; i < 16 * 64; ++i) T[i] = (float)(i % 251);
        hipMemcpy(dA, T.data(), T.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_tr16, dim3(1), dim3(64), 0, 0, dA, dC);
        hipMemcpy(O.data(), dC, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { const int g = l >> 4, p = l & 15; if (O[l * 4 + j] != T[(4 * g + j) * 64 + p]) ++bad; }
        printf("probe ds_read_b64_tr_b16: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
        if (bad) for (int l = 0; l < 20; ++l) printf("  lane %d got %g %g %g %g\n", l, O[l * 4], O[l * 4 + 1], O[l * 4 + 2], O[l * 4 + 3]);
        fails += bad != 0;
    }
    hipError_t e = hipDeviceSynchronize();
    printf("probe status: %s, hip: %s\n", fails ? "FAIL" : "PASS", hipGetErrorString(e));
    return fails;
}
