// 32x32 MFMA tile primitives shared by the GEMM and attention kernels (gfx950).
//
// One "fragment" is the 16 bytes a lane holds of a K-contiguous operand:
//   bf16 : 8 consecutive k  -> one v_mfma_f32_32x32x16_bf16 (lane l: row l&31, k = (l>>5)*8..+8)
//   f32  : 4 consecutive k  -> four v_mfma_f32_32x32x2_f32   (lane l: row l&31, k-pair {e, 4+e})
// The k-order inside a fragment step differs between the two types but is the same for the A and
// the B operand, so the contraction is unchanged (only the f32 summation order).
// Accumulator layout (both): col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include "pa_common.h"

namespace pa {

template <typename T> struct Frag;
template <> struct Frag<bf16> { typedef bf16x8 type; static constexpr int K = 16; };
template <> struct Frag<float> { typedef f32x4 type; static constexpr int K = 8; };

template <typename T>
__device__ __forceinline__ void mma32(f32x16& acc, const typename Frag<T>::type& a,
                                      const typename Frag<T>::type& b);
template <>
__device__ __forceinline__ void mma32<bf16>(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma32<float>(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

// first product of an accumulation chain: acc = a * b (C operand = the inline constant 0: no zero-filled registers)
template <typename T>
__device__ __forceinline__ void mma32_first(f32x16& acc, const typename Frag<T>::type& a, const typename Frag<T>::type& b);
template <>
__device__ __forceinline__ void mma32_first<bf16>(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma32_first<float>(f32x16& acc, const f32x4& a, const f32x4& b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], z, 0, 0, 0);
#pragma unroll
    for (int e = 1; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

// first product of a chain that starts from a given C operand: acc = a * b + c  (acc and c are different registers:
// attention keeps a block of per-row offsets (-running max, -lse, -delta) and gets "score - offset" from the matrix
// pipe instead of one VALU instruction per score)
template <typename T>
__device__ __forceinline__ void mma32_c(f32x16& acc, const typename Frag<T>::type& a, const typename Frag<T>::type& b, const f32x16& c);
template <>
__device__ __forceinline__ void mma32_c<bf16>(f32x16& acc, const bf16x8& a, const bf16x8& b, const f32x16& c) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma32_c<float>(f32x16& acc, const f32x4& a, const f32x4& b, const f32x16& c) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
#pragma unroll
    for (int e = 1; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

// row of accumulator register r for this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// LDS tile layouts.  A tile is [rows][ROWBYTES] with ROWBYTES = 128 (8 x 16-byte chunks) or 256
// (16 chunks); chunk c of row `row` lives at physical chunk c ^ f(row).  f is chosen so that BOTH
// access patterns of the MFMA operands are bank-conflict free (guide sec. 2 lane groups, T2, T10):
//  * row access  (ds_read_b128, 32 lanes = 32 consecutive rows at one logical chunk), and
//  * column access of bf16 tiles by ds_read_b64_tr_b16 (a 16-lane group = 4 consecutive rows x
//    32 bytes; rows r..r+3 with r%4==0 must land on different bank quarters).
// 128-byte rows: f = bits (b1,b2,b3) of row with b1 moved to the top: rows r,r+1 vs r+2,r+3 flip
// the 64-byte half (column access) while any 16 rows of a b128 lane group still get 8 distinct
// f values x 2 row parities = 16 distinct 16-byte bank slots (row access).
__device__ __forceinline__ int swz_f128(int row) {
    const int y = (row >> 1) & 7;
    return ((y & 1) << 2) | (y & 2) | (y >> 2);
}
__device__ __forceinline__ int swz128(int row, int c) { return row * 128 + ((c ^ swz_f128(row)) << 4); }
__device__ __forceinline__ int swz256(int row, int c) { return row * 256 + ((c ^ (row & 15)) << 4); }

template <int ROWBYTES> __device__ __forceinline__ int swz(int row, int c) {
    if constexpr (ROWBYTES == 128) return swz128(row, c);
    else return swz256(row, c);
}

// Build the register-operand fragment (B operand of P*V-like products) from accumulator
// registers.  Step `s` consumes the k-slots this lane already owns:
//   bf16: regs 8s..8s+7   (keys/rows 16s + 8*(j>>2) + (j&3) + 4*(lane>>5), j = 0..7)
//   f32 : regs 4s..4s+3   (rows 8s + e + 4*(lane>>5), e = 0..3)
template <typename T> __device__ __forceinline__ typename Frag<T>::type acc_frag(const f32x16& p, int s);
template <> __device__ __forceinline__ bf16x8 acc_frag<bf16>(const f32x16& p, int s) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)p[8 * s + j];
    return f;
}
template <> __device__ __forceinline__ f32x4 acc_frag<float>(const f32x16& p, int s) {
    f32x4 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = p[4 * s + e];
    return f;
}
// number of acc_frag steps covering the 32 rows of one accumulator tile
template <typename T> struct AccSteps;
template <> struct AccSteps<bf16> { static constexpr int N = 2; };
template <> struct AccSteps<float> { static constexpr int N = 4; };

// hardware transpose read: 4 x bf16 down a column of a 4x16 row-major block; every lane of a
// 16-lane group passes the address of ITS 8-byte piece (row p>>2, cols (p&3)*4..+4 of the block,
// p = lane&15) and receives block[0..3][p]  (ds_read_b64_tr_b16; guide T10, ck_tile Quad16).
__device__ __forceinline__ bf16x4 lds_tr16(const char* lds_ptr) {
    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4;
    v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) v4*)(lds_ptr));
    bf16x4 o;
    o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
    return o;
}

// The same read as inline asm with an immediate offset.  The compiler cannot tell which LDS bytes the intrinsic
// form touches, so with an LDS-DMA (global_load_lds) in flight it puts s_waitcnt vmcnt(0) in front of every such
// read -- which serialises a prefetch against the reads of the tile being consumed.  This form is invisible
// to that analysis: the CALLER must s_waitcnt lgkmcnt(0) before using the result and must order the read
// against the DMA that filled the tile itself (both kernels that use it do so with explicit waits + barriers).
template <int OFF>
__device__ __forceinline__ bf16x4 lds_tr16_asm(uint32_t lds_addr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
    return __builtin_bit_cast(bf16x4, r);
}
// same, the immediate given as an argument that must fold to a constant after inlining / unrolling
__device__ __forceinline__ bf16x4 lds_tr16_asm_imm(uint32_t lds_addr, int off) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "i"(off));
    return __builtin_bit_cast(bf16x4, r);
}

}  // namespace pa
