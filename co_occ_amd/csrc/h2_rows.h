// "H2 rows": fp32 activations as two f16 halves (hi + lo * 2^-11), the operand format of the split-f16 GEMMs (gemm_h2.hip).
// A row of C channels (C % 32 == 0) is C/32 chunks of 128 bytes [32 x f16 hi | 32 x f16 lo].  Shared by every kernel that writes
// them: the GEMM epilogues, the Winograd transforms, the trilinear upsample-add, the conversion pass.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define H2_LO_SCALE 2048.f

__device__ __forceinline__ void split_h2(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * H2_LO_SCALE);
}
// Range guard of everything written as a 16-bit operand: the hi half of an H2 row (and an f16 twin) overflows to inf at
// |v| >= 65520 and the layer that consumes it would emit inf / NaN without any error.  Every H2 / f16 writer compares against
// H2_GUARD (half the f16 range: a margin for the rows a later transform amplifies is taken by the producer's own scale) and
// raises a host-visible flag (a plain store of 1 into host-mapped memory: no PCIe atomics needed, no cost when nothing
// overflows); the host reads it at its next synchronisation point (coocc_h2_overflow; core.check_h2_overflow raises).
#define H2_GUARD 32768.f
__device__ __forceinline__ void h2_guard(int* flag, f32x4 v) {
  const float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  if (flag && !(mx < H2_GUARD)) *(volatile int*)flag = 1;        // !(x < g): NaN counts
}
__device__ __forceinline__ void store_h2(void* out, size_t row, int out_stride, int n, f32x4 v) {
  f16x4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)((v[e] - (float)hi[e]) * H2_LO_SCALE);
  }
  char* o = (char*)out + row * (size_t)out_stride * 4 + (n >> 5) * 128 + (n & 31) * 2;
  *(f16x4*)o = hi;
  *(f16x4*)(o + 64) = lo;
}
// Pair form of store_h2 for kernels whose lanes l, l ^ 1 hold ADJACENT channel quads (n, n + 4; n % 8 == 0 in the even lane) of the
// SAME row and take the same branch: the even lane stores the 16 bytes of hi halves of both, the odd lane the 16 bytes of lo halves
// (one DPP quad_perm exchange of two dwords) -- every store instruction of a wave then writes whole 64-byte runs as 16-byte pieces
// instead of two 8-byte-per-lane instructions.  Same bytes in memory as store_h2.
__device__ __forceinline__ void store_h2_pair(void* out, size_t row, int out_stride, int n, f32x4 v) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f16x4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)((v[e] - (float)hi[e]) * H2_LO_SCALE);
  }
  const bool odd = (n >> 2) & 1;
  const u32x2 mh = __builtin_bit_cast(u32x2, hi), ml = __builtin_bit_cast(u32x2, lo);
  const u32x2 send = odd ? mh : ml;                      // what the partner stores
  u32x2 recv;
  recv[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[0], 0xB1, 0xF, 0xF, false);     // quad_perm [1,0,3,2]: lane ^ 1
  recv[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send[1], 0xB1, 0xF, 0xF, false);
  const u32x4 o = odd ? u32x4{recv[0], recv[1], ml[0], ml[1]} : u32x4{mh[0], mh[1], recv[0], recv[1]};
  char* p = (char*)out + row * (size_t)out_stride * 4 + (n >> 5) * 128 + ((n & 31) & ~7) * 2 + (odd ? 64 : 0);
  *(u32x4*)p = o;
}
// epilogue option out16: a second, f16 copy of the output rows ([rows][out16_stride] f16) -- the operand of the next layer on the
// one-term f16 path, written by the producer instead of a conversion pass
__device__ __forceinline__ void store_f16(void* out16, size_t row, int stride, int n, f32x4 v) {
  f16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
  *(f16x4*)((char*)out16 + (row * (size_t)stride + n) * 2) = o;
}
