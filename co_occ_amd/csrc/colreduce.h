// Deterministic per-channel (column) sums over rows [M, C] -- the reductions of training-mode BatchNorm (statistics, dgamma /
// dbeta) and of the bias gradients.  Shared by backward.hip and conv_bwd.hip.
#pragma once
#include "common.h"

// Fast form (C % 4 == 0, C / 4 a divisor of 256: every trunk width): a workgroup takes COL_ROWS rows, its threads are
// (row lane, channel quad) pairs with float4 loads -- every lane of the wave is busy and 64 / (C / 4) rows are in flight per wave
// (the generic kernels below run C threads of 256 with one dependent load chain each: 188 us for the 82 MB of a 128-channel
// backward pass, 36 times per training step).  Sums stay fp64 in a fixed order: row lane r adds rows r, r + R, ... of the block,
// lane 0 adds the lanes' sums in lane order, the final pass adds the blocks' partials 64 at a time, again in a fixed order.
constexpr int COL_ROWS = 64;
__host__ __device__ inline bool col_fast(int C) { return C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0; }

template <int NS>
__device__ __forceinline__ void col_block_reduce(double (&acc)[NS][4], int q, int r, int cq, int C, double* __restrict__ part) {
  __shared__ double sh[256][NS * 4];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[threadIdx.x][k * 4 + e] = acc[k][e];
  __syncthreads();
  if (r == 0) {
    const int R = 256 / q;
    for (int rr = 1; rr < R; ++rr)
#pragma unroll
      for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] += sh[rr * q + cq][k * 4 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < NS; ++k) part[((size_t)blockIdx.x * C + 4 * cq + e) * NS + k] = acc[k][e];
  }
}

// part[block][c][NS] -> out[k][c] = sum over blocks (fp64, fixed order); one workgroup per 4 channels, 64 lanes over the blocks
template <int NS>
__device__ __forceinline__ void col_final(const double* __restrict__ part, int nparts, int C, double (&tot)[NS]) {
  __shared__ double sh[64][4][NS];
  const int c = blockIdx.x * 4 + (threadIdx.x & 3), lane = threadIdx.x >> 2;
  double a[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) a[k] = 0;
  if (c < C)
    for (int p = lane; p < nparts; p += 64)
#pragma unroll
      for (int k = 0; k < NS; ++k) a[k] += part[((size_t)p * C + c) * NS + k];
#pragma unroll
  for (int k = 0; k < NS; ++k) sh[lane][threadIdx.x & 3][k] = a[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NS; ++k) tot[k] = 0;
  if (lane == 0)
    for (int l = 0; l < 64; ++l)
#pragma unroll
      for (int k = 0; k < NS; ++k) tot[k] += sh[l][threadIdx.x & 3][k];
}

typedef float bn_f4 __attribute__((ext_vector_type(4)));
