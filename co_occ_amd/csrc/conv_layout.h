// Packed-weight layout shared by the forward GEMM (conv3d.hip) and the backward kernels (conv_bwd.hip).
#pragma once
#include "common.h"

#define KC 32      // K chunk (floats)
#define LDS_ST 36  // LDS row stride in floats: 144 B rows keep ds_read_b128 conflict-free
#define NPAD_TO 128

// K-chunks are ordered channel-chunk major, tap minor (c = kc*taps + tap): all 27 taps of one 32-channel
// slice are consumed back to back, so the activation lines they share are re-read from L2 at a reuse
// distance of one tile-slab slice (~1.5 MB) instead of a whole channel sweep.
// Packed weight layout ("fragment-major"): for K-chunk c and 128-column group g,
// one 16 KB block [wn 0..3][q 0..3][lane 0..63][4]: lane (h = lane>>5, li = lane&31) of wave wn holds
// W[n = 128g + 32wn + li][k = 32c' + 8q + 4h + 0..3] -- exactly the B operand of four MFMA k-steps, so
// k_conv2 loads B fragments straight from global memory (no LDS), and k_conv stages the same block.
__host__ __device__ inline size_t wfrag_index(size_t chunk, int ngroups, int n, int kk) {
  const int g = n >> 7, wn = (n >> 5) & 3, li = n & 31, q = kk >> 3, h = (kk >> 2) & 1, e = kk & 3;
  return ((chunk * ngroups + g) * 4 + wn) * 1024 + (size_t)q * 256 + (size_t)(h * 32 + li) * 4 + e;
}

