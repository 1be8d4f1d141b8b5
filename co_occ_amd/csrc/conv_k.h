// Kernel-side launch record of the implicit-GEMM family (conv3d.hip, gemm_h2.hip) and the shared epilogue.
#pragma once
#include "common.h"
#include "conv_layout.h"

struct ConvK {
  const float* in; const float* w; float* out; const float* scale; const float* bias;
  const float* res; const int32_t* gather; const int32_t* out_rows; float* ws;
  int M, Cin, Cout, Npad, taps, kchunks;
  int in_stride, out_stride, res_stride;
  int Xi, Yi, Zi, Xo, Yo, Zo, stride;
  int kx, ky, kz, px, py, pz;   // per-axis kernel extent / padding (taps = kx*ky*kz, tap index t = (dx*ky + dy)*kz + dz)
  int wgroup_rows;              // > 0: output rows [g*wgroup_rows, (g+1)*wgroup_rows) use weight pack g (Winograd points)
  size_t wgroup_floats;         // floats per weight pack
  int relu, res_mode, iters_per_split, total_iters, splitk;   // relu: 0 none | 1 every column | c >= 4: columns [0, c) only (merged heads)
  int mtiles, ntiles, mtiles_per_xcd;
  size_t in_bytes;              // total input bytes (k_conv2 bases its buffer descriptor at the tile's first row)
  unsigned w_bytes;             // bytes of one weight pack
  const void* zrow;             // k_conv_bf16w / k_gemm_h2z: 16 zero bytes in global memory (source of padded / out-of-range rows)
  float alpha;                  // k_gemm_h2z: accumulators are multiplied by alpha before the epilogue (1 / operand scale)
  const int32_t* M_dev;         // row-table kernels: actual row count on the device (<= M, the grid's capacity), or NULL
  int gstride;                  // row-table kernels: entries per tap of `gather` (>= M)
  int out_h2;                   // split-f16 kernels: write the output rows in H2 format (the next layer's operand)
  void* out16;                  // split-f16 / f16 kernels: second, f16 copy of the output rows, or NULL
  int out16_stride;             // its row stride in f16 elements
  void* out_h2t;                // split-f16 kernels: H2 twin of the fp32 output rows ([rows][Cout] as H2 chunks; Cout % 32 == 0), or NULL
  int32_t* tile_sem;            // split-f16 kernels, split-K: per-(M tile, N tile) arrival counters (zero on entry, left zero): the last
                                // workgroup of a tile sums the slabs in slice order and runs the epilogue -- no k_conv_reduce launch
  int* h2_flag;                 // host-mapped word, OR-ed with 1 when a value written as an H2 / f16 operand leaves the guarded range
  const float* alpha_dev;       // split-f16 kernels: the accumulators are also multiplied by *alpha_dev (the inverse of an operand scale that
                                // was chosen on the device: the gradient operand of training's dgrad GEMMs), or NULL
};

__device__ __forceinline__ float epilogue(const ConvK& p, float v, int n, size_t rrow) {
  if (p.res_mode == 3) v += p.res[rrow * p.res_stride + n];   // raw partial sum of an earlier K-slice pass
  if (p.scale) v *= p.scale[n];
  if (p.bias) v += p.bias[n];
  if (p.res_mode == 1) v += p.res[rrow * p.res_stride + n];
  if (p.relu && (p.relu == 1 || n < p.relu)) v = fmaxf(v, 0.f);     // relu > 1: only columns [0, relu) (a multiple of 4)
  if (p.res_mode == 2) v *= p.res[rrow * p.res_stride + n];
  return v;
}


// gemm_h2.hip: fp32-accurate GEMM on the f16 matrix cores (operands split hi + lo * 2^-11); called by coocc_conv_fwd for mfma_dtype 3
int coocc_launch_h2(ConvK& k, const coocc_conv_desc* d, hipStream_t s);
int coocc_zero_row(const void** out);
int coocc_h2_flag_ptr(int** out);      // gemm_h2.hip: the host-mapped range-guard flag of the 16-bit operand writers
