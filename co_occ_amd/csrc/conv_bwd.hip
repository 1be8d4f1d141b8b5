// Backward of the implicit-GEMM conv family: data gradient, weight gradient and the
// epilogue (folded-BN scale, ReLU, residual) gradient, for frozen-statistics BN (eval-mode BN folded into
// scale/bias, as the forward).
//
//   forward   out[o][n]  = relu( scale[n] * sum_{t,c} in[src(o,t)][c] W[n][c][t] + bias[n] + res[o][n] )
//   epilogue  dpre       = dout * (out > 0);   dres = dpre;   dacc = dpre * scale;   dbias = sum_o dpre
//   dgrad     din[i][c]  = sum_{t,n} dacc[o(i,t)][n] W[n][c][t]     -> the FORWARD kernel (coocc_conv_fwd) on
//             re-packed weights: stride 1 = taps flipped (mode 2); stride 2 = a row table o(i,t) (mode 3)
//   wgrad     dW[n][c][t] = sum_o in[src(o,t)][c] dacc[o][n]       -> k_wgrad below (K = M on the MFMA)
#include "conv_layout.h"
#include "colreduce.h"
#include "conv_k.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ device-side weight packing
// mode 0: w[Cout][Cin][taps] (torch Conv3d)        -> forward pack
// mode 1: w[Cout][taps][Cin] (tap-major Linear)    -> forward pack
// mode 2: dgrad pack, taps flipped:   W'[n' = c][c' = n][t' = taps-1-t] = w[n][c][t]   (stride-1 convs)
// mode 3: dgrad pack, taps unflipped: W'[n' = c][c' = n][t' = t]        = w[n][c][t]   (row-table dgrad)
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, int Cout, int Cin, int taps, int mode,
                                                       int Npad, float* __restrict__ packed) {
  const size_t total = (size_t)Cout * Cin * taps;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int n, c, t;
  if (mode == 1) { c = (int)(i % Cin); size_t r = i / Cin; t = (int)(r % taps); n = (int)(r / taps); }
  else { t = (int)(i % taps); size_t r = i / taps; c = (int)(r % Cin); n = (int)(r / Cin); }
  const float v = w[i];
  if (mode <= 1) packed[wfrag_index((size_t)(c / KC) * taps + t, Npad >> 7, n, c % KC)] = v;
  else {
    const int tt = mode == 2 ? taps - 1 - t : t;
    packed[wfrag_index((size_t)(n / KC) * taps + tt, Npad >> 7, c, n % KC)] = v;
  }
}

extern "C" int64_t coocc_conv_pack_weights_dev(const float* w, int Cout, int Cin, int taps, int mode, float* packed,
                                               void* stream) {
  if (Cout <= 0 || Cin <= 0 || taps <= 0 || mode < 0 || mode > 3) return coocc_set_error(COOCC_EINVAL, "pack_weights_dev: bad args");
  const int N = mode >= 2 ? Cin : Cout, K = mode >= 2 ? Cout : Cin;
  const int kch = (K + KC - 1) / KC, Npad = (N + NPAD_TO - 1) / NPAD_TO * NPAD_TO;
  const int64_t total = (int64_t)taps * kch * Npad * KC;
  if (!packed) return total;
  if (!w) return coocc_set_error(COOCC_EINVAL, "pack_weights_dev: null weights");
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)total, s) != hipSuccess)
    return coocc_set_error(COOCC_EHIP, "pack_weights_dev: memset failed");
  hipLaunchKernelGGL(k_pack_weights, dim3(cdiv((long long)Cout * Cin * taps, 256)), dim3(256), 0, s, w, Cout, Cin, taps, mode,
                     Npad, packed);
  if (hipGetLastError() != hipSuccess) return coocc_set_error(COOCC_EHIP, "pack_weights_dev: launch failed");
  return total;
}

// The same packs for the split-f16 engine (csrc/gemm_h2.hip; layout of core.PackedConv._h2_layout):
// [(K / 32 chunk, tap)][Npad / 32][2 k16 steps][hi | lo][64 lanes][8 f16], lane l of step s holds k = 32 chunk + 16 s + 8 (l >> 5) + 0..7
// of column 32 nt + (l & 31); hi = f16(w), lo = f16((w - hi) 2^11).  Training re-packs from the live parameter every step, so its
// direct (strided, 1x1x1, small-grid) forward and stride-1 dgrad GEMMs can run on the f16 matrix cores like inference's.
// One thread per 16-byte unit of the pack (8 consecutive k of one column, both planes): the stores are whole 16-byte runs, 512
// bytes contiguous per half-wave; the eight strided weight reads hit L2 (a layer's weights are a few MB).  The first version ran
// one thread per WEIGHT with two scattered 2-byte stores each: 21 us per layer, 42 layers per training step.
__global__ __launch_bounds__(256) void k_pack_weights_h2(const float* __restrict__ w, int Cout, int Cin, int taps, int mode, int Npad,
                                                          _Float16* __restrict__ packed, int* __restrict__ flag) {
  const int K = mode >= 2 ? Cout : Cin, N = mode >= 2 ? Cin : Cout;
  // unit index: ((((chunk * taps + tt) * (Npad / 32) + nt) * 2 + s) * 2 + hf) * 32 + li   (plane handled inside)
  const size_t units = (size_t)(K >> 5) * taps * (Npad >> 5) * 2 * 2 * 32;
  size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (u >= units) return;
  const int li = (int)(u & 31); u >>= 5;
  const int hf = (int)(u & 1); u >>= 1;
  const int sidx = (int)(u & 1); u >>= 1;
  const int nt = (int)(u % (Npad >> 5)); u /= (Npad >> 5);
  const int tt = (int)(u % taps); const int chunk = (int)(u / taps);
  const int nn = nt * 32 + li, kk0 = chunk * 32 + sidx * 16 + hf * 8;
  const int t = mode == 2 ? taps - 1 - tt : tt;          // source tap of pack tap tt
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 hi, lo;
  float mx = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (nn < N) {
      const int kk = kk0 + e;
      // GEMM roles: forward K = Cin (c), N = Cout (n); dgrad K = Cout (n), N = Cin (c)
      const int n = mode >= 2 ? kk : nn, c = mode >= 2 ? nn : kk;
      v = mode == 1 ? w[((size_t)n * taps + t) * Cin + c] : w[((size_t)n * Cin + c) * taps + t];
    }
    const _Float16 h = (_Float16)v;
    hi[e] = h;
    lo[e] = (_Float16)((v - (float)h) * 2048.0f);          // exact in fp32: v - hi has at most 13 significant bits
    mx = fmaxf(mx, fabsf(v));
  }
  _Float16* o = packed + ((((size_t)chunk * taps + tt) * (Npad >> 5) + nt) * 2 + sidx) * 2 * 512 + (size_t)(hf * 32 + li) * 8;
  *(h8*)o = hi;
  *(h8*)(o + 512) = lo;
  if (flag && !(mx < 32768.0f)) *(volatile int*)flag = 1;
}

extern "C" int64_t coocc_conv_pack_weights_h2_dev(const float* w, int Cout, int Cin, int taps, int mode, void* packed, void* stream) {
  if (Cout <= 0 || Cin <= 0 || taps <= 0 || mode < 0 || mode > 3) return coocc_set_error(COOCC_EINVAL, "pack_weights_h2_dev: bad args");
  const int N = mode >= 2 ? Cin : Cout, K = mode >= 2 ? Cout : Cin;
  if (K % 32) return coocc_set_error(COOCC_EINVAL, "pack_weights_h2_dev: the GEMM's K (Cin forward, Cout dgrad) must be a multiple of 32");
  const int Npad = (N + NPAD_TO - 1) / NPAD_TO * NPAD_TO;
  const int64_t total_floats = (int64_t)taps * (K / 32) * Npad * 32;          // 4 bytes (hi + lo) per (k, n, tap)
  if (!packed) return total_floats;
  if (!w) return coocc_set_error(COOCC_EINVAL, "pack_weights_h2_dev: null weights");
  hipStream_t s = as_stream(stream);
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  const long long units = (long long)(K / 32) * taps * (Npad / 32) * 2 * 2 * 32;       // padding columns are written (as zeros) too
  hipLaunchKernelGGL(k_pack_weights_h2, dim3(cdiv(units, 256)), dim3(256), 0, s, w, Cout, Cin, taps, mode, Npad, (_Float16*)packed, flag);
  if (hipGetLastError() != hipSuccess) return coocc_set_error(COOCC_EHIP, "pack_weights_h2_dev: launch failed");
  return total_floats;
}

// ------------------------------------------------------------------ row tables
// fwd table  [taps][Mo]: input row read by output voxel o for tap t (or -1: zero padding)
// dgrad table[taps][Mi]: output row o with o*stride - pad + t == i (or -1)
__global__ __launch_bounds__(256) void k_tap_table(int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int ksize,
                                                    int stride, int pad, int dgrad, int32_t* __restrict__ table) {
  const int M = dgrad ? B * Xi * Yi * Zi : B * Xo * Yo * Zo;
  const int taps = ksize * ksize * ksize;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * taps) return;
  const int t = (int)(i / M), m = (int)(i % M);
  const int kw = t % ksize, kh = (t / ksize) % ksize, kd = t / (ksize * ksize);
  int r = -1;
  if (!dgrad) {
    const int z = m % Zo, y = (m / Zo) % Yo, x = (m / (Zo * Yo)) % Xo, b = m / (Zo * Yo * Xo);
    const int ix = x * stride - pad + kd, iy = y * stride - pad + kh, iz = z * stride - pad + kw;
    if ((unsigned)ix < (unsigned)Xi && (unsigned)iy < (unsigned)Yi && (unsigned)iz < (unsigned)Zi)
      r = ((b * Xi + ix) * Yi + iy) * Zi + iz;
  } else {
    const int z = m % Zi, y = (m / Zi) % Yi, x = (m / (Zi * Yi)) % Xi, b = m / (Zi * Yi * Xi);
    const int ox = x + pad - kd, oy = y + pad - kh, oz = z + pad - kw;
    if (ox >= 0 && oy >= 0 && oz >= 0 && ox % stride == 0 && oy % stride == 0 && oz % stride == 0 && ox / stride < Xo &&
        oy / stride < Yo && oz / stride < Zo)
      r = ((b * Xo + ox / stride) * Yo + oy / stride) * Zo + oz / stride;
  }
  table[i] = r;
}

extern "C" int coocc_conv_tap_table(int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int ksize, int stride, int pad,
                                    int dgrad, int32_t* table, void* stream) {
  COOCC_CHECK_ARG(table && B > 0 && Xi > 0 && Yi > 0 && Zi > 0 && Xo > 0 && Yo > 0 && Zo > 0 && ksize > 0 && stride > 0 && pad >= 0,
                  "conv_tap_table: bad args");
  const long long M = dgrad ? (long long)B * Xi * Yi * Zi : (long long)B * Xo * Yo * Zo;
  COOCC_CHECK_ARG(M * ksize * ksize * ksize < (1ll << 31) && (long long)B * Xi * Yi * Zi < (1ll << 31), "conv_tap_table: too large");
  hipLaunchKernelGGL(k_tap_table, dim3(cdiv(M * ksize * ksize * ksize, 256)), dim3(256), 0, as_stream(stream), B, Xi, Yi, Zi, Xo,
                     Yo, Zo, ksize, stride, pad, dgrad, table);
  COOCC_LAUNCH_CHECK("k_tap_table");
  return COOCC_OK;
}

// ------------------------------------------------------------------ epilogue backward
constexpr int AMAX_SLOTS = 64, AMAX_STRIDE = 32;       // coocc_conv_epilogue_bwd_ex: 64 words, 128 bytes apart (8 KB, COOCC_AMAX_WORDS)
// one thread per 4 channels; dbias partials: per-block column sums -> second pass
__global__ __launch_bounds__(256) void k_epilogue_bwd(const float* __restrict__ dout, int dout_stride,
                                                       const float* __restrict__ out, int out_stride,
                                                       const float* __restrict__ scale, int M, int C, int relu,
                                                       float* __restrict__ dacc, int dacc_stride, float* __restrict__ dres,
                                                       int dres_stride, int dres_accumulate, uint32_t* __restrict__ amax_word) {
  const int c4 = (C + 3) >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < (long long)M * c4;
  if (!live && !amax_word) return;
  const int m = live ? (int)(i / c4) : 0, c = live ? (int)(i % c4) * 4 : 0;      // (a dead thread of the last block re-reads row 0, stores nothing)
  if (((C | dout_stride | out_stride | dacc_stride | dres_stride) & 3) == 0) {   // whole rows of dwordx4 (the usual case)
    if (amax_word) {      // only asked for in this form (the training path's rows)
      f32x4 g = *(const f32x4*)(dout + (size_t)m * dout_stride + c);
      if (relu) {
        const f32x4 o = *(const f32x4*)(out + (size_t)m * out_stride + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
      }
      if (dres && live) {
        f32x4* d = (f32x4*)(dres + (size_t)m * dres_stride + c);
        *d = dres_accumulate ? *d + g : g;
      }
      if (scale) g = g * *(const f32x4*)(scale + c);
      if (dacc && live) *(f32x4*)(dacc + (size_t)m * dacc_stride + c) = g;
      if (!live) g = f32x4{0.f, 0.f, 0.f, 0.f};
      // max |dacc| of the workgroup (bit patterns of non-negative floats order like the values), then ONE atomic per workgroup
      // into one of AMAX_SLOTS words 128 bytes apart, and only when a look at the slot says it is needed.  (One word for
      // everybody -- an atomic, or just an L2 read, per wave: 40 k same-address accesses per pass -- cost 3-8 ms per training step.)
      __shared__ uint32_t wmax[4];
      uint32_t a = __float_as_uint(fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(g[3]))));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a = max(a, (uint32_t)__shfl_xor((int)a, off));
      if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = a;
      __syncthreads();
      if (threadIdx.x == 0) {
        a = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        uint32_t* slot = amax_word + (blockIdx.x % AMAX_SLOTS) * AMAX_STRIDE;
        if (a > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, a);
      }
      return;
    }
    f32x4 g = *(const f32x4*)(dout + (size_t)m * dout_stride + c);
    if (relu) {
      const f32x4 o = *(const f32x4*)(out + (size_t)m * out_stride + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if (dres) {
      f32x4* d = (f32x4*)(dres + (size_t)m * dres_stride + c);
      *d = dres_accumulate ? *d + g : g;
    }
    if (dacc) *(f32x4*)(dacc + (size_t)m * dacc_stride + c) = scale ? g * *(const f32x4*)(scale + c) : g;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (c + e >= C) break;
    float g = dout[(size_t)m * dout_stride + c + e];
    if (relu && !(out[(size_t)m * out_stride + c + e] > 0.f)) g = 0.f;
    if (dres) {
      float* d = dres + (size_t)m * dres_stride + c + e;
      *d = dres_accumulate ? *d + g : g;
    }
    if (dacc) dacc[(size_t)m * dacc_stride + c + e] = scale ? g * scale[c + e] : g;
  }
}

// deterministic column sums of dpre = dout * (out > 0): 256 rows per block, then one block sums the partials
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ dout, int dout_stride,
                                                      const float* __restrict__ out, int out_stride, int M, int C, int relu,
                                                      float* __restrict__ part) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int m0 = blockIdx.x * 256, m1 = min(M, m0 + 256);
  float s = 0.f;
  for (int m = m0; m < m1; ++m) {
    float g = dout[(size_t)m * dout_stride + c];
    if (relu && !(out[(size_t)m * out_stride + c] > 0.f)) g = 0.f;
    s += g;
  }
  part[(size_t)blockIdx.x * C + c] = s;
}

__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, int nparts, int C,
                                                       float* __restrict__ dbias, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(size_t)p * C + c];
  dbias[c] = accumulate ? dbias[c] + s : s;
}

__global__ __launch_bounds__(256) void k_colsum_part4(const float* __restrict__ dout, int dout_stride, const float* __restrict__ out,
                                                       int out_stride, int M, int C, int relu, double* __restrict__ part) {
  const int q = C >> 2, cq = threadIdx.x % q, r = threadIdx.x / q, R = 256 / q;
  const int m0 = blockIdx.x * COL_ROWS, m1 = min(M, m0 + COL_ROWS);
  double acc[1][4] = {};
  for (int m = m0 + r; m < m1; m += R) {
    const bn_f4 g4 = *(const bn_f4*)(dout + (size_t)m * dout_stride + 4 * cq);
    bn_f4 y4 = {1.f, 1.f, 1.f, 1.f};
    if (relu) y4 = *(const bn_f4*)(out + (size_t)m * out_stride + 4 * cq);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] += (relu && !(y4[e] > 0.f)) ? 0.f : g4[e];
  }
  col_block_reduce<1>(acc, q, r, cq, C, part);
}

__global__ __launch_bounds__(256) void k_colsum_final4(const double* __restrict__ part, int nparts, int C, float* __restrict__ dbias,
                                                        int accumulate) {
  double t[1];
  col_final<1>(part, nparts, C, t);
  const int c = blockIdx.x * 4 + (threadIdx.x & 3);
  if ((threadIdx.x >> 2) == 0 && c < C) dbias[c] = accumulate ? dbias[c] + (float)t[0] : (float)t[0];
}

// scale2 = {2^k, 2^-k} with amax * 2^k in [target / 2, target) (k = 0 for an all-zero gradient); the slots are left zero for the next pass
__global__ __launch_bounds__(64) void k_amax_scale(uint32_t* __restrict__ amax_word, float target, float* __restrict__ scale2) {
  uint32_t v = amax_word[threadIdx.x * AMAX_STRIDE];
  amax_word[threadIdx.x * AMAX_STRIDE] = 0u;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
  if (threadIdx.x) return;
  const float a = __uint_as_float(v);
  int k = 0;
  if (a > 0.f && a < 3.0e38f) {
    int ea, et;
    (void)frexpf(a, &ea);          // a = m 2^ea, m in [0.5, 1)
    (void)frexpf(target, &et);
    k = min(max(et - ea, -100), 100);
  }
  scale2[0] = ldexpf(1.f, k);
  scale2[1] = ldexpf(1.f, -k);
}

extern "C" int coocc_conv_epilogue_bwd_ex(const float* dout, int dout_stride, const float* out, int out_stride,
                                          const float* scale, int M, int C, int relu, float* dacc, int dacc_stride,
                                          float* dres, int dres_stride, int dres_accumulate, float* dbias,
                                          int dbias_accumulate, float* ws, int64_t ws_floats, uint32_t* amax_word, float* scale2,
                                          float target, void* stream) {
  COOCC_CHECK_ARG(dout && M > 0 && C > 0 && (!relu || out), "conv_epilogue_bwd: bad args");
  COOCC_CHECK_ARG((amax_word == nullptr) == (scale2 == nullptr), "conv_epilogue_bwd_ex: amax_word and scale2 come together");
  COOCC_CHECK_ARG(!amax_word || (dacc && target > 0.f && ((C | dout_stride | out_stride | dacc_stride | dres_stride) & 3) == 0),
                  "conv_epilogue_bwd_ex: the operand scale needs dacc, target > 0 and rows of whole dwordx4");
  hipStream_t s = as_stream(stream);
  if (dacc || dres) {
    hipLaunchKernelGGL(k_epilogue_bwd, dim3(cdiv((long long)M * ((C + 3) / 4), 256)), dim3(256), 0, s, dout, dout_stride, out,
                       out_stride, scale, M, C, relu, dacc, dacc_stride, dres, dres_stride, dres_accumulate, amax_word);
    COOCC_LAUNCH_CHECK("k_epilogue_bwd");
    if (amax_word) {
      hipLaunchKernelGGL(k_amax_scale, dim3(1), dim3(AMAX_SLOTS), 0, s, amax_word, target, scale2);
      COOCC_LAUNCH_CHECK("k_amax_scale");
    }
  }
  if (dbias) {
    const int nparts4 = cdiv(M, COL_ROWS);
    const bool fast = col_fast(C) && dout_stride % 4 == 0 && (!relu || out_stride % 4 == 0) &&
                      (((uintptr_t)dout | (uintptr_t)(relu ? out : nullptr) | (uintptr_t)ws) & 15) == 0 &&
                      ws && ws_floats >= 2 * (int64_t)nparts4 * C;
    if (fast) {          // fp64 partials, every lane busy (colreduce.h)
      hipLaunchKernelGGL(k_colsum_part4, dim3(nparts4), dim3(256), 0, s, dout, dout_stride, out, out_stride, M, C, relu, (double*)ws);
      hipLaunchKernelGGL(k_colsum_final4, dim3(cdiv(C, 4)), dim3(256), 0, s, (const double*)ws, nparts4, C, dbias, dbias_accumulate);
    } else {
      const int nparts = (M + 255) / 256;
      COOCC_CHECK_ARG(ws && ws_floats >= (int64_t)nparts * C, "conv_epilogue_bwd: workspace too small for dbias");
      hipLaunchKernelGGL(k_colsum_part, dim3(nparts, cdiv(C, 256)), dim3(256), 0, s, dout, dout_stride, out, out_stride, M, C,
                         relu, ws);
      hipLaunchKernelGGL(k_colsum_final, dim3(cdiv(C, 256)), dim3(256), 0, s, ws, nparts, C, dbias, dbias_accumulate);
    }
    COOCC_LAUNCH_CHECK("k_colsum");
  }
  return COOCC_OK;
}

extern "C" int coocc_conv_epilogue_bwd(const float* dout, int dout_stride, const float* out, int out_stride,
                                       const float* scale, int M, int C, int relu, float* dacc, int dacc_stride,
                                       float* dres, int dres_stride, int dres_accumulate, float* dbias,
                                       int dbias_accumulate, float* ws, int64_t ws_floats, void* stream) {
  return coocc_conv_epilogue_bwd_ex(dout, dout_stride, out, out_stride, scale, M, C, relu, dacc, dacc_stride, dres, dres_stride,
                                    dres_accumulate, dbias, dbias_accumulate, ws, ws_floats, nullptr, nullptr, 0.f, stream);
}

// ------------------------------------------------------------------ weight gradient
// GEMM view: D[c][n] (Cin x Cout, per tap) = sum_m A[m][c] * Bm[m][n], A = gathered input rows, Bm = dacc rows;
// the reduction index is the voxel m, so both MFMA operands are read along their contiguous channel axis
// straight from global memory (lane (li,h): A[m+h][c0+li], Bm[m+h][n0+li]) -- no LDS, no transposition.
// Workgroup = 4 waves (2 x 2), tile 128 (Cin) x 128 (Cout) for one tap and one M-slice; each wave owns
// 64 x 64 = 2 x 2 MFMA blocks; UNR k-steps (2 voxels each) are loaded ahead of their 4*UNR MFMAs.
// M-slices write partial slabs that k_wgrad_reduce sums in slice order (deterministic) into the torch
// weight layout [Cout][Cin][taps].
constexpr int WG_UNR = 8;

__device__ __forceinline__ float bload_f32(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
constexpr unsigned OOB = 0xFFFFFFFFu;   // beyond num_records: the buffer load returns 0 without a branch

template <bool TABLE>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ in, int in_stride, unsigned in_bytes,
                                                const float* __restrict__ dacc, int dacc_stride, unsigned dacc_bytes,
                                                const int32_t* __restrict__ table, int M, int Cin, int Cout,
                                                int taps, int ctiles, int ntiles, int mslice, float* __restrict__ slabs) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, h = lane >> 5;
  int tile = blockIdx.x;
  const int nt = tile % ntiles; tile /= ntiles;
  const int ct = tile % ctiles; const int t = tile / ctiles;
  const int c0 = ct * 128 + (wave >> 1) * 64, n0 = nt * 128 + (wave & 1) * 64;
  const int mbeg = blockIdx.y * mslice, mend = min(M, mbeg + mslice);
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, in_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_da = __builtin_amdgcn_make_buffer_rsrc((void*)dacc, 0, dacc_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_tb = __builtin_amdgcn_make_buffer_rsrc((void*)(table ? table + (size_t)t * M : nullptr), 0,
                                                                    table ? (unsigned)M * 4u : 0u, 0x00020000);
  // per-lane channel byte offsets + all-ones masks for channels beyond the tensor.  Validity is folded into the
  // offset with shifts and ORs (no compares): a select here makes hipcc sink the address math into divergent
  // branches around every load.
  const unsigned ca0 = (unsigned)(c0 + li) * 4u, ca1 = ca0 + 128u, nb0 = (unsigned)(n0 + li) * 4u, nb1 = nb0 + 128u;
  const unsigned cm0 = (unsigned)((Cin - 1 - (c0 + li)) >> 31), cm1 = (unsigned)((Cin - 1 - (c0 + 32 + li)) >> 31);
  const unsigned nm0 = (unsigned)((Cout - 1 - (n0 + li)) >> 31), nm1 = (unsigned)((Cout - 1 - (n0 + 32 + li)) >> 31);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // three-stage software pipeline: row ids two steps ahead, operands one step ahead of their MFMAs
  constexpr int STEP = 2 * WG_UNR;
  int rows[WG_UNR];
  float a0[WG_UNR][2], b0[WG_UNR][2], a1[WG_UNR][2], b1[WG_UNR][2];   // two statically named operand sets
  auto load_rows = [&](int m) {
#pragma unroll
    for (int u = 0; u < WG_UNR; ++u) {
      const int mm = m + 2 * u + h;
      const int r = TABLE ? __builtin_amdgcn_raw_buffer_load_b32(rs_tb, (int)((unsigned)min(mm, M - 1) * 4u), 0, 0) : mm;
      rows[u] = r | ((mend - 1 - mm) >> 31);          // -1 beyond the slice
    }
  };
  auto load_ops = [&](int m, float (&a)[WG_UNR][2], float (&b)[WG_UNR][2]) {
#pragma unroll
    for (int u = 0; u < WG_UNR; ++u) {
      const int mm = m + 2 * u + h;
      const unsigned rmask = (unsigned)(rows[u] >> 31), mmask = (unsigned)((mend - 1 - mm) >> 31);
      const unsigned ra = (unsigned)rows[u] * (unsigned)in_stride * 4u, rb = (unsigned)mm * (unsigned)dacc_stride * 4u;
      a[u][0] = bload_f32(rs_in, (ra + ca0) | rmask | cm0);
      a[u][1] = bload_f32(rs_in, (ra + ca1) | rmask | cm1);
      b[u][0] = bload_f32(rs_da, (rb + nb0) | mmask | nm0);
      b[u][1] = bload_f32(rs_da, (rb + nb1) | mmask | nm1);
    }
  };
  auto mfmas = [&](const float (&a)[WG_UNR][2], const float (&b)[WG_UNR][2]) {
#pragma unroll
    for (int u = 0; u < WG_UNR; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
  };
  load_rows(mbeg);
  load_ops(mbeg, a0, b0);
  load_rows(mbeg + STEP);
  for (int m = mbeg; m < mend; m += 2 * STEP) {
    load_ops(m + STEP, a1, b1);      // uses the row ids fetched one step ago
    load_rows(m + 2 * STEP);
    mfmas(a0, b0);
    load_ops(m + 2 * STEP, a0, b0);
    load_rows(m + 3 * STEP);
    mfmas(a1, b1);                   // beyond mend these operands are all zeros
  }
  // slab layout [slice][t][Cin][Cout]
  float* sl = slabs + ((size_t)blockIdx.y * taps + t) * (size_t)Cin * Cout;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + j * 32 + li;
        if (c < Cin && n < Cout) sl[(size_t)c * Cout + n] = acc[i][j][r];
      }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slabs, int nslices, int Cin, int Cout,
                                                       int taps, float* __restrict__ dw, int accumulate) {
  const size_t per = (size_t)taps * Cin * Cout;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // index in [t][c][n]
  if (i >= per) return;
  float s = 0.f;
  for (int p = 0; p < nslices; ++p) s += slabs[(size_t)p * per + i];
  const int n = (int)(i % Cout); const size_t r = i / Cout;
  const int c = (int)(r % Cin), t = (int)(r / Cin);
  float* d = dw + ((size_t)n * Cin + c) * taps + t;
  *d = accumulate ? *d + s : s;
}

// The same sum with coalesced stores: a workgroup owns 16 (Cout) x 16 (Cin) weights with all their taps, sums the slices with n
// fastest (64-byte reads), turns the tile in LDS and writes each output channel's 16 * taps floats as one contiguous run -- the
// kernel above scatters 4-byte stores at a stride of Cin * taps floats (47 us per layer, 37 layers per training step).
constexpr int WRED_MAX_TAPS = 27;
__global__ __launch_bounds__(256) void k_wgrad_reduce_tiled(const float* __restrict__ slabs, int nslices, int Cin, int Cout, int taps,
                                                             float* __restrict__ dw, int accumulate) {
  __shared__ float tile[16][16 * WRED_MAX_TAPS + 1];
  const int ln = threadIdx.x & 15, lc = threadIdx.x >> 4;
  const int n0 = blockIdx.x * 16, c0 = blockIdx.y * 16;
  const size_t per = (size_t)taps * Cin * Cout;
  const bool ok = n0 + ln < Cout && c0 + lc < Cin;
  for (int t = 0; t < taps; ++t) {
    float v = 0.f;
    if (ok) {
      const float* q = slabs + ((size_t)t * Cin + c0 + lc) * Cout + n0 + ln;
      for (int p = 0; p < nslices; ++p) v += q[(size_t)p * per];
    }
    tile[ln][lc * taps + t] = v;
  }
  __syncthreads();
  const int run = min(16, Cin - c0) * taps;                 // floats per output channel in this tile
  for (int idx = threadIdx.x; idx < 16 * run; idx += 256) {
    const int nl = idx / run, rem = idx - nl * run;
    if (n0 + nl < Cout) {
      float* d = dw + ((size_t)(n0 + nl) * Cin + c0) * taps + rem;
      *d = accumulate ? *d + tile[nl][rem] : tile[nl][rem];
    }
  }
}

static void launch_wgrad_reduce(const float* slabs, int nslices, int Cin, int Cout, int taps, float* dw, int accumulate, hipStream_t s) {
  if (taps <= WRED_MAX_TAPS)
    hipLaunchKernelGGL(k_wgrad_reduce_tiled, dim3(cdiv(Cout, 16), cdiv(Cin, 16)), dim3(256), 0, s, slabs, nslices, Cin, Cout, taps, dw,
                       accumulate);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv((long long)taps * Cin * Cout, 256)), dim3(256), 0, s, slabs, nslices, Cin, Cout, taps, dw,
                       accumulate);
}

extern "C" int coocc_conv_wgrad(const float* in, int in_rows, int in_stride, const float* dacc, int dacc_stride,
                                const int32_t* table, int M, int Cin, int Cout, int taps, float* dw, int accumulate,
                                float* ws, int64_t ws_floats, void* stream) {
  COOCC_CHECK_ARG(in && dacc && dw && ws && M > 0 && Cin > 0 && Cout > 0 && taps > 0 && in_rows > 0, "conv_wgrad: bad args");
  const unsigned long long in_bytes = (unsigned long long)in_rows * in_stride * 4ull, da_bytes = (unsigned long long)M * dacc_stride * 4ull;
  COOCC_CHECK_ARG(in_bytes < 0xFFFFFF00ull && da_bytes < 0xFFFFFF00ull, "conv_wgrad: operand larger than 4 GB");
  COOCC_CHECK_ARG(table || in_rows >= M, "conv_wgrad: identity rows need in_rows >= M");
  COOCC_CHECK_ARG(table || taps == 1, "conv_wgrad: taps > 1 needs the forward row table (coocc_conv_tap_table)");
  const int ctiles = (Cin + 127) / 128, ntiles = (Cout + 127) / 128;
  const long long tiles = (long long)ctiles * ntiles * taps;
  const int64_t per = (int64_t)taps * Cin * Cout;
  static const int wg_target = getenv("COOCC_WGRAD_WGS") ? atoi(getenv("COOCC_WGRAD_WGS")) : 512;
  long long nslices = (wg_target + tiles - 1) / tiles;                  // ~2 workgroups per CU (the layers left on this kernel are the small ones)
  nslices = std::min<long long>(nslices, (M + 255) / 256);              // >= 256 voxels per slice
  nslices = std::min<long long>(nslices, ws_floats / per);
  COOCC_CHECK_ARG(nslices >= 1, "conv_wgrad: workspace smaller than one weight slab");
  int mslice = (int)((M + nslices - 1) / nslices);
  mslice = (mslice + 2 * WG_UNR - 1) / (2 * WG_UNR) * (2 * WG_UNR);
  nslices = (M + mslice - 1) / mslice;
  hipStream_t s = as_stream(stream);
  if (table)
    hipLaunchKernelGGL(k_wgrad<true>, dim3((unsigned)tiles, (unsigned)nslices), dim3(256), 0, s, in, in_stride,
                       (unsigned)in_bytes, dacc, dacc_stride, (unsigned)da_bytes, table, M, Cin, Cout, taps, ctiles, ntiles,
                       mslice, ws);
  else
    hipLaunchKernelGGL(k_wgrad<false>, dim3((unsigned)tiles, (unsigned)nslices), dim3(256), 0, s, in, in_stride,
                       (unsigned)in_bytes, dacc, dacc_stride, (unsigned)da_bytes, table, M, Cin, Cout, taps, ctiles, ntiles,
                       mslice, ws);
  launch_wgrad_reduce(ws, (int)nslices, Cin, Cout, taps, dw, accumulate, s);
  COOCC_LAUNCH_CHECK("k_wgrad");
  return COOCC_OK;
}

// the second passes, for csrc/wgrad_h2.hip (same slab layouts)
int coocc_wgrad_reduce_launch(const float* slabs, int nslices, int Cin, int Cout, int taps, float* dw, int accumulate, hipStream_t s) {
  launch_wgrad_reduce(slabs, nslices, Cin, Cout, taps, dw, accumulate, s);
  COOCC_LAUNCH_CHECK("k_wgrad_reduce");
  return COOCC_OK;
}

// ------------------------------------------------------------------ Winograd-domain weight gradient
// dU[p][dz][c][n] = sum over the rows of transform point p of V[p][row + dz - 1][c] * dM[p][row][n] (z taps stay direct),
// then dg = G^T dU G.  One k_wgrad<true> launch over all (tile+2)^2 * G rows with taps = 3: the M slices are cut so that
// no slice straddles two points, i.e. slab [p*S + s] holds a partial dU[p]; the reduce kernel sums the S slices of
// each point in order (deterministic) and applies the inverse transform into the torch layout [Cout][Cin][3][3][3].
// 4x (F(4x4)) / 2.25x (F(2x2)) fewer multiplies than the direct wgrad.
__global__ __launch_bounds__(256) void k_wino_ztap_table(long long rows_total, int Z, int32_t* __restrict__ table) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * rows_total) return;
  const int dz = (int)(i / rows_total);
  const long long m = i - (long long)dz * rows_total;
  const int z = (int)(m % Z) + dz - 1;
  table[i] = (z >= 0 && z < Z) ? (int32_t)(m + dz - 1) : -1;
}

extern "C" int coocc_wino_ztap_table(int64_t rows_total, int Z, int32_t* table, void* stream) {
  COOCC_CHECK_ARG(table && rows_total > 0 && rows_total < (1ll << 31) && Z > 0 && rows_total % Z == 0, "wino_ztap_table: bad args");
  hipLaunchKernelGGL(k_wino_ztap_table, dim3(cdiv(3 * rows_total, 256)), dim3(256), 0, as_stream(stream), (long long)rows_total,
                     Z, table);
  COOCC_LAUNCH_CHECK("k_wino_ztap_table");
  return COOCC_OK;
}

__constant__ double c_GW4[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
__constant__ double c_GW5[5][3] = {{1, 0, 0}, {-2. / 9, 2. / 9, -2. / 9}, {1. / 9, 2. / 9, 4. / 9}, {-8. / 9, -4. / 9, -2. / 9}, {0, 0, 1}};
__constant__ double c_GW6[6][3] = {{1, 0, 0}, {1. / 3, 1. / 3, 1. / 3}, {-1. / 3, 1. / 3, -1. / 3}, {-16. / 15, -8. / 15, -4. / 15},
                                   {1. / 15, -2. / 15, 4. / 15}, {0, 0, 1}};

template <int N>
__global__ __launch_bounds__(256) void k_wino_wgrad_reduce(const float* __restrict__ slabs, int S, int Cin, int Cout,
                                                            float* __restrict__ dw, int accumulate) {
  const size_t per = (size_t)3 * Cin * Cout;                 // one slab: [dz][c][n]
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const int n = (int)(i % Cout); const size_t r = i / Cout;
  const int c = (int)(r % Cin), dz = (int)(r / Cin);
  const double (*G)[3] = N == 4 ? c_GW4 : (N == 5 ? c_GW5 : c_GW6);
  double g[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int xi = 0; xi < N; ++xi) {
    double row[3] = {0, 0, 0};                                // sum_eta G[eta][b] dU[xi][eta]
    for (int eta = 0; eta < N; ++eta) {
      float u = 0.f;
      for (int s = 0; s < S; ++s) u += slabs[((size_t)(xi * N + eta) * S + s) * per + i];
#pragma unroll
      for (int b = 0; b < 3; ++b) row[b] += G[eta][b] * (double)u;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] += G[xi][a] * row[b];
  }
  float* d = dw + ((size_t)n * Cin + c) * 27 + dz;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float* q = d + (a * 3 + b) * 3;
      *q = accumulate ? *q + (float)g[a][b] : (float)g[a][b];
    }
}

extern "C" int coocc_wino_wgrad(const float* V, const float* dM, int64_t group_rows, int Z, int Cin, int Cout, int tile,
                                const int32_t* ztap_table, float* dw, int accumulate, float* ws, int64_t ws_floats,
                                void* stream) {
  COOCC_CHECK_ARG(V && dM && dw && ws && ztap_table && group_rows > 0 && Z > 0 && Cin > 0 && Cout > 0, "wino_wgrad: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4 && group_rows % 16 == 0 && group_rows % Z == 0, "wino_wgrad: tile 2..4, group_rows % 16 == 0");
  const int pts = (tile + 2) * (tile + 2);
  const long long M = (long long)pts * group_rows;
  const unsigned long long v_bytes = (unsigned long long)M * Cin * 4ull, m_bytes = (unsigned long long)M * Cout * 4ull;
  COOCC_CHECK_ARG(M < (1ll << 31) && v_bytes < 0xFFFFFF00ull && m_bytes < 0xFFFFFF00ull, "wino_wgrad: operand larger than 4 GB");
  const int ctiles = (Cin + 127) / 128, ntiles = (Cout + 127) / 128;
  const long long tiles = (long long)ctiles * ntiles * 3;
  const int64_t per = (int64_t)3 * Cin * Cout;
  // S slices per transform point: at least one workgroup per CU in total (longer slices beat more of them: 72 -> 77
  // TFLOP/s), >= 256 rows per slice, slice length a multiple of 16
  int S = 1;
  static const int wg_target = getenv("COOCC_WINO_WGRAD_WGS") ? atoi(getenv("COOCC_WINO_WGRAD_WGS")) : 256;
  while (S < 64 && tiles * pts * S < wg_target && group_rows % (2 * S * 16) == 0 && group_rows / (2 * S) >= 256 &&
         (int64_t)pts * 2 * S * per <= ws_floats) S *= 2;
  COOCC_CHECK_ARG((int64_t)pts * S * per <= ws_floats, "wino_wgrad: workspace smaller than (tile+2)^2 weight slabs");
  const int mslice = (int)(group_rows / S);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(k_wgrad<true>, dim3((unsigned)tiles, (unsigned)(pts * S)), dim3(256), 0, s, V, Cin, (unsigned)v_bytes, dM,
                     Cout, (unsigned)m_bytes, ztap_table, (int)M, Cin, Cout, 3, ctiles, ntiles, mslice, ws);
  const dim3 grid(cdiv(per, 256));
  if (tile == 2) hipLaunchKernelGGL(k_wino_wgrad_reduce<4>, grid, dim3(256), 0, s, ws, S, Cin, Cout, dw, accumulate);
  else if (tile == 3) hipLaunchKernelGGL(k_wino_wgrad_reduce<5>, grid, dim3(256), 0, s, ws, S, Cin, Cout, dw, accumulate);
  else hipLaunchKernelGGL(k_wino_wgrad_reduce<6>, grid, dim3(256), 0, s, ws, S, Cin, Cout, dw, accumulate);
  COOCC_LAUNCH_CHECK("k_wino_wgrad");
  return COOCC_OK;
}

int coocc_wino_wgrad_reduce_launch(const float* slabs, int S, int Cin, int Cout, int tile, float* dw, int accumulate, hipStream_t s) {
  const dim3 grid(cdiv((long long)3 * Cin * Cout, 256));
  if (tile == 2) hipLaunchKernelGGL(k_wino_wgrad_reduce<4>, grid, dim3(256), 0, s, slabs, S, Cin, Cout, dw, accumulate);
  else if (tile == 3) hipLaunchKernelGGL(k_wino_wgrad_reduce<5>, grid, dim3(256), 0, s, slabs, S, Cin, Cout, dw, accumulate);
  else hipLaunchKernelGGL(k_wino_wgrad_reduce<6>, grid, dim3(256), 0, s, slabs, S, Cin, Cout, dw, accumulate);
  COOCC_LAUNCH_CHECK("k_wino_wgrad_reduce");
  return COOCC_OK;
}
