// OccHead fine-branch MLP chain in one kernel (occ_head.py:70-83, 224-233):
//   y1   = ReLU(GroupNorm16(Linear(128->64)(img_sample)))
//   h    = ReLU(GroupNorm16(Linear(192->64)(cat[voxel_sample, y1])))
//   out  = Linear(64->ncls)(h)
// for every fine point.  The unfused chain moves each 64-wide activation through HBM four times (GEMM out, GroupNorm
// in/out, GEMM in); here a wave owns 64 fine points and only reads their two 128-channel samples and writes ncls logits.
//
// The GEMMs are computed TRANSPOSED, out^T[ch x points] = W[ch x K] . X^T[K x points], with v_mfma_f32_32x32x2_f32:
//   A operand = weights  (lane (li,h) supplies W[32i+li][k]),   straight from the nn.Linear [out,in] array
//   B operand = points   (lane (li,h) supplies X[point li][k]), straight from the row-major sample arrays
//   D[row = (r&3) + 8(r>>2) + 4h][col = li]: lane (li,h) ends up with 16 channels of ITS point, in runs of 4
//   consecutive channels -> (1) a GroupNorm(16, 64) group (4 channels) lives in one lane: no cross-lane traffic;
//   (2) those registers are exactly the B operand of the next layer (k = 8g + 4h + s <-> register 4g + s), so the
//   chain never leaves the register file: no LDS, no barriers.
// K order (8q + 4h + s, q ascending) and the GroupNorm arithmetic are the ones of the unfused path
// (k_conv / k_groupnorm_rows), so both paths produce the same bits.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FineMlp {
  const float* samp; const float* vox; float* out;
  const float* w_img; const float* b_img; const float* g_img; const float* be_img;
  const float* w_f0; const float* b_f0; const float* g_f0; const float* be_f0;
  const float* w_f3; const float* b_f3;
  long long nf;
  const int32_t* n_dev;       // optional device-side count (coarse voxels); nf = min(nf, *n_dev * n_mul)
  int n_mul;
  int samp_stride, vox_stride, ncls;
  float eps_img, eps_f0;
};

__device__ __forceinline__ f32x4 bl4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

__device__ __forceinline__ void mfma_group(f32x16 (&acc)[2][2], const f32x4 (&a)[2], const f32x4 (&b)[2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[t][s], acc[i][t], 0, 0, 0);
}

// acc[i][t] += W[32i + li][kw + 8q + 4h + s] * X[point 32t + li][8q + 4h + s], q < nq (even), both operands from memory,
// one k-group (8 channels) prefetched ahead.  Offsets are bytes; a prefetch past the last group reads the next row or
// out of range (-> 0) and is discarded.
__device__ __forceinline__ void seg_mem(f32x16 (&acc)[2][2], __amdgpu_buffer_rsrc_t rw, unsigned wo0, unsigned wo1,
                                        __amdgpu_buffer_rsrc_t rx, unsigned xo0, unsigned xo1, int nq) {
  f32x4 a0[2], b0[2], a1[2], b1[2];
  a0[0] = bl4(rw, wo0); a0[1] = bl4(rw, wo1);
  b0[0] = bl4(rx, xo0); b0[1] = bl4(rx, xo1);
  for (int q = 0; q < nq; q += 2) {
    unsigned d = (unsigned)(q + 1) * 32u;
    a1[0] = bl4(rw, wo0 + d); a1[1] = bl4(rw, wo1 + d);
    b1[0] = bl4(rx, xo0 + d); b1[1] = bl4(rx, xo1 + d);
    mfma_group(acc, a0, b0);
    d += 32u;
    a0[0] = bl4(rw, wo0 + d); a0[1] = bl4(rw, wo1 + d);
    b0[0] = bl4(rx, xo0 + d); b0[1] = bl4(rx, xo1 + d);
    mfma_group(acc, a1, b1);
  }
}

// bias + GroupNorm (groups of 4 consecutive channels = registers 4g..4g+3) + ReLU, arithmetic of k_groupnorm_rows
__device__ __forceinline__ void bias_gn_relu(f32x16& v, const float* __restrict__ bias, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, float eps, int c0) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 bi = *(const f32x4*)(bias + c0 + 8 * g), ga = *(const f32x4*)(gamma + c0 + 8 * g),
                be = *(const f32x4*)(beta + c0 + 8 * g);
    float x[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = v[4 * g + s] + bi[s];
    float mean = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) mean += x[s];
    mean /= 4.0f;
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) { float d = x[s] - mean; var += d * d; }
    var /= 4.0f;
    const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
    for (int s = 0; s < 4; ++s) v[4 * g + s] = fmaxf((x[s] - mean) * rstd * ga[s] + be[s], 0.f);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <bool PRE>
__global__ __launch_bounds__(256, 2) void k_fine_mlp(FineMlp p) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int li = lane & 31, h = lane >> 5;
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * 64;
  if (p.n_dev) p.nf = min(p.nf, (long long)*p.n_dev * p.n_mul);
  if (r0 >= p.nf) return;
  const unsigned rows = (unsigned)(p.nf - r0 < 64 ? p.nf - r0 : 64);
  // the wave's 64 points: descriptors based at its first row, rows past the end read 0
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.samp + r0 * p.samp_stride), 0,
                                                                      rows * (unsigned)p.samp_stride * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vox + r0 * p.vox_stride), 0,
                                                                      rows * (unsigned)p.vox_stride * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwi = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_img, 0, 64u * 128u * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_f0, 0, 64u * 192u * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw3 =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.w_f3, 0, (unsigned)p.ncls * 64u * 4u, 0x00020000);

  f32x16 y[2][2], acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // D-layout load of a 64-channel row block: register 4g+s of tile j <-> channel 32j + 8g + 4h + s of the lane's point
  auto load_rows64 = [&](f32x16 (&dst)[2][2], __amdgpu_buffer_rsrc_t r, int stride) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = bl4(r, (unsigned)((32 * t + li) * stride + 32 * j + 8 * g + 4 * h) * 4u);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) dst[j][t][4 * g + s_] = v[s_];
        }
  };
  const unsigned w0a = (unsigned)(li * 192 + 4 * h) * 4u, w0b = (unsigned)((32 + li) * 192 + 4 * h) * 4u;
  if (PRE) {
    // both Linear layers that precede a resampling were applied BEFORE it (they commute with the interpolation):
    // samp = bilinear sample of W_img . img features, vox = trilinear sample of W_f0[:, :128] . voxel features
    load_rows64(acc, rs, p.samp_stride);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bias_gn_relu(acc[i][t], p.b_img, p.g_img, p.be_img, p.eps_img, 32 * i + 4 * h);
        y[i][t] = acc[i][t];
      }
    load_rows64(acc, rv, p.vox_stride);          // the accumulators of fine_mlp[0] start from the voxel term
  } else {
    // ---- img_mlp: Linear(128 -> 64) + GN + ReLU
    seg_mem(acc, rwi, (unsigned)(li * 128 + 4 * h) * 4u, (unsigned)((32 + li) * 128 + 4 * h) * 4u, rs,
            (unsigned)(li * p.samp_stride + 4 * h) * 4u, (unsigned)((32 + li) * p.samp_stride + 4 * h) * 4u, 16);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bias_gn_relu(acc[i][t], p.b_img, p.g_img, p.be_img, p.eps_img, 32 * i + 4 * h);
        y[i][t] = acc[i][t];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
      }
    // ---- fine_mlp[0]: Linear(192 -> 64) over cat[voxel sample (128), y1 (64)] + GN + ReLU
    seg_mem(acc, rw0, w0a, w0b, rv, (unsigned)(li * p.vox_stride + 4 * h) * 4u,
            (unsigned)((32 + li) * p.vox_stride + 4 * h) * 4u, 16);
  }
  {
    f32x4 an0 = bl4(rw0, w0a + 128u * 4u), an1 = bl4(rw0, w0b + 128u * 4u);
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // k-group q of y1: channel tile j = q >> 2, run g = q & 3
      const int j = q >> 2, g = q & 3;
      const f32x4 a0 = an0, a1 = an1;
      if (q < 7) {
        const unsigned d = (unsigned)(128 + 8 * (q + 1)) * 4u;
        an0 = bl4(rw0, w0a + d); an1 = bl4(rw0, w0b + d);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], y[j][t][4 * g + s], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], y[j][t][4 * g + s], acc[1][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bias_gn_relu(acc[i][t], p.b_f0, p.g_f0, p.be_f0, p.eps_f0, 32 * i + 4 * h);
      y[i][t] = acc[i][t];
    }

  // ---- fine_mlp[3]: Linear(64 -> ncls <= 32); weight rows >= ncls read 0
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  const unsigned w3 = (unsigned)(li * 64 + 4 * h) * 4u;
  {
    f32x4 an = bl4(rw3, w3);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = q >> 2, g = q & 3;
      const f32x4 a = an;
      if (q < 7) an = bl4(rw3, w3 + (unsigned)(8 * (q + 1)) * 4u);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], y[j][t][4 * g + s], o[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long long row = r0 + 32 * t + li;
    if (row >= p.nf) continue;
    float* dst = p.out + row * p.ncls;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < p.ncls) dst[c] = o[t][r] + p.b_f3[c];
    }
  }
}

extern "C" int coocc_fine_mlp(const float* samp, int samp_stride, const float* vox, int vox_stride, int64_t nfine,
                              const float* w_img, const float* b_img, const float* gn_img_w, const float* gn_img_b,
                              float eps_img, const float* w_f0, const float* b_f0, const float* gn_f0_w,
                              const float* gn_f0_b, float eps_f0, const float* w_f3, const float* b_f3, int ncls,
                              float* out, void* stream) {
  if (nfine == 0) return COOCC_OK;          // nothing selected: empty tensors carry null pointers
  COOCC_CHECK_ARG(samp && vox && out && w_img && b_img && gn_img_w && gn_img_b && w_f0 && b_f0 && gn_f0_w && gn_f0_b &&
                      w_f3 && b_f3, "fine_mlp: null pointer");
  COOCC_CHECK_ARG(nfine >= 0 && ncls >= 1 && ncls <= 32 && samp_stride >= 128 && vox_stride >= 128 &&
                      samp_stride % 4 == 0 && vox_stride % 4 == 0 && samp_stride <= (1 << 20) && vox_stride <= (1 << 20),
                  "fine_mlp: bad args");
  COOCC_CHECK_ARG(((uintptr_t)samp | (uintptr_t)vox | (uintptr_t)w_img | (uintptr_t)w_f0 | (uintptr_t)w_f3 |
                   (uintptr_t)b_img | (uintptr_t)b_f0 | (uintptr_t)gn_img_w | (uintptr_t)gn_img_b | (uintptr_t)gn_f0_w |
                   (uintptr_t)gn_f0_b) % 16 == 0, "fine_mlp: arrays must be 16-byte aligned");
  if (nfine == 0) return COOCC_OK;
  FineMlp p;
  p.samp = samp; p.vox = vox; p.out = out;
  p.w_img = w_img; p.b_img = b_img; p.g_img = gn_img_w; p.be_img = gn_img_b;
  p.w_f0 = w_f0; p.b_f0 = b_f0; p.g_f0 = gn_f0_w; p.be_f0 = gn_f0_b;
  p.w_f3 = w_f3; p.b_f3 = b_f3;
  p.nf = nfine; p.samp_stride = samp_stride; p.vox_stride = vox_stride; p.ncls = ncls;
  p.n_dev = nullptr; p.n_mul = 1;
  p.eps_img = eps_img; p.eps_f0 = eps_f0;
  hipLaunchKernelGGL(k_fine_mlp<false>, dim3(cdiv(nfine, 256)), dim3(256), 0, as_stream(stream), p);
  COOCC_LAUNCH_CHECK("k_fine_mlp");
  return COOCC_OK;
}

static int fine_mlp_pre_impl(const float* samp64, int samp_stride, const float* vox64, int vox_stride, int64_t nfine,
                             const int32_t* n_dev, int n_mul, const float* b_img, const float* gn_img_w, const float* gn_img_b,
                             float eps_img, const float* w_f0, const float* b_f0, const float* gn_f0_w, const float* gn_f0_b,
                             float eps_f0, const float* w_f3, const float* b_f3, int ncls, float* out, void* stream);
extern "C" int coocc_fine_mlp_pre(const float* samp64, int samp_stride, const float* vox64, int vox_stride, int64_t nfine,
                                  const float* b_img, const float* gn_img_w, const float* gn_img_b, float eps_img,
                                  const float* w_f0, const float* b_f0, const float* gn_f0_w, const float* gn_f0_b,
                                  float eps_f0, const float* w_f3, const float* b_f3, int ncls, float* out, void* stream) {
  return fine_mlp_pre_impl(samp64, samp_stride, vox64, vox_stride, nfine, nullptr, 1, b_img, gn_img_w, gn_img_b, eps_img, w_f0, b_f0,
                           gn_f0_w, gn_f0_b, eps_f0, w_f3, b_f3, ncls, out, stream);
}
extern "C" int coocc_fine_mlp_pre_dev(const float* samp64, int samp_stride, const float* vox64, int vox_stride, int64_t nfine_cap,
                                      const int32_t* n_dev, int n_mul, const float* b_img, const float* gn_img_w,
                                      const float* gn_img_b, float eps_img, const float* w_f0, const float* b_f0,
                                      const float* gn_f0_w, const float* gn_f0_b, float eps_f0, const float* w_f3, const float* b_f3,
                                      int ncls, float* out, void* stream) {
  COOCC_CHECK_ARG(n_dev && n_mul > 0, "fine_mlp_pre_dev: null device count");
  return fine_mlp_pre_impl(samp64, samp_stride, vox64, vox_stride, nfine_cap, n_dev, n_mul, b_img, gn_img_w, gn_img_b, eps_img, w_f0,
                           b_f0, gn_f0_w, gn_f0_b, eps_f0, w_f3, b_f3, ncls, out, stream);
}
static int fine_mlp_pre_impl(const float* samp64, int samp_stride, const float* vox64, int vox_stride, int64_t nfine,
                             const int32_t* n_dev, int n_mul, const float* b_img, const float* gn_img_w, const float* gn_img_b,
                             float eps_img, const float* w_f0, const float* b_f0, const float* gn_f0_w, const float* gn_f0_b,
                             float eps_f0, const float* w_f3, const float* b_f3, int ncls, float* out, void* stream) {
  if (nfine == 0) return COOCC_OK;
  COOCC_CHECK_ARG(samp64 && vox64 && out && b_img && gn_img_w && gn_img_b && w_f0 && b_f0 && gn_f0_w && gn_f0_b && w_f3 && b_f3,
                  "fine_mlp_pre: null pointer");
  COOCC_CHECK_ARG(nfine >= 0 && ncls >= 1 && ncls <= 32 && samp_stride >= 64 && vox_stride >= 64 && samp_stride % 4 == 0 &&
                      vox_stride % 4 == 0 && samp_stride <= (1 << 20) && vox_stride <= (1 << 20), "fine_mlp_pre: bad args");
  COOCC_CHECK_ARG(((uintptr_t)samp64 | (uintptr_t)vox64 | (uintptr_t)w_f0 | (uintptr_t)w_f3 | (uintptr_t)b_img | (uintptr_t)b_f0 |
                   (uintptr_t)gn_img_w | (uintptr_t)gn_img_b | (uintptr_t)gn_f0_w | (uintptr_t)gn_f0_b) % 16 == 0,
                  "fine_mlp_pre: arrays must be 16-byte aligned");
  if (nfine == 0) return COOCC_OK;
  FineMlp p;
  p.samp = samp64; p.vox = vox64; p.out = out;
  p.w_img = nullptr; p.b_img = b_img; p.g_img = gn_img_w; p.be_img = gn_img_b;
  p.w_f0 = w_f0; p.b_f0 = b_f0; p.g_f0 = gn_f0_w; p.be_f0 = gn_f0_b;
  p.w_f3 = w_f3; p.b_f3 = b_f3;
  p.nf = nfine; p.samp_stride = samp_stride; p.vox_stride = vox_stride; p.ncls = ncls;
  p.n_dev = n_dev; p.n_mul = n_mul;
  p.eps_img = eps_img; p.eps_f0 = eps_f0;
  hipLaunchKernelGGL(k_fine_mlp<true>, dim3(cdiv(nfine, 256)), dim3(256), 0, as_stream(stream), p);
  COOCC_LAUNCH_CHECK("k_fine_mlp<pre>");
  return COOCC_OK;
}
