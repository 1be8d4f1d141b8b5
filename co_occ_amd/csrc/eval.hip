// On-device semantic evaluation: COOCC_Ray.evaluation_semantic (coocc_ray.py:659-684)
// + fast_hist (:726-730) without the .cpu().numpy() round trip.  One pass over the ground-truth grid:
// trilinear resample of the class logits to the gt size (F.interpolate, align_corners=False), argmax
// (first maximum), and three confusion matrices indexed [label][pred]:
//   SC  2x2  (label != empty, pred != empty)        over gt != 255
//   SSC CxC                                         over gt != 255
//   OCC CxC                                         over gt != 255 and visible != 0   (optional)
// Counts accumulate in LDS per workgroup and are flushed with 64-bit atomics, so a whole validation
// set can be accumulated on the device and read back once.
#include "common.h"

struct Lin { int i0, i1; float w0, w1; };

__device__ __forceinline__ Lin lin_src1(int dst, int in, int out) {
  Lin r;
  if (in == out) { r.i0 = r.i1 = dst; r.w0 = 1.f; r.w1 = 0.f; return r; }
  float scale = (float)in / (float)out;
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  r.i0 = (int)s;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  r.w1 = s - (float)r.i0;
  r.w0 = 1.f - r.w1;
  return r;
}

// logits resampled to voxel i of the [H,W,D] grid (F.interpolate trilinear, align_corners=False), first maximum
__device__ __forceinline__ int resample_argmax(const float* __restrict__ pred, long long sc, long long sx, long long sy,
                                               long long sz, int C, int h, int w, int d, int H, int W, int D, int i) {
  const int z = i % D, y = (i / D) % W, x = i / (D * W);
  const Lin lx = lin_src1(x, h, H), ly = lin_src1(y, w, W), lz = lin_src1(z, d, D);
  const long long o000 = lx.i0 * sx + ly.i0 * sy + lz.i0 * sz, o001 = lx.i0 * sx + ly.i0 * sy + lz.i1 * sz;
  const long long o010 = lx.i0 * sx + ly.i1 * sy + lz.i0 * sz, o011 = lx.i0 * sx + ly.i1 * sy + lz.i1 * sz;
  const long long o100 = lx.i1 * sx + ly.i0 * sy + lz.i0 * sz, o101 = lx.i1 * sx + ly.i0 * sy + lz.i1 * sz;
  const long long o110 = lx.i1 * sx + ly.i1 * sy + lz.i0 * sz, o111 = lx.i1 * sx + ly.i1 * sy + lz.i1 * sz;
  float best = -INFINITY;
  int arg = 0;
  for (int c = 0; c < C; ++c) {
    const float* p = pred + c * sc;
    const float v = lx.w0 * (ly.w0 * (lz.w0 * p[o000] + lz.w1 * p[o001]) + ly.w1 * (lz.w0 * p[o010] + lz.w1 * p[o011])) +
                    lx.w1 * (ly.w0 * (lz.w0 * p[o100] + lz.w1 * p[o101]) + ly.w1 * (lz.w0 * p[o110] + lz.w1 * p[o111]));
    if (v > best || c == 0) { best = v; arg = c; }
  }
  return arg;
}

constexpr int EVAL_MAX_C = 32;

__global__ __launch_bounds__(256) void k_eval_semantic(const float* __restrict__ pred, long long sc, long long sx,
                                                        long long sy, long long sz, int C, int h, int w, int d,
                                                        const uint8_t* __restrict__ gt,
                                                        const uint8_t* __restrict__ visible, int H, int W, int D,
                                                        int empty_idx, unsigned long long* __restrict__ hist) {
  __shared__ unsigned int s_hist[4 + 2 * EVAL_MAX_C * EVAL_MAX_C];
  const int nbins = 4 + 2 * C * C;
  for (int i = threadIdx.x; i < nbins; i += 256) s_hist[i] = 0;
  __syncthreads();
  const int total = H * W * D;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total) {
    const int label = gt[i];
    if (label != 255) {
      const int arg = resample_argmax(pred, sc, sx, sy, sz, C, h, w, d, H, W, D, i);
      atomicAdd(&s_hist[(label != empty_idx ? 2 : 0) + (arg != empty_idx ? 1 : 0)], 1u);
      if (label < C) {
        atomicAdd(&s_hist[4 + label * C + arg], 1u);
        if (visible && visible[i] != 0) atomicAdd(&s_hist[4 + C * C + label * C + arg], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += 256) {
    const unsigned int n = s_hist[b];
    if (n) atomicAdd(&hist[b], (unsigned long long)n);
  }
}

extern "C" int coocc_eval_semantic(const float* pred, int64_t stride_c, int64_t stride_x, int64_t stride_y,
                                   int64_t stride_z, int C, int h, int w, int d, const uint8_t* gt,
                                   const uint8_t* visible, int H, int W, int D, int empty_idx, int accumulate,
                                   int64_t* hist, void* stream) {
  COOCC_CHECK_ARG(pred && gt && hist && C > 0 && C <= EVAL_MAX_C && h > 0 && w > 0 && d > 0 && H > 0 && W > 0 && D > 0,
                  "eval_semantic: bad args (C <= 32)");
  COOCC_CHECK_ARG((long long)H * W * D < (1ll << 31) && empty_idx >= 0 && empty_idx < C, "eval_semantic: sizes");
  hipStream_t s = as_stream(stream);
  if (!accumulate) COOCC_HIP(hipMemsetAsync(hist, 0, sizeof(int64_t) * (size_t)(4 + 2 * C * C), s));
  hipLaunchKernelGGL(k_eval_semantic, dim3(cdiv((size_t)H * W * D, 256)), dim3(256), 0, s, pred, stride_c, stride_x,
                     stride_y, stride_z, C, h, w, d, gt, visible, H, W, D, empty_idx, (unsigned long long*)hist);
  COOCC_LAUNCH_CHECK("k_eval_semantic");
  return COOCC_OK;
}

// Prediction labels for the dump formats (P/coocc/apis/test.py:67-68,198-201: F.interpolate(trilinear) + argmax(dim=1),
// then .astype(np.uint8) in save_output_nuscenes, P/coocc/apis/utils.py:65): one u8 label per voxel of the [H,W,D] grid.
__global__ __launch_bounds__(256) void k_predict_labels(const float* __restrict__ pred, long long sc, long long sx,
                                                         long long sy, long long sz, int C, int h, int w, int d, int H,
                                                         int W, int D, uint8_t* __restrict__ labels) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < H * W * D) labels[i] = (uint8_t)resample_argmax(pred, sc, sx, sy, sz, C, h, w, d, H, W, D, i);
}

extern "C" int coocc_predict_labels(const float* pred, int64_t stride_c, int64_t stride_x, int64_t stride_y,
                                    int64_t stride_z, int C, int h, int w, int d, int H, int W, int D, uint8_t* labels,
                                    void* stream) {
  COOCC_CHECK_ARG(pred && labels && C > 0 && C <= 256 && h > 0 && w > 0 && d > 0 && H > 0 && W > 0 && D > 0,
                  "predict_labels: bad args (C <= 256)");
  COOCC_CHECK_ARG((long long)H * W * D < (1ll << 31), "predict_labels: grid too large");
  hipLaunchKernelGGL(k_predict_labels, dim3(cdiv((size_t)H * W * D, 256)), dim3(256), 0, as_stream(stream), pred, stride_c,
                     stride_x, stride_y, stride_z, C, h, w, d, H, W, D, labels);
  COOCC_LAUNCH_CHECK("k_predict_labels");
  return COOCC_OK;
}
