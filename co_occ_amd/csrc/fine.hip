// OccHead cascade ("fine") branch, occ_head.py:173-237 (C4): occupied coarse voxels ->
// ratio^3 fine coordinates -> trilinear sample of the mixed voxel features + bilinear
// sample of the 6 camera feature maps -> small MLPs (GEMMs in conv3d.hip) with GroupNorm.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// coarse_occ.argmax(1) != empty_idx (occ_head.py:182); torch.argmax keeps the first maximum
__global__ __launch_bounds__(256) void k_argmax_flags(const float* __restrict__ logits, int V, int ncls, int stride,
                                                       int empty_idx, uint8_t* __restrict__ flags) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const float* r = logits + (size_t)v * stride;
  float best = r[0];
  int bi = 0;
  for (int c = 1; c < ncls; ++c)
    if (r[c] > best) { best = r[c]; bi = c; }
  flags[v] = bi != empty_idx;
}

extern "C" int coocc_argmax_flags(const float* logits, int V, int ncls, int stride, int empty_idx, uint8_t* flags,
                                  void* stream) {
  COOCC_CHECK_ARG(logits && flags && V > 0 && ncls > 0 && stride >= ncls, "argmax_flags: bad args");
  hipLaunchKernelGGL(k_argmax_flags, dim3(cdiv(V, 256)), dim3(256), 0, as_stream(stream), logits, V, ncls, stride,
                     empty_idx, flags);
  COOCC_LAUNCH_CHECK("k_argmax_flags");
  return COOCC_OK;
}

// coarse_to_fine_coordinates (coordinate_transform.py:3-21, eval branch) + the voxel
// grid_sample of occ_head.py:205-214.  Fine point f = o*n + i (offset-major), offset
// o = (a*r + b)*r + c.  Normalised coordinate g = (fine/(final-1) - 0.5)*2; the sampled
// volume is out_voxel_feats.permute(0,1,4,3,2) so grid x walks our X axis, y -> Y, z -> Z.
// grid_sample(bilinear, zeros, align_corners=False): pix = ((g+1)*size - 1)/2.
// One wave per fine point, two channels per lane per step.
__global__ COOCC_SCALAR_FP32 __launch_bounds__(256) void k_fine_sample_voxel(const float* __restrict__ vol, int C, int X, int Y, int Z,
                                                            const int32_t* __restrict__ coarse_lin, int n, int ratio,
                                                            float fx1, float fy1, float fz1,
                                                            int64_t* __restrict__ fine_xyz, float* __restrict__ feat,
                                                            int out_stride, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);
  const int r3 = ratio * ratio * ratio;
  const long long nf = (long long)n * r3;
  const long long f = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (f >= nf) return;
  const int o = (int)(f / n), i = (int)(f - (long long)o * n);
  int l = coarse_lin[i];
  const int cz = l % Z; l /= Z;
  const int cy = l % Y; const int cx = l / Y;   // B == 1
  const int oc = o % ratio, ob = (o / ratio) % ratio, oa = o / (ratio * ratio);
  const int qx = cx * ratio + oa, qy = cy * ratio + ob, qz = cz * ratio + oc;
  if (lane == 0) { fine_xyz[f] = qx; fine_xyz[nf + f] = qy; fine_xyz[2 * nf + f] = qz; }
  float gx = ((float)qx / fx1 - 0.5f) * 2.f, gy = ((float)qy / fy1 - 0.5f) * 2.f, gz = ((float)qz / fz1 - 0.5f) * 2.f;
  float px = ((gx + 1.f) * (float)X - 1.f) / 2.f, py = ((gy + 1.f) * (float)Y - 1.f) / 2.f,
        pz = ((gz + 1.f) * (float)Z - 1.f) / 2.f;
  float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
  int x0 = (int)flx, y0 = (int)fly, z0 = (int)flz;
  float tx = px - flx, ty = py - fly, tz = pz - flz;
  float wx[2] = {1.f - tx, tx}, wy[2] = {1.f - ty, ty}, wz[2] = {1.f - tz, tz};
  for (int c = lane * 2; c < C; c += 128) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          int x = x0 + a, y = y0 + b, z = z0 + d;
          if ((unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z) {
            f32x2 v = *(const f32x2*)(vol + (((size_t)x * Y + y) * Z + z) * C + c);
            acc = acc + v * (wx[a] * wy[b] * wz[d]);
          }
        }
    *(f32x2*)(feat + (size_t)f * out_stride + c) = acc;
  }
}

// ratio == 2: one wave per COARSE voxel computes its 8 children together.  Their 8-corner stencils overlap in a
// 3x3x3 neighbourhood, which is read once (27 rows instead of 64), and the volume is streamed once instead of once
// per child offset (the offset-major point order made the one-wave-per-point kernel re-read it 8 times: 808 MB
// of HBM fetch for a 41 MB volume).  Output order is unchanged (f = o*n + i).
__global__ __launch_bounds__(256, 4) void k_fine_sample_voxel_r2(const float* __restrict__ vol, int C, int X, int Y, int Z,
                                                               const int32_t* __restrict__ coarse_lin, int n,
                                                               float fx1, float fy1, float fz1,
                                                               int64_t* __restrict__ fine_xyz, float* __restrict__ feat,
                                                               int out_stride, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);
  // C <= 64: two channels per lane fill only half a wave, so each half-wave takes its own coarse voxel (no cross-lane
  // operation below: every lane derives the voxel's stencil itself)
  const int wv = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const bool halfw = C <= 64;
  const int i = halfw ? wv * 2 + (int)((threadIdx.x & 63) >> 5) : wv;
  const int lane = halfw ? (threadIdx.x & 31) : (threadIdx.x & 63);
  const int cstep = halfw ? 64 : 128;
  if (i >= n) return;
  const long long nf = (long long)n * 8;
  int l = coarse_lin[i];
  const int cz = l % Z; l /= Z;
  const int cy = l % Y; const int cx = l / Y;   // B == 1
  // per axis and child bit: base index and fraction, with the expressions of k_fine_sample_voxel
  int i0[3][2]; float t[3][2];
  const int cc[3] = {cx, cy, cz}; const float f1[3] = {fx1, fy1, fz1}; const int S[3] = {X, Y, Z};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int q = cc[ax] * 2 + a;
      const float g = ((float)q / f1[ax] - 0.5f) * 2.f;
      const float p = ((g + 1.f) * (float)S[ax] - 1.f) / 2.f;
      const float fl = floorf(p);
      i0[ax][a] = (int)fl; t[ax][a] = p - fl;
    }
  if (lane < 8) {
    const int oa = lane >> 2, ob = (lane >> 1) & 1, oc = lane & 1;
    const long long f = (long long)lane * n + i;
    fine_xyz[f] = cx * 2 + oa; fine_xyz[nf + f] = cy * 2 + ob; fine_xyz[2 * nf + f] = cz * 2 + oc;
  }
  // tap weight of window position x for a child with base index b0 and fraction t
  auto tapw = [](int b0, float t, int x) { return (x == b0 ? 1.f - t : 0.f) + (x == b0 + 1 ? t : 0.f); };
  const int wx0 = min(i0[0][0], i0[0][1]), wy0 = min(i0[1][0], i0[1][1]), wz0 = min(i0[2][0], i0[2][1]);
  for (int c = lane * 2; c < C; c += cstep) {
    f32x2 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = f32x2{0.f, 0.f};
    // runtime loops on purpose: fully unrolled, hipcc hoists 216 weight products and spills
#pragma unroll 1
    for (int kx = 0; kx < 3; ++kx) {
      const int x = wx0 + kx;
      if ((unsigned)x >= (unsigned)X) continue;
      const float ax0 = tapw(i0[0][0], t[0][0], x), ax1 = tapw(i0[0][1], t[0][1], x);
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) {
        const int y = wy0 + ky;
        if ((unsigned)y >= (unsigned)Y) continue;
        const float by0 = tapw(i0[1][0], t[1][0], y), by1 = tapw(i0[1][1], t[1][1], y);
        const float w00 = ax0 * by0, w01 = ax0 * by1, w10 = ax1 * by0, w11 = ax1 * by1;
        const float* rowp = vol + (((size_t)x * Y + y) * Z) * C + c;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          const int z = wz0 + kz;
          if ((unsigned)z >= (unsigned)Z) continue;
          const float cz0 = tapw(i0[2][0], t[2][0], z), cz1 = tapw(i0[2][1], t[2][1], z);
          const f32x2 v = *(const f32x2*)(rowp + (size_t)z * C);
          acc[0] = acc[0] + v * (w00 * cz0); acc[1] = acc[1] + v * (w00 * cz1);
          acc[2] = acc[2] + v * (w01 * cz0); acc[3] = acc[3] + v * (w01 * cz1);
          acc[4] = acc[4] + v * (w10 * cz0); acc[5] = acc[5] + v * (w10 * cz1);
          acc[6] = acc[6] + v * (w11 * cz0); acc[7] = acc[7] + v * (w11 * cz1);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) *(f32x2*)(feat + ((size_t)o * n + i) * out_stride + c) = acc[o];
  }
}

// ratio R > 2 (OpenOccupancy: cascade_ratio 4, 64 children): same idea, one (half-)wave per coarse voxel.  With
// final == R * coarse a child's base index along an axis is c-1 (a < R/2) or c (p = q*S/(R*S-1) - 1/2), so all R^3 stencils
// still live in the 3x3x3 neighbourhood.  The R x-children are walked one after the other with R*R accumulators each; the
// window rows are re-read from L1 (R x 18 row loads per coarse voxel instead of 8 R^3, and the volume is streamed from HBM
// once instead of R^3 times).  Same tap weights, products and accumulation order as k_fine_sample_voxel (zero-weight window
// positions add v * 0).
template <int R>
__global__ __launch_bounds__(256) void k_fine_sample_voxel_rn(const float* __restrict__ vol, int C, int X, int Y, int Z,
                                                               const int32_t* __restrict__ coarse_lin, int n,
                                                               float fx1, float fy1, float fz1,
                                                               int64_t* __restrict__ fine_xyz, float* __restrict__ feat,
                                                               int out_stride, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);
  constexpr int R3 = R * R * R;
  const int wv = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const bool halfw = C <= 64;
  const int i = halfw ? wv * 2 + (int)((threadIdx.x & 63) >> 5) : wv;
  const int lane = halfw ? (threadIdx.x & 31) : (threadIdx.x & 63);
  const int nl = halfw ? 32 : 64;
  const int cstep = 2 * nl;
  if (i >= n) return;
  const long long nf = (long long)n * R3;
  int l = coarse_lin[i];
  const int cz = l % Z; l /= Z;
  const int cy = l % Y; const int cx = l / Y;   // B == 1
  int i0[3][R]; float t[3][R];
  const int cc[3] = {cx, cy, cz}; const float f1[3] = {fx1, fy1, fz1}; const int S[3] = {X, Y, Z};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int a = 0; a < R; ++a) {
      const int q = cc[ax] * R + a;
      const float g = ((float)q / f1[ax] - 0.5f) * 2.f;
      const float p = ((g + 1.f) * (float)S[ax] - 1.f) / 2.f;
      const float fl = floorf(p);
      i0[ax][a] = (int)fl; t[ax][a] = p - fl;
    }
  for (int o = lane; o < R3; o += nl) {
    const int oa = o / (R * R), ob = (o / R) % R, oc = o % R;
    const long long f = (long long)o * n + i;
    fine_xyz[f] = cx * R + oa; fine_xyz[nf + f] = cy * R + ob; fine_xyz[2 * nf + f] = cz * R + oc;
  }
  auto tapw = [](int b0, float t, int x) { return (x == b0 ? 1.f - t : 0.f) + (x == b0 + 1 ? t : 0.f); };
  int wy0 = i0[1][0], wz0 = i0[2][0];
#pragma unroll
  for (int a = 1; a < R; ++a) { wy0 = min(wy0, i0[1][a]); wz0 = min(wz0, i0[2][a]); }
  for (int c = lane * 2; c < C; c += cstep) {
#pragma unroll 1
    for (int a = 0; a < R; ++a) {
      // runtime-indexed copies of this x-child's base / fraction (R is small: select chain instead of scratch)
      int bx = i0[0][0]; float txa = t[0][0];
#pragma unroll
      for (int k = 1; k < R; ++k) if (a == k) { bx = i0[0][k]; txa = t[0][k]; }
      f32x2 acc[R * R];
#pragma unroll
      for (int o = 0; o < R * R; ++o) acc[o] = f32x2{0.f, 0.f};
#pragma unroll 1
      for (int kx = 0; kx < 2; ++kx) {
        const int x = bx + kx;
        if ((unsigned)x >= (unsigned)X) continue;
        const float wxa = kx ? txa : 1.f - txa;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
          const int y = wy0 + ky;
          if ((unsigned)y >= (unsigned)Y) continue;
          float wxy[R];
#pragma unroll
          for (int b = 0; b < R; ++b) wxy[b] = wxa * tapw(i0[1][b], t[1][b], y);
          const float* rowp = vol + (((size_t)x * Y + y) * Z) * C + c;
#pragma unroll
          for (int kz = 0; kz < 3; ++kz) {
            const int z = wz0 + kz;
            if ((unsigned)z >= (unsigned)Z) continue;
            const f32x2 v = *(const f32x2*)(rowp + (size_t)z * C);
#pragma unroll
            for (int d = 0; d < R; ++d) {
              const float czd = tapw(i0[2][d], t[2][d], z);
#pragma unroll
              for (int b = 0; b < R; ++b) acc[b * R + d] = acc[b * R + d] + v * (wxy[b] * czd);
            }
          }
        }
      }
#pragma unroll
      for (int o = 0; o < R * R; ++o)
        *(f32x2*)(feat + ((size_t)(a * R * R + o) * n + i) * out_stride + c) = acc[o];
    }
  }
}

static int g_fine_pointwise = 0;   // test hook: 1 = always the one-wave-per-fine-point kernels
extern "C" void coocc_fine_set_pointwise(int on) { g_fine_pointwise = on; }

static int fine_sample_voxel_impl(const float* vol, int C, int X, int Y, int Z, const int32_t* coarse_lin, int n, const int32_t* n_dev,
                                  int ratio, const int* final_size_host, int64_t* fine_xyz, float* feat, int out_stride, void* stream);
extern "C" int coocc_fine_sample_voxel(const float* vol, int C, int X, int Y, int Z, const int32_t* coarse_lin, int n,
                                       int ratio, const int* final_size_host, int64_t* fine_xyz, float* feat,
                                       int out_stride, void* stream) {
  return fine_sample_voxel_impl(vol, C, X, Y, Z, coarse_lin, n, nullptr, ratio, final_size_host, fine_xyz, feat, out_stride, stream);
}
extern "C" int coocc_fine_sample_voxel_dev(const float* vol, int C, int X, int Y, int Z, const int32_t* coarse_lin, int n_cap,
                                           const int32_t* n_dev, int ratio, const int* final_size_host, int64_t* fine_xyz,
                                           float* feat, int out_stride, void* stream) {
  COOCC_CHECK_ARG(n_dev, "fine_sample_voxel_dev: null device count");
  return fine_sample_voxel_impl(vol, C, X, Y, Z, coarse_lin, n_cap, n_dev, ratio, final_size_host, fine_xyz, feat, out_stride, stream);
}
static int fine_sample_voxel_impl(const float* vol, int C, int X, int Y, int Z, const int32_t* coarse_lin, int n, const int32_t* n_dev,
                                  int ratio, const int* final_size_host, int64_t* fine_xyz, float* feat, int out_stride, void* stream) {
  COOCC_CHECK_ARG(vol && coarse_lin && final_size_host && fine_xyz && feat && C % 2 == 0 && ratio >= 1 && n >= 0,
                  "fine_sample_voxel: bad args");
  if (n == 0) return COOCC_OK;
  long long nf = (long long)n * ratio * ratio * ratio;
  // the 3-wide window of the grouped kernel needs floor(p) of the two children of an axis to differ by <= 1:
  // true for final == ratio * coarse (p = q*S/(2S-1) - 1/2)
  if (!g_fine_pointwise && ratio == 2 && final_size_host[0] == 2 * X && final_size_host[1] == 2 * Y && final_size_host[2] == 2 * Z) {
    hipLaunchKernelGGL(k_fine_sample_voxel_r2, dim3(cdiv((long long)(C <= 64 ? (n + 1) / 2 : n) * 64, 256)), dim3(256), 0, as_stream(stream), vol, C, X,
                       Y, Z, coarse_lin, n, (float)(final_size_host[0] - 1), (float)(final_size_host[1] - 1),
                       (float)(final_size_host[2] - 1), fine_xyz, feat, out_stride, n_dev);
    COOCC_LAUNCH_CHECK("k_fine_sample_voxel_r2");
    return COOCC_OK;
  }
  if (!g_fine_pointwise && ratio == 4 && final_size_host[0] == 4 * X && final_size_host[1] == 4 * Y && final_size_host[2] == 4 * Z) {
    hipLaunchKernelGGL(k_fine_sample_voxel_rn<4>, dim3(cdiv((long long)(C <= 64 ? (n + 1) / 2 : n) * 64, 256)), dim3(256), 0,
                       as_stream(stream), vol, C, X, Y, Z, coarse_lin, n, (float)(final_size_host[0] - 1),
                       (float)(final_size_host[1] - 1), (float)(final_size_host[2] - 1), fine_xyz, feat, out_stride, n_dev);
    COOCC_LAUNCH_CHECK("k_fine_sample_voxel_rn");
    return COOCC_OK;
  }
  hipLaunchKernelGGL(k_fine_sample_voxel, dim3(cdiv(nf * 64, 256)), dim3(256), 0, as_stream(stream), vol, C, X, Y, Z,
                     coarse_lin, n, ratio, (float)(final_size_host[0] - 1), (float)(final_size_host[1] - 1),
                     (float)(final_size_host[2] - 1), fine_xyz, feat, out_stride, n_dev);
  COOCC_LAUNCH_CHECK("k_fine_sample_voxel");
  return COOCC_OK;
}

// project_points_on_img (coordinate_transform.py:25-65, nuScenes branch) fused with the
// per-camera bilinear grid_sample(align_corners=True, zeros) * mask, summed over cameras
// (occ_head.py:222-234).  params layout (floats):
//   [0:9] inv(bda)  [9:12] voxel_size  [12:15] range_lo  [15] W_img-1  [16] H_img-1
//   then per camera 27: inv(rots)[9], trans[3], intrins[9], post_rots[:2,:2][4], post_trans[:2][2]
#define FINE_CAM_STRIDE 27
#define FINE_HDR 17
__global__ COOCC_SCALAR_FP32 __launch_bounds__(256) void k_fine_sample_img(const float* __restrict__ img, int ncam, int Ci, int Hf, int Wf,
                                                          const float* __restrict__ prm,
                                                          const int64_t* __restrict__ fine_xyz, long long nf,
                                                          float* __restrict__ feat, int out_stride,
                                                          const int32_t* __restrict__ n_dev, int n_mul) {
  if (n_dev) nf = min(nf, (long long)*n_dev * n_mul);
  const long long f = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (f >= nf) return;
  float p0 = (float)fine_xyz[f] * prm[9] + prm[12];
  float p1 = (float)fine_xyz[nf + f] * prm[10] + prm[13];
  float p2 = (float)fine_xyz[2 * nf + f] * prm[11] + prm[14];
  float bx = prm[0] * p0 + prm[1] * p1 + prm[2] * p2;
  float by = prm[3] * p0 + prm[4] * p1 + prm[5] * p2;
  float bz = prm[6] * p0 + prm[7] * p1 + prm[8] * p2;
  const float wimg1 = prm[15], himg1 = prm[16];
  f32x2 acc[4];
  const int nstep = (Ci + 127) / 128;  // <= 4 steps (Ci <= 512)
#pragma unroll
  for (int s = 0; s < 4; ++s) acc[s] = f32x2{0.f, 0.f};
  for (int cam = 0; cam < ncam; ++cam) {
    const float* q = prm + FINE_HDR + cam * FINE_CAM_STRIDE;
    float tx = bx - q[9], ty = by - q[10], tz = bz - q[11];
    float cx = q[0] * tx + q[1] * ty + q[2] * tz;
    float cy = q[3] * tx + q[4] * ty + q[5] * tz;
    float cz = q[6] * tx + q[7] * ty + q[8] * tz;
    float ix = q[12] * cx + q[13] * cy + q[14] * cz;
    float iy = q[15] * cx + q[16] * cy + q[17] * cz;
    float d = q[18] * cx + q[19] * cy + q[20] * cz;
    float u = ix / (d + 1e-5f), v = iy / (d + 1e-5f);
    float u2 = q[21] * u + q[22] * v + q[25];
    float v2 = q[23] * u + q[24] * v + q[26];
    u2 = (u2 / wimg1 - 0.5f) * 2.f;
    v2 = (v2 / himg1 - 0.5f) * 2.f;
    bool m = d > 1e-5f && u2 > -1.f && u2 < 1.f && v2 > -1.f && v2 < 1.f;
    if (!m) continue;
    float px = (u2 + 1.f) / 2.f * (float)(Wf - 1), py = (v2 + 1.f) / 2.f * (float)(Hf - 1);
    float flx = floorf(px), fly = floorf(py);
    int x0 = (int)flx, y0 = (int)fly;
    float ax = px - flx, ay = py - fly;
    const float* base = img + (size_t)cam * Hf * Wf * Ci;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int c = s * 128 + lane * 2;
      if (s >= nstep || c >= Ci) continue;
      f32x2 a = acc[s];
#pragma unroll
      for (int yy = 0; yy < 2; ++yy)
#pragma unroll
        for (int xx = 0; xx < 2; ++xx) {
          int x = x0 + xx, y = y0 + yy;
          if ((unsigned)x < (unsigned)Wf && (unsigned)y < (unsigned)Hf) {
            float w = (xx ? ax : 1.f - ax) * (yy ? ay : 1.f - ay);
            f32x2 val = *(const f32x2*)(base + ((size_t)y * Wf + x) * Ci + c);
            a = a + val * w;
          }
        }
      acc[s] = a;
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    int c = s * 128 + lane * 2;
    if (s < nstep && c < Ci) *(f32x2*)(feat + (size_t)f * out_stride + c) = acc[s];
  }
}

// Parameter block of k_fine_sample_img built on the device: the 3x3 inverses (adjugate in fp64) and the packing
// that the host used to do with ~20 tiny torch launches and a torch.inverse (which synchronises to read its
// status word and so stalled the whole enqueue-ahead pipeline in the middle of the fine branch).
__device__ __forceinline__ void inv3x3(const float* __restrict__ m, float* __restrict__ o) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C, r = 1.0 / det;
  o[0] = (float)(A * r); o[1] = (float)(-(b * i - c * h) * r); o[2] = (float)((b * f - c * e) * r);
  o[3] = (float)(B * r); o[4] = (float)((a * i - c * g) * r);  o[5] = (float)(-(a * f - c * d) * r);
  o[6] = (float)(C * r); o[7] = (float)(-(a * h - b * g) * r); o[8] = (float)((a * e - b * d) * r);
}

__global__ void k_projection_params(const float* __restrict__ rots, const float* __restrict__ trans,
                                    const float* __restrict__ intrins, const float* __restrict__ post_rots,
                                    const float* __restrict__ post_trans, const float* __restrict__ bda, int ncam,
                                    float vs0, float vs1, float vs2, float lo0, float lo1, float lo2, float wimg1,
                                    float himg1, float* __restrict__ prm) {
  const int t = threadIdx.x;
  if (t == 0) {
    inv3x3(bda, prm);
    prm[9] = vs0; prm[10] = vs1; prm[11] = vs2; prm[12] = lo0; prm[13] = lo1; prm[14] = lo2;
    prm[15] = wimg1; prm[16] = himg1;
  } else if (t <= ncam) {
    const int cam = t - 1;
    float* q = prm + FINE_HDR + cam * FINE_CAM_STRIDE;
    inv3x3(rots + cam * 9, q);
    for (int k = 0; k < 3; ++k) q[9 + k] = trans[cam * 3 + k];
    for (int k = 0; k < 9; ++k) q[12 + k] = intrins[cam * 9 + k];
    q[21] = post_rots[cam * 9 + 0]; q[22] = post_rots[cam * 9 + 1];
    q[23] = post_rots[cam * 9 + 3]; q[24] = post_rots[cam * 9 + 4];
    q[25] = post_trans[cam * 3 + 0]; q[26] = post_trans[cam * 3 + 1];
  }
}

extern "C" int coocc_projection_params(const float* rots, const float* trans, const float* intrins, const float* post_rots,
                                       const float* post_trans, const float* bda, int ncam, const float* hdr_host,
                                       float* params, void* stream) {
  COOCC_CHECK_ARG(rots && trans && intrins && post_rots && post_trans && bda && hdr_host && params && ncam > 0 && ncam < 64,
                  "projection_params: bad args");
  const float* h = hdr_host;
  hipLaunchKernelGGL(k_projection_params, dim3(1), dim3(64), 0, as_stream(stream), rots, trans, intrins, post_rots, post_trans,
                     bda, ncam, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], params);
  COOCC_LAUNCH_CHECK("k_projection_params");
  return COOCC_OK;
}

// Grouped form for the offset-major fine list of the head (f = o*n + i, R^3 children per coarse voxel): one wave per
// coarse voxel, 8 children per round (R^3 / 8 rounds).  Lane o8*ncam + cam projects child g*8 + o8 into camera cam (the
// 8*ncam projections run in parallel instead of every lane repeating all of them); then lanes = channels and the (child,
// camera) pairs that see the point are walked with wave-uniform readlanes.  Same expressions and accumulation order as
// k_fine_sample_img.
template <int R>
__global__ COOCC_SCALAR_FP32 __launch_bounds__(256) void k_fine_sample_img_grp(const float* __restrict__ img, int ncam, int Ci, int Hf, int Wf,
                                                              const float* __restrict__ prm,
                                                              const int64_t* __restrict__ fine_xyz, int n,
                                                              float* __restrict__ feat, int out_stride,
                                                              const int32_t* __restrict__ n_dev,
                                                              const int32_t* __restrict__ coarse_lin = nullptr, int Yc = 0, int Zc = 0) {
  if (n_dev) n = min(n, *n_dev);
  constexpr int R3 = R * R * R;
  const int i = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const long long nf = (long long)n * R3;
  // child o = (a*R + b)*R + c of coarse voxel i sits at (child 0) + (a, b, c): three wave-uniform reads instead of
  // 3 x 48 scattered 8-byte ones per round (f = o*n + i); or, with the coarse list (B == 1), child 0 = R x the coarse voxel
  long long x00, y00, z00;
  if (coarse_lin) {
    int l = coarse_lin[i];
    z00 = (long long)(l % Zc) * R; l /= Zc;
    y00 = (long long)(l % Yc) * R; x00 = (long long)(l / Yc) * R;
  } else {
    x00 = fine_xyz[i]; y00 = fine_xyz[nf + i]; z00 = fine_xyz[2 * nf + i];
  }
#pragma unroll 1
  for (int g = 0; g < R3 / 8; ++g) {
    int m = 0, x0 = 0, y0 = 0;
    float ax = 0.f, ay = 0.f;
    if (lane < 8 * ncam) {
      const int o8 = lane / ncam, cam = lane - o8 * ncam;
      const int o = g * 8 + o8;
      const long long fx = x00 + o / (R * R), fy = y00 + (o / R) % R, fz = z00 + o % R;
      float p0 = (float)fx * prm[9] + prm[12];
      float p1 = (float)fy * prm[10] + prm[13];
      float p2 = (float)fz * prm[11] + prm[14];
      float bx = prm[0] * p0 + prm[1] * p1 + prm[2] * p2;
      float by = prm[3] * p0 + prm[4] * p1 + prm[5] * p2;
      float bz = prm[6] * p0 + prm[7] * p1 + prm[8] * p2;
      const float wimg1 = prm[15], himg1 = prm[16];
      const float* q = prm + FINE_HDR + cam * FINE_CAM_STRIDE;
      float tx = bx - q[9], ty = by - q[10], tz = bz - q[11];
      float cx = q[0] * tx + q[1] * ty + q[2] * tz;
      float cy = q[3] * tx + q[4] * ty + q[5] * tz;
      float cz = q[6] * tx + q[7] * ty + q[8] * tz;
      float ix = q[12] * cx + q[13] * cy + q[14] * cz;
      float iy = q[15] * cx + q[16] * cy + q[17] * cz;
      float d = q[18] * cx + q[19] * cy + q[20] * cz;
      float u = ix / (d + 1e-5f), v = iy / (d + 1e-5f);
      float u2 = q[21] * u + q[22] * v + q[25];
      float v2 = q[23] * u + q[24] * v + q[26];
      u2 = (u2 / wimg1 - 0.5f) * 2.f;
      v2 = (v2 / himg1 - 0.5f) * 2.f;
      m = (d > 1e-5f && u2 > -1.f && u2 < 1.f && v2 > -1.f && v2 < 1.f) ? 1 : 0;
      float px = (u2 + 1.f) / 2.f * (float)(Wf - 1), py = (v2 + 1.f) / 2.f * (float)(Hf - 1);
      float flx = floorf(px), fly = floorf(py);
      x0 = (int)flx; y0 = (int)fly;
      ax = px - flx; ay = py - fly;
    }
    // per (child, camera) lane: the four tap offsets (rows of the [ncam*Hf*Wf, Ci] map, clamped into it) and weights (0 outside
    // the map or when the camera does not see the child), so the channel loop below only broadcasts them (readlane)
    int toff[4]; float tw[4];
    const int cbase = lane < 8 * ncam ? (lane % ncam) * Hf * Wf : 0;
#pragma unroll
    for (int yy = 0; yy < 2; ++yy)
#pragma unroll
      for (int xx = 0; xx < 2; ++xx) {
        const int x = x0 + xx, y = y0 + yy;
        const bool in = (unsigned)x < (unsigned)Wf && (unsigned)y < (unsigned)Hf;
        toff[yy * 2 + xx] = cbase + (in ? y * Wf + x : 0);
        tw[yy * 2 + xx] = (in && m) ? (xx ? ax : 1.f - ax) * (yy ? ay : 1.f - ay) : 0.f;
      }
    // Cameras that see each child, as wave-uniform bit fields.  Round r takes every child's r-th seeing camera (ascending:
    // the accumulation order of k_fine_sample_img) and issues the 8 x 4 tap loads together -- walking the (child, camera)
    // pairs one after the other put 30-40 dependent L2 round trips in a row (~20 k cycles per wave).  A child without an
    // r-th camera rides along with weight 0 (x + 0 * v: finite features assumed, as everywhere in this branch).
    const unsigned long long seen = __ballot(m != 0);
    unsigned sub[8];
    int rounds = 0;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      sub[o] = (unsigned)(seen >> (o * ncam)) & ((1u << ncam) - 1u);
      rounds = max(rounds, __popc(sub[o]));
    }
    // every lane stays in the channel loop (a lane past Ci works on a clamped column and skips the store): the readlanes
    // below must not sit under a divergent branch -- registers of lanes that are inactive there are not preserved
    for (int c0 = 0; c0 < Ci; c0 += 64) {
      const int c = c0 + lane;
      float acc[8];
      unsigned left[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) { acc[o] = 0.f; left[o] = sub[o]; }
      const float* base = img + min(c, Ci - 1);
#pragma unroll 1
      for (int r = 0; r < rounds; ++r) {
        float v[8][4], w[8][4];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
          const bool has = left[o] != 0u;                                  // wave-uniform
          const int k = has ? o * ncam + (__ffs((int)left[o]) - 1) : 0;
          left[o] &= left[o] - 1u;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int off = __builtin_amdgcn_readlane(toff[t], k);
            const float wt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw[t]), k));
            w[o][t] = has ? wt : 0.f;
            v[o][t] = base[(size_t)off * Ci];
          }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[o] = acc[o] + v[o][t] * w[o][t];
      }
      if (c < Ci) {
#pragma unroll
        for (int o = 0; o < 8; ++o) feat[((size_t)(g * 8 + o) * n + i) * out_stride + c] = acc[o];
      }
    }
  }
}

static int fine_sample_img_impl(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params, const int64_t* fine_xyz,
                                int64_t nfine, const int32_t* n_dev, float* feat, int out_stride, int group, void* stream);
extern "C" int coocc_fine_sample_img(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params,
                                     const int64_t* fine_xyz, int64_t nfine, float* feat, int out_stride, int group,
                                     void* stream) {
  return fine_sample_img_impl(img_nhwc, ncam, Ci, Hf, Wf, params, fine_xyz, nfine, nullptr, feat, out_stride, group, stream);
}
extern "C" int coocc_fine_sample_img_dev(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params,
                                         const int64_t* fine_xyz, int64_t nfine_cap, const int32_t* n_dev, float* feat,
                                         int out_stride, int group, void* stream) {
  COOCC_CHECK_ARG(n_dev && (group == 1 || group == 2 || group == 4), "fine_sample_img_dev: needs the device count of COARSE voxels and group 2 | 4");
  return fine_sample_img_impl(img_nhwc, ncam, Ci, Hf, Wf, params, fine_xyz, nfine_cap, n_dev, feat, out_stride, group, stream);
}
// The grouped sampler straight from the foreground list: the fine coordinates need not exist yet (cascade ratio 2 | 4, B == 1,
// final grid = ratio x coarse grid: child 0 of coarse voxel (x, y, z) is fine voxel ratio x (x, y, z)).  n_dev optional.
extern "C" int coocc_fine_sample_img_lin(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params,
                                         const int32_t* coarse_lin, int Yc, int Zc, int n_cap, const int32_t* n_dev, float* feat,
                                         int out_stride, int ratio, void* stream) {
  COOCC_CHECK_ARG(img_nhwc && params && coarse_lin && feat && ncam > 0 && ncam <= 8 && Ci > 0 && Ci % 2 == 0 && Ci <= 512 && Yc > 0 && Zc > 0,
                  "fine_sample_img_lin: bad args (Ci even, <= 512, <= 8 cameras)");
  COOCC_CHECK_ARG(ratio == 2 || ratio == 4, "fine_sample_img_lin: cascade ratio 2 | 4");
  if (n_cap <= 0) return COOCC_OK;
  if (ratio == 2)
    hipLaunchKernelGGL(k_fine_sample_img_grp<2>, dim3(cdiv((long long)n_cap * 64, 256)), dim3(256), 0, as_stream(stream), img_nhwc, ncam,
                       Ci, Hf, Wf, params, (const int64_t*)nullptr, n_cap, feat, out_stride, n_dev, coarse_lin, Yc, Zc);
  else
    hipLaunchKernelGGL(k_fine_sample_img_grp<4>, dim3(cdiv((long long)n_cap * 64, 256)), dim3(256), 0, as_stream(stream), img_nhwc, ncam,
                       Ci, Hf, Wf, params, (const int64_t*)nullptr, n_cap, feat, out_stride, n_dev, coarse_lin, Yc, Zc);
  COOCC_LAUNCH_CHECK("k_fine_sample_img_grp");
  return COOCC_OK;
}

static int fine_sample_img_impl(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params, const int64_t* fine_xyz,
                                int64_t nfine, const int32_t* n_dev, float* feat, int out_stride, int group, void* stream) {
  COOCC_CHECK_ARG(img_nhwc && params && fine_xyz && feat && ncam > 0 && Ci > 0 && Ci % 2 == 0 && Ci <= 512,
                  "fine_sample_img: bad args (Ci even, <= 512)");
  if (nfine == 0) return COOCC_OK;
  // group: 1 / 2 -> children of ratio 2, 4 -> ratio 4, anything else -> the list is taken point by point
  const int R = (group == 1 || group == 2) ? 2 : group == 4 ? 4 : 0;
  const int r3 = R * R * R;
  if (R && !g_fine_pointwise && nfine % r3 == 0 && ncam <= 8 && nfine / r3 < (1ll << 31)) {
    const int n = (int)(nfine / r3);
    if (R == 2)
      hipLaunchKernelGGL(k_fine_sample_img_grp<2>, dim3(cdiv((long long)n * 64, 256)), dim3(256), 0, as_stream(stream), img_nhwc,
                         ncam, Ci, Hf, Wf, params, fine_xyz, n, feat, out_stride, n_dev);
    else
      hipLaunchKernelGGL(k_fine_sample_img_grp<4>, dim3(cdiv((long long)n * 64, 256)), dim3(256), 0, as_stream(stream), img_nhwc,
                         ncam, Ci, Hf, Wf, params, fine_xyz, n, feat, out_stride, n_dev);
    COOCC_LAUNCH_CHECK("k_fine_sample_img_grp");
    return COOCC_OK;
  }
  hipLaunchKernelGGL(k_fine_sample_img, dim3(cdiv(nfine * 64, 256)), dim3(256), 0, as_stream(stream), img_nhwc, ncam, Ci,
                     Hf, Wf, params, fine_xyz, (long long)nfine, feat, out_stride, n_dev, r3 ? r3 : 1);
  COOCC_LAUNCH_CHECK("k_fine_sample_img");
  return COOCC_OK;
}

// nn.GroupNorm over rows [n, C] (2-D input: statistics per row and group) + optional ReLU
__global__ __launch_bounds__(256) void k_groupnorm_rows(float* __restrict__ x, long long n, int C, int stride, int groups,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int relu) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * groups) return;
  long long row = i / groups;
  int g = (int)(i - row * groups);
  int cpg = C / groups;
  float* p = x + row * stride + g * cpg;
  float mean = 0.f;
  for (int c = 0; c < cpg; ++c) mean += p[c];
  mean /= (float)cpg;
  float var = 0.f;
  for (int c = 0; c < cpg; ++c) { float d = p[c] - mean; var += d * d; }
  var /= (float)cpg;
  float rstd = 1.f / sqrtf(var + eps);
  for (int c = 0; c < cpg; ++c) {
    float v = (p[c] - mean) * rstd * gamma[g * cpg + c] + beta[g * cpg + c];
    p[c] = relu ? fmaxf(v, 0.f) : v;
  }
}

extern "C" int coocc_groupnorm_rows(float* x, int64_t n, int C, int stride, int groups, const float* gamma,
                                    const float* beta, float eps, int relu, void* stream) {
  COOCC_CHECK_ARG(x && gamma && beta && C > 0 && groups > 0 && C % groups == 0 && stride >= C, "groupnorm_rows: bad args");
  if (n == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_groupnorm_rows, dim3(cdiv(n * groups, 256)), dim3(256), 0, as_stream(stream), x, (long long)n, C,
                     stride, groups, gamma, beta, eps, relu);
  COOCC_LAUNCH_CHECK("k_groupnorm_rows");
  return COOCC_OK;
}

// nn.GroupNorm over an NHWC image batch [N, HW, C]: statistics per (image, group) over
// HW * C/groups values (occ_head.py:64-68), then normalise (+ReLU).  One block per (n, group).
__global__ __launch_bounds__(256) void k_groupnorm_nhwc(float* __restrict__ x, int HW, int C, int groups,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int relu) {
  __shared__ double s_a[4], s_b[4];
  __shared__ float s_mean, s_rstd;
  const int n = blockIdx.y, g = blockIdx.x, cpg = C / groups;
  float* base = x + (size_t)n * HW * C + g * cpg;
  const int total = HW * cpg;
  double sum = 0, sq = 0;
  for (int i = threadIdx.x; i < total; i += 256) {
    float v = base[(size_t)(i / cpg) * C + (i % cpg)];
    sum += v; sq += (double)v * v;
  }
  for (int m = 32; m > 0; m >>= 1) { sum += __shfl_xor(sum, m); sq += __shfl_xor(sq, m); }
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = sum; s_b[threadIdx.x >> 6] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = s_a[0] + s_a[1] + s_a[2] + s_a[3], b = s_b[0] + s_b[1] + s_b[2] + s_b[3];
    double mean = a / total, var = b / total - mean * mean;
    s_mean = (float)mean;
    s_rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
  }
  __syncthreads();
  const float mean = s_mean, rstd = s_rstd;
  for (int i = threadIdx.x; i < total; i += 256) {
    int c = i % cpg;
    float* p = base + (size_t)(i / cpg) * C + c;
    float v = (*p - mean) * rstd * gamma[g * cpg + c] + beta[g * cpg + c];
    *p = relu ? fmaxf(v, 0.f) : v;
  }
}

extern "C" int coocc_groupnorm_nhwc(float* x, int N, int HW, int C, int groups, const float* gamma, const float* beta,
                                    float eps, int relu, void* stream) {
  COOCC_CHECK_ARG(x && gamma && beta && N > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "groupnorm_nhwc: bad args");
  hipLaunchKernelGGL(k_groupnorm_nhwc, dim3(groups, N), dim3(256), 0, as_stream(stream), x, HW, C, groups, gamma, beta, eps,
                     relu);
  COOCC_LAUNCH_CHECK("k_groupnorm_nhwc");
  return COOCC_OK;
}

// simple_test fine scatter (coocc_ray.py:546-550): grid [ncls,Xf,Yf,Zf] pre-filled with
// empty_idx, fine logits written at their coordinates.
__global__ __launch_bounds__(256) void k_fill(float* __restrict__ p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// 256 fine points per block: their logits are read as one contiguous span into LDS, then every class plane gets one
// store per lane whose addresses follow the fine coordinates (consecutive points = neighbouring z), instead of one
// lane per (point, class) scattering a wave's stores over 17 planes.
__global__ __launch_bounds__(256) void k_scatter_fine(const float* __restrict__ logits, long long nf, int ncls, int stride,
                                                       const int64_t* __restrict__ fine_xyz, float* __restrict__ grid,
                                                       int Xf, int Yf, int Zf, const int32_t* __restrict__ n_dev, int n_mul) {
  if (n_dev) nf = min(nf, (long long)*n_dev * n_mul);
  if ((long long)blockIdx.x * 256 >= nf) return;
  extern __shared__ float s_log[];   // [256][ncls | 1]: odd row pitch -> conflict-free column reads
  const int pitch = ncls | 1;
  const long long f0 = (long long)blockIdx.x * 256;
  const int rows = (int)(nf - f0 < 256 ? nf - f0 : 256);
  if (stride == ncls) {
    for (int i = threadIdx.x; i < rows * ncls; i += 256) s_log[(i / ncls) * pitch + i % ncls] = logits[f0 * ncls + i];
  } else {
    for (int i = threadIdx.x; i < rows * ncls; i += 256) s_log[(i / ncls) * pitch + i % ncls] = logits[(f0 + i / ncls) * stride + i % ncls];
  }
  __syncthreads();
  const long long f = f0 + threadIdx.x;
  if (f >= nf) return;
  const long long x = fine_xyz[f], y = fine_xyz[nf + f], z = fine_xyz[2 * nf + f];
  const size_t plane = (size_t)Xf * Yf * Zf;
  float* g = grid + ((size_t)x * Yf + y) * Zf + z;
  for (int c = 0; c < ncls; ++c) g[c * plane] = s_log[threadIdx.x * pitch + c];
}

static int scatter_fine_impl(const float* fine_logits, int64_t nfine, const int32_t* n_dev, int n_mul, int ncls, int stride,
                             const int64_t* fine_xyz, float* grid, int Xf, int Yf, int Zf, float empty_val, void* stream);
extern "C" int coocc_scatter_fine(const float* fine_logits, int64_t nfine, int ncls, int stride, const int64_t* fine_xyz,
                                  float* grid, int Xf, int Yf, int Zf, float empty_val, void* stream) {
  return scatter_fine_impl(fine_logits, nfine, nullptr, 1, ncls, stride, fine_xyz, grid, Xf, Yf, Zf, empty_val, stream);
}
extern "C" int coocc_scatter_fine_dev(const float* fine_logits, int64_t nfine_cap, const int32_t* n_dev, int n_mul, int ncls, int stride,
                                      const int64_t* fine_xyz, float* grid, int Xf, int Yf, int Zf, float empty_val, void* stream) {
  COOCC_CHECK_ARG(n_dev && n_mul > 0, "scatter_fine_dev: null device count");
  return scatter_fine_impl(fine_logits, nfine_cap, n_dev, n_mul, ncls, stride, fine_xyz, grid, Xf, Yf, Zf, empty_val, stream);
}
static int scatter_fine_impl(const float* fine_logits, int64_t nfine, const int32_t* n_dev, int n_mul, int ncls, int stride,
                             const int64_t* fine_xyz, float* grid, int Xf, int Yf, int Zf, float empty_val, void* stream) {
  COOCC_CHECK_ARG(grid && ncls > 0 && Xf > 0 && Yf > 0 && Zf > 0, "scatter_fine: bad args");
  size_t total = (size_t)ncls * Xf * Yf * Zf;
  hipLaunchKernelGGL(k_fill, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), grid, total, empty_val);
  if (nfine > 0) {
    COOCC_CHECK_ARG(fine_logits && fine_xyz && stride >= ncls, "scatter_fine: null pointer");
    COOCC_CHECK_ARG(ncls <= 128, "scatter_fine: at most 128 classes");
    hipLaunchKernelGGL(k_scatter_fine, dim3(cdiv(nfine, 256)), dim3(256), 256 * (ncls | 1) * sizeof(float), as_stream(stream), fine_logits,
                       (long long)nfine, ncls, stride, fine_xyz, grid, Xf, Yf, Zf, n_dev, n_mul);
  }
  COOCC_LAUNCH_CHECK("scatter_fine");
  return COOCC_OK;
}

__global__ __launch_bounds__(256) void k_lin_ordinal_map(const int32_t* __restrict__ lin, int n, const int32_t* __restrict__ n_dev,
                                                          int32_t* __restrict__ map) {
  if (n_dev) n = min(n, *n_dev);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) map[lin[i]] = i;
}

// The same grid when the fine points are the R^3 children of a LIST of coarse voxels (the head's own output: point f = o n + i,
// offset o = (a R + b) R + c, final grid = R x coarse grid), written OUTPUT-major in one pass: every output voxel looks its coarse
// voxel up in a voxel -> list-ordinal map and stores either its logits row or the empty value -- ncls coalesced plane stores per
// wave, no fill pass and no scattered 4-byte stores over ncls planes (OpenOccupancy cascade: 713 MB filled + 231 MB scattered in
// 1.1 ms; configs[1]: 43 MB, 57 us in two launches).  Same values as coocc_scatter_fine on the head's coordinates.
__global__ __launch_bounds__(256) void k_scatter_fine_grouped(const float* __restrict__ logits, int stride, int ncls,
                                                               const int32_t* __restrict__ map, int n, const int32_t* __restrict__ n_dev,
                                                               int R, int Xc, int Yc, int Zc, float empty, float* __restrict__ grid) {
  const int Yf = Yc * R, Zf = Zc * R;
  const size_t plane = (size_t)Xc * R * Yf * Zf;
  const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= plane) return;
  if (n_dev) n = min(n, *n_dev);
  const int zf = (int)(v % Zf); size_t q = v / Zf;
  const int yf = (int)(q % Yf); const int xf = (int)(q / Yf);
  const int i = map[((size_t)(xf / R) * Yc + yf / R) * Zc + zf / R];
  float* g = grid + v;
  if (i >= 0 && i < n) {
    const int o = ((xf % R) * R + yf % R) * R + zf % R;
    const float* row = logits + ((size_t)o * n + i) * stride;
    for (int c = 0; c < ncls; ++c) g[c * plane] = row[c];
  } else {
    for (int c = 0; c < ncls; ++c) g[c * plane] = empty;
  }
}

// (a fill KERNEL, not hipMemsetAsync: this entry point is captured into hipGraphs, and a memset NODE in the middle of a captured
// chain was found unreliable when several graphs replay concurrently -- csrc/knn.hip coocc_voxel_index_map_dev)
__global__ __launch_bounds__(256) void k_fill_m1(int32_t* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = -1;
}

extern "C" int coocc_scatter_fine_grouped(const float* fine_logits, int ncls, int stride, const int32_t* coarse_lin, int n_cap,
                                          const int32_t* n_dev, int R, int Xc, int Yc, int Zc, float* grid, float empty_val,
                                          int32_t* map_ws, void* stream) {
  COOCC_CHECK_ARG(grid && map_ws && ncls > 0 && R >= 1 && Xc > 0 && Yc > 0 && Zc > 0 && n_cap >= 0 && stride >= ncls,
                  "scatter_fine_grouped: bad args");
  COOCC_CHECK_ARG(n_cap == 0 || fine_logits, "scatter_fine_grouped: null pointer");
  COOCC_CHECK_ARG((long long)Xc * Yc * Zc < (1ll << 31), "scatter_fine_grouped: coarse grid too large");
  hipStream_t s = as_stream(stream);
  const int V = Xc * Yc * Zc;
  if (coarse_lin) {                   // coarse_lin == NULL: map_ws already holds the voxel -> ordinal table (coocc_compact_flags_ex)
    hipLaunchKernelGGL(k_fill_m1, dim3(cdiv(V, 256)), dim3(256), 0, s, map_ws, V);   // -1 everywhere
    if (n_cap > 0) {
      hipLaunchKernelGGL(k_lin_ordinal_map, dim3(cdiv(n_cap, 256)), dim3(256), 0, s, coarse_lin, n_cap, n_dev, map_ws);
      COOCC_LAUNCH_CHECK("k_lin_ordinal_map");
    }
  }
  const size_t plane = (size_t)V * R * R * R;
  hipLaunchKernelGGL(k_scatter_fine_grouped, dim3(cdiv(plane, 256)), dim3(256), 0, s, fine_logits, stride, ncls, map_ws, n_cap, n_dev, R,
                     Xc, Yc, Zc, empty_val, grid);
  COOCC_LAUNCH_CHECK("k_scatter_fine_grouped");
  return COOCC_OK;
}
