// Lift-splat side of the path: frustum geometry (P1) and voxel pooling (P2).
//
// The reference pools with argsort(ranks) + an interval-sum kernel (bev_pool.py:83-97,
// bev_pool_cuda.cu:20-42); argsort is unstable there, so the fp32 summation order inside a
// voxel is unspecified.  Here the (voxel, point id) pairs go through a STABLE LSD radix sort
// on the voxel key only (rocPRIM device primitive), so every voxel sums its rows in ascending
// point id -- deterministic and equal to the oracle's order.  The [EXT] entry points that take
// already-sorted intervals are provided as well.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ P1: get_geometry
// ViewTransformerLSSBEVDepth.py:117-150.  mats[cam] = {A=inv(post_rots)[9], post_trans[3],
// Cm=rots@inv(intrins[:3,:3])[9], trans[3], bda[:3,:3][9], shift[3] = intrins[:3,3] of a 3x4/4x4 KITTI intrinsic (:136-139,
// else 0), bda[:3,3] of a 4x4 bda (:145-148, else 0)} (COOCC_CAM_FLOATS = 39 floats); xs/ys/ds = the frustum axes of
// create_frustum (:104-115) computed by the host with the same torch calls.  The same chain is the module-level
// get_frustum of the detector (P/coocc/detectors/coocc_ray.py:732-776).
__device__ __forceinline__ void geometry_point(const float* __restrict__ mats, const float* __restrict__ xs,
                                               const float* __restrict__ ys, const float* __restrict__ ds, size_t i, int D,
                                               int fH, int fW, float& gx, float& gy, float& gz) {
  int w = (int)(i % fW); size_t r = i / fW;
  int h = (int)(r % fH); r /= fH;
  int d = (int)(r % D); int cam = (int)(r / D);
  const float* m = mats + (size_t)cam * COOCC_CAM_FLOATS;
  float px = xs[w] - m[9], py = ys[h] - m[10], pz = ds[d] - m[11];
  float qx = m[0] * px + m[1] * py + m[2] * pz;
  float qy = m[3] * px + m[4] * py + m[5] * pz;
  float qz = m[6] * px + m[7] * py + m[8] * pz;
  qx *= qz; qy *= qz;
  qx -= m[33]; qy -= m[34]; qz -= m[35];
  float ex = m[12] * qx + m[13] * qy + m[14] * qz + m[21];
  float ey = m[15] * qx + m[16] * qy + m[17] * qz + m[22];
  float ez = m[18] * qx + m[19] * qy + m[20] * qz + m[23];
  gx = m[24] * ex + m[25] * ey + m[26] * ez + m[36];
  gy = m[27] * ex + m[28] * ey + m[29] * ez + m[37];
  gz = m[30] * ex + m[31] * ey + m[32] * ez + m[38];
}

__global__ __launch_bounds__(256) void k_get_geometry(const float* __restrict__ mats, const float* __restrict__ xs,
                                                       const float* __restrict__ ys, const float* __restrict__ ds,
                                                       int BN, int D, int fH, int fW, float* __restrict__ geom) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)BN * D * fH * fW;
  if (i >= total) return;
  float* o = geom + i * 3;
  geometry_point(mats, xs, ys, ds, i, D, fH, fW, o[0], o[1], o[2]);
}

extern "C" int coocc_get_geometry(const float* mats, const float* xs, const float* ys, const float* ds, int BN, int D,
                                  int fH, int fW, float* geom, void* stream) {
  COOCC_CHECK_ARG(mats && xs && ys && ds && geom && BN > 0 && D > 0 && fH > 0 && fW > 0, "get_geometry: bad args");
  size_t total = (size_t)BN * D * fH * fW;
  hipLaunchKernelGGL(k_get_geometry, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), mats, xs, ys, ds, BN, D, fH,
                     fW, geom);
  COOCC_LAUNCH_CHECK("k_get_geometry");
  return COOCC_OK;
}

// ------------------------------------------------------------------ [EXT] interval kernels
// one wave per interval, lanes along channels (rows are read as whole coalesced lines)
__global__ __launch_bounds__(256) void k_bev_pool_fwd(int d, int h, int w, int c, int n_intervals,
                                                       const float* __restrict__ x, const int32_t* __restrict__ geom,
                                                       const int32_t* __restrict__ starts,
                                                       const int32_t* __restrict__ lengths, float* __restrict__ out) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= n_intervals) return;
  const int s = starts[it], len = lengths[it];
  const int32_t* g = geom + (size_t)s * 4;
  float* o = out + (((size_t)g[3] * d + g[2]) * h + g[0]) * (size_t)w * c + (size_t)g[1] * c;
  for (int cc = lane; cc < c; cc += 64) {
    float psum = 0.f;
    for (int i = 0; i < len; ++i) psum += x[((size_t)s + i) * c + cc];
    o[cc] = psum;
  }
}

__global__ __launch_bounds__(256) void k_bev_pool_bwd(int d, int h, int w, int c, int n_intervals,
                                                       const float* __restrict__ out_grad,
                                                       const int32_t* __restrict__ geom,
                                                       const int32_t* __restrict__ starts,
                                                       const int32_t* __restrict__ lengths, float* __restrict__ x_grad) {
  const int it = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (it >= n_intervals) return;
  const int s = starts[it], len = lengths[it];
  const int32_t* g = geom + (size_t)s * 4;
  const float* o = out_grad + (((size_t)g[3] * d + g[2]) * h + g[0]) * (size_t)w * c + (size_t)g[1] * c;
  for (int cc = lane; cc < c; cc += 64) {
    float v = o[cc];
    for (int i = 0; i < len; ++i) x_grad[((size_t)s + i) * c + cc] = v;
  }
}

extern "C" int coocc_bev_pool_forward(const float* x, const int32_t* geom, const int32_t* interval_lengths,
                                      const int32_t* interval_starts, int b, int d, int h, int w, int n, int c,
                                      int n_intervals, float* out, void* stream) {
  COOCC_CHECK_ARG(out && b > 0 && d > 0 && h > 0 && w > 0 && c > 0 && n >= 0 && n_intervals >= 0, "bev_pool_forward: bad args");
  COOCC_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)b * d * h * w * c, as_stream(stream)));
  if (n_intervals == 0) return COOCC_OK;
  COOCC_CHECK_ARG(x && geom && interval_lengths && interval_starts, "bev_pool_forward: null pointer");
  hipLaunchKernelGGL(k_bev_pool_fwd, dim3(cdiv(n_intervals, 4)), dim3(256), 0, as_stream(stream), d, h, w, c, n_intervals,
                     x, geom, interval_starts, interval_lengths, out);
  COOCC_LAUNCH_CHECK("k_bev_pool_fwd");
  return COOCC_OK;
}

extern "C" int coocc_bev_pool_backward(const float* out_grad, const int32_t* geom, const int32_t* interval_lengths,
                                       const int32_t* interval_starts, int b, int d, int h, int w, int n, int c,
                                       int n_intervals, float* x_grad, void* stream) {
  COOCC_CHECK_ARG(x_grad && b > 0 && d > 0 && h > 0 && w > 0 && c > 0 && n >= 0 && n_intervals >= 0, "bev_pool_backward: bad args");
  COOCC_HIP(hipMemsetAsync(x_grad, 0, sizeof(float) * (size_t)n * c, as_stream(stream)));
  if (n_intervals == 0) return COOCC_OK;
  COOCC_CHECK_ARG(out_grad && geom && interval_lengths && interval_starts, "bev_pool_backward: null pointer");
  hipLaunchKernelGGL(k_bev_pool_bwd, dim3(cdiv(n_intervals, 4)), dim3(256), 0, as_stream(stream), d, h, w, c, n_intervals,
                     out_grad, geom, interval_starts, interval_lengths, x_grad);
  COOCC_LAUNCH_CHECK("k_bev_pool_bwd");
  return COOCC_OK;
}

// ------------------------------------------------------------------ P2: sort-by-voxel pooling
// keys: voxel row (b,x,y,z order) or nvox for dropped points
// ((geom - (bx - dx/2)) / dx).long(): truncation toward zero BEFORE the range filter
// (ViewTransformerLSSVoxel.py:107,113-115), so (-1,0) lands in voxel 0.
__device__ __forceinline__ uint32_t voxel_key(float x, float y, float z, int b, float lox, float loy, float loz, float dx,
                                              float dy, float dz, int X, int Y, int Z, int nvox) {
  float gx = __fdiv_rn(x - lox, dx), gy = __fdiv_rn(y - loy, dy), gz = __fdiv_rn(z - loz, dz);
  long long ix = (long long)gx, iy = (long long)gy, iz = (long long)gz;
  bool kept = ix >= 0 && ix < X && iy >= 0 && iy < Y && iz >= 0 && iz < Z;
  return kept ? (uint32_t)((((size_t)b * X + ix) * Y + iy) * Z + iz) : (uint32_t)nvox;
}

__global__ __launch_bounds__(256) void k_quantize_geom(const float* __restrict__ geom, int npts, int pts_per_batch,
                                                        float lox, float loy, float loz, float dx, float dy, float dz,
                                                        int X, int Y, int Z, int nvox, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ ids) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  keys[i] = voxel_key(geom[(size_t)i * 3 + 0], geom[(size_t)i * 3 + 1], geom[(size_t)i * 3 + 2], i / pts_per_batch, lox,
                      loy, loz, dx, dy, dz, X, Y, Z, nvox);
  ids[i] = (uint32_t)i;
}

// geometry computed in-kernel from the camera matrices (no [npts,3] geom tensor)
__global__ __launch_bounds__(256) void k_quantize_cams(const float* __restrict__ mats, const float* __restrict__ xs,
                                                        const float* __restrict__ ys, const float* __restrict__ ds, int D,
                                                        int fH, int fW, int npts, int pts_per_batch, float lox, float loy,
                                                        float loz, float dx, float dy, float dz, int X, int Y, int Z,
                                                        int nvox, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  float gx, gy, gz;
  geometry_point(mats, xs, ys, ds, (size_t)i, D, fH, fW, gx, gy, gz);
  keys[i] = voxel_key(gx, gy, gz, i / pts_per_batch, lox, loy, loz, dx, dy, dz, X, Y, Z, nvox);
  ids[i] = (uint32_t)i;
}

// keys from integer coords (bev_pool drop-in): coords [n,4] = (x,y,z,b) int64
__global__ __launch_bounds__(256) void k_keys_from_coords(const int64_t* __restrict__ coords, int n, int B, int X, int Y,
                                                           int Z, int nvox, uint32_t* __restrict__ keys,
                                                           uint32_t* __restrict__ ids) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long x = coords[(size_t)i * 4 + 0], y = coords[(size_t)i * 4 + 1], z = coords[(size_t)i * 4 + 2],
            b = coords[(size_t)i * 4 + 3];
  bool ok = x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z && b >= 0 && b < B;
  keys[i] = ok ? (uint32_t)(((b * X + x) * Y + y) * Z + z) : (uint32_t)nvox;
  ids[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_segment_bounds(const uint32_t* __restrict__ keys, int npts, int nvox,
                                                         int32_t* __restrict__ seg_start, int32_t* __restrict__ seg_end) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts) return;
  uint32_t k = keys[i];
  if (k < (uint32_t)nvox && (i == 0 || keys[i - 1] != k)) seg_start[k] = i;
  if (k < (uint32_t)nvox && (i == npts - 1 || keys[i + 1] != k)) seg_end[k] = i + 1;
}

// One wave per voxel row, 4 channels per lane.  Rows are summed in sorted (= ascending id) order with a
// single accumulator, the association of the reference's interval kernel, but the loads run ahead: the
// wave fetches 64 ids at once, broadcasts them through SGPRs (v_readlane) and keeps POOL_BATCH independent
// row loads in flight before the dependent adds.  The longest voxel holds 484 (r50) / 2576 (r101) points,
// so the serial chain, not bandwidth, bounds this kernel.
constexpr int POOL_BATCH = 16;

template <int VEC> struct PoolVec;
template <> struct PoolVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct PoolVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// A lone wave issues roughly one instruction per 5 cycles, so the per-point instruction count matters as
// much as the load latency: the id -> (context row, depth) decode is done once per 64 ids in the vector
// lanes and broadcast with v_readlane (3 per point); VEC = 2 keeps all 64 lanes busy at C <= 128.
template <bool LIFT, int VEC>
__device__ __forceinline__ void pool_row(const float* __restrict__ x, const float* __restrict__ depth,
                                         const uint32_t* __restrict__ ids, int s, int e, int lane, int C, int D, int HW,
                                         float* __restrict__ orow) {
  typedef typename PoolVec<VEC>::type vec;
  for (int c0 = 0; c0 < C; c0 += 64 * VEC) {
    const int c = c0 + lane * VEC;
    const bool lane_on = c < C;
    const float* xc = x + (lane_on ? c : 0);   // idle lanes re-read channel 0 instead of branching
    vec acc = (vec)(0.f);
    for (int base = s; base < e; base += 64) {
      const int nb = min(64, e - base);
      const uint32_t myid = ids[base + min(lane, nb - 1)];
      uint32_t myrow = myid;
      float mydp = 1.f;
      if (LIFT) {
        myrow = (myid / (uint32_t)(D * HW)) * (uint32_t)HW + myid % (uint32_t)HW;
        mydp = depth[myid];
      }
      for (int j0 = 0; j0 < nb; j0 += POOL_BATCH) {
        vec r[POOL_BATCH];
#pragma unroll
        for (int j = 0; j < POOL_BATCH; ++j) {
#pragma clang fp contract(off)  // LIFT: the product is rounded before the add, as the materialised volume is
          const int jj = min(j0 + j, nb - 1);
          const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)myrow, jj);
          vec v = *(const vec*)(xc + (size_t)row * C);
          if (LIFT) v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mydp), jj)) * v;
          r[j] = v;
        }
#pragma unroll
        for (int j = 0; j < POOL_BATCH; ++j)
          if (j0 + j < nb) acc = acc + r[j];
      }
    }
    if (lane_on) *(vec*)(orow + c) = acc;
  }
}

__global__ __launch_bounds__(256) void k_pool_sum(const float* __restrict__ x, const uint32_t* __restrict__ ids,
                                                   const int32_t* __restrict__ seg_start,
                                                   const int32_t* __restrict__ seg_end, int nvox, int C,
                                                   float* __restrict__ out, int out_stride) {
  const int v = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (v >= nvox) return;
  if (C <= 128) pool_row<false, 2>(x, nullptr, ids, seg_start[v], seg_end[v], lane, C, 0, 0, out + (size_t)v * out_stride);
  else pool_row<false, 4>(x, nullptr, ids, seg_start[v], seg_end[v], lane, C, 0, 0, out + (size_t)v * out_stride);
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int npts) {
  size_t tmp = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, (size_t)npts, 0, 32, (hipStream_t)0);
  return tmp;
}

extern "C" size_t coocc_voxel_pool_ws(int npts, int nvox) {
  if (npts <= 0 || nvox <= 0) return 0;
  return 4 * align256(sizeof(uint32_t) * (size_t)npts) + 2 * align256(sizeof(int32_t) * (size_t)nvox) +
         align256(sort_temp_bytes(npts)) + 256;
}

static int pool_sorted(const float* x, int npts, int C, int nvox, float* out, int out_stride, uint32_t* k_in,
                       uint32_t* k_out, uint32_t* i_in, uint32_t* i_out, int32_t* seg_s, int32_t* seg_e, void* tmp,
                       size_t tmp_bytes, hipStream_t s) {
  int bits = 1;
  while ((1ll << bits) <= nvox) ++bits;  // keys go up to nvox inclusive
  COOCC_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, i_in, i_out, (size_t)npts, 0, bits, s));
  COOCC_HIP(hipMemsetAsync(seg_s, 0, 2 * align256(sizeof(int32_t) * (size_t)nvox), s));
  hipLaunchKernelGGL(k_segment_bounds, dim3(cdiv(npts, 256)), dim3(256), 0, s, k_out, npts, nvox, seg_s, seg_e);
  hipLaunchKernelGGL(k_pool_sum, dim3(cdiv(nvox, 4)), dim3(256), 0, s, x, i_out, seg_s, seg_e, nvox, C, out, out_stride);
  COOCC_LAUNCH_CHECK("voxel_pool");
  return COOCC_OK;
}

struct PoolWs { uint32_t *k_in, *k_out, *i_in, *i_out; int32_t *seg_s, *seg_e; void* tmp; size_t tmp_bytes; };

static int carve(void* ws, size_t ws_bytes, int npts, int nvox, PoolWs* p) {
  size_t need = coocc_voxel_pool_ws(npts, nvox);
  if (!ws || ws_bytes < need) return coocc_set_error(COOCC_ENOMEM, "voxel_pool: workspace %zu < %zu bytes", ws_bytes, need);
  char* c = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  size_t a = align256(sizeof(uint32_t) * (size_t)npts), v = align256(sizeof(int32_t) * (size_t)nvox);
  p->k_in = (uint32_t*)c; c += a; p->k_out = (uint32_t*)c; c += a;
  p->i_in = (uint32_t*)c; c += a; p->i_out = (uint32_t*)c; c += a;
  p->seg_s = (int32_t*)c; c += v; p->seg_e = (int32_t*)c; c += v;  // contiguous: one memset clears both
  p->tmp = c; p->tmp_bytes = sort_temp_bytes(npts);
  return COOCC_OK;
}

extern "C" int coocc_voxel_pool(const float* x, const float* geom, int npts, int pts_per_batch, int C,
                                const float* lo_dx_host, int B, int X, int Y, int Z, float* out, int out_stride,
                                void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(x && geom && out && lo_dx_host && npts > 0 && pts_per_batch > 0 && C > 0 && C % 4 == 0, "voxel_pool: bad args");
  COOCC_CHECK_ARG(((uintptr_t)x & 15) == 0 && out_stride % 4 == 0 && out_stride >= C, "voxel_pool: alignment");
  const long long nvox_ll = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(nvox_ll > 0 && nvox_ll < (1ll << 31), "voxel_pool: grid too large");
  const int nvox = (int)nvox_ll;
  PoolWs p;
  int rc = carve(ws, ws_bytes, npts, nvox, &p);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  const float* l = lo_dx_host;
  hipLaunchKernelGGL(k_quantize_geom, dim3(cdiv(npts, 256)), dim3(256), 0, s, geom, npts, pts_per_batch, l[0], l[1], l[2],
                     l[3], l[4], l[5], X, Y, Z, nvox, p.k_in, p.i_in);
  return pool_sorted(x, npts, C, nvox, out, out_stride, p.k_in, p.k_out, p.i_in, p.i_out, p.seg_s, p.seg_e, p.tmp,
                     p.tmp_bytes, s);
}

// ------------------------------------------------------------------ fused lift (x) splat
// ViewTransformerLSSVoxel.py:135-143 computes volume = depth_prob[n,d,h,w] * img_feat[n,c,h,w] (242 MB at
// r50, 1.93 GB at r101) and pools it.  Here the product is formed on the fly inside the per-voxel sum:
// point id = ((n*D + d)*H + h)*W + w indexes depth directly and selects the context row (n,h,w).
// Products are rounded to fp32 before the add (no FMA), in ascending point id: bit-equal to pooling the
// materialised volume with the stable order.
__global__ __launch_bounds__(256) void k_lift_pool_sum(const float* __restrict__ depth, const float* __restrict__ feat,
                                                        const uint32_t* __restrict__ ids,
                                                        const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ seg_end, int nvox, int C, int D, int HW,
                                                        float* __restrict__ out, int out_stride) {
  const int v = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (v >= nvox) return;
  if (C <= 128) pool_row<true, 2>(feat, depth, ids, seg_start[v], seg_end[v], lane, C, D, HW, out + (size_t)v * out_stride);
  else pool_row<true, 4>(feat, depth, ids, seg_start[v], seg_end[v], lane, C, D, HW, out + (size_t)v * out_stride);
}

static int lift_splat_impl(const float* depth, const float* feat_nhwc, const float* geom, const float* mats,
                           const float* xs, const float* ys, const float* ds, int N, int D, int H, int W, int C,
                           int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                           int out_stride, void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(depth && feat_nhwc && out && lo_dx_host && N > 0 && D > 0 && H > 0 && W > 0, "lift_splat: bad args");
  COOCC_CHECK_ARG(geom || (mats && xs && ys && ds), "lift_splat: geometry missing");
  COOCC_CHECK_ARG(C > 0 && C % 4 == 0 && ((uintptr_t)feat_nhwc & 15) == 0 && out_stride % 4 == 0 && out_stride >= C &&
                      ((uintptr_t)out & 15) == 0,
                  "lift_splat: C % 4 == 0 and 16-byte aligned rows");
  const long long npts_ll = (long long)N * D * H * W, nvox_ll = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(npts_ll < (1ll << 31) && nvox_ll > 0 && nvox_ll < (1ll << 31) && pts_per_batch > 0, "lift_splat: sizes");
  const int npts = (int)npts_ll, nvox = (int)nvox_ll;
  PoolWs p;
  int rc = carve(ws, ws_bytes, npts, nvox, &p);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  const float* l = lo_dx_host;
  if (geom)
    hipLaunchKernelGGL(k_quantize_geom, dim3(cdiv(npts, 256)), dim3(256), 0, s, geom, npts, pts_per_batch, l[0], l[1], l[2],
                       l[3], l[4], l[5], X, Y, Z, nvox, p.k_in, p.i_in);
  else
    hipLaunchKernelGGL(k_quantize_cams, dim3(cdiv(npts, 256)), dim3(256), 0, s, mats, xs, ys, ds, D, H, W, npts,
                       pts_per_batch, l[0], l[1], l[2], l[3], l[4], l[5], X, Y, Z, nvox, p.k_in, p.i_in);
  int bits = 1;
  while ((1ll << bits) <= nvox) ++bits;
  COOCC_HIP(rocprim::radix_sort_pairs(p.tmp, p.tmp_bytes, p.k_in, p.k_out, p.i_in, p.i_out, (size_t)npts, 0, bits, s));
  COOCC_HIP(hipMemsetAsync(p.seg_s, 0, 2 * align256(sizeof(int32_t) * (size_t)nvox), s));
  hipLaunchKernelGGL(k_segment_bounds, dim3(cdiv(npts, 256)), dim3(256), 0, s, p.k_out, npts, nvox, p.seg_s, p.seg_e);
  hipLaunchKernelGGL(k_lift_pool_sum, dim3(cdiv(nvox, 4)), dim3(256), 0, s, depth, feat_nhwc, p.i_out, p.seg_s, p.seg_e,
                     nvox, C, D, H * W, out, out_stride);
  COOCC_LAUNCH_CHECK("lift_splat");
  return COOCC_OK;
}

extern "C" int coocc_lift_splat(const float* depth, const float* feat_nhwc, const float* geom, int N, int D, int H, int W,
                                int C, int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                                int out_stride, void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(geom, "lift_splat: null geom");
  return lift_splat_impl(depth, feat_nhwc, geom, nullptr, nullptr, nullptr, nullptr, N, D, H, W, C, pts_per_batch,
                         lo_dx_host, B, X, Y, Z, out, out_stride, ws, ws_bytes, stream);
}

extern "C" int coocc_lift_splat_cams(const float* depth, const float* feat_nhwc, const float* mats, const float* xs,
                                     const float* ys, const float* ds, int N, int D, int H, int W, int C,
                                     int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                                     int out_stride, void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(mats && xs && ys && ds, "lift_splat_cams: null camera data");
  return lift_splat_impl(depth, feat_nhwc, nullptr, mats, xs, ys, ds, N, D, H, W, C, pts_per_batch, lo_dx_host, B, X, Y,
                         Z, out, out_stride, ws, ws_bytes, stream);
}

extern "C" int coocc_bev_pool_coords(const float* x, const int64_t* coords, int n, int C, int B, int X, int Y, int Z,
                                     float* out, int out_stride, void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(x && coords && out && n > 0 && C > 0 && C % 4 == 0, "bev_pool_coords: bad args");
  COOCC_CHECK_ARG(((uintptr_t)x & 15) == 0 && out_stride % 4 == 0 && out_stride >= C, "bev_pool_coords: alignment");
  const long long nvox_ll = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(nvox_ll > 0 && nvox_ll < (1ll << 31), "bev_pool_coords: grid too large");
  const int nvox = (int)nvox_ll;
  PoolWs p;
  int rc = carve(ws, ws_bytes, n, nvox, &p);
  if (rc) return rc;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(k_keys_from_coords, dim3(cdiv(n, 256)), dim3(256), 0, s, coords, n, B, X, Y, Z, nvox, p.k_in, p.i_in);
  return pool_sorted(x, n, C, nvox, out, out_stride, p.k_in, p.k_out, p.i_in, p.i_out, p.seg_s, p.seg_e, p.tmp, p.tmp_bytes,
                     s);
}
