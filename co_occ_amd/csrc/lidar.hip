// LiDAR-side producer of the fuser's pts_voxel_feats: hard voxelisation
// (M/ops/voxel/src/voxelization_cpu.cpp:44-104 semantics = the deterministic CUDA path), HardSimpleVFE
// (M/models/voxel_encoders/voxel_encoder.py:43-45) and the rule-book builders of the sparse encoder
// (P/coocc/voxel_encoder/sparse_lidar_enc.py; spconv 2.3.6 is un-vendored upstream, its published semantics are
// restated: SubMConv3d keeps the active set, SparseConv3d(k,s,p) activates every output whose receptive field holds an
// active input).  The convolutions themselves run through coocc_conv_fwd in row-table mode: a rule book IS a
// [taps][M] table of input rows (-1 = inactive).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <rocprim/rocprim.hpp>

#include "common.h"

// ------------------------------------------------------------------ hard voxelisation
// Reference (sequential): for each point in order, c = floor((p - lo) / vs) per axis (skip if outside), voxel index =
// order of first appearance (new voxels beyond max_voxels are dropped), the first max_points points of a voxel are kept.
// Parallel restatement: stable sort of (voxel key, point index); the head of each key segment is the voxel's first
// point; an exclusive scan of "is a first point" over the point order gives the appearance rank.
__global__ __launch_bounds__(256) void k_vox_keys(const float* __restrict__ pts, int n, int F, float lox, float loy, float loz,
                                                   float vx, float vy, float vz, int gx, int gy, int gz,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + (size_t)i * F;
  const int cx = (int)floorf((p[0] - lox) / vx), cy = (int)floorf((p[1] - loy) / vy), cz = (int)floorf((p[2] - loz) / vz);
  const bool ok = cx >= 0 && cx < gx && cy >= 0 && cy < gy && cz >= 0 && cz < gz;
  keys[i] = ok ? (uint32_t)(((size_t)cz * gy + cy) * gx + cx) : 0xFFFFFFFFu;     // (z, y, x) order of the coors
  ids[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_vox_heads(const uint32_t* __restrict__ keys_s, const uint32_t* __restrict__ ids_s, int n,
                                                    int32_t* __restrict__ isfirst, int32_t* __restrict__ seg_head) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys_s[i];
  const bool head = k != 0xFFFFFFFFu && (i == 0 || keys_s[i - 1] != k);
  if (head) isfirst[ids_s[i]] = 1;
  // position of this element's segment head: heads mark themselves, the rest is filled by a max-scan
  seg_head[i] = head ? i : 0;
}

__global__ __launch_bounds__(256) void k_vox_fill(const float* __restrict__ pts, int n, int F, const uint32_t* __restrict__ keys_s,
                                                   const uint32_t* __restrict__ ids_s, const int32_t* __restrict__ seg_head,
                                                   const int32_t* __restrict__ rank, int gx, int gy, int max_points,
                                                   int max_voxels, float* __restrict__ voxels, int32_t* __restrict__ coors,
                                                   int32_t* __restrict__ num_points) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys_s[i];
  if (k == 0xFFFFFFFFu) return;
  const int h = seg_head[i];
  const int r = rank[ids_s[h]];                 // appearance rank of the voxel
  if (r >= max_voxels) return;
  const int j = i - h;                          // position inside the voxel, in point order (stable sort)
  if (j == 0) {
    const int x = (int)(k % (uint32_t)gx), y = (int)((k / (uint32_t)gx) % (uint32_t)gy), z = (int)(k / ((uint32_t)gx * gy));
    coors[r * 3 + 0] = z; coors[r * 3 + 1] = y; coors[r * 3 + 2] = x;
  }
  if (j < max_points) {
    const float* p = pts + (size_t)ids_s[i] * F;
    float* o = voxels + ((size_t)r * max_points + j) * F;
    for (int f = 0; f < F; ++f) o[f] = p[f];
    atomicAdd(num_points + r, 1);
  }
}

__global__ void k_vox_count(const int32_t* __restrict__ rank, const int32_t* __restrict__ isfirst, int n, int max_voxels,
                            int32_t* __restrict__ count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int total = n > 0 ? rank[n - 1] + isfirst[n - 1] : 0;
    *count = total < max_voxels ? total : max_voxels;
  }
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t coocc_voxelize_ws(int n) {
  if (n <= 0) return 256;
  size_t tmp = 0, tmp2 = 0, tmp3 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (size_t)n, 0, 32, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, tmp2, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)n, rocprim::plus<int32_t>(),
                                (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, tmp3, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, rocprim::maximum<int32_t>(),
                                (hipStream_t)0);
  return 7 * al256(sizeof(uint32_t) * (size_t)n) + al256(std::max(tmp, std::max(tmp2, tmp3))) + 256;
}

extern "C" int coocc_voxelize_hard(const float* points, int n, int F, const float* range_host, const float* voxel_size_host,
                                   int max_points, int max_voxels, float* voxels, int32_t* coors, int32_t* num_points,
                                   int32_t* count, void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG((points || n == 0) && range_host && voxel_size_host && voxels && coors && num_points && count && n >= 0 && F >= 3 &&
                      max_points > 0 && max_voxels > 0,
                  "voxelize_hard: bad args");
  const float* rg = range_host; const float* vs = voxel_size_host;
  const int gx = (int)roundf((rg[3] - rg[0]) / vs[0]), gy = (int)roundf((rg[4] - rg[1]) / vs[1]), gz = (int)roundf((rg[5] - rg[2]) / vs[2]);
  COOCC_CHECK_ARG(gx > 0 && gy > 0 && gz > 0 && (long long)gx * gy * gz < 0xFFFFFFFFll, "voxelize_hard: bad grid");
  hipStream_t s = as_stream(stream);
  COOCC_HIP(hipMemsetAsync(voxels, 0, sizeof(float) * (size_t)max_voxels * max_points * F, s));
  COOCC_HIP(hipMemsetAsync(num_points, 0, sizeof(int32_t) * (size_t)max_voxels, s));
  COOCC_HIP(hipMemsetAsync(coors, 0, sizeof(int32_t) * 3 * (size_t)max_voxels, s));
  COOCC_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), s));
  if (n == 0) return COOCC_OK;
  if (!ws || ws_bytes < coocc_voxelize_ws(n)) return coocc_set_error(COOCC_ENOMEM, "voxelize_hard: workspace too small");
  char* c = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const size_t a = al256(sizeof(uint32_t) * (size_t)n);
  uint32_t* k_in = (uint32_t*)c; c += a;
  uint32_t* k_out = (uint32_t*)c; c += a;
  uint32_t* i_in = (uint32_t*)c; c += a;
  uint32_t* i_out = (uint32_t*)c; c += a;
  int32_t* isfirst = (int32_t*)c; c += a;
  int32_t* rank = (int32_t*)c; c += a;
  int32_t* seg = (int32_t*)c; c += a;
  void* tmp = c;
  size_t tmp_bytes = ws_bytes - (size_t)(c - (char*)ws);
  hipLaunchKernelGGL(k_vox_keys, dim3(cdiv(n, 256)), dim3(256), 0, s, points, n, F, rg[0], rg[1], rg[2], vs[0], vs[1], vs[2], gx, gy,
                     gz, k_in, i_in);
  COOCC_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, i_in, i_out, (size_t)n, 0, 32, s));
  COOCC_HIP(hipMemsetAsync(isfirst, 0, sizeof(int32_t) * (size_t)n, s));
  hipLaunchKernelGGL(k_vox_heads, dim3(cdiv(n, 256)), dim3(256), 0, s, k_out, i_out, n, isfirst, seg);
  COOCC_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, seg, seg, (size_t)n, rocprim::maximum<int32_t>(), s));
  COOCC_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, isfirst, rank, 0, (size_t)n, rocprim::plus<int32_t>(), s));
  hipLaunchKernelGGL(k_vox_fill, dim3(cdiv(n, 256)), dim3(256), 0, s, points, n, F, k_out, i_out, seg, rank, gx, gy, max_points,
                     max_voxels, voxels, coors, num_points);
  hipLaunchKernelGGL(k_vox_count, dim3(1), dim3(64), 0, s, rank, isfirst, n, max_voxels, count);
  COOCC_LAUNCH_CHECK("voxelize_hard");
  return COOCC_OK;
}

// HardSimpleVFE: mean of the first nf features over the valid points of each voxel -> rows [M, out_stride]
__global__ __launch_bounds__(256) void k_vfe_mean(const float* __restrict__ voxels, const int32_t* __restrict__ num_points, int M,
                                                   int max_points, int F, int nf, float* __restrict__ out, int out_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M * nf) return;
  const int m = i / nf, f = i % nf;
  float s = 0.f;
  for (int j = 0; j < max_points; ++j) s += voxels[((size_t)m * max_points + j) * F + f];   // padded slots are zero, as upstream
  out[(size_t)m * out_stride + f] = s / (float)num_points[m];
}

extern "C" int coocc_vfe_mean(const float* voxels, const int32_t* num_points, int M, int max_points, int F, int nf, float* out,
                              int out_stride, void* stream) {
  COOCC_CHECK_ARG(voxels && num_points && out && M >= 0 && nf > 0 && nf <= F && out_stride >= nf, "vfe_mean: bad args");
  if (M == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_vfe_mean, dim3(cdiv((long long)M * nf, 256)), dim3(256), 0, as_stream(stream), voxels, num_points, M,
                     max_points, F, nf, out, out_stride);
  COOCC_LAUNCH_CHECK("k_vfe_mean");
  return COOCC_OK;
}

// ------------------------------------------------------------------ sparse rule books (dense index map per resolution)
// coors: [M,3] (z, y, x) int32 (batch 1); dims = (D, H, W) = spatial shape in (z, y, x) order.
__global__ __launch_bounds__(256) void k_sp_index_map(const int32_t* __restrict__ coors, int M, int H, int W,
                                                       int32_t* __restrict__ map) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  map[((size_t)coors[i * 3] * H + coors[i * 3 + 1]) * W + coors[i * 3 + 2]] = i;
}

extern "C" int coocc_sparse_index_map(const int32_t* coors, int M, int D, int H, int W, int32_t* map, void* stream) {
  COOCC_CHECK_ARG(coors && map && M >= 0 && D > 0 && H > 0 && W > 0, "sparse_index_map: bad args");
  hipStream_t s = as_stream(stream);
  COOCC_HIP(hipMemsetAsync(map, 0xFF, sizeof(int32_t) * (size_t)D * H * W, s));
  if (M == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_sp_index_map, dim3(cdiv(M, 256)), dim3(256), 0, s, coors, M, H, W, map);
  COOCC_LAUNCH_CHECK("k_sp_index_map");
  return COOCC_OK;
}

// table[t][o] = input row feeding output o through tap t = (kd*k + kh)*k + kw (spconv KRSC order), or -1.
// Input position = o * stride - pad + (kd, kh, kw); SubMConv3d: stride 1, pad k/2, outputs = inputs.
__global__ __launch_bounds__(256) void k_sp_table(const int32_t* __restrict__ out_coors, int Mo, int Di, int Hi, int Wi, int k,
                                                   int stride, int pad, const int32_t* __restrict__ in_map,
                                                   int32_t* __restrict__ table) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int taps = k * k * k;
  if (i >= (long long)Mo * taps) return;
  const int t = (int)(i / Mo), o = (int)(i % Mo);
  const int kw = t % k, kh = (t / k) % k, kd = t / (k * k);
  const int z = out_coors[o * 3] * stride - pad + kd, y = out_coors[o * 3 + 1] * stride - pad + kh,
            x = out_coors[o * 3 + 2] * stride - pad + kw;
  int r = -1;
  if ((unsigned)z < (unsigned)Di && (unsigned)y < (unsigned)Hi && (unsigned)x < (unsigned)Wi) r = in_map[((size_t)z * Hi + y) * Wi + x];
  table[i] = r;
}

extern "C" int coocc_sparse_conv_table(const int32_t* out_coors, int Mo, int Di, int Hi, int Wi, int ksize, int stride, int pad,
                                       const int32_t* in_map, int32_t* table, void* stream) {
  COOCC_CHECK_ARG(out_coors && in_map && table && Mo >= 0 && ksize > 0 && stride > 0 && pad >= 0, "sparse_conv_table: bad args");
  if (Mo == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_sp_table, dim3(cdiv((long long)Mo * ksize * ksize * ksize, 256)), dim3(256), 0, as_stream(stream), out_coors, Mo,
                     Di, Hi, Wi, ksize, stride, pad, in_map, table);
  COOCC_LAUNCH_CHECK("k_sp_table");
  return COOCC_OK;
}

// SparseConv3d active-output flags: every output o = (i + pad - tap) / stride (exact, in range) of an active input i
__global__ __launch_bounds__(256) void k_sp_down_flags(const int32_t* __restrict__ coors, int M, int k, int stride, int pad, int Do,
                                                        int Ho, int Wo, uint8_t* __restrict__ flags) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int taps = k * k * k;
  if (i >= (long long)M * taps) return;
  const int m = (int)(i / taps), t = (int)(i % taps);
  const int kw = t % k, kh = (t / k) % k, kd = t / (k * k);
  const int z = coors[m * 3] + pad - kd, y = coors[m * 3 + 1] + pad - kh, x = coors[m * 3 + 2] + pad - kw;
  if (z < 0 || y < 0 || x < 0 || z % stride || y % stride || x % stride) return;
  const int oz = z / stride, oy = y / stride, ox = x / stride;
  if (oz < Do && oy < Ho && ox < Wo) flags[((size_t)oz * Ho + oy) * Wo + ox] = 1;
}

extern "C" int coocc_sparse_down_flags(const int32_t* coors, int M, int ksize, int stride, int pad, int Do, int Ho, int Wo,
                                       uint8_t* flags, void* stream) {
  COOCC_CHECK_ARG(coors && flags && M >= 0 && ksize > 0 && stride > 0 && Do > 0 && Ho > 0 && Wo > 0, "sparse_down_flags: bad args");
  hipStream_t s = as_stream(stream);
  COOCC_HIP(hipMemsetAsync(flags, 0, (size_t)Do * Ho * Wo, s));
  if (M == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_sp_down_flags, dim3(cdiv((long long)M * ksize * ksize * ksize, 256)), dim3(256), 0, s, coors, M, ksize, stride,
                     pad, Do, Ho, Wo, flags);
  COOCC_LAUNCH_CHECK("k_sp_down_flags");
  return COOCC_OK;
}

// linear (z*H + y)*W + x ids -> coors [n,3]; and coors -> channels-last row ids (x*Y + y)*Z + z of the dense
// [B,C,W,H,D] = (x,y,z) volume that SparseConvTensor.dense().permute(0,1,4,3,2) produces
__global__ __launch_bounds__(256) void k_sp_lin_to_coors(const int32_t* __restrict__ lin, int n, int H, int W,
                                                          int32_t* __restrict__ coors, int32_t* __restrict__ dense_rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int l = lin[i];
  const int x = l % W, y = (l / W) % H, z = l / (W * H);
  coors[i * 3] = z; coors[i * 3 + 1] = y; coors[i * 3 + 2] = x;
  if (dense_rows) dense_rows[i] = (x * H + y) * (int)gridDim.y + z;   // gridDim.y carries D
}

extern "C" int coocc_sparse_lin_to_coors(const int32_t* lin, int n, int D, int H, int W, int32_t* coors, int32_t* dense_rows,
                                         void* stream) {
  COOCC_CHECK_ARG(lin && coors && n >= 0 && D > 0 && D < 65536 && H > 0 && W > 0, "sparse_lin_to_coors: bad args");
  if (n == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_sp_lin_to_coors, dim3(cdiv(n, 256), D), dim3(256), 0, as_stream(stream), lin, n, H, W, coors, dense_rows);
  COOCC_LAUNCH_CHECK("k_sp_lin_to_coors");
  return COOCC_OK;
}
