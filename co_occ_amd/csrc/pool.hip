// Lift-splat side of the path: frustum geometry (P1) and voxel pooling (P2).
//
// The reference pools with argsort(ranks) + an interval-sum kernel (bev_pool.py:83-97,
// bev_pool_cuda.cu:20-42); argsort is unstable there, so the fp32 summation order inside a
// voxel is unspecified.  Here every voxel sums its rows in ASCENDING POINT ID -- deterministic
// and equal to the oracle's (stable) order -- without any sort of the point set:
//   keys + per-voxel histogram (integer atomics, one per wave-level group of equal keys; each point keeps
//   its slot) -> exclusive scan -> CSR fill (order inside a voxel arbitrary) -> the sum kernel orders each
//   voxel's few ids itself
//   (wave-level rank for <= 64 points: 4.5 points per voxel on average at r50, 23 at r101; an LDS
//   bitonic sort by a whole workgroup for the few hundred voxels next to the cameras).
// Round 1 ran rocPRIM's stable radix sort over all (key, id) pairs: three passes over 473 k -
// 3.8 M pairs were 80 % of the kernel time.  The [EXT] entry points that take already-sorted
// intervals are provided as well.
#include <cstring>

#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ P1: get_geometry
// ViewTransformerLSSBEVDepth.py:117-150.  mats[cam] = {A=inv(post_rots)[9], post_trans[3],
// Cm=rots@inv(intrins[:3,:3])[9], trans[3], bda[:3,:3][9], shift[3] = intrins[:3,3] of a 3x4/4x4 KITTI intrinsic (:136-139,
// else 0), bda[:3,3] of a 4x4 bda (:145-148, else 0)} (COOCC_CAM_FLOATS = 39 floats); xs/ys/ds = the frustum axes of
// create_frustum (:104-115) computed by the host with the same torch calls.  The same chain is the module-level
// get_frustum of the detector (P/coocc/detectors/coocc_ray.py:732-776).
#include "geometry.h"

// The 39 per-camera constants from the raw calibration tensors in ONE tiny launch (3x3 inverses by the adjugate in fp64,
// rounded to fp32).  The host-side torch form needs two torch.inverse calls, which synchronise with the host to read their
// status word -- 0.24 ms per sample in front of a 0.05 ms pooling kernel, and a stall of the enqueue-ahead pipeline.
__device__ __forceinline__ void inv3x3(const float* a, float* o) {
  const double a00 = a[0], a01 = a[1], a02 = a[2], a10 = a[3], a11 = a[4], a12 = a[5], a20 = a[6], a21 = a[7], a22 = a[8];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02, r = 1.0 / det;
  o[0] = (float)(c00 * r); o[1] = (float)((a02 * a21 - a01 * a22) * r); o[2] = (float)((a01 * a12 - a02 * a11) * r);
  o[3] = (float)(c01 * r); o[4] = (float)((a00 * a22 - a02 * a20) * r); o[5] = (float)((a02 * a10 - a00 * a12) * r);
  o[6] = (float)(c02 * r); o[7] = (float)((a01 * a20 - a00 * a21) * r); o[8] = (float)((a00 * a11 - a01 * a10) * r);
}

__global__ void k_camera_mats(const float* __restrict__ rots, const float* __restrict__ trans, const float* __restrict__ intrins,
                              const float* __restrict__ post_rots, const float* __restrict__ post_trans,
                              const float* __restrict__ bda, int B, int N, int kdim, int bdim, float* __restrict__ mats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  float* m = mats + (size_t)i * COOCC_CAM_FLOATS;
  inv3x3(post_rots + (size_t)i * 9, m);
  for (int j = 0; j < 3; ++j) m[9 + j] = post_trans[(size_t)i * 3 + j];
  float K[9], Ki[9];
  const float* kin = intrins + (size_t)i * kdim * kdim;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) K[r * 3 + c] = kin[r * kdim + c];
  inv3x3(K, Ki);
  const float* R = rots + (size_t)i * 9;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
#pragma clang fp contract(off)
      m[12 + r * 3 + c] = (R[r * 3 + 0] * Ki[0 * 3 + c] + R[r * 3 + 1] * Ki[1 * 3 + c]) + R[r * 3 + 2] * Ki[2 * 3 + c];
    }
  for (int j = 0; j < 3; ++j) m[21 + j] = trans[(size_t)i * 3 + j];
  const float* bd = bda + (size_t)(i / N) * bdim * bdim;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) m[24 + r * 3 + c] = bd[r * bdim + c];
  for (int j = 0; j < 3; ++j) {
    m[33 + j] = kdim == 4 ? kin[j * 4 + 3] : 0.f;
    m[36 + j] = bdim == 4 ? bd[j * 4 + 3] : 0.f;
  }
}

extern "C" int coocc_camera_mats(const float* rots, const float* trans, const float* intrins, const float* post_rots,
                                 const float* post_trans, const float* bda, int B, int N, int intrin_dim, int bda_dim,
                                 float* mats, void* stream) {
  COOCC_CHECK_ARG(rots && trans && intrins && post_rots && post_trans && bda && mats && B > 0 && N > 0, "camera_mats: bad args");
  COOCC_CHECK_ARG((intrin_dim == 3 || intrin_dim == 4) && (bda_dim == 3 || bda_dim == 4), "camera_mats: 3x3 or 4x4 matrices");
  hipLaunchKernelGGL(k_camera_mats, dim3(cdiv(B * N, 64)), dim3(64), 0, as_stream(stream), rots, trans, intrins, post_rots,
                     post_trans, bda, B, N, intrin_dim, bda_dim, mats);
  COOCC_LAUNCH_CHECK("k_camera_mats");
  return COOCC_OK;
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

// ------------------------------------------------------------------ CSR build: histogram -> scan -> fill
// One atomic pass: the lanes of a wave that hold the same key are combined (a voxel next to a camera receives up to 2600
// points, consecutive ids mostly -- one atomicAdd per lane would serialise on that address), the group leader adds the
// group size to count[key] and every member keeps (old value + its position in the group) as its slot inside the voxel.
// After the scan the fill is a plain scatter: ids[start[key] + slot] = id.
constexpr int POOL_MEDIUM = 256;     // <= 64 points: wave rank sort in registers; <= 256: wave rank sort through LDS; else workgroup

// Launch 1 of 3: voxel key of every point (written for the fill pass) + the histogram above, in one kernel.
// MODE 0: geometry tensor [npts,3]; 1: geometry from the camera matrices; 2: integer coords [n,4] (bev_pool drop-in)
struct KeySrc {
  const float* geom; const float* mats; const float* xs; const float* ys; const float* ds; const int64_t* coords;
  int D, fH, fW, pts_per_batch, B;
  float lox, loy, loz, dx, dy, dz;
};
template <int MODE>
__global__ __launch_bounds__(256) void k_keys_hist(KeySrc a, int npts, int X, int Y, int Z, int nvox, uint32_t* __restrict__ keys,
                                                    int32_t* __restrict__ count, int32_t* __restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t k = 0xFFFFFFFFu;
  if (i < npts) {
    if (MODE == 0) {
      k = voxel_key(a.geom[(size_t)i * 3 + 0], a.geom[(size_t)i * 3 + 1], a.geom[(size_t)i * 3 + 2], i / a.pts_per_batch, a.lox, a.loy,
                    a.loz, a.dx, a.dy, a.dz, X, Y, Z, nvox);
    } else if (MODE == 1) {
      float gx, gy, gz;
      geometry_point(a.mats, a.xs, a.ys, a.ds, (size_t)i, a.D, a.fH, a.fW, gx, gy, gz);
      k = voxel_key(gx, gy, gz, i / a.pts_per_batch, a.lox, a.loy, a.loz, a.dx, a.dy, a.dz, X, Y, Z, nvox);
    } else {
      long long x = a.coords[(size_t)i * 4 + 0], y = a.coords[(size_t)i * 4 + 1], z = a.coords[(size_t)i * 4 + 2],
                b = a.coords[(size_t)i * 4 + 3];
      bool ok = x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z && b >= 0 && b < a.B;
      k = ok ? (uint32_t)(((b * X + x) * Y + y) * Z + z) : (uint32_t)nvox;
    }
    keys[i] = k;
  }
  // Lanes that hold the same key form a group: its leader adds the group size to count[key], every member takes
  // (old value + its position in the group) as its slot.  The grouping loop is pure ALU (one ballot per DISTINCT key of the
  // wave); the atomics of all leaders are then issued TOGETHER and waited for once -- issuing one per loop iteration and waiting
  // for its return value put up to 64 memory round trips in series per wave (far depth bins: 64 pixels, 64 voxels).
  const bool valid = k < (uint32_t)nvox;
  unsigned long long remaining = __ballot(valid);
  int leader_of = lane, pos = 0, gsize = 0;
  while (remaining) {
    const int leader = (int)__ffsll((long long)remaining) - 1;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)k, leader);
    const bool mine = valid && k == k0;
    const unsigned long long grp = __ballot(mine);
    if (mine) { leader_of = leader; pos = (int)__popcll(grp & ((1ull << lane) - 1ull)); }
    if (lane == leader) gsize = (int)__popcll(grp);
    remaining &= ~grp;
  }
  int base = 0;
  if (valid && lane == leader_of) base = atomicAdd(&count[k], gsize);
  base = __shfl(base, leader_of);
  if (valid) slot[i] = base + pos;
}

// Launches 2 and 3 of 4: the CSR.  k_scan_local: exclusive scan of count[] inside each 1024-voxel chunk (lstart[]) + the chunk
// totals (tops[]).  k_csr_fill: every workgroup first scans the (<= POOL_MAX_CHUNKS) chunk totals into LDS, then
// (a) scatters its points, ids[lstart[key] + prefix[key >> 10] + slot] = id, no atomics, four independent points per thread and
// iteration (the chain key -> lstart[key] -> store is pure latency), and (b) for the voxels of its own index range writes the
// global start[] the sums read, appends the voxels with more than POOL_MEDIUM points to long_list and zeroes count[] again -- so
// the NEXT call on the workspace needs no memset.  Readers of (a) never read what (b) writes: no ordering between workgroups.
// (A first version of this round ran scan + fill as ONE kernel with two grid barriers: 3 launches, fine alone -- and 10x slower
// inside the pipeline at r101 / stress200, where its 232-256 spinning workgroups wait for CU slots behind the dense graphs'
// workgroups: 17 instead of 190 samples/s.  No grid barriers on a GPU that is shared between streams.)
constexpr int POOL_MAX_CHUNKS = 8192;       // 8.4 M voxels (32 KB of LDS for the chunk prefix)

__global__ __launch_bounds__(1024) void k_scan_local(const int32_t* __restrict__ count, int nvox, int32_t* __restrict__ lstart,
                                                      int32_t* __restrict__ tops, int32_t* __restrict__ nlong) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, v = blockIdx.x * 1024 + tid;
  if (v == 0) *nlong = 0;                       // the previous call's list length (its consumer ran before this launch)
  const int c = v < nvox ? count[v] : 0;
  int inc = c;
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o);
    if ((tid & 63) >= o) inc += n;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  int off = inc - c;
  for (int w = 0; w < (tid >> 6); ++w) off += wsum[w];
  if (v < nvox) lstart[v] = off;
  if (tid == 1023) tops[blockIdx.x] = off + c;
}

__global__ __launch_bounds__(256) void k_csr_fill(const uint32_t* __restrict__ keys, int npts, int nvox, int nblk,
                                                   const int32_t* __restrict__ lstart, const int32_t* __restrict__ tops,
                                                   const int32_t* __restrict__ slot, int32_t* __restrict__ count,
                                                   int32_t* __restrict__ start, int32_t* __restrict__ long_list,
                                                   int32_t* __restrict__ nlong, uint32_t* __restrict__ ids, int slot_mask,
                                                   const float* __restrict__ wts, float* __restrict__ wcsr) {
  extern __shared__ int prefix[];               // [nblk + 1] exclusive prefix of the chunk totals
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int tid = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < nblk; base += 256) {
    const int i = base + tid;
    const int c = i < nblk ? tops[i] : 0;
    int inc = c;
    for (int o = 1; o < 64; o <<= 1) {
      int n = __shfl_up(inc, o);
      if ((tid & 63) >= o) inc += n;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int off = carry + inc - c;
    for (int w = 0; w < (tid >> 6); ++w) off += wsum[w];
    if (i < nblk) prefix[i] = off;
    if (tid == 255) carry_s = off + c;
    __syncthreads();
    carry = carry_s;
  }
  if (tid == 0) prefix[nblk] = carry;
  __syncthreads();
  // (b) this workgroup's share of the voxel-side bookkeeping
  const long long gsz = (long long)gridDim.x * 256;
  for (long long v = (long long)blockIdx.x * 256 + tid; v <= nvox; v += gsz) {
    if (v == nvox) { start[nvox] = prefix[nblk]; break; }
    const int cn = count[v];
    start[v] = lstart[v] + prefix[v >> 10];
    if (cn > POOL_MEDIUM) long_list[atomicAdd(nlong, 1)] = (int)v;
    count[v] = 0;
  }
  // (a) the scatter
  for (long long i0 = (long long)blockIdx.x * 256 + tid; i0 < npts; i0 += 4 * gsz) {
    uint32_t k[4];
    int sl[4], st[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * gsz;
      k[u] = i < npts ? keys[i] : 0xFFFFFFFFu;
      sl[u] = i < npts ? (slot[i] & slot_mask) : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) st[u] = k[u] < (uint32_t)nvox ? lstart[k[u]] + prefix[k[u] >> 10] : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k[u] < (uint32_t)nvox) {
        ids[st[u] + sl[u]] = (uint32_t)(i0 + u * gsz);
        if (wcsr) wcsr[st[u] + sl[u]] = wts[i0 + u * gsz];          // segment form: the weights in CSR order (coalesced reads later)
      }
  }
}

// One wave per voxel row, VEC channels per lane.  Rows are summed in ascending point id with a single accumulator, the
// association of the reference's interval kernel over a stably sorted input.  The segment arrives in arbitrary order (atomic
// slots).  n <= 64: each lane takes one id, its rank is the number of smaller ids in the segment (n broadcast compares,
// n = 4.5 on average) and one ds_permute puts the ids in order.  64 < n <= 256: four ids per lane, ranks by n LDS broadcast
// reads, the sorted ids go through the wave's 1 KB LDS slice.  Then the wave fetches the sorted ids' rows POOL_BATCH at a
// time so that the loads run ahead of the dependent adds.  n > 256 (24 voxels of 57 k at r50, 554 of 73 k at r101, up to
// 2614 points each, next to the cameras): a whole workgroup bitonic-sorts the ids in LDS and thread c sums channel c; these
// workgroups are the FIRST blocks of the same launch, so the long tail runs under the short voxels instead of after them.
constexpr int POOL_BATCH = 16;     // row loads in flight per lane and batch (8: 7 instead of 6 waves per SIMD, r50 the same, r101 0.35 -> 0.40 ms: measured)
constexpr int POOL_LONG_CAP = 4096;   // ids a workgroup sorts in LDS; beyond: a slow selection path (never seen)

template <int VEC> struct PoolVec;
template <> struct PoolVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct PoolVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

// sum of nb (<= 64) rows whose (row, depth) sit in the lanes' registers in ascending id order
template <bool LIFT, int VEC>
__device__ __forceinline__ void pool_accumulate(const float* __restrict__ xc, int C, uint32_t myrow, float mydp, int nb,
                                                typename PoolVec<VEC>::type& acc) {
  typedef typename PoolVec<VEC>::type vec;
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

template <bool LIFT, int VEC>
__device__ __forceinline__ void pool_row(const float* __restrict__ x, const float* __restrict__ depth,
                                         const uint32_t* __restrict__ ids, int s, int e, int lane, int C, int D, int HW,
                                         float* __restrict__ orow, uint32_t* __restrict__ lds /* 256 ids, wave-private */) {
  typedef typename PoolVec<VEC>::type vec;
  const int n = e - s;                      // 0 .. POOL_MEDIUM
  if (n <= 64) {
    uint32_t myid = lane < n ? ids[s + lane] : 0xFFFFFFFFu;
    if (n > 1) {
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (uint32_t)__builtin_amdgcn_readlane((int)myid, j) < myid;   // ids are distinct
      // idle lanes (id = 0xFFFFFFFF) all rank n: they land on lane n, which is not read
      myid = (uint32_t)__builtin_amdgcn_ds_permute(rank << 2, (int)myid);
    }
    uint32_t myrow = myid;
    float mydp = 1.f;
    if (LIFT && lane < n) {
      myrow = (myid / (uint32_t)(D * HW)) * (uint32_t)HW + myid % (uint32_t)HW;
      mydp = depth[myid];
    }
    for (int c0 = 0; c0 < C; c0 += 64 * VEC) {
      const int c = c0 + lane * VEC;
      const bool lane_on = c < C;
      vec acc = (vec)(0.f);
      pool_accumulate<LIFT, VEC>(x + (lane_on ? c : 0), C, myrow, mydp, n, acc);   // idle lanes re-read channel 0
      if (lane_on) *(vec*)(orow + c) = acc;
    }
    return;
  }
  // medium voxel: ids -> LDS (unsorted) -> ranks -> LDS (sorted, after every lane has read what it needs)
  uint32_t mine[POOL_MEDIUM / 64];
  int rank[POOL_MEDIUM / 64];
#pragma unroll
  for (int q = 0; q < POOL_MEDIUM / 64; ++q) {
    mine[q] = q * 64 + lane < n ? ids[s + q * 64 + lane] : 0xFFFFFFFFu;
    lds[q * 64 + lane] = mine[q];
    rank[q] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int j = 0; j < n; ++j) {
    const uint32_t o = lds[j];              // broadcast read
#pragma unroll
    for (int q = 0; q < POOL_MEDIUM / 64; ++q) rank[q] += o < mine[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < POOL_MEDIUM / 64; ++q)
    if (q * 64 + lane < n) lds[rank[q]] = mine[q];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int c0 = 0; c0 < C; c0 += 64 * VEC) {
    const int c = c0 + lane * VEC;
    const bool lane_on = c < C;
    vec acc = (vec)(0.f);
    for (int base = 0; base < n; base += 64) {
      const int nb = min(64, n - base);
      const uint32_t myid = lds[base + min(lane, nb - 1)];
      uint32_t myrow = myid;
      float mydp = 1.f;
      if (LIFT) {
        myrow = (myid / (uint32_t)(D * HW)) * (uint32_t)HW + myid % (uint32_t)HW;
        mydp = depth[myid];
      }
      pool_accumulate<LIFT, VEC>(x + (lane_on ? c : 0), C, myrow, mydp, nb, acc);
    }
    if (lane_on) *(vec*)(orow + c) = acc;
  }
}

// n > POOL_MEDIUM: one workgroup per voxel
template <bool LIFT>
__device__ __forceinline__ void pool_long(const float* __restrict__ x, const float* __restrict__ depth, uint32_t* __restrict__ seg,
                                          int n, int C, int D, int HW, float* __restrict__ orow, uint32_t* __restrict__ sid,
                                          float* __restrict__ sdp) {
  const int tid = threadIdx.x;
  if (n <= POOL_LONG_CAP) {
    int np2 = 512;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) sid[i] = i < n ? seg[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < np2; i += 256) {
          const int l = i ^ j;
          if (l > i) {
            const uint32_t a = sid[i], b = sid[l];
            const bool up = (i & k) == 0;
            if ((a > b) == up) { sid[i] = b; sid[l] = a; }
          }
        }
        __syncthreads();
      }
    // decode once (two integer divisions per point), not once per point and channel thread
    for (int i = tid; i < n; i += 256) {
      const uint32_t id = sid[i];
      if (LIFT) {
        sdp[i] = depth[id];
        sid[i] = (id / (uint32_t)(D * HW)) * (uint32_t)HW + id % (uint32_t)HW;
      }
    }
    __syncthreads();
    // the serial chain of the longest voxel (2614 points at r101) bounds the whole launch: 64 row loads in flight per batch
    // (one memory latency per 64 points instead of per 16), then the 64 dependent adds
    constexpr int LB = 64;        // (128, and a rank sort instead of the bitonic network, measured SLOWER in round 4: 233 -> 303 us at r101)
    for (int c = tid; c < C; c += 256) {
      float acc = 0.f;
      for (int j0 = 0; j0 < n; j0 += LB) {
        float r[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {
#pragma clang fp contract(off)
          const int jj = min(j0 + j, n - 1);
          float val = x[(size_t)sid[jj] * C + c];
          if (LIFT) val = sdp[jj] * val;
          r[j] = val;
        }
#pragma unroll
        for (int j = 0; j < LB; ++j)
          if (j0 + j < n) acc = acc + r[j];
      }
      orow[c] = acc;
    }
    __syncthreads();
    return;
  }
  // never seen in practice (> 4096 points in one voxel): the next id in ascending order is found by a block-wide min over
  // the unsorted segment, one point at a time -- O(n^2 / 256), correct for any n
  for (int c = tid; c < C; c += 256) orow[c] = 0.f;
  uint32_t prev = 0;
  bool first = true;
  for (int t = 0; t < n; ++t) {
    uint32_t best = 0xFFFFFFFFu;
    for (int i = tid; i < n; i += 256) {
      const uint32_t me = seg[i];
      if ((first || me > prev) && me < best) best = me;
    }
    sid[tid] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) sid[tid] = min(sid[tid], sid[tid + o]);
      __syncthreads();
    }
    const uint32_t id = sid[0];
    __syncthreads();
    prev = id; first = false;
    const uint32_t row = LIFT ? (id / (uint32_t)(D * HW)) * (uint32_t)HW + id % (uint32_t)HW : id;
    const float dp = LIFT ? depth[id] : 1.f;
    for (int c = tid; c < C; c += 256) {
#pragma clang fp contract(off)
      float val = x[(size_t)row * C + c];
      if (LIFT) val = dp * val;
      orow[c] = orow[c] + val;
    }
  }
  __syncthreads();
}

// blocks [0, long_blocks): the long voxels (persistent over long_list); the rest: one wave per voxel
template <bool LIFT>
__global__ __launch_bounds__(256) void k_pool_sum_csr(const float* __restrict__ x, const float* __restrict__ depth,
                                                       uint32_t* __restrict__ ids, const int32_t* __restrict__ start,
                                                       const int32_t* __restrict__ long_list,
                                                       const int32_t* __restrict__ nlong_p, int long_blocks, int nvox, int C,
                                                       int D, int HW, float* __restrict__ out, int out_stride) {
  __shared__ uint32_t sid[POOL_LONG_CAP];
  __shared__ float sdp[LIFT ? POOL_LONG_CAP : 1];
  if ((int)blockIdx.x < long_blocks) {
    const int nlong = *nlong_p;
    for (int li = blockIdx.x; li < nlong; li += long_blocks) {
      const int v = long_list[li];
      const int s = start[v];
      pool_long<LIFT>(x, depth, ids + s, start[v + 1] - s, C, D, HW, out + (size_t)v * out_stride, sid, sdp);
    }
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int v = __builtin_amdgcn_readfirstlane(((int)blockIdx.x - long_blocks) * 4 + wave);
  if (v >= nvox) return;
  const int s = start[v], e = start[v + 1];
  if (e - s > POOL_MEDIUM) return;
  uint32_t* lds = sid + wave * POOL_MEDIUM;
  if (C <= 128) pool_row<LIFT, 2>(x, depth, ids, s, e, lane, C, D, HW, out + (size_t)v * out_stride, lds);
  else pool_row<LIFT, 4>(x, depth, ids, s, e, lane, C, D, HW, out + (size_t)v * out_stride, lds);
}

// ------------------------------------------------------------------ round 5: ray-segment form of the fused lift (x) splat
// The reference's order inside a voxel is unspecified (bev_pool.py:92 is an unstable argsort; bev_pool_cuda.cu:37-40 then adds
// serially), so the contract is "deterministic and within fp32 rounding of any order", not "ascending point id".  The fused form
// uses that freedom.  Points are walked PIXEL-major (t = pixel * D + d, lanes along the depth axis of one pixel's ray): the
// consecutive depth bins of a ray that fall into the same voxel form one SEGMENT, whose weight is the sum of its depth
// probabilities (added in ascending d), and a voxel's row is sum over its segments, in ascending t, of weight * context row.
// Against the per-point form: 1.5-2x fewer CSR entries and row gathers (a 0.8 m voxel holds 1-3 bins of 0.5 m), the depth value
// is no longer a dependent gather of the sum kernel (the weight sits at wts[t]), and a long voxel (up to 2600 points next to a
// camera) is summed as four partial sums by the four waves of its workgroup instead of one serial chain.
//   launch 1  k_seg_hist     key per point, segment leaders, weights, histogram (one atomic per SEGMENT)
//   launch 2  k_scan_local   (as above)
//   launch 3  k_csr_fill     (as above; the slot word also carries the segment length: reuse form)
//   launch 4  k_pool_sum_seg per-voxel sums: ids ranked inside the voxel (atomic slots arrive in any order), then FMAs in that
//                            order -- the same bits on every run
constexpr int SEG_SLOT_BITS = 24;          // slot inside the voxel (< 16.7 M segments per voxel); bits 24..30: segment length (<= 64)
constexpr int SEG_SLOT_MASK = (1 << SEG_SLOT_BITS) - 1;

// A workgroup owns a STRIP of SEG_STRIP consecutive pixels of one camera: the strip's depth values [D][SEG_STRIP] are staged in
// LDS with coalesced 128-byte rows (the depth tensor is [N, D, H, W]: walking a pixel's ray reads one value per 4 H W bytes --
// a first version did exactly that and fetched every line eight times, once per XCD), then the threads walk the strip
// pixel-major, 64 consecutive depth bins of one ray per wave.  Strips are numbered so that each XCD (workgroup id mod 8) owns
// a contiguous eighth of them.  Geometry tensor mode: the three coordinates of the strip are staged the same way.
constexpr int SEG_STRIP = 8;               // 8 pixels x 112 bins = 3.5 passes of the workgroup; 32-byte runs per depth row (the neighbouring strips run on the same XCD)
constexpr int SEG_HASH_BITS = 11, SEG_HASH = 1 << SEG_HASH_BITS;     // LDS hash cells per round of 1024 points
constexpr int SEG_LDS_ENTRIES = 2048;      // (row, weight) pairs of a long voxel kept in LDS between its two passes
constexpr int SEG_DMAX = 256;              // depth bins staged per strip (32 KB of LDS + padding); more: the ascending-point-id form

// rowmax[row] = max |x[row][:]| by the 32 lanes of a half-wave
__device__ __forceinline__ void row_absmax(const float* __restrict__ x, int C, size_t row, int l32, float* __restrict__ rowmax) {
  const float* r = x + row * (size_t)C;
  float m = 0.f;
  for (int c = l32 * 4; c < C; c += 128) {
    const f32x4 v = *(const f32x4*)(r + c);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (l32 == 0) rowmax[row] = m;
}

// the same for every row, for the reuse form (cached binning, this frame's context rows)
__global__ __launch_bounds__(256) void k_row_absmax(const float* __restrict__ x, int C, int nrows, float* __restrict__ rowmax) {
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row < nrows) row_absmax(x, C, (size_t)row, threadIdx.x & 31, rowmax);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_seg_hist(KeySrc a, const float* __restrict__ depth, int npts, int X, int Y, int Z, int nvox,
                                                   int strips_per_cam, int nstrips, int stage_geom, uint32_t* __restrict__ keys,
                                                   int32_t* __restrict__ count, int32_t* __restrict__ slot, float* __restrict__ wts,
                                                   const float* __restrict__ feat, int C, float* __restrict__ rowmax) {
  extern __shared__ float seg_lds[];       // [D][SEG_STRIP + 1] depth (+ 3 more planes of the same shape in geometry-tensor mode)
  const int tid = threadIdx.x, lane = tid & 63;
  const int per = (nstrips + 7) >> 3;
  const int strip = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (strip >= nstrips) return;
  const int D = a.D, HW = a.fH * a.fW;
  const int n = strip / strips_per_cam, hw0 = (strip - n * strips_per_cam) * SEG_STRIP;
  const int np = min(SEG_STRIP, HW - hw0);                       // pixels in this strip
  constexpr int P = SEG_STRIP + 1;
  const size_t base = (size_t)n * D * HW + hw0;                  // point id of (n, d = 0, hw0)
  for (int e = tid; e < D * SEG_STRIP; e += 256) {
    const int d = e / SEG_STRIP, j = e - d * SEG_STRIP;
    if (j < np) {
      const size_t i = base + (size_t)d * HW + j;
      seg_lds[d * P + j] = depth[i];
      if (MODE == 0 && stage_geom) {
        seg_lds[(D + d) * P + j] = a.geom[i * 3 + 0];
        seg_lds[(2 * D + d) * P + j] = a.geom[i * 3 + 1];
        seg_lds[(3 * D + d) * P + j] = a.geom[i * 3 + 2];
      }
    }
  }
  // max |context row| of the strip's pixels: the long voxels' exact sums take their scale from it (seg_long)
  for (int j = tid >> 5; j < np; j += 8) row_absmax(feat, C, (size_t)n * HW + hw0 + j, tid & 31, rowmax);
  __syncthreads();
  const int total = np * D;
  const size_t t0 = ((size_t)n * HW + hw0) * D;                   // pixel-major index of the strip's first point
  // Histogram through an LDS hash of the strip: next to a camera the 8 pixels of a strip put dozens of segments into the same
  // voxel, and one device-scope atomic per segment made the hottest voxel's counter (1600 atomics at r101) the critical path of
  // the whole launch (87 us); now a workgroup issues ONE global atomic per distinct voxel of <= 1024 points and hands out the
  // slots inside it from the LDS counter.
  constexpr int U = 4;                                           // passes (of 256 points) per hash round
  uint32_t* hkey = (uint32_t*)(seg_lds + (size_t)(MODE == 0 && stage_geom ? 4 : 1) * D * P);
  int* hcnt = (int*)(hkey + SEG_HASH);
  int* hbase = hcnt + SEG_HASH;
  for (int e0 = 0; e0 < total; e0 += 256 * U) {                  // whole waves stay together: the shuffles below need all lanes
    uint32_t ku[U];
    float wu[U];
    int lenu[U], baseu[U], hu[U];
    bool leadu[U];
    for (int i = tid; i < SEG_HASH; i += 256) { hkey[i] = 0xFFFFFFFFu; hcnt[i] = 0; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * 256 + tid;
      uint32_t k = (uint32_t)nvox;
      int pj = -1;
      float dp = 0.f;
      if (e < total) {
        pj = e / D;
        const int d = e - pj * D;
        const int hw = hw0 + pj;
        float gx, gy, gz;
        const size_t i = base + (size_t)d * HW + pj;
        if (MODE == 0) {
          if (stage_geom) { gx = seg_lds[(D + d) * P + pj]; gy = seg_lds[(2 * D + d) * P + pj]; gz = seg_lds[(3 * D + d) * P + pj]; }
          else { gx = a.geom[i * 3 + 0]; gy = a.geom[i * 3 + 1]; gz = a.geom[i * 3 + 2]; }
        } else {
          const int h = hw / a.fW, w = hw - h * a.fW;
          geometry_sample(a.mats + (size_t)n * COOCC_CAM_FLOATS, a.xs[w], a.ys[h], a.ds[d], gx, gy, gz);
        }
        k = voxel_key(gx, gy, gz, (int)(i / (size_t)a.pts_per_batch), a.lox, a.loy, a.loz, a.dx, a.dy, a.dz, X, Y, Z, nvox);
        dp = seg_lds[d * P + pj];
      }
      const bool valid = k < (uint32_t)nvox;
      const uint32_t pk = (uint32_t)__shfl_up((int)k, 1);
      const int pp = __shfl_up(pj, 1);
      const bool leader = valid && (lane == 0 || pk != k || pp != pj);
      // weight of the run that starts at a leader: its own bins in ascending d (runs are 1-3 bins long; one that crosses the
      // wave boundary simply becomes two segments)
      float w = dp;
      int len = 1;
      bool cont = leader;
      for (int o = 1; o < 64; ++o) {
        const uint32_t ko = (uint32_t)__shfl_down((int)k, o);
        const int po = __shfl_down(pj, o);
        const float dv = __shfl_down(dp, o);
        cont = cont && lane + o < 64 && ko == k && po == pj;
        if (!__any(cont)) break;
        if (cont) { w = w + dv; ++len; }
      }
      ku[u] = k; wu[u] = w; lenu[u] = len; leadu[u] = leader;
      baseu[u] = 0; hu[u] = 0;
      if (leader) {
        unsigned h = (k * 2654435761u) >> (32 - SEG_HASH_BITS);
        for (;;) {                                               // open addressing; <= 1024 leaders in SEG_HASH = 2048 cells
          const uint32_t prev = atomicCAS(&hkey[h], 0xFFFFFFFFu, k);
          if (prev == 0xFFFFFFFFu || prev == k) break;
          h = (h + 1) & (SEG_HASH - 1);
        }
        hu[u] = (int)h;
        baseu[u] = atomicAdd(&hcnt[h], 1);
      }
    }
    __syncthreads();
    for (int i = tid; i < SEG_HASH; i += 256)
      if (hkey[i] != 0xFFFFFFFFu) hbase[i] = atomicAdd(&count[hkey[i]], hcnt[i]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * 256 + tid;
      if (e < total) {
        const size_t t = t0 + e;
        keys[t] = leadu[u] ? ku[u] : (uint32_t)nvox;             // the fill pass scatters leaders only
        if (leadu[u]) {
          slot[t] = (hbase[hu[u]] + baseu[u]) | (lenu[u] << SEG_SLOT_BITS);
          wts[t] = wu[u];
        }
      }
    }
    __syncthreads();                                             // the next round re-initialises the hash
  }
}

// weight of segment t in the reuse form (the CSR of an earlier frame, this frame's depth): the leader's run re-added in the order
// k_seg_hist used
__device__ __forceinline__ float seg_weight_from_depth(const float* __restrict__ depth, const int32_t* __restrict__ slot, uint32_t t,
                                                       int D, int HW) {
  const int len = slot[t] >> SEG_SLOT_BITS;
  const uint32_t pix = t / (uint32_t)D, d = t - pix * (uint32_t)D;
  const uint32_t n = pix / (uint32_t)HW, hw = pix - n * (uint32_t)HW;
  const float* q = depth + ((size_t)n * D + d) * HW + hw;
  float w = q[0];
  for (int j = 1; j < len; ++j) w = w + q[(size_t)j * HW];
  return w;
}

// sum of nb (<= 64) weighted rows whose (row, weight) sit in the lanes' registers in ascending t; NB loads in flight
template <int VEC, int NB>
__device__ __forceinline__ void seg_accumulate(const float* __restrict__ xc, int C, uint32_t myrow, float myw, int nb,
                                               typename PoolVec<VEC>::type& acc) {
  typedef typename PoolVec<VEC>::type vec;
  for (int j0 = 0; j0 < nb; j0 += NB) {
    vec r[NB];
    float wj[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int jj = min(j0 + j, nb - 1);
      const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)myrow, jj);
      wj[j] = j0 + j < nb ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myw), jj)) : 0.f;
      r[j] = *(const vec*)(xc + (size_t)row * C);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = __fmaf_rn(wj[j], r[j][e], acc[e]);      // a tail entry repeats the last row with weight 0
  }
}

// wcsr: the segments' weights in CSR order (k_csr_fill); REUSE: recomputed from this frame's depth instead
template <int VEC, bool REUSE>
__device__ __forceinline__ void seg_row(const float* __restrict__ x, const float* __restrict__ depth, const float* __restrict__ wcsr,
                                        const int32_t* __restrict__ slot, const uint32_t* __restrict__ ids, int s, int n, int lane,
                                        int C, int D, int HW, float* __restrict__ orow, uint32_t* __restrict__ lds,
                                        float* __restrict__ ldw) {
  typedef typename PoolVec<VEC>::type vec;
  if (n == 0) {
    for (int c = lane * VEC; c < C; c += 64 * VEC) *(vec*)(orow + c) = (vec)(0.f);
    return;
  }
  if (n <= 64) {
    uint32_t myid = lane < n ? ids[s + lane] : 0xFFFFFFFFu;
    float myw = (!REUSE && lane < n) ? wcsr[s + lane] : 0.f;
    if (n > 1) {
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (uint32_t)__builtin_amdgcn_readlane((int)myid, j) < myid;   // ids are distinct
      myid = (uint32_t)__builtin_amdgcn_ds_permute(rank << 2, (int)myid);     // idle lanes all rank n: lane n is not read
      if (!REUSE) myw = __int_as_float(__builtin_amdgcn_ds_permute(rank << 2, __float_as_int(myw)));
    }
    uint32_t myrow = 0;
    if (lane < n) {
      myrow = myid / (uint32_t)D;
      if (REUSE) myw = seg_weight_from_depth(depth, slot, myid, D, HW);
    } else myw = 0.f;
    for (int c0 = 0; c0 < C; c0 += 64 * VEC) {
      const int c = c0 + lane * VEC;
      const bool lane_on = c < C;
      vec acc = (vec)(0.f);
      const float* xc = x + (lane_on ? c : 0);
      if (n <= 4) seg_accumulate<VEC, 4>(xc, C, myrow, myw, n, acc);
      else seg_accumulate<VEC, 8>(xc, C, myrow, myw, n, acc);
      if (lane_on) *(vec*)(orow + c) = acc;
    }
    return;
  }
  // medium voxel (64 < n <= POOL_MEDIUM): ids -> LDS (unsorted) -> ranks -> LDS (sorted, after every lane has read what it needs)
  uint32_t mine[POOL_MEDIUM / 64];
  float minew[POOL_MEDIUM / 64];
  int rank[POOL_MEDIUM / 64];
#pragma unroll
  for (int q = 0; q < POOL_MEDIUM / 64; ++q) {
    mine[q] = q * 64 + lane < n ? ids[s + q * 64 + lane] : 0xFFFFFFFFu;
    minew[q] = (!REUSE && q * 64 + lane < n) ? wcsr[s + q * 64 + lane] : 0.f;
    lds[q * 64 + lane] = mine[q];
    rank[q] = 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int j = 0; j < n; ++j) {
    const uint32_t o = lds[j];              // broadcast read
#pragma unroll
    for (int q = 0; q < POOL_MEDIUM / 64; ++q) rank[q] += o < mine[q];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < POOL_MEDIUM / 64; ++q)
    if (q * 64 + lane < n) { lds[rank[q]] = mine[q]; ldw[rank[q]] = minew[q]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int c0 = 0; c0 < C; c0 += 64 * VEC) {
    const int c = c0 + lane * VEC;
    const bool lane_on = c < C;
    vec acc = (vec)(0.f);
    for (int base = 0; base < n; base += 64) {
      const int nb = min(64, n - base);
      const uint32_t myid = lds[base + min(lane, nb - 1)];
      const uint32_t myrow = myid / (uint32_t)D;
      const float myw = REUSE ? seg_weight_from_depth(depth, slot, myid, D, HW) : ldw[base + min(lane, nb - 1)];
      seg_accumulate<VEC, 8>(x + (lane_on ? c : 0), C, myrow, myw, nb, acc);
    }
    if (lane_on) *(vec*)(orow + c) = acc;
  }
}

// Four voxels per wave, 16 lanes each (C = 128: 8 channels per lane, C = 64: 4): at r50 a voxel holds 2.6 segments on average, and
// with one wave per voxel the launch was 80 k waves x four dependent memory round trips (44 us).  A group whose voxel has more
// than 16 segments sits this pass out (the caller runs the whole-wave code for it).  (s, n): the group's CSR range.
template <int V8, bool REUSE>
__device__ __forceinline__ void seg_rows16(const float* __restrict__ x, const float* __restrict__ depth, const float* __restrict__ wcsr,
                                           const int32_t* __restrict__ slot, const uint32_t* __restrict__ ids, int s, int n, bool on,
                                           int lane, int C, int D, int HW, float* __restrict__ orow) {
  const int l16 = lane & 15, gbase = lane & 48;
  const bool small = on && n <= 16;
  const bool mine = small && l16 < n;
  uint32_t myid = mine ? ids[s + l16] : 0xFFFFFFFFu;
  float myw = (!REUSE && mine) ? wcsr[s + l16] : 0.f;
  int nmax = 0;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ng = __builtin_amdgcn_readlane(small ? n : 0, g * 16);
    nmax = max(nmax, ng);
  }
  if (nmax > 1) {
    int rank = 0;
    for (int j = 0; j < nmax; ++j) rank += (uint32_t)__shfl((int)myid, j, 16) < myid;      // ids are distinct; idle lanes rank n
    const int dst = (gbase + min(rank, 15)) << 2;
    myid = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)myid);
    if (!REUSE) myw = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(myw)));
  }
  uint32_t myrow = 0;
  if (mine) {
    myrow = myid / (uint32_t)D;
    if (REUSE) myw = seg_weight_from_depth(depth, slot, myid, D, HW);
  } else myw = 0.f;
  float acc[V8];
#pragma unroll
  for (int e = 0; e < V8; ++e) acc[e] = 0.f;
  const float* xc = x + l16 * V8;
  constexpr int NB = 4;
  for (int j0 = 0; j0 < nmax; j0 += NB) {
    f32x4 r[NB][V8 / 4];
    float wj[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int jj = min(j0 + j, 15);
      const uint32_t row = (uint32_t)__shfl((int)myrow, jj, 16);
      wj[j] = j0 + j < nmax ? __shfl(myw, jj, 16) : 0.f;           // lanes past a group's n hold weight 0 and row 0
#pragma unroll
      for (int q = 0; q < V8 / 4; ++q) r[j][q] = *(const f32x4*)(xc + (size_t)row * C + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < V8 / 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = __fmaf_rn(wj[j], r[j][q][e], acc[4 * q + e]);
  }
  if (small) {
#pragma unroll
    for (int q = 0; q < V8 / 4; ++q) *(f32x4*)(orow + l16 * V8 + 4 * q) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  }
}

// n > POOL_MEDIUM: one workgroup per voxel, NO sort.  The long voxels sit next to the cameras (up to 2600 points / 1700 segments
// at r101) and ordering their ids -- a 66-stage bitonic network -- was most of the launch (r50: 40 us, r101: 130 us, whatever
// the shape of the sums).  Instead their sums are made EXACT, and therefore independent of order and of how the list is split:
//   * bound = max over the voxel's segments of weight * rowmax[row]   (rowmax[pixel] = max |context row|, from k_seg_hist; a
//     max is order-free), E = its binary exponent, quantum q = 2^(E - bits), bits = 50 - ceil(log2 n);
//   * every product w * x (rounded to fp32, the input of the short voxels' FMA chain) is snapped to a multiple of q
//     ((p + 1.5 * 2^52 q) - 1.5 * 2^52 q in fp64) and added in fp64: all partial sums are multiples of q below 2^52 q, so every
//     addition is exact -- 16 quarter-waves sum sixteenths of the UNSORTED list, their partial rows are added in LDS, one
//     rounding to fp32 at the end.  The snap costs at most q / 2 = 2^-40 of the bound per term at n = 2048 -- four orders of
//     magnitude below one fp32 rounding of the result.
// Same bits on every run for any arrival order of the atomic slots; no cap on n (the list is walked from global memory).
template <bool REUSE>
__device__ __forceinline__ void seg_long(const float* __restrict__ x, const float* __restrict__ depth, const float* __restrict__ wseg,
                                         const int32_t* __restrict__ slot, const uint32_t* __restrict__ seg, int n, int C, int D, int HW,
                                         const float* __restrict__ rowmax, float* __restrict__ orow, float* __restrict__ red /* [256] */,
                                         double* __restrict__ part /* [16][128] */, uint32_t* __restrict__ erow, float* __restrict__ ew) {
  const int tid = threadIdx.x;
  // pass 1: the bound; the first SEG_LDS_ENTRIES (row, weight) pairs stay in LDS for pass 2 (one memory round trip per batch of
  // rows instead of two)
  float bnd = 0.f;
  for (int i0 = 0; i0 < n; i0 += 256 * 4) {
    uint32_t idv[4];
    float wv[4], rm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256 + tid; idv[u] = i < n ? seg[i] : 0u; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256 + tid;
      wv[u] = i < n ? (REUSE ? seg_weight_from_depth(depth, slot, idv[u], D, HW) : wseg[i]) : 0.f;
      idv[u] = idv[u] / (uint32_t)D;
      rm[u] = i < n ? rowmax[idv[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256 + tid;
      bnd = fmaxf(bnd, fabsf(wv[u]) * rm[u]);
      if (i < n && i < SEG_LDS_ENTRIES) { erow[i] = idv[u]; ew[i] = wv[u]; }
    }
  }
  red[tid] = bnd;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  bnd = red[0];
  __syncthreads();
  int ex = 0;
  frexpf(bnd, &ex);                                              // bnd < 2^ex
  int lg = 0;
  while ((1 << lg) < n) ++lg;
  const double q = ldexp(1.0, ex - (50 - lg));
  const double magic = 6755399441055744.0 * q;                   // 1.5 * 2^52 * q
  const int g16 = tid >> 4, l16 = tid & 15;
  const int q16 = (n + 15) >> 4, lo = min(n, g16 * q16), hi = min(n, lo + q16);
  for (int c0 = 0; c0 < C; c0 += 128) {                          // 16 lanes x 8 channels per pass
    const int c = c0 + l16 * 8;
    const bool on = c < C;                                       // C % 4 == 0: a lane's second quad may be off
    const bool on2 = c + 4 < C;
    double acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0;
    const float* xc = x + (on ? c : 0);
    constexpr int LB = 4;             // (8: 118 registers for the whole kernel -- the short voxels' waves would drop from 6 to 4 per SIMD)
    for (int j0 = lo; j0 < hi; j0 += LB) {
      f32x4 r0[LB], r1[LB];
      float wj[LB];
#pragma unroll
      for (int j = 0; j < LB; ++j) {
        const int jj = min(j0 + j, hi - 1);
        uint32_t row;
        float wv;
        if (jj < SEG_LDS_ENTRIES) { row = erow[jj]; wv = ew[jj]; }
        else {
          const uint32_t id = seg[jj];
          row = id / (uint32_t)D;
          wv = REUSE ? seg_weight_from_depth(depth, slot, id, D, HW) : wseg[jj];
        }
        wj[j] = (on && j0 + j < hi) ? wv : 0.f;
        const float* rp = xc + (size_t)row * C;
        r0[j] = *(const f32x4*)rp;
        r1[j] = on2 ? *(const f32x4*)(rp + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < LB; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double p0 = (double)__fmul_rn(wj[j], r0[j][e]), p1 = (double)__fmul_rn(wj[j], r1[j][e]);
          acc[e] = __dadd_rn(acc[e], __dsub_rn(__dadd_rn(p0, magic), magic));
          acc[4 + e] = __dadd_rn(acc[4 + e], __dsub_rn(__dadd_rn(p1, magic), magic));
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[g16 * 128 + l16 * 8 + e] = acc[e];
    __syncthreads();
    if (tid < 128 && c0 + tid < C) {
      double t = part[tid];
#pragma unroll
      for (int w = 1; w < 16; ++w) t = __dadd_rn(t, part[w * 128 + tid]);      // exact: any order gives these bits
      orow[c0 + tid] = (float)t;
    }
    __syncthreads();
  }
}

// blocks [0, long_blocks): the long voxels (persistent over long_list); the rest: one wave per voxel, four voxels per workgroup.
// XCD-aware order of the short blocks: workgroup b runs on XCD b % 8, and XCD x walks the x-th eighth of the voxel list front to
// back -- the voxels in flight on one XCD are neighbours, so the context rows they gather are shared through that XCD's L2
// (with the plain order every XCD swept the whole grid and fetched every context row itself: 3.6x the algorithmic bytes at r101).
template <bool REUSE, int V8 /* 0: one voxel per wave; 4 | 8: four voxels per wave, V8 channels per lane (C = 16 V8) */>
__global__ __launch_bounds__(256) void k_pool_sum_seg(const float* __restrict__ x, const float* __restrict__ depth,
                                                       const float* __restrict__ wts, const int32_t* __restrict__ slot,
                                                       const uint32_t* __restrict__ ids, const int32_t* __restrict__ start,
                                                       const int32_t* __restrict__ long_list, const int32_t* __restrict__ nlong_p,
                                                       int long_blocks, int nvox, int C, int D, int HW, const float* __restrict__ rowmax,
                                                       float* __restrict__ out, int out_stride, int ablate, int gX, int gY, int gZ) {
  __shared__ __attribute__((aligned(16))) double part[16 * 128];      // long voxels: 16 partial rows of 128 channels (16 KB)
  __shared__ uint32_t erow[SEG_LDS_ENTRIES];                           // ... and their (row, weight) pairs
  __shared__ float ew[SEG_LDS_ENTRIES];
  uint32_t* sid = (uint32_t*)part;                                     // short blocks: the waves' medium-voxel id / weight slices
  float* sw = (float*)(sid + 4 * POOL_MEDIUM);
  float* red = (float*)(sid + 8 * POOL_MEDIUM);
  if ((int)blockIdx.x < long_blocks) {
    const int nlong = *nlong_p;
    for (int li = blockIdx.x; li < nlong; li += long_blocks) {
      const int v = long_list[li];
      const int s = start[v];
      seg_long<REUSE>(x, depth, wts + s, slot, ids + s, start[v + 1] - s, C, D, HW, rowmax, out + (size_t)v * out_stride, red, part, erow, ew);
    }
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = (int)blockIdx.x - long_blocks, nb = (int)gridDim.x - long_blocks;      // nb is a multiple of 8
  // plain order: a contiguous eighth of the voxel list per XCD (one workgroup id in eight) left the XCDs that own the centre of
  // the grid with most of the segments -- 134 against 111 us at r101 -- and did not lower the fetched bytes
  const int chunk = (ablate & 1) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
  if (ablate & 2) D = 0x7fffffff;                                 // timing experiment: every row gather reads row 0
  uint32_t* lds = sid + wave * POOL_MEDIUM;
  float* ldw = sw + wave * POOL_MEDIUM;
  // gX > 0 (COOCC_POOL_XCD=1, one batch): workgroup b runs on XCD b % 8 and takes its voxels from that XCD's own sector of the
  // grid -- quadrant (b & 1, (b >> 1) & 1) of the x-y plane, x parity (b >> 2) inside it -- so the context rows an XCD gathers
  // belong to the two or three cameras that look into its quadrant and stay in its L2; the x parity split keeps the XCDs of
  // one quadrant equally loaded.  t = position in the sector's own (x / 2, y, z) order.
  auto sector_voxel = [&](int t) -> int {
    const int hx = (gX + 1) >> 1, hy = (gY + 1) >> 1, k = b & 7;
    const int z = t % gZ; int q = t / gZ;
    const int j = q % hy, i = q / hy;
    const int x = (k & 1) * hx + 2 * i + (k >> 2), y = ((k >> 1) & 1) * hy + j;
    const bool ok = x < min(((k & 1) + 1) * hx, gX) && y < min((((k >> 1) & 1) + 1) * hy, gY);
    return ok ? (x * gY + y) * gZ + z : nvox;
  };
  if (V8) {
    const int vg = gX > 0 ? sector_voxel(((b >> 3) * 4 + wave) * 4 + (lane >> 4)) : (chunk * 4 + wave) * 4 + (lane >> 4);         // this 16-lane group's voxel
    const bool on = vg < nvox;
    const int sg = on ? start[vg] : 0, ng = on ? start[vg + 1] - sg : 0;
    seg_rows16<V8 ? V8 : 4, REUSE>(x, depth, wts, slot, ids, sg, ng, on, lane, C, D, HW, out + (size_t)(on ? vg : 0) * out_stride);
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {                                // the groups that sat out: whole-wave code, one after the other
      const int n = __builtin_amdgcn_readlane(ng, g * 16);
      if (n <= 16 || n > POOL_MEDIUM) continue;
      const int s = __builtin_amdgcn_readlane(sg, g * 16);
      const int v = __builtin_amdgcn_readlane(vg, g * 16);
      seg_row<2, REUSE>(x, depth, wts, slot, ids, s, n, lane, C, D, HW, out + (size_t)v * out_stride, lds, ldw);
    }
    return;
  }
  const int v = __builtin_amdgcn_readfirstlane(gX > 0 ? sector_voxel((b >> 3) * 4 + wave) : chunk * 4 + wave);
  if (v >= nvox) return;
  const int s = start[v], n = start[v + 1] - s;
  if (n > POOL_MEDIUM) return;
  if (C <= 128) seg_row<2, REUSE>(x, depth, wts, slot, ids, s, n, lane, C, D, HW, out + (size_t)v * out_stride, lds, ldw);
  else seg_row<4, REUSE>(x, depth, wts, slot, ids, s, n, lane, C, D, HW, out + (size_t)v * out_stride, lds, ldw);
}

// (Round 4 tried the sums as TWO launches -- short / medium voxels, 8 waves per SIMD instead of 3, and the long voxels on a side
// stream -- and measured no gain at r50 (0.118 ms either way: that round's gain was the batched histogram atomics) and a loss at
// r101 (0.325 -> 0.410 ms: the long voxels' serial chains, 233 us, are the launch and they run slower next to a denser short
// kernel); a rank sort instead of the bitonic network and 128 rows in flight made the long path slower still (233 -> 303 us).
// profiles/r4_pool_experiments.txt.)
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// workspace: keys[npts] | ids[npts] | slot[npts] | count[nvox+1] | nlong [64] (one memset clears these two when the caller does
// not vouch for them) | start[nvox+1] | lstart[nvox+1] | long_list[nvox] | tops[nvox / 1024 + 2] | wts[npts] | wcsr[npts] | rowmax[npts / D <= npts] (segment form)
extern "C" size_t coocc_voxel_pool_ws(int npts, int nvox) {
  if (npts <= 0 || nvox <= 0) return 0;
  return 6 * align256(sizeof(uint32_t) * (size_t)npts) + 4 * align256(sizeof(int32_t) * ((size_t)nvox + 1)) + 256 +
         align256(sizeof(int32_t) * ((size_t)nvox / 1024 + 2)) + 8192 + 256;
}

struct PoolWs { uint32_t *keys, *ids; int32_t *slot, *count, *nlong, *start, *lstart, *long_list, *tops; float *wts, *wcsr, *rowmax; size_t zero_bytes; };

static int carve(void* ws, size_t ws_bytes, int npts, int nvox, PoolWs* p) {
  size_t need = coocc_voxel_pool_ws(npts, nvox);
  if (!ws || ws_bytes < need) return coocc_set_error(COOCC_ENOMEM, "voxel_pool: workspace %zu < %zu bytes", ws_bytes, need);
  char* c = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  size_t a = align256(sizeof(uint32_t) * (size_t)npts), v = align256(sizeof(int32_t) * ((size_t)nvox + 1));
  p->keys = (uint32_t*)c; c += a; p->ids = (uint32_t*)c; c += a; p->slot = (int32_t*)c; c += a;
  p->count = (int32_t*)c; c += v; p->nlong = (int32_t*)c; c += 256;
  p->zero_bytes = v + 256;
  p->start = (int32_t*)c; c += v; p->lstart = (int32_t*)c; c += v; p->long_list = (int32_t*)c; c += v; p->tops = (int32_t*)c;
  c += align256(sizeof(int32_t) * ((size_t)nvox / 1024 + 2)) + 8192;
  p->wts = (float*)c; c += a;
  p->wcsr = (float*)c; c += a;
  p->rowmax = (float*)c;
  return COOCC_OK;
}

// per-voxel sums over the CSR (start, ids, long-voxel list) in the workspace: launch 3 of 3, and the only launch of the
// geometry-cached form.  LIFT: x = context rows [N*H*W, C], depth [npts], product formed inside the sum (fused lift (x) splat).
template <bool LIFT>
static int pool_sums(const float* x, const float* depth, int C, int D, int HW, int nvox, float* out, int out_stride,
                     const PoolWs& p, hipStream_t s) {
  // workgroups that walk the list of long voxels (> POOL_MEDIUM points; 24 at r50, 554 at r101, up to 2614 points each): enough of
  // them for one voxel each at r101; idle ones exit after one load
  static const int long_blocks = getenv("COOCC_POOL_LONG_BLOCKS") ? atoi(getenv("COOCC_POOL_LONG_BLOCKS")) : 1024;
  hipLaunchKernelGGL(k_pool_sum_csr<LIFT>, dim3(long_blocks + cdiv(nvox, 4)), dim3(256), 0, s, x, depth, p.ids, p.start,
                     p.long_list, p.nlong, long_blocks, nvox, C, D, HW, out, out_stride);
  COOCC_LAUNCH_CHECK("voxel_pool");
  return COOCC_OK;
}

// points -> keys + histogram (launch 1) -> chunk scan (2) -> CSR fill + global starts (3) -> per-voxel sums in ascending point id (4).
// ws_clean != 0: the caller vouches that this workspace was last written by a pooling call of the SAME (npts, nvox) that returned
// COOCC_OK (every such call leaves the count array zeroed); otherwise one memset clears it first.
template <bool LIFT, int MODE>
static int pool_build(const KeySrc& ks, const float* x, const float* depth, int npts, int C, int D, int HW, int X, int Y, int Z, int nvox,
                      float* out, int out_stride, const PoolWs& p, int ws_clean, hipStream_t s) {
  COOCC_CHECK_ARG((nvox + 1023) / 1024 <= POOL_MAX_CHUNKS, "voxel_pool: grids above 8.4 M voxels are not supported");
  if (!ws_clean) COOCC_HIP(hipMemsetAsync(p.count, 0, p.zero_bytes, s));
  hipLaunchKernelGGL(k_keys_hist<MODE>, dim3(cdiv(npts, 256)), dim3(256), 0, s, ks, npts, X, Y, Z, nvox, p.keys, p.count, p.slot);
  const int nblk = (nvox + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(1024), 0, s, p.count, nvox, p.lstart, p.tops, p.nlong);
  // a quarter of the points per grid pass (four per thread and iteration); every workgroup re-scans the nblk chunk totals
  const int fill_blocks = max(cdiv(npts, 1024), 1);
  hipLaunchKernelGGL(k_csr_fill, dim3(fill_blocks), dim3(256), sizeof(int) * ((size_t)nblk + 1), s, p.keys, npts, nvox, nblk, p.lstart, p.tops,
                     p.slot, p.count, p.start, p.long_list, p.nlong, p.ids, 0x7FFFFFFF, nullptr, nullptr);
  return pool_sums<LIFT>(x, depth, C, D, HW, nvox, out, out_stride, p, s);
}

// ---- segment form (fused lift (x) splat only; COOCC_POOL_SEG=0 restores the ascending-point-id form above)
static bool pool_seg_on() {               // read per call: tests compare the two forms inside one process
  const char* e = getenv("COOCC_POOL_SEG");
  return !(e && atoi(e) == 0);
}

template <bool REUSE>
static int pool_sums_seg(const float* x, const float* depth, int C, int D, int HW, int nvox, long long npts, float* out, int out_stride,
                         const PoolWs& p, hipStream_t s, int X = 0, int Y = 0, int Z = 0) {
  static const bool xcd = getenv("COOCC_POOL_XCD") && atoi(getenv("COOCC_POOL_XCD")) != 0;
  const bool sect = xcd && X > 0 && (long long)X * Y * Z == nvox;                   // one batch: the sector order is per grid
  const int gX = sect ? X : 0, gY = sect ? Y : 0, gZ = sect ? Z : 0;
  const int sector = sect ? (((X + 1) / 2 + 1) / 2) * ((Y + 1) / 2) * Z : 0;         // voxels per XCD sector
  static const int long_blocks = getenv("COOCC_POOL_LONG_BLOCKS") ? atoi(getenv("COOCC_POOL_LONG_BLOCKS")) : 1024;
  static const bool g16 = !(getenv("COOCC_POOL_G16") && atoi(getenv("COOCC_POOL_G16")) == 0);
  static const int ablate = getenv("COOCC_POOL_ABLATE") ? atoi(getenv("COOCC_POOL_ABLATE")) : 0;      // timing experiments (results wrong)
  // four voxels per wave pay when most voxels hold <= 16 segments: r50 has 5.9 points per voxel (2.6 segments per non-empty
  // voxel), r101 47 (15 segments; there the groups that sit out cost more than the packing saves: 134 against 117 us)
  if (g16 && (C == 128 || C == 64) && npts < 16ll * nvox) {
    const int short_blocks = sect ? 8 * cdiv(sector, 16) : 8 * cdiv(cdiv(nvox, 16), 8);
    if (C == 128)
      hipLaunchKernelGGL((k_pool_sum_seg<REUSE, 8>), dim3(long_blocks + short_blocks), dim3(256), 0, s, x, depth, p.wcsr, p.slot, p.ids,
                         p.start, p.long_list, p.nlong, long_blocks, nvox, C, D, HW, p.rowmax, out, out_stride, ablate, gX, gY, gZ);
    else
      hipLaunchKernelGGL((k_pool_sum_seg<REUSE, 4>), dim3(long_blocks + short_blocks), dim3(256), 0, s, x, depth, p.wcsr, p.slot, p.ids,
                         p.start, p.long_list, p.nlong, long_blocks, nvox, C, D, HW, p.rowmax, out, out_stride, ablate, gX, gY, gZ);
  } else {
    const int short_blocks = sect ? 8 * cdiv(sector, 4) : 8 * cdiv(cdiv(nvox, 4), 8);
    hipLaunchKernelGGL((k_pool_sum_seg<REUSE, 0>), dim3(long_blocks + short_blocks), dim3(256), 0, s, x, depth, p.wcsr, p.slot, p.ids,
                       p.start, p.long_list, p.nlong, long_blocks, nvox, C, D, HW, p.rowmax, out, out_stride, ablate, gX, gY, gZ);
  }
  COOCC_LAUNCH_CHECK("voxel_pool (segments)");
  return COOCC_OK;
}

template <int MODE>
static int pool_build_seg(const KeySrc& ks, const float* x, const float* depth, int npts, int C, int D, int HW, int X, int Y, int Z,
                          int nvox, float* out, int out_stride, const PoolWs& p, int ws_clean, hipStream_t s) {
  COOCC_CHECK_ARG((nvox + 1023) / 1024 <= POOL_MAX_CHUNKS, "voxel_pool: grids above 8.4 M voxels are not supported");
  if (!ws_clean) COOCC_HIP(hipMemsetAsync(p.count, 0, p.zero_bytes, s));
  const int strips_per_cam = cdiv(HW, SEG_STRIP), nstrips = strips_per_cam * (npts / (D * HW));
  const size_t plane = sizeof(float) * (size_t)D * (SEG_STRIP + 1);
  const int stage_geom = MODE == 0 && 4 * plane + 12 * SEG_HASH <= 60 * 1024;
  hipLaunchKernelGGL(k_seg_hist<MODE>, dim3(8 * cdiv(nstrips, 8)), dim3(256), (stage_geom ? 4 * plane : plane) + 12 * SEG_HASH, s, ks, depth, npts, X, Y, Z,
                     nvox, strips_per_cam, nstrips, stage_geom, p.keys, p.count, p.slot, p.wts, x, C, p.rowmax);
  const int nblk = (nvox + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_local, dim3(nblk), dim3(1024), 0, s, p.count, nvox, p.lstart, p.tops, p.nlong);
  const int fill_blocks = max(cdiv(npts, 1024), 1);
  hipLaunchKernelGGL(k_csr_fill, dim3(fill_blocks), dim3(256), sizeof(int) * ((size_t)nblk + 1), s, p.keys, npts, nvox, nblk, p.lstart, p.tops,
                     p.slot, p.count, p.start, p.long_list, p.nlong, p.ids, SEG_SLOT_MASK, p.wts, p.wcsr);
  return pool_sums_seg<false>(x, depth, C, D, HW, nvox, npts, out, out_stride, p, s, X, Y, Z);
}

extern "C" int coocc_voxel_pool(const float* x, const float* geom, int npts, int pts_per_batch, int C,
                                const float* lo_dx_host, int B, int X, int Y, int Z, float* out, int out_stride,
                                void* ws, size_t ws_bytes, int ws_clean, void* stream) {
  COOCC_CHECK_ARG(x && geom && out && lo_dx_host && npts > 0 && pts_per_batch > 0 && C > 0 && C % 4 == 0, "voxel_pool: bad args");
  COOCC_CHECK_ARG(((uintptr_t)x & 15) == 0 && out_stride % 4 == 0 && out_stride >= C, "voxel_pool: alignment");
  const long long nvox_ll = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(nvox_ll > 0 && nvox_ll < (1ll << 31), "voxel_pool: grid too large");
  const int nvox = (int)nvox_ll;
  PoolWs p;
  int rc = carve(ws, ws_bytes, npts, nvox, &p);
  if (rc) return rc;
  const float* l = lo_dx_host;
  KeySrc ks = {};
  ks.geom = geom; ks.pts_per_batch = pts_per_batch; ks.B = B;
  ks.lox = l[0]; ks.loy = l[1]; ks.loz = l[2]; ks.dx = l[3]; ks.dy = l[4]; ks.dz = l[5];
  return pool_build<false, 0>(ks, x, nullptr, npts, C, 0, 0, X, Y, Z, nvox, out, out_stride, p, ws_clean, as_stream(stream));
}

// ------------------------------------------------------------------ fused lift (x) splat
// ViewTransformerLSSVoxel.py:135-143 computes volume = depth_prob[n,d,h,w] * img_feat[n,c,h,w] (242 MB at
// r50, 1.93 GB at r101) and pools it.  Here the product is formed on the fly inside the per-voxel sum:
// point id = ((n*D + d)*H + h)*W + w indexes depth directly and selects the context row (n,h,w).
// Products are rounded to fp32 before the add (no FMA), in ascending point id: bit-equal to pooling the
// materialised volume with the stable order.
static int lift_splat_impl(const float* depth, const float* feat_nhwc, const float* geom, const float* mats,
                           const float* xs, const float* ys, const float* ds, int N, int D, int H, int W, int C,
                           int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                           int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream) {
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
  KeySrc ks = {};
  ks.geom = geom; ks.mats = mats; ks.xs = xs; ks.ys = ys; ks.ds = ds; ks.D = D; ks.fH = H; ks.fW = W;
  ks.pts_per_batch = pts_per_batch; ks.B = B;
  ks.lox = l[0]; ks.loy = l[1]; ks.loz = l[2]; ks.dx = l[3]; ks.dy = l[4]; ks.dz = l[5];
  if (pool_seg_on() && (long long)N * H * W < (1ll << SEG_SLOT_BITS) && D <= SEG_DMAX) {
    if (geom) return pool_build_seg<0>(ks, feat_nhwc, depth, npts, C, D, H * W, X, Y, Z, nvox, out, out_stride, p, ws_clean, s);
    return pool_build_seg<1>(ks, feat_nhwc, depth, npts, C, D, H * W, X, Y, Z, nvox, out, out_stride, p, ws_clean, s);
  }
  if (geom) return pool_build<true, 0>(ks, feat_nhwc, depth, npts, C, D, H * W, X, Y, Z, nvox, out, out_stride, p, ws_clean, s);
  return pool_build<true, 1>(ks, feat_nhwc, depth, npts, C, D, H * W, X, Y, Z, nvox, out, out_stride, p, ws_clean, s);
}

extern "C" int coocc_lift_splat(const float* depth, const float* feat_nhwc, const float* geom, int N, int D, int H, int W,
                                int C, int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                                int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream) {
  COOCC_CHECK_ARG(geom, "lift_splat: null geom");
  return lift_splat_impl(depth, feat_nhwc, geom, nullptr, nullptr, nullptr, nullptr, N, D, H, W, C, pts_per_batch,
                         lo_dx_host, B, X, Y, Z, out, out_stride, ws, ws_bytes, ws_clean, stream);
}

extern "C" int coocc_lift_splat_cams(const float* depth, const float* feat_nhwc, const float* mats, const float* xs,
                                     const float* ys, const float* ds, int N, int D, int H, int W, int C,
                                     int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                                     int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream) {
  COOCC_CHECK_ARG(mats && xs && ys && ds, "lift_splat_cams: null camera data");
  return lift_splat_impl(depth, feat_nhwc, nullptr, mats, xs, ys, ds, N, D, H, W, C, pts_per_batch, lo_dx_host, B, X, Y,
                         Z, out, out_stride, ws, ws_bytes, ws_clean, stream);
}

// The per-voxel sums alone over the CSR a previous coocc_lift_splat[_cams] call left in `ws` (same N, D, H, W, grid and
// GEOMETRY: a calibrated rig does not move between frames): one launch instead of eight, bit-equal to the full call.
extern "C" int coocc_lift_splat_reuse(const float* depth, const float* feat_nhwc, int N, int D, int H, int W, int C, int B, int X,
                                      int Y, int Z, float* out, int out_stride, void* ws, size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(depth && feat_nhwc && out && N > 0 && D > 0 && H > 0 && W > 0, "lift_splat_reuse: bad args");
  COOCC_CHECK_ARG(C > 0 && C % 4 == 0 && ((uintptr_t)feat_nhwc & 15) == 0 && out_stride % 4 == 0 && out_stride >= C &&
                      ((uintptr_t)out & 15) == 0, "lift_splat_reuse: C % 4 == 0 and 16-byte aligned rows");
  const long long npts_ll = (long long)N * D * H * W, nvox_ll = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(npts_ll < (1ll << 31) && nvox_ll > 0 && nvox_ll < (1ll << 31), "lift_splat_reuse: sizes");
  PoolWs p;
  int rc = carve(ws, ws_bytes, (int)npts_ll, (int)nvox_ll, &p);
  if (rc) return rc;
  if (pool_seg_on() && (long long)N * H * W < (1ll << SEG_SLOT_BITS) && D <= SEG_DMAX) {    // the rule of lift_splat_impl
    hipLaunchKernelGGL(k_row_absmax, dim3(cdiv(N * H * W, 8)), dim3(256), 0, as_stream(stream), feat_nhwc, C, N * H * W, p.rowmax);
    return pool_sums_seg<true>(feat_nhwc, depth, C, D, H * W, (int)nvox_ll, npts_ll, out, out_stride, p, as_stream(stream));
  }
  return pool_sums<true>(feat_nhwc, depth, C, D, H * W, (int)nvox_ll, out, out_stride, p, as_stream(stream));
}

extern "C" int coocc_bev_pool_coords(const float* x, const int64_t* coords, int n, int C, int B, int X, int Y, int Z,
                                     float* out, int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream) {
  COOCC_CHECK_ARG(x && coords && out && n > 0 && C > 0 && C % 4 == 0, "bev_pool_coords: bad args");
  COOCC_CHECK_ARG(((uintptr_t)x & 15) == 0 && out_stride % 4 == 0 && out_stride >= C, "bev_pool_coords: alignment");
  const long long nvox_ll = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(nvox_ll > 0 && nvox_ll < (1ll << 31), "bev_pool_coords: grid too large");
  const int nvox = (int)nvox_ll;
  PoolWs p;
  int rc = carve(ws, ws_bytes, n, nvox, &p);
  if (rc) return rc;
  KeySrc ks = {};
  ks.coords = coords; ks.B = B; ks.pts_per_batch = 1;
  return pool_build<false, 2>(ks, x, nullptr, n, C, 0, 0, X, Y, Z, nvox, out, out_stride, p, ws_clean, as_stream(stream));
}
