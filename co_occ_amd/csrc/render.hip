// Volume-rendering regulariser (R2, R3, L1).
//
// R2 = the inline render block of COOCC_Ray (coocc_ray.py:575-616): nearest-voxel lookup
// along the D frustum samples of every feature-map pixel, alpha compositing, expected
// depth bin.  The sigma/rgb heads are pointwise MLPs, so they are evaluated once per voxel
// into a [V,4] table (MFMA GEMMs, conv3d.hip) and the ray kernel gathers 16 B per sample
// instead of C*4 B.  Reference quirks kept: out-of-bounds samples read voxel (0,0,0) for
// sigma (:586,:597) and get rgb = sigmoid(0) (:595-596); step length is the distance
// between consecutive TRUNCATED voxel indices (:600-601), last step 1e10 (:603);
// z_vals = linspace(0, D, D) (:614); bounds hard-coded (:577).
#include <stdlib.h>

#include "common.h"
#include "geometry.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RT_MAX 32  // rays (consecutive w) per block

// Wave-wide float sum without the LDS crossbar: DPP inside rows of 16 lanes, v_readlane across rows.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// DPP move whose lanes without a valid source keep the multiplicative identity
template <int CTRL>
__device__ __forceinline__ float dpp_f32_id(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3F800000, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);   // row_half_mirror
  v += dpp_f32<0x140>(v);   // row_mirror
  float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

// sums of the two half-waves (lanes 0-31 -> lo, lanes 32-63 -> hi), wave-uniform results
__device__ __forceinline__ void half_sums(float v, float& lo, float& hi) {
  v += dpp_f32<0xB1>(v);
  v += dpp_f32<0x4E>(v);
  v += dpp_f32<0x141>(v);
  v += dpp_f32<0x140>(v);
  float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  lo = a + b;
  hi = c + d;
}

struct __attribute__((packed, aligned(4))) Geo3 { float x, y, z; };

// Block = (camera n, row h, tile of rt <= 32 consecutive rays).
// Phase 1, lane per sample with lanes along w: one 12-byte load per sample (a depth bin of the tile
// is one contiguous segment of geom), quantise, store the packed voxel index (bit 31 = outside the
// bounds) transposed to LDS [ray][D+1] -- 4 bytes per sample, so ~10 workgroups fit a CU and their
// load latencies overlap.
// Phase 2, one wave per ray with lanes along depth (CH consecutive samples per lane): 16-byte gather
// of the voxel's (sigma, rgb) from the L2-resident table, alpha, exclusive transmittance by a
// wave-level multiplicative scan, weighted sums by DPP wave reductions.
// ACT = 1: the table already holds sigmoid(rgb) (coocc_render_activate_table: 3 exp + 3 rcp per VOXEL instead of
// per ray sample); ACT = 0: raw logits (the form the backward kernel differentiates).
// (Two persistent variants were measured slower at r101, 1344 tiles: prefetching the next tile's geometry into
// registers under phase 2 -- 38-48 us vs 27, the 48 extra VGPRs cost more occupancy than the overlap returned --
// and swapping the two phases between wave pairs over a double-buffered LDS tile -- 38 us at 2 tiles per
// workgroup, 59 at 4: with so few tiles, one workgroup per tile, all resident at once, is the better use of the chip.)
// GEO: the sample positions are not read from a [N,D,H,W,3] tensor but evaluated in place from the 39 per-camera constants and
// the frustum axes (geom = mats, xs / ys / ds passed separately) with get_geometry's own chain (geometry.h): the kernel then
// reads 16 bytes of table per sample and nothing else (12 bytes per sample less: 45 of the 47 MB it moves at r101).
template <int ACT, int CHT, bool GEO = false>
__global__ __launch_bounds__(256) void k_render_nearest(const float* __restrict__ table, int Y, int Z,
                                                         const float* __restrict__ geom,
                                                         const float* __restrict__ zvals, int D, int H, int W, int rt,
                                                         float lox, float loy, float loz, float dx, float dy, float dz,
                                                         float nx, float ny, float nz, float* __restrict__ maps,
                                                         const float* __restrict__ xs = nullptr, const float* __restrict__ ys = nullptr,
                                                         const float* __restrict__ ds = nullptr) {
  extern __shared__ int s_pos[];    // [rt][D+1]
  const int DS = D + 1;
  const int wt = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  const int w0 = wt * rt;
  const int nray = min(rt, W - w0);
  const int tid = threadIdx.x;

  // lane -> (ray r = tid & 31, depth phase tid >> 5): no integer division in the loop
  // GEO: the ray's frustum coordinates are loop invariants (this lane always works on ray tid & 31), the camera's 39 constants
  // are read through a uniform base; a sample costs get_geometry's ~45 flops and no index arithmetic.  x / 1.0f == x exactly, so
  // the three IEEE divisions (a dozen instructions each) are skipped for unit voxels (every shipped config renders on the
  // hard-coded 1 m grid of coocc_ray.py:577); the branch is uniform.
  const float* mcam = GEO ? geom + (size_t)n * COOCC_CAM_FLOATS : nullptr;
  const float xw = GEO && (tid & 31) < nray ? xs[w0 + (tid & 31)] : 0.f, yh = GEO ? ys[h] : 0.f;
  const bool unit = dx == 1.f && dy == 1.f && dz == 1.f;
#pragma unroll 4
  for (int d = tid >> 5, r = tid & 31; d < D; d += 8) {
    if (r >= nray) continue;
    Geo3 g;
    if (GEO) geometry_sample(mcam, xw, yh, ds[d], g.x, g.y, g.z);
    else g = *(const Geo3*)(geom + ((((size_t)n * D + d) * H + h) * W + w0 + r) * 3);
    float gx, gy, gz;
    if (unit) { gx = g.x - lox; gy = g.y - loy; gz = g.z - loz; }
    else { gx = __fdiv_rn(g.x - lox, dx); gy = __fdiv_rn(g.y - loy, dy); gz = __fdiv_rn(g.z - loz, dz); }
    bool in = gx >= 0.f && gx < nx && gy >= 0.f && gy < ny && gz >= 0.f && gz < nz;
    int ix = in ? (int)gx : 0, iy = in ? (int)gy : 0, iz = in ? (int)gz : 0;
    s_pos[r * DS + d] = ix | (iy << 10) | (iz << 20) | (in ? 0 : (1 << 31));
  }
  __syncthreads();

  // Phase 2: each HALF-wave (32 lanes) composites one ray, lanes along depth with CH consecutive
  // samples per lane -- two rays per wave instruction stream, a 5-step scan and row-local DPP
  // reductions instead of 64-lane ones (the phase is instruction-bound, not memory-bound:
  // profiles/r1_render_ablation.txt).
  const int lane = tid & 63, wave = tid >> 6, hl = lane & 31, half = lane >> 5;
  const int CH = (D + 31) / 32;                  // <= CHT
  const int d0 = hl * CH;
  float zv[CHT];                                 // this lane's z_vals, shared by every ray
#pragma unroll
  for (int j = 0; j < CHT; ++j) zv[j] = (j < CH && d0 + j < D) ? zvals[d0 + j] : 0.f;
  for (int r0 = wave * 2; r0 < nray; r0 += 8) {
    const int r = r0 + half;
    const bool live = r < nray;
    int pk[CHT + 1];                              // this lane's CH packed positions + the one after (for the step length)
#pragma unroll
    for (int j = 0; j <= CHT; ++j) pk[j] = (j <= CH && d0 + j < D && live) ? s_pos[r * DS + d0 + j] : 0;
    float al[CHT], cr[CHT], cg[CHT], cb[CHT], prod = 1.f;
#pragma unroll
    for (int j = 0; j < CHT; ++j) {
      al[j] = 0.f; cr[j] = cg[j] = cb[j] = 0.f;
      const int d = d0 + j;
      if (j < CH && d < D && live) {
        const int p0 = pk[j];
        const int x0 = p0 & 1023, y0 = (p0 >> 10) & 1023, z0 = (p0 >> 20) & 1023;
        const f32x4 t = *(const f32x4*)(table + (((size_t)x0 * Y + y0) * Z + z0) * 4);
        float dist = 1e10f;
        if (d + 1 < D) {
          const int p1 = pk[j + 1];
          float ex = (float)((p1 & 1023) - x0), ey = (float)(((p1 >> 10) & 1023) - y0), ez = (float)(((p1 >> 20) & 1023) - z0);
          dist = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez);   // v_sqrt_f32 (1 ulp) on small integers
        }
        const bool in = p0 >= 0;
        // hardware exp / rcp (<= 1 ulp-class error, far inside the 1e-4 parity bound)
        al[j] = 1.f - __expf(-fmaxf(fmaxf(t[0], 0.f) * dist, 0.f));
        if (ACT) {
          cr[j] = in ? t[1] : 0.5f; cg[j] = in ? t[2] : 0.5f; cb[j] = in ? t[3] : 0.5f;
        } else {
          cr[j] = in ? __frcp_rn(1.f + __expf(-t[1])) : 0.5f;
          cg[j] = in ? __frcp_rn(1.f + __expf(-t[2])) : 0.5f;
          cb[j] = in ? __frcp_rn(1.f + __expf(-t[3])) : 0.5f;
        }
        prod *= 1.f - al[j] + 1e-10f;
      }
    }
    // exclusive multiplicative scan inside each half-wave, all in DPP (no LDS crossbar): row_shr 1/2/4/8 inside the
    // rows of 16 lanes (lanes without a source keep the identity 1.0), row_bcast:15 carries row 0 -> 1 and 2 -> 3,
    // wave_shr:1 turns the inclusive scan into the exclusive one (the first lane of each half restarts at 1)
    float inc = prod;
    inc *= dpp_f32_id<0x111>(inc);
    inc *= dpp_f32_id<0x112>(inc);
    inc *= dpp_f32_id<0x114>(inc);
    inc *= dpp_f32_id<0x118>(inc);
    inc *= __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3F800000, __builtin_bit_cast(int, inc), 0x142, 0xA, 0xF, false));
    float T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3F800000, __builtin_bit_cast(int, inc), 0x138, 0xF, 0xF, false));
    if (hl == 0) T = 1.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};             // r, g, b, depth
#pragma unroll
    for (int j = 0; j < CHT; ++j) {
      const int d = d0 + j;
      if (j < CH && d < D) {
        const float wgt = al[j] * T;
        acc = acc + wgt * f32x4{cr[j], cg[j], cb[j], zv[j]};
        T *= 1.f - al[j] + 1e-10f;
      }
    }
    // half-wave sums without leaving the vector lanes: row totals (4 DPP steps), then row 0 -> row 1 and row 2 -> row 3
    // (row_bcast:15); lanes 16 and 48 hold the totals of their half and store them
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = acc[k];
      v += dpp_f32<0xB1>(v);
      v += dpp_f32<0x4E>(v);
      v += dpp_f32<0x141>(v);
      v += dpp_f32<0x140>(v);
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
      acc[k] = v;
    }
    if (hl == 16 && live) *(f32x4*)(maps + (((size_t)n * H + h) * W + w0 + r) * 4) = acc;
  }
}

// The in-kernel-geometry form WITHOUT the LDS transpose (round 4).  With the sample positions computed instead of streamed there
// is nothing to coalesce in a first phase: the lane that composites samples d0 .. d0+CH-1 of a ray evaluates get_geometry's chain
// for exactly those samples itself (same expressions, same bits as phase 1 of k_render_nearest<.., true>), and takes the position
// that follows its last one (the step length of its last sample) from its right neighbour's first with one shuffle.  No LDS, no
// barrier; a workgroup is 4 waves x 2 rays, one pass, (W / 8) x H x N workgroups.
template <int ACT, int CHT>
__global__ __launch_bounds__(256) void k_render_rays_geo(const float* __restrict__ table, int Y, int Z, const float* __restrict__ mats,
                                                          const float* __restrict__ xs, const float* __restrict__ ys,
                                                          const float* __restrict__ ds, const float* __restrict__ zvals, int D, int H, int W,
                                                          float lox, float loy, float loz, float dx, float dy, float dz, float nx, float ny,
                                                          float nz, float* __restrict__ maps) {
  const int h = blockIdx.y, n = blockIdx.z;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, hl = lane & 31, half = lane >> 5;
  const int r = blockIdx.x * 8 + wave * 2 + half;          // this half-wave's ray = pixel column w
  const bool live = r < W;
  const int CH = (D + 31) / 32;                  // <= CHT
  const int d0 = hl * CH;
  const float* mcam = mats + (size_t)n * COOCC_CAM_FLOATS;
  const float xw = live ? xs[r] : 0.f, yh = ys[h];
  const bool unit = dx == 1.f && dy == 1.f && dz == 1.f;
  int pk[CHT + 1];
  float zv[CHT];
#pragma unroll
  for (int j = 0; j < CHT; ++j) {
    const int d = d0 + j;
    pk[j] = 0;
    zv[j] = 0.f;
    if (j < CH && d < D && live) {
      zv[j] = zvals[d];
      float px, py, pz;
      geometry_sample(mcam, xw, yh, ds[d], px, py, pz);
      float gx, gy, gz;
      if (unit) { gx = px - lox; gy = py - loy; gz = pz - loz; }
      else { gx = __fdiv_rn(px - lox, dx); gy = __fdiv_rn(py - loy, dy); gz = __fdiv_rn(pz - loz, dz); }
      const bool in = gx >= 0.f && gx < nx && gy >= 0.f && gy < ny && gz >= 0.f && gz < nz;
      const int ix = in ? (int)gx : 0, iy = in ? (int)gy : 0, iz = in ? (int)gz : 0;
      pk[j] = ix | (iy << 10) | (iz << 20) | (in ? 0 : (1 << 31));
    }
  }
  {
    const int nxt = __shfl_down(pk[0], 1);       // the right neighbour's first sample = the one after this lane's last
#pragma unroll
    for (int j = 1; j <= CHT; ++j)
      if (j == CH) pk[j] = nxt;
  }
  float al[CHT], cr[CHT], cg[CHT], cb[CHT], prod = 1.f;
#pragma unroll
  for (int j = 0; j < CHT; ++j) {
    al[j] = 0.f; cr[j] = cg[j] = cb[j] = 0.f;
    const int d = d0 + j;
    if (j < CH && d < D && live) {
      const int p0 = pk[j];
      const int x0 = p0 & 1023, y0 = (p0 >> 10) & 1023, z0 = (p0 >> 20) & 1023;
      const f32x4 t = *(const f32x4*)(table + (((size_t)x0 * Y + y0) * Z + z0) * 4);
      float dist = 1e10f;
      if (d + 1 < D) {
        const int p1 = pk[j + 1];
        float ex = (float)((p1 & 1023) - x0), ey = (float)(((p1 >> 10) & 1023) - y0), ez = (float)(((p1 >> 20) & 1023) - z0);
        dist = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez);
      }
      const bool in = p0 >= 0;
      al[j] = 1.f - __expf(-fmaxf(fmaxf(t[0], 0.f) * dist, 0.f));
      if (ACT) {
        cr[j] = in ? t[1] : 0.5f; cg[j] = in ? t[2] : 0.5f; cb[j] = in ? t[3] : 0.5f;
      } else {
        cr[j] = in ? __frcp_rn(1.f + __expf(-t[1])) : 0.5f;
        cg[j] = in ? __frcp_rn(1.f + __expf(-t[2])) : 0.5f;
        cb[j] = in ? __frcp_rn(1.f + __expf(-t[3])) : 0.5f;
      }
      prod *= 1.f - al[j] + 1e-10f;
    }
  }
  float inc = prod;
  inc *= dpp_f32_id<0x111>(inc);
  inc *= dpp_f32_id<0x112>(inc);
  inc *= dpp_f32_id<0x114>(inc);
  inc *= dpp_f32_id<0x118>(inc);
  inc *= __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3F800000, __builtin_bit_cast(int, inc), 0x142, 0xA, 0xF, false));
  float T = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x3F800000, __builtin_bit_cast(int, inc), 0x138, 0xF, 0xF, false));
  if (hl == 0) T = 1.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < CHT; ++j) {
    const int d = d0 + j;
    if (j < CH && d < D) {
      const float wgt = al[j] * T;
      acc = acc + wgt * f32x4{cr[j], cg[j], cb[j], zv[j]};
      T *= 1.f - al[j] + 1e-10f;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = acc[k];
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    acc[k] = v;
  }
  if (hl == 16 && live) *(f32x4*)(maps + (((size_t)n * H + h) * W + r) * 4) = acc;
}

// sigmoid of the rgb logits once per voxel (in place on columns 1..3 of the [V,4] table)
__global__ __launch_bounds__(256) void k_render_activate_table(float* __restrict__ table, int V) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  f32x4 t = *(f32x4*)(table + (size_t)v * 4);
  t[1] = __frcp_rn(1.f + __expf(-t[1]));
  t[2] = __frcp_rn(1.f + __expf(-t[2]));
  t[3] = __frcp_rn(1.f + __expf(-t[3]));
  *(f32x4*)(table + (size_t)v * 4) = t;
}

extern "C" int coocc_render_activate_table(float* table, int V, void* stream) {
  COOCC_CHECK_ARG(table && V > 0, "render_activate_table: bad args");
  hipLaunchKernelGGL(k_render_activate_table, dim3(cdiv(V, 256)), dim3(256), 0, as_stream(stream), table, V);
  COOCC_LAUNCH_CHECK("k_render_activate_table");
  return COOCC_OK;
}

static int render_nearest_impl(const float* table, int X, int Y, int Z, const float* geom, const float* xs, const float* ys,
                               const float* ds, const float* zvals, int N, int D, int H, int W, const float* bounds_host,
                               int activated, float* maps, void* stream);
extern "C" int coocc_render_nearest(const float* table, int X, int Y, int Z, const float* geom,
                                    const float* zvals, int N, int D, int H, int W, const float* bounds_host,
                                    int activated, float* maps, void* stream) {
  return render_nearest_impl(table, X, Y, Z, geom, nullptr, nullptr, nullptr, zvals, N, D, H, W, bounds_host, activated, maps, stream);
}
extern "C" int coocc_render_nearest_cams(const float* table, int X, int Y, int Z, const float* mats, const float* xs, const float* ys,
                                         const float* ds, const float* zvals, int N, int D, int H, int W,
                                         const float* bounds_host, int activated, float* maps, void* stream) {
  COOCC_CHECK_ARG(xs && ys && ds, "render_nearest_cams: null frustum axes");
  return render_nearest_impl(table, X, Y, Z, mats, xs, ys, ds, zvals, N, D, H, W, bounds_host, activated, maps, stream);
}
static int render_nearest_impl(const float* table, int X, int Y, int Z, const float* geom, const float* xs, const float* ys,
                               const float* ds, const float* zvals, int N, int D, int H, int W, const float* bounds_host,
                               int activated, float* maps, void* stream) {
  COOCC_CHECK_ARG(table && geom && zvals && maps && bounds_host, "render_nearest: null pointer");
  COOCC_CHECK_ARG(N > 0 && D > 0 && D <= 256 && H > 0 && W > 0, "render_nearest: bad sizes (D <= 256)");
  const float* bd = bounds_host;  // xbound(3), ybound(3), zbound(3) = lo, hi, step (coocc_ray.py:577)
  float dx = bd[2], dy = bd[5], dz = bd[8];
  // dx/bx/nx exactly as coocc_ray.py:579-581 then bx - dx/2 (:582), all in fp32
  float bx = bd[0] + bd[2] / 2.0f, by = bd[3] + bd[5] / 2.0f, bz = bd[6] + bd[8] / 2.0f;
  float lox = bx - dx / 2.f, loy = by - dy / 2.f, loz = bz - dz / 2.f;
  float nx = (bd[1] - bd[0]) / bd[2], ny = (bd[4] - bd[3]) / bd[5], nz = (bd[7] - bd[6]) / bd[8];
  // the reference would raise IndexError where the hard-coded bounds exceed the volume
  COOCC_CHECK_ARG(nx <= (float)X && ny <= (float)Y && nz <= (float)Z && X <= 1024 && Y <= 1024 && Z <= 1024,
                  "render_nearest: render bounds exceed the voxel volume");
  // full 32-ray tiles plus one remainder tile (W = 100 -> 32,32,32,4): phase 1 keeps all 32 lanes of a depth
  // phase busy and phase 2 walks 8 rays per step, so 25-ray balanced tiles wasted 22 % of both (4 steps for 25 rays)
  const int rt = W < RT_MAX ? W : RT_MAX;
  const int tiles = (W + rt - 1) / rt;
  size_t lds = sizeof(int) * (size_t)rt * (D + 1);
  dim3 grid(tiles, H, N);
#define RN_LAUNCH(ACT, CHT, GEO)                                                                                                  \
  hipLaunchKernelGGL((k_render_nearest<ACT, CHT, GEO>), grid, dim3(256), lds, as_stream(stream), table, Y, Z, geom, zvals, D, H, W, rt, \
                     lox, loy, loz, dx, dy, dz, nx, ny, nz, maps, xs, ys, ds)
  static const int geo_lds = getenv("COOCC_RENDER_GEO_LDS") ? atoi(getenv("COOCC_RENDER_GEO_LDS")) : 0;   // 1: round 3's two-phase form
  if (xs && !geo_lds) {
    dim3 g2((W + 7) / 8, H, N);
#define RG_LAUNCH(ACT, CHT)                                                                                                         \
  hipLaunchKernelGGL((k_render_rays_geo<ACT, CHT>), g2, dim3(256), 0, as_stream(stream), table, Y, Z, geom, xs, ys, ds, zvals, D, H, W, lox, \
                     loy, loz, dx, dy, dz, nx, ny, nz, maps)
    if (D > 128) { if (activated) RG_LAUNCH(1, 8); else RG_LAUNCH(0, 8); }
    else { if (activated) RG_LAUNCH(1, 4); else RG_LAUNCH(0, 4); }
#undef RG_LAUNCH
  } else if (xs) {
    if (D > 128) { if (activated) RN_LAUNCH(1, 8, true); else RN_LAUNCH(0, 8, true); }
    else { if (activated) RN_LAUNCH(1, 4, true); else RN_LAUNCH(0, 4, true); }
  } else {
    if (D > 128) { if (activated) RN_LAUNCH(1, 8, false); else RN_LAUNCH(0, 8, false); }    // up to 8 samples per lane
    else { if (activated) RN_LAUNCH(1, 4, false); else RN_LAUNCH(0, 4, false); }
  }
#undef RN_LAUNCH
  COOCC_LAUNCH_CHECK("k_render_nearest");
  return COOCC_OK;
}

// ------------------------------------------------------------------ R3: library renderer
// volume_sampling (P/utils/render_ray.py:28-48): F.grid_sample(features [1,C,D,W,H], pts)
// trilinear, align_corners=True, padding_mode='border'.  grid x indexes the LAST volume dim.
// vol: channels-last rows [d0*d1*d2, C] of the [1,C,d0,d1,d2] reference volume.
__global__ __launch_bounds__(256) void k_volume_sampling(const float* __restrict__ vol, int C, int d0, int d1, int d2,
                                                          const float* __restrict__ pts, int n, float ax, float ay,
                                                          float az, float sx, float sy, float sz,
                                                          float* __restrict__ feat, uint8_t* __restrict__ mask) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  // norm_pts = (p - aabb[0]) * (1/size*2) - 1   (:41-43)
  float gx = (pts[wave * 3 + 0] - ax) * sx - 1.f;
  float gy = (pts[wave * 3 + 1] - ay) * sy - 1.f;
  float gz = (pts[wave * 3 + 2] - az) * sz - 1.f;
  if (lane == 0 && mask) mask[wave] = (gx < 1.f && gx > -1.f && gy < 1.f && gy > -1.f && gz < 1.f && gz > -1.f) ? 1 : 0;
  // unnormalise (align_corners=True): ((g+1)/2)*(size-1); border: clamp to [0,size-1]
  float fx = fminf(fmaxf((gx + 1.f) / 2.f * (float)(d2 - 1), 0.f), (float)(d2 - 1));
  float fy = fminf(fmaxf((gy + 1.f) / 2.f * (float)(d1 - 1), 0.f), (float)(d1 - 1));
  float fz = fminf(fmaxf((gz + 1.f) / 2.f * (float)(d0 - 1), 0.f), (float)(d0 - 1));
  int x0 = (int)floorf(fx), y0 = (int)floorf(fy), z0 = (int)floorf(fz);
  float tx = fx - x0, ty = fy - y0, tz = fz - z0;
  int x1 = min(x0 + 1, d2 - 1), y1 = min(y0 + 1, d1 - 1), z1 = min(z0 + 1, d0 - 1);
  // ATen weights: (x1-x)*(y1-y)*(z1-z) with x1 = x0+1 (out-of-range corners carry weight 0 here)
  float wx0 = 1.f - tx, wy0 = 1.f - ty, wz0 = 1.f - tz;
  auto row = [&](int z, int y, int x) { return vol + (((size_t)z * d1 + y) * d2 + x) * C; };
  const float *r000 = row(z0, y0, x0), *r001 = row(z0, y0, x1), *r010 = row(z0, y1, x0), *r011 = row(z0, y1, x1);
  const float *r100 = row(z1, y0, x0), *r101 = row(z1, y0, x1), *r110 = row(z1, y1, x0), *r111 = row(z1, y1, x1);
  for (int c = lane; c < C; c += 64) {
    float v = r000[c] * (wx0 * wy0 * wz0) + r001[c] * (tx * wy0 * wz0) + r010[c] * (wx0 * ty * wz0) +
              r011[c] * (tx * ty * wz0) + r100[c] * (wx0 * wy0 * tz) + r101[c] * (tx * wy0 * tz) +
              r110[c] * (wx0 * ty * tz) + r111[c] * (tx * ty * tz);
    feat[(size_t)wave * C + c] = v;
  }
}

extern "C" int coocc_volume_sampling(const float* vol, int C, int d0, int d1, int d2, const float* pts, int n,
                                     const float* aabb_host, float* feat, uint8_t* mask, void* stream) {
  COOCC_CHECK_ARG(vol && pts && feat && aabb_host && C > 0 && n >= 0, "volume_sampling: bad args");
  if (n == 0) return COOCC_OK;
  const float* a = aabb_host;
  float sx = 1.0f / (a[3] - a[0]) * 2, sy = 1.0f / (a[4] - a[1]) * 2, sz = 1.0f / (a[5] - a[2]) * 2;
  hipLaunchKernelGGL(k_volume_sampling, dim3(cdiv((long long)n * 64, 256)), dim3(256), 0, as_stream(stream), vol, C,
                     d0, d1, d2, pts, n, a[0], a[1], a[2], sx, sy, sz, feat, mask);
  COOCC_LAUNCH_CHECK("k_volume_sampling");
  return COOCC_OK;
}

// raw2outputs (render_ray.py:198-249): alpha = 1 - exp(-sigma) (no interval), exclusive
// cumprod of (1 - alpha + 1e-10), rgb / renormalised clamped depth.  One wave per ray.
__global__ __launch_bounds__(256) void k_raw2outputs(const float* __restrict__ raw, const float* __restrict__ z, int R,
                                                      int S, int white_bkgd, float zmin, float zmax,
                                                      float* __restrict__ rgb, float* __restrict__ depth,
                                                      float* __restrict__ weights) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (r >= R) return;
  const int CH = (S + 63) / 64, d0 = lane * CH;
  float prod = 1.f;
  for (int j = 0; j < CH; ++j) {
    int d = d0 + j;
    if (d < S) prod *= 1.f - (1.f - expf(-raw[((size_t)r * S + d) * 4 + 3])) + 1e-10f;
  }
  float inc = prod;
  for (int o = 1; o < 64; o <<= 1) {
    float v = __shfl_up(inc, o);
    if (lane >= o) inc *= v;
  }
  float T = __shfl_up(inc, 1);
  if (lane == 0) T = 1.f;
  float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, aw = 0.f;
  for (int j = 0; j < CH; ++j) {
    int d = d0 + j;
    if (d < S) {
      const float* q = raw + ((size_t)r * S + d) * 4;
      float alpha = 1.f - expf(-q[3]);
      float w = alpha * T;
      if (weights) weights[(size_t)r * S + d] = w;
      ar += w * q[0]; ag += w * q[1]; ab += w * q[2];
      ad += w * z[(size_t)r * S + d];
      aw += w;
      T *= 1.f - alpha + 1e-10f;
    }
  }
  ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); ad = wave_sum(ad); aw = wave_sum(aw);
  if (lane == 0) {
    float bg = white_bkgd ? 1.f - aw : 0.f;
    rgb[r * 3 + 0] = ar + bg; rgb[r * 3 + 1] = ag + bg; rgb[r * 3 + 2] = ab + bg;
    float dm = ad / (aw + 1e-8f);
    depth[r] = fminf(fmaxf(dm, zmin), zmax);
  }
}

extern "C" int coocc_raw2outputs(const float* raw, const float* z, int R, int S, int white_bkgd, float zmin,
                                 float zmax, float* rgb, float* depth, float* weights, void* stream) {
  COOCC_CHECK_ARG(raw && z && rgb && depth && R >= 0 && S > 0, "raw2outputs: bad args");
  if (R == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_raw2outputs, dim3(cdiv((long long)R * 64, 256)), dim3(256), 0, as_stream(stream), raw, z, R, S,
                     white_bkgd, zmin, zmax, rgb, depth, weights);
  COOCC_LAUNCH_CHECK("k_raw2outputs");
  return COOCC_OK;
}

// ------------------------------------------------------------------ L1: render losses
// coocc_ray.py:423-433.  out[0] = mse(depths[fg]/D, gt_bin[fg]/D), out[1] = mse(rgbs, rgb_gt).
// Deterministic two-pass reduction: every block writes fp64 partial sums of its pixel slice, one block adds the
// partials in block order (a single block walking 1-9 M pixels took 1.5 / 11 ms at r50 / r101).
__global__ __launch_bounds__(256) void k_render_losses_part(const float* __restrict__ rgbs, const float* __restrict__ depths,
                                                             const float* __restrict__ rgb_gt,
                                                             const float* __restrict__ depth_gt, size_t npix, float D,
                                                             double* __restrict__ part) {
  __shared__ double s_d[4], s_c[4], s_n[4];
  double sd = 0, sc = 0, sn = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
    float g = (depth_gt[i] - (2.f - 0.5f / 2.f)) / 0.5f;
    g = fminf(fmaxf(g, 0.f), D);
    if (g > 0.f) {
      float e = depths[i] / D - g / D;
      sd += (double)e * e;
      sn += 1.0;
    }
    for (int k = 0; k < 3; ++k) {
      float e = rgbs[i * 3 + k] - rgb_gt[i * 3 + k];
      sc += (double)e * e;
    }
  }
  for (int m = 32; m > 0; m >>= 1) {
    sd += __shfl_xor(sd, m); sc += __shfl_xor(sc, m); sn += __shfl_xor(sn, m);
  }
  if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = sd; s_c[threadIdx.x >> 6] = sc; s_n[threadIdx.x >> 6] = sn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 3 + 0] = s_d[0] + s_d[1] + s_d[2] + s_d[3];
    part[blockIdx.x * 3 + 1] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    part[blockIdx.x * 3 + 2] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
  }
}

__global__ __launch_bounds__(64) void k_render_losses_final(const double* __restrict__ part, int nblocks, size_t npix,
                                                             float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  double a = 0, b = 0, c = 0;
  for (int i = 0; i < nblocks; ++i) { a += part[i * 3]; b += part[i * 3 + 1]; c += part[i * 3 + 2]; }
  out[0] = (float)(a / c);           // mean over the foreground pixels (NaN when there are none, like torch)
  out[1] = (float)(b / (3.0 * (double)npix));
  out[2] = (float)c;                 // foreground count (consumed by coocc_render_losses_bwd)
}

extern "C" int coocc_render_losses(const float* rgbs, const float* depths, const float* rgb_gt,
                                   const float* depth_gt, int64_t npix, int D, float* out, void* ws, size_t ws_bytes,
                                   void* stream) {
  COOCC_CHECK_ARG(rgbs && depths && rgb_gt && depth_gt && out && npix > 0 && D > 0, "render_losses: bad args");
  int nblocks = (int)std::min<int64_t>(1024, (npix + 1023) / 1024);
  COOCC_CHECK_ARG(ws && ws_bytes >= sizeof(double) * 3 * (size_t)nblocks, "render_losses: workspace too small (24 KB)");
  hipLaunchKernelGGL(k_render_losses_part, dim3(nblocks), dim3(256), 0, as_stream(stream), rgbs, depths, rgb_gt, depth_gt,
                     (size_t)npix, (float)D, (double*)ws);
  hipLaunchKernelGGL(k_render_losses_final, dim3(1), dim3(64), 0, as_stream(stream), (const double*)ws, nblocks, (size_t)npix, out);
  COOCC_LAUNCH_CHECK("k_render_losses");
  return COOCC_OK;
}
