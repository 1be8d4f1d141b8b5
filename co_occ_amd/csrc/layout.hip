// Layout conversion at the reference boundary ([B,C,X,Y,Z] <-> channels-last voxel rows),
// the BiFuser_N prologue (K1) and stream compaction (torch.nonzero replacement).
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ error state
static thread_local char g_err[512] = "";

int coocc_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* coocc_last_error(void) { return g_err; }
extern "C" int coocc_abi_version(void) { return 1; }

// ------------------------------------------------------------------ transposes
// One block = 64 voxels x (up to) 128 channels staged through LDS so that both the
// [C][V] side (lanes along V) and the row side (lanes along C) are coalesced.
#define TV 64
#define TC 128

__global__ __launch_bounds__(256) void k_ncdhw_to_ndhwc(const float* __restrict__ src,
                                                         float* __restrict__ dst, int C, int V,
                                                         int dst_stride, int dst_coff) {
  __shared__ float tile[TV][TC + 1];
  const int b = blockIdx.y;
  const int v0 = blockIdx.x * TV;
  const int t = threadIdx.x;
  // one 128-channel slab per blockIdx.z: small volumes with many channels (camera feature maps: 6 x 512 x 704)
  // still spread over the whole chip
  const int c0 = blockIdx.z * TC;
  const int cn = min(TC, C - c0);
  if (cn == TC) {
    // full slab (every shipped shape): 32 independent loads per thread in flight, no division in the store loop
    const int v = v0 + (t & 63);
    const float* sp = src + ((size_t)b * C + c0 + (t >> 6)) * V + v;
    float r[TC / 4];
#pragma unroll
    for (int k = 0; k < TC / 4; ++k) r[k] = v < V ? sp[(size_t)(4 * k) * V] : 0.f;
#pragma unroll
    for (int k = 0; k < TC / 4; ++k) tile[t & 63][(t >> 6) + 4 * k] = r[k];
    __syncthreads();
#pragma unroll 8
    for (int i = t; i < TV * TC; i += 256) {
      const int vv = i >> 7, c = i & (TC - 1);
      if (v0 + vv < V) dst[((size_t)b * V + v0 + vv) * dst_stride + dst_coff + c0 + c] = tile[vv][c];
    }
    return;
  }
  for (int c = t >> 6; c < cn; c += 4) {
    int v = v0 + (t & 63);
    tile[t & 63][c] = v < V ? src[((size_t)b * C + c0 + c) * V + v] : 0.f;
  }
  __syncthreads();
  for (int i = t; i < TV * cn; i += 256) {
    int v = i / cn, c = i - v * cn;
    if (v0 + v < V) dst[((size_t)b * V + v0 + v) * dst_stride + dst_coff + c0 + c] = tile[v][c];
  }
}

__global__ __launch_bounds__(256) void k_ndhwc_to_ncdhw(const float* __restrict__ src,
                                                         float* __restrict__ dst, int C, int V,
                                                         int src_stride, int src_coff) {
  __shared__ float tile[TV][TC + 1];
  const int b = blockIdx.y;
  const int v0 = blockIdx.x * TV;
  const int t = threadIdx.x;
  for (int c0 = 0; c0 < C; c0 += TC) {
    const int cn = min(TC, C - c0);
    for (int i = t; i < TV * cn; i += 256) {
      int v = i / cn, c = i - v * cn;
      tile[v][c] = (v0 + v < V) ? src[((size_t)b * V + v0 + v) * src_stride + src_coff + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int c = t >> 6; c < cn; c += 4) {
      int v = v0 + (t & 63);
      if (v < V) dst[((size_t)b * C + c0 + c) * V + v] = tile[t & 63][c];
    }
    __syncthreads();
  }
}

extern "C" int coocc_ncdhw_to_ndhwc(const float* src, float* dst, int B, int C, int V, int dst_stride,
                                    int dst_coff, void* stream) {
  COOCC_CHECK_ARG(src && dst && B > 0 && C > 0 && V > 0 && dst_stride >= dst_coff + C, "ncdhw_to_ndhwc: bad args");
  dim3 grid(cdiv(V, TV), B, cdiv(C, TC));
  hipLaunchKernelGGL(k_ncdhw_to_ndhwc, grid, dim3(256), 0, as_stream(stream), src, dst, C, V, dst_stride, dst_coff);
  COOCC_LAUNCH_CHECK("k_ncdhw_to_ndhwc");
  return COOCC_OK;
}

extern "C" int coocc_ndhwc_to_ncdhw(const float* src, float* dst, int B, int C, int V, int src_stride,
                                    int src_coff, void* stream) {
  COOCC_CHECK_ARG(src && dst && B > 0 && C > 0 && V > 0 && src_stride >= src_coff + C, "ndhwc_to_ncdhw: bad args");
  dim3 grid(cdiv(V, TV), B);
  hipLaunchKernelGGL(k_ndhwc_to_ncdhw, grid, dim3(256), 0, as_stream(stream), src, dst, C, V, src_stride, src_coff);
  COOCC_LAUNCH_CHECK("k_ndhwc_to_ncdhw");
  return COOCC_OK;
}

// ------------------------------------------------------------------ K1 prologue
// bifuser_n.py:129-135 + the zero-filled halves of :164-168, fused: one read of each
// input volume, one write of the [img|pts|0|0] rows, and the non-empty flags.
__global__ __launch_bounds__(256) void k_fuser_prepare(const float* __restrict__ img,
                                                        const float* __restrict__ pts,
                                                        float* __restrict__ cat4,
                                                        uint8_t* __restrict__ flag_img,
                                                        uint8_t* __restrict__ flag_pts, int C, int V) {
  __shared__ float tile[TV][TC + 1];
  const int b = blockIdx.y;
  const int v0 = blockIdx.x * TV;
  const int t = threadIdx.x;
  const int stride = 4 * C;
  for (int mod = 0; mod < 2; ++mod) {
    const float* src = mod ? pts : img;
    uint8_t* flags = mod ? flag_pts : flag_img;
    float rsum = 0.f;  // channel sum of voxel t (threads 0..63), ascending c, fp32
    for (int c0 = 0; c0 < C; c0 += TC) {
      const int cn = min(TC, C - c0);
      for (int c = t >> 6; c < cn; c += 4) {
        int v = v0 + (t & 63);
        tile[t & 63][c] = v < V ? src[((size_t)b * C + c0 + c) * V + v] : 0.f;
      }
      __syncthreads();
      if (t < TV)
        for (int c = 0; c < cn; ++c) rsum += tile[t][c];
      for (int i = t; i < TV * cn; i += 256) {
        int v = i / cn, c = i - v * cn;
        if (v0 + v < V) cat4[((size_t)b * V + v0 + v) * stride + mod * C + c0 + c] = tile[v][c];
      }
      __syncthreads();
    }
    if (t < TV && v0 + t < V) flags[(size_t)b * V + v0 + t] = rsum != 0.f ? 1 : 0;
  }
  // fused_feats_img / fused_feats_pts start as zeros (bifuser_n.py:164,168)
  for (int i = t; i < TV * 2 * C; i += 256) {
    int v = i / (2 * C), c = i - v * 2 * C;
    if (v0 + v < V) cat4[((size_t)b * V + v0 + v) * stride + 2 * C + c] = 0.f;
  }
}

extern "C" int coocc_fuser_prepare(const float* img, const float* pts, float* cat4, uint8_t* flag_img,
                                   uint8_t* flag_pts, int B, int C, int V, void* stream) {
  COOCC_CHECK_ARG(img && pts && cat4 && flag_img && flag_pts && B > 0 && C > 0 && V > 0, "fuser_prepare: bad args");
  dim3 grid(cdiv(V, TV), B);
  hipLaunchKernelGGL(k_fuser_prepare, grid, dim3(256), 0, as_stream(stream), img, pts, cat4, flag_img, flag_pts, C, V);
  COOCC_LAUNCH_CHECK("k_fuser_prepare");
  return COOCC_OK;
}

// Same prologue when a producer already hands over channels-last rows (the fused lift-splat writes [V, stride] rows; the
// sparse LiDAR encoder scatters rows): *_rows != 0 -> src is [B*V, *_stride] rows instead of NCDHW.  A row source that IS
// the destination slot (lift-splat wrote straight into cat4[:, 0:C]) is only read for its flag, not copied.
__global__ __launch_bounds__(256) void k_fuser_prepare_rows(const float* __restrict__ img, int img_rows, int img_stride,
                                                             const float* __restrict__ pts, int pts_rows, int pts_stride,
                                                             float* __restrict__ cat4, uint8_t* __restrict__ flag_img,
                                                             uint8_t* __restrict__ flag_pts, int C, int V) {
  __shared__ float tile[TV][TC + 1];
  const int b = blockIdx.y;
  const int v0 = blockIdx.x * TV;
  const int t = threadIdx.x;
  const int stride = 4 * C;
  for (int mod = 0; mod < 2; ++mod) {
    const float* src = mod ? pts : img;
    const int rows = mod ? pts_rows : img_rows, sstride = mod ? pts_stride : img_stride;
    uint8_t* flags = mod ? flag_pts : flag_img;
    const bool in_place = rows && src == cat4 + mod * C && sstride == stride;
    float rsum = 0.f;  // channel sum of voxel t (threads 0..63), ascending c, fp32
    for (int c0 = 0; c0 < C; c0 += TC) {
      const int cn = min(TC, C - c0);
      if (rows) {
        if (cn == TC) {            // full channel chunk: TC / 4 lanes x 16 B per voxel row, constant divisors
          for (int i = t; i < TV * (TC / 4); i += 256) {
            const int v = i / (TC / 4), c = (i % (TC / 4)) * 4;
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
            if (v0 + v < V) q = *(const f32x4*)(src + ((size_t)b * V + v0 + v) * sstride + c0 + c);
            tile[v][c] = q[0]; tile[v][c + 1] = q[1]; tile[v][c + 2] = q[2]; tile[v][c + 3] = q[3];
          }
        } else {
          for (int i = t; i < TV * cn; i += 256) {
            int v = i / cn, c = i - v * cn;
            tile[v][c] = v0 + v < V ? src[((size_t)b * V + v0 + v) * sstride + c0 + c] : 0.f;
          }
        }
      } else {
        for (int c = t >> 6; c < cn; c += 4) {
          int v = v0 + (t & 63);
          tile[t & 63][c] = v < V ? src[((size_t)b * C + c0 + c) * V + v] : 0.f;
        }
      }
      __syncthreads();
      if (t < TV)
        for (int c = 0; c < cn; ++c) rsum += tile[t][c];
      if (!in_place)
        for (int i = t; i < TV * cn; i += 256) {
          int v = i / cn, c = i - v * cn;
          if (v0 + v < V) cat4[((size_t)b * V + v0 + v) * stride + mod * C + c0 + c] = tile[v][c];
        }
      __syncthreads();
    }
    if (t < TV && v0 + t < V) flags[(size_t)b * V + v0 + t] = rsum != 0.f ? 1 : 0;
  }
  for (int i = t; i < TV * 2 * C; i += 256) {
    int v = i / (2 * C), c = i - v * 2 * C;
    if (v0 + v < V) cat4[((size_t)b * V + v0 + v) * stride + 2 * C + c] = 0.f;
  }
}

// Fast form for C % 32 == 0 and 16-byte aligned rows (every shipped configuration): 256 voxels per workgroup, ONE thread per
// voxel for the flag -- its channel sum runs over all C channels in ascending order exactly as above, but 256 threads do it
// instead of 64 -- channel chunks of 32 staged through LDS so that every global access is a 16-byte vector of a 128-byte run
// (row sources: 8 lanes per voxel; NCDHW sources: 256 consecutive voxels of one channel), vector stores for the slot copy and
// for the zero fill of slots 2 / 3.  The first version (64 voxels per workgroup, 64 summing threads, scalar stores with integer
// divisions in the index math) ran at 0.9 TB/s: 160-187 us per sample at configs[1].
constexpr int FV = 256, FC = 32;
__global__ __launch_bounds__(256) void k_fuser_prepare_rows_fast(const float* __restrict__ img, int img_rows, int img_stride,
                                                                  const float* __restrict__ pts, int pts_rows, int pts_stride,
                                                                  float* __restrict__ cat4, uint8_t* __restrict__ flag_img,
                                                                  uint8_t* __restrict__ flag_pts, int C, int V) {
  __shared__ float tile[FV][FC + 1];
  const int b = blockIdx.y;
  const int v0 = blockIdx.x * FV;
  const int t = threadIdx.x;
  const int stride = 4 * C;
  const int rv = t >> 3, rc = (t & 7) * 4;        // row-form accesses: 32 voxels x 8 lanes x 16 B per pass
  for (int mod = 0; mod < 2; ++mod) {
    const float* src = mod ? pts : img;
    const int rows = mod ? pts_rows : img_rows, sstride = mod ? pts_stride : img_stride;
    uint8_t* flags = mod ? flag_pts : flag_img;
    const bool in_place = rows && src == cat4 + mod * C && sstride == stride;
    float rsum = 0.f;                             // channel sum of voxel v0 + t, ascending c, fp32
    for (int c0 = 0; c0 < C; c0 += FC) {
      if (rows) {
#pragma unroll
        for (int ps = 0; ps < FV / 32; ++ps) {
          const int v = ps * 32 + rv;
          f32x4 q = {0.f, 0.f, 0.f, 0.f};
          if (v0 + v < V) q = *(const f32x4*)(src + ((size_t)b * V + v0 + v) * sstride + c0 + rc);
          tile[v][rc] = q[0]; tile[v][rc + 1] = q[1]; tile[v][rc + 2] = q[2]; tile[v][rc + 3] = q[3];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < FC; ++c) rsum += tile[t][c];
      } else {
        // NCDHW: lanes = consecutive voxels of one channel; the value goes into the sum directly and into LDS for the copy
#pragma unroll 8
        for (int c = 0; c < FC; ++c) {
          const float x = v0 + t < V ? src[((size_t)b * C + c0 + c) * V + v0 + t] : 0.f;
          rsum += x;
          tile[t][c] = x;
        }
        __syncthreads();
      }
      if (!in_place) {
#pragma unroll
        for (int ps = 0; ps < FV / 32; ++ps) {
          const int v = ps * 32 + rv;
          if (v0 + v < V) {
            const f32x4 q = {tile[v][rc], tile[v][rc + 1], tile[v][rc + 2], tile[v][rc + 3]};
            *(f32x4*)(cat4 + ((size_t)b * V + v0 + v) * stride + mod * C + c0 + rc) = q;
          }
        }
      }
      __syncthreads();
    }
    if (v0 + t < V) flags[(size_t)b * V + v0 + t] = rsum != 0.f ? 1 : 0;
  }
  // slots 2 / 3: 2C floats per voxel = C/2 vectors
  const int vec_per_row = C >> 1;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (vec_per_row <= 256 && 256 % vec_per_row == 0) {      // a thread keeps its column: no division in the loop
    const int vpp = 256 / vec_per_row, tv = t / vec_per_row, c = (t - tv * vec_per_row) * 4;
    for (int v = tv; v < FV; v += vpp)
      if (v0 + v < V) *(f32x4*)(cat4 + ((size_t)b * V + v0 + v) * stride + 2 * C + c) = z;
  } else {
    for (int i = t; i < FV * vec_per_row; i += 256) {
      const int v = i / vec_per_row, c = (i - v * vec_per_row) * 4;
      if (v0 + v < V) *(f32x4*)(cat4 + ((size_t)b * V + v0 + v) * stride + 2 * C + c) = z;
    }
  }
}

extern "C" int coocc_fuser_prepare_rows(const float* img, int img_rows, int img_stride, const float* pts, int pts_rows,
                                        int pts_stride, float* cat4, uint8_t* flag_img, uint8_t* flag_pts, int B, int C, int V,
                                        void* stream) {
  COOCC_CHECK_ARG(img && pts && cat4 && flag_img && flag_pts && B > 0 && C > 0 && V > 0, "fuser_prepare_rows: bad args");
  COOCC_CHECK_ARG((!img_rows || img_stride >= C) && (!pts_rows || pts_stride >= C), "fuser_prepare_rows: row stride < C");
  const bool vec_ok = C % FC == 0 && ((uintptr_t)cat4 & 15) == 0 && (!img_rows || (img_stride % 4 == 0 && ((uintptr_t)img & 15) == 0)) &&
                      (!pts_rows || (pts_stride % 4 == 0 && ((uintptr_t)pts & 15) == 0));
  if (vec_ok) {
    hipLaunchKernelGGL(k_fuser_prepare_rows_fast, dim3(cdiv(V, FV), B), dim3(256), 0, as_stream(stream), img, img_rows, img_stride,
                       pts, pts_rows, pts_stride, cat4, flag_img, flag_pts, C, V);
    COOCC_LAUNCH_CHECK("k_fuser_prepare_rows_fast");
    return COOCC_OK;
  }
  dim3 grid(cdiv(V, TV), B);
  hipLaunchKernelGGL(k_fuser_prepare_rows, grid, dim3(256), 0, as_stream(stream), img, img_rows, img_stride, pts, pts_rows,
                     pts_stride, cat4, flag_img, flag_pts, C, V);
  COOCC_LAUNCH_CHECK("k_fuser_prepare_rows");
  return COOCC_OK;
}

// ------------------------------------------------------------------ compaction
#define CB 1024  // flags per block

__global__ __launch_bounds__(256) void k_flag_count(const uint8_t* __restrict__ flags, int total,
                                                     int32_t* __restrict__ blk) {
  __shared__ int wsum[4];
  int base = blockIdx.x * CB + threadIdx.x * 4;
  int c = 0;
  for (int j = 0; j < 4; ++j)
    if (base + j < total) c += flags[base + j] != 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of nblk block counts by one block; writes total to *count
__global__ __launch_bounds__(1024) void k_scan_blocks(int32_t* __restrict__ blk, int nblk,
                                                       int32_t* __restrict__ count) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < nblk ? blk[i] : 0;
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) {
      int n = __shfl_up(inc, o);
      if ((threadIdx.x & 63) >= o) inc += n;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
    int carry = carry_s;
    if (i < nblk) blk[i] = carry + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = carry_s;
}

__global__ __launch_bounds__(256) void k_flag_write(const uint8_t* __restrict__ flags, int total,
                                                     const int32_t* __restrict__ blk,
                                                     int32_t* __restrict__ lin, int32_t* __restrict__ map) {
  __shared__ int wsum[4];
  int base = blockIdx.x * CB + threadIdx.x * 4;
  int f[4], c = 0;
  for (int j = 0; j < 4; ++j) {
    f[j] = (base + j < total) && flags[base + j] != 0;
    c += f[j];
  }
  int inc = c;
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o);
    if ((threadIdx.x & 63) >= o) inc += n;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  int off = blk[blockIdx.x] + inc - c;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
  for (int j = 0; j < 4; ++j) {
    if (map && base + j < total) map[base + j] = f[j] ? off : -1;        // the inverse: element -> its ordinal in lin, or -1
    if (f[j]) lin[off++] = base + j;
  }
}

// Sum of v over the block's 256 threads, returned to every thread (s: 4 ints of LDS).
__device__ __forceinline__ int block_sum256(int v, int* s) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();                                  // s may still be read from a previous use
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  return s[0] + s[1] + s[2] + s[3];
}

__device__ __forceinline__ int nonzero_bytes(uint32_t x) {
  return __popc((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);
}

// The last pass with the block's start ordinal computed by the block itself (no scan launch):
//   FROM_FLAGS: counts the non-zero flags of every earlier block (16 flags per load; flags 16-byte aligned, totals up to
//               ONE_PASS_MAX: at most 32 coalesced loads per thread, eight in flight) -- the whole compaction is this ONE launch;
//   otherwise : adds up the earlier blocks' counts left by k_flag_count.
// The last block also writes the total.  Same lin / map as the three-launch form (integers: no order to differ in).
#define ONE_PASS_MAX (128 * CB)
#define TWO_PASS_MAX_BLOCKS 2048                     // 2 M flags: at most 8 count loads per thread in the last block
template <bool FROM_FLAGS>
__global__ __launch_bounds__(256) void k_flag_write_px(const uint8_t* __restrict__ flags, int total,
                                                        const int32_t* __restrict__ blk, int32_t* __restrict__ count,
                                                        int32_t* __restrict__ lin, int32_t* __restrict__ map) {
  __shared__ int wsum[4], psum[4];
  int before = 0;
  if (FROM_FLAGS) {
    const uint4* w = (const uint4*)flags;
    const int nw = blockIdx.x * (CB / 16);          // whole 16-byte words: every earlier block is full
    for (int i0 = threadIdx.x; i0 < nw; i0 += 256 * 8) {
      uint4 x[8];                                   // eight independent loads per round trip (at most four round trips)
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = i0 + k * 256 < nw ? w[i0 + k * 256] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int k = 0; k < 8; ++k) before += nonzero_bytes(x[k].x) + nonzero_bytes(x[k].y) + nonzero_bytes(x[k].z) + nonzero_bytes(x[k].w);
    }
  } else {
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) before += blk[i];
  }
  before = block_sum256(before, psum);
  int base = blockIdx.x * CB + threadIdx.x * 4;
  int f[4], c = 0;
  for (int j = 0; j < 4; ++j) {
    f[j] = (base + j < total) && flags[base + j] != 0;
    c += f[j];
  }
  int inc = c;
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(inc, o);
    if ((threadIdx.x & 63) >= o) inc += n;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  int off = before + inc - c;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *count = off + c;
  for (int j = 0; j < 4; ++j) {
    if (map && base + j < total) map[base + j] = f[j] ? off : -1;        // the inverse: element -> its ordinal in lin, or -1
    if (f[j]) lin[off++] = base + j;
  }
}

extern "C" int coocc_compact_flags_ex(const uint8_t* flags, int total, int32_t* lin, int32_t* count, int32_t* map, void* ws,
                                      size_t ws_bytes, void* stream);
extern "C" int coocc_compact_flags(const uint8_t* flags, int total, int32_t* lin, int32_t* count, void* ws,
                                   size_t ws_bytes, void* stream) {
  return coocc_compact_flags_ex(flags, total, lin, count, nullptr, ws, ws_bytes, stream);
}
// ... and, with map != NULL, the inverse table map[element] = ordinal in lin (or -1) written by the same last pass.
// Launches: 1 up to ONE_PASS_MAX flags (the fused grids of every shipped config), 2 up to 2 M flags, 3 above; COOCC_COMPACT_SCAN=1 keeps the
// count / scan / write form of rounds 1-4 (read per call: the direct test compares the forms).
extern "C" int coocc_compact_flags_ex(const uint8_t* flags, int total, int32_t* lin, int32_t* count, int32_t* map, void* ws,
                                      size_t ws_bytes, void* stream) {
  COOCC_CHECK_ARG(flags && lin && count && ws && total > 0, "compact_flags: bad args");
  int nblk = (int)cdiv(total, CB);
  if (ws_bytes < sizeof(int32_t) * (size_t)nblk) return coocc_set_error(COOCC_ENOMEM, "compact_flags: workspace too small");
  int32_t* blk = (int32_t*)ws;
  const char* e = getenv("COOCC_COMPACT_SCAN");
  if (e && e[0] == '1') {
    hipLaunchKernelGGL(k_flag_count, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, as_stream(stream), blk, nblk, count);
    hipLaunchKernelGGL(k_flag_write, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk, lin, map);
  } else if (total <= ONE_PASS_MAX && ((uintptr_t)flags & 15) == 0) {
    hipLaunchKernelGGL(k_flag_write_px<true>, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk, count, lin, map);
  } else if (nblk <= TWO_PASS_MAX_BLOCKS) {
    hipLaunchKernelGGL(k_flag_count, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk);
    hipLaunchKernelGGL(k_flag_write_px<false>, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk, count, lin, map);
  } else {                                          // the per-block prefix sums are quadratic in the block count: scan launch
    hipLaunchKernelGGL(k_flag_count, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, as_stream(stream), blk, nblk, count);
    hipLaunchKernelGGL(k_flag_write, dim3(nblk), dim3(256), 0, as_stream(stream), flags, total, blk, lin, map);
  }
  COOCC_LAUNCH_CHECK("compact_flags");
  return COOCC_OK;
}

__global__ void k_lin_to_coords(const int32_t* __restrict__ lin, int n, int X, int Y, int Z,
                                float* __restrict__ xyz, int64_t* __restrict__ bxyz) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int l = lin[i];
  int z = l % Z; l /= Z;
  int y = l % Y; l /= Y;
  int x = l % X; int b = l / X;
  if (xyz) { xyz[i * 3 + 0] = (float)x; xyz[i * 3 + 1] = (float)y; xyz[i * 3 + 2] = (float)z; }
  if (bxyz) { bxyz[i * 4 + 0] = b; bxyz[i * 4 + 1] = x; bxyz[i * 4 + 2] = y; bxyz[i * 4 + 3] = z; }
}

extern "C" int coocc_lin_to_coords(const int32_t* lin, int n, int X, int Y, int Z, float* xyz, int64_t* bxyz,
                                   void* stream) {
  COOCC_CHECK_ARG(lin && n >= 0 && X > 0 && Y > 0 && Z > 0, "lin_to_coords: bad args");
  if (n == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_lin_to_coords, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), lin, n, X, Y, Z, xyz, bxyz);
  COOCC_LAUNCH_CHECK("k_lin_to_coords");
  return COOCC_OK;
}

// ------------------------------------------------------------------ CU-partitioned streams
// The FPS chains are latency-bound single workgroups; sharing a CU with convolution waves doubles
// their time (measured).  hipExtStreamCreateWithCUMask gives them private CUs: one stream restricted
// to a few reserved CUs for the FPS kernels, one stream restricted to all the others for the rest.
extern "C" int coocc_device_cu_count(int* n) {
  COOCC_CHECK_ARG(n, "device_cu_count: null");
  int dev = 0;
  COOCC_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  COOCC_HIP(hipGetDeviceProperties(&prop, dev));
  *n = prop.multiProcessorCount;
  return COOCC_OK;
}

extern "C" int coocc_stream_create_cu_mask(const uint32_t* mask_host, int nwords, void** stream_out) {
  COOCC_CHECK_ARG(mask_host && nwords > 0 && stream_out, "stream_create_cu_mask: bad args");
  hipStream_t s = nullptr;
  COOCC_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask_host));
  *stream_out = (void*)s;
  return COOCC_OK;
}

extern "C" int coocc_stream_destroy(void* stream) {
  COOCC_CHECK_ARG(stream, "stream_destroy: null");
  COOCC_HIP(hipStreamDestroy(as_stream(stream)));
  return COOCC_OK;
}

// ------------------------------------------------------------------ device-scope events
// A default HIP event performs a SYSTEM-scope release when it is recorded (hip_runtime_api.h: "cache writeback and invalidation, and
// the performance impact of those actions on the execution of following work") -- what a host or another device needs in order to see
// the stream's writes.  The serving loop records an event after every dense-stage replay and around every search only to order
// streams of the SAME device, where the kernels' own agent-scope release / acquire suffices; with three dense graphs sharing the chip
// those fences cost 24 % of the throughput (profiles/r6_serving_probe_events.txt: 277 -> 343 samples/s without them).  These events
// carry hipEventDisableSystemFence; flags bit 0 keeps the timestamps (hipEventElapsedTime; else hipEventDisableTiming), bit 1 makes
// coocc_event_synchronize sleep instead of spin (hipEventBlockingSync).
// A host that reads results still synchronises the STREAM (or copies device -> host, which is stream-ordered).
extern "C" int coocc_event_create(int flags, void** event_out) {
  COOCC_CHECK_ARG(event_out && (flags & ~3) == 0, "event_create: bad args");
  hipEvent_t e = nullptr;
  COOCC_HIP(hipEventCreateWithFlags(&e, hipEventDisableSystemFence | ((flags & 1) ? 0u : hipEventDisableTiming) |
                                            ((flags & 2) ? hipEventBlockingSync : 0u)));
  *event_out = (void*)e;
  return COOCC_OK;
}
extern "C" int coocc_event_destroy(void* event) {
  COOCC_CHECK_ARG(event, "event_destroy: null");
  COOCC_HIP(hipEventDestroy((hipEvent_t)event));
  return COOCC_OK;
}
extern "C" int coocc_event_record(void* event, void* stream) {
  COOCC_CHECK_ARG(event, "event_record: null");
  COOCC_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
  return COOCC_OK;
}
extern "C" int coocc_stream_wait_event(void* stream, void* event) {
  COOCC_CHECK_ARG(event, "stream_wait_event: null");
  COOCC_HIP(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
  return COOCC_OK;
}
extern "C" int coocc_event_synchronize(void* event) {
  COOCC_CHECK_ARG(event, "event_synchronize: null");
  COOCC_HIP(hipEventSynchronize((hipEvent_t)event));
  return COOCC_OK;
}
// 1: complete, 0: not yet, negative: error
extern "C" int coocc_event_query(void* event) {
  COOCC_CHECK_ARG(event, "event_query: null");
  const hipError_t e = hipEventQuery((hipEvent_t)event);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return coocc_set_error(COOCC_EHIP, "hipEventQuery failed: %s", hipGetErrorString(e));
}
extern "C" int coocc_event_elapsed_ms(void* start, void* stop, float* ms) {
  COOCC_CHECK_ARG(start && stop && ms, "event_elapsed_ms: null");
  COOCC_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return COOCC_OK;
}
