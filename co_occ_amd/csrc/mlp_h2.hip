// R1 in ONE launch: the per-voxel sigma and rgb heads of the render block (P/utils/nerf_mlp.py:14-105 as instantiated at
// coocc_ray.py:111-113 and evaluated at :583-590 -- here once per VOXEL instead of once per ray sample) on the
// split-f16 engine of gemm_h2.hip.
//
//   sigma = w_s1 . relu(W_s0 x + b_s0) + b_s1                       x: 128 channels, hidden width 256
//   rgb   = W_ro . relu(W_r2 relu(W_r1 relu(W_r0 x + b_r0) + b_r1) + b_r2) + b_ro      (net_depth hidden layers, 1..4)
//
// Layer by layer this was six GEMM launches (0.33 ms at 80 000 voxels): every hidden activation [V, 256] went to HBM as H2 rows
// and came back (82 MB each way), and the 1- and 3-column output layers were padded to a 128-column MFMA tile.  Here a workgroup
// owns 64 voxels from x to the finished table row:
//   * x (H2 rows from the producer's twin, 32 KB per tile) is staged once by global_load_lds (source-side bank swizzle, as in
//     k_gemm_h2z); a hidden activation lives in the SAME 64 KB of LDS as H2 rows [chunk][64 rows][hi 64 B | lo 64 B] -- written by
//     the epilogue of the layer that made it (bias, ReLU, split, 8-byte ds_write), read as MFMA fragments by the next layer;
//   * 4 waves, wave w owns hidden columns 64 w .. 64 w + 63 of all 64 rows: 2 x 2 accumulator tiles of 32 x 32 in two sets (hi x hi
//     and the two cross terms): 128 registers; weights stream from L2 straight into registers (the 0.8 MB of H2 packs stay
//     L2-resident), one k16 step ahead;
//   * the MFMA is issued transposed (weights first), so a lane ends a layer holding 4-channel runs of ITS rows: the output layers
//     (1 / 3 columns) are plain fp32 dot products of those registers with the output weights -- on the fp32 hidden values, not on
//     their H2 rounding -- reduced across the half-waves by one DPP-class shuffle and across the four waves through 4 KB of LDS in
//     wave order (deterministic).
// 64 KB of LDS per workgroup: two workgroups per CU.
#include <string.h>
#include <type_traits>

#include "conv_k.h"
#include "h2_rows.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MLP_ROWS 64
#define MLP_W 256
#define MLP_MAXH 4

struct HeadsK {
  const char* x;                 // H2 rows [V][CIN]
  int V;
  const char* ws0; const float* bs0; const float* ws1; const float* bs1;       // sigma: H2 pack of [W][CIN], b [W]; w [W], b [1]
  const char* wr[MLP_MAXH]; const float* br[MLP_MAXH]; int nh;                 // rgb hidden layers: H2 packs, biases
  const float* wro; const float* bro;                                          // rgb output [3][W], [3]
  float* table;                  // [V][4] = (sigma, r, g, b)
  int activate;                  // 1: rgb columns hold sigmoid(logit) (coocc_render_activate_table's expression)
  const char* zrow;
};

__device__ __forceinline__ void mlp_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// acc (hh: hi x hi; xx: the two cross terms, scaled by 2^-11 at the end) = W[256 x 32 NCH] . act[64 rows x 32 NCH] for this wave's 64 columns
template <int NCH>
__device__ __forceinline__ void mlp_layer(const char* __restrict__ w, const char* As, int wave, int lane, f32x16 (&hh)[2][2], f32x16 (&xx)[2][2]) {
  const int li = lane & 31, h = lane >> 5;
  const unsigned swz = (unsigned)((li >> 1) & 7);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hh[t][i][r] = 0.f; xx[t][i][r] = 0.f; }
  // pack: [chunk][Npad/32 = 8][2 k16 steps][hi | lo][64 lanes][8 f16]: 4 KB per (chunk, n tile), 32 KB per chunk
  const char* wp = w + (size_t)(2 * wave) * 4096 + lane * 16;
  f16x8 b[2][2][2];      // [set][n tile][plane]
  auto loadB = [&](int set, int c, int s) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) b[set][t][pl] = *(const f16x8*)(wp + (size_t)c * 32768 + t * 4096 + (s * 2 + pl) * 1024);
  };
  loadB(0, 0, 0);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cur = (c * 2 + s) & 1;
      if (s == 0) loadB(cur ^ 1, c, 1);
      else if (c + 1 < NCH) loadB(cur ^ 1, c + 1, 0);
      f16x8 ahi[2], alo[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned base = (unsigned)c * (MLP_ROWS * 128) + (unsigned)(li + 32 * i) * 128;
        ahi[i] = *(const f16x8*)&As[base + (((unsigned)(2 * s + h) ^ swz) << 4)];
        alo[i] = *(const f16x8*)&As[base + (((unsigned)(4 + 2 * s + h) ^ swz) << 4)];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          hh[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cur][t][0], ahi[i], hh[t][i], 0, 0, 0);
          xx[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cur][t][1], ahi[i], xx[t][i], 0, 0, 0);
          xx[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[cur][t][0], alo[i], xx[t][i], 0, 0, 0);
        }
    }
  }
}

// relu(acc + bias) of this wave's columns: in place in hh (fp32)
__device__ __forceinline__ void mlp_bias_relu(const float* __restrict__ bias, int wave, int lane, f32x16 (&hh)[2][2], const f32x16 (&xx)[2][2]) {
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bi = *(const f32x4*)(bias + 64 * wave + 32 * t + 8 * j + 4 * h);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          hh[t][i][4 * j + e] = fmaxf(hh[t][i][4 * j + e] + xx[t][i][4 * j + e] * (1.f / H2_LO_SCALE) + bi[e], 0.f);
    }
}

// the hidden activation (fp32 in hh) -> H2 rows in LDS: chunk 2 wave + t, row li + 32 i, channels 8 j + 4 h + e
__device__ __forceinline__ void mlp_store_hidden(char* As, int wave, int lane, const f32x16 (&hh)[2][2]) {
  const int li = lane & 31, h = lane >> 5;
  const unsigned swz = (unsigned)((li >> 1) & 7);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned base = (unsigned)(2 * wave + t) * (MLP_ROWS * 128) + (unsigned)(li + 32 * i) * 128 + 8 * h;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 a, b;
          split_h2(hh[t][i][4 * j + e], a, b);
          hi[e] = a; lo[e] = b;
        }
        *(f16x4*)&As[base + (((unsigned)j ^ swz) << 4)] = hi;
        *(f16x4*)&As[base + (((unsigned)(4 + j) ^ swz) << 4)] = lo;
      }
    }
}

// NOUT fp32 dot products of the hidden values (in hh) with rows of wout [NOUT][256]; per wave partials -> red[wave][row][4]
template <int NOUT>
__device__ __forceinline__ void mlp_out_partials(const float* __restrict__ wout, float* red, int wave, int lane, const f32x16 (&hh)[2][2]) {
  const int li = lane & 31, h = lane >> 5;
  float acc[2][NOUT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[i][o] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const f32x4 wv = *(const f32x4*)(wout + o * MLP_W + 64 * wave + 32 * t + 8 * j + 4 * h);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][o] = fmaf(hh[t][i][4 * j + e], wv[e], acc[i][o]);
      }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const float other = __shfl_xor(acc[i][o], 32);            // the other half-wave holds the other 32 channels of the same row
      const float s = h == 0 ? acc[i][o] + other : other + acc[i][o];      // (lower channels) + (upper channels): same order in both halves
      if (h == 0) red[(wave * MLP_ROWS + li + 32 * i) * 4 + o] = s;
    }
}

template <int CIN>
__global__ __launch_bounds__(256, 2) void k_render_heads_h2(HeadsK p) {
  constexpr int KC0 = CIN / 32;
  __shared__ __attribute__((aligned(256))) char As[MLP_ROWS * 128 * (MLP_W / 32)];       // 64 KB
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m0 = blockIdx.x * MLP_ROWS;
  // x -> LDS chunks 0 .. KC0-1 ([chunk][64 rows][128 B], slot c of row r holds slot c ^ ((r >> 1) & 7) of the chunk)
  {
    const int srow = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int u = 0; u < (KC0 * 8) / 4; ++u) {
      const int q = u * 4 + wave;            // (chunk, 8-row group)
      const int c = q >> 3, g = q & 7;
      const int row = g * 8 + srow;
      const int m = m0 + row;
      const char* src = m < p.V ? p.x + (size_t)m * (CIN * 4) + c * 128 + ((slot ^ ((row >> 1) & 7)) << 4) : p.zrow;
      mlp_glds16(src, &As[c * (MLP_ROWS * 128) + g * 1024]);
    }
  }
  __syncthreads();
  f32x16 hh[2][2], xx[2][2];
  float* red = (float*)&As[KC0 * (MLP_ROWS * 128)];           // 4 KB behind x: free until the first hidden activation is stored
  // ---- sigma head
  mlp_layer<KC0>(p.ws0, As, wave, lane, hh, xx);
  mlp_bias_relu(p.bs0, wave, lane, hh, xx);
  mlp_out_partials<1>(p.ws1, red, wave, lane, hh);
  __syncthreads();
  float sigma = 0.f;
  if (tid < MLP_ROWS) {
    sigma = p.bs1[0];
#pragma unroll
    for (int w = 0; w < 4; ++w) sigma += red[(w * MLP_ROWS + tid) * 4];
  }
  if (!p.wr[0]) {              // depth-only branch (coocc_ray.py:436-484): no colour head
    if (tid < MLP_ROWS && m0 + tid < p.V) *(f32x4*)(p.table + (size_t)(m0 + tid) * 4) = f32x4{sigma, 0.f, 0.f, 0.f};
    return;
  }
  // ---- rgb head
  mlp_layer<KC0>(p.wr[0], As, wave, lane, hh, xx);
  mlp_bias_relu(p.br[0], wave, lane, hh, xx);
  for (int l = 1; l < p.nh; ++l) {
    __syncthreads();           // every wave has read the previous activation (and the sigma partials) for the last time
    mlp_store_hidden(As, wave, lane, hh);
    __syncthreads();
    mlp_layer<MLP_W / 32>(p.wr[l], As, wave, lane, hh, xx);
    mlp_bias_relu(p.br[l], wave, lane, hh, xx);
  }
  __syncthreads();
  red = (float*)As;
  mlp_out_partials<3>(p.wro, red, wave, lane, hh);
  __syncthreads();
  if (tid < MLP_ROWS && m0 + tid < p.V) {
    f32x4 o = {sigma, p.bro[0], p.bro[1], p.bro[2]};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      o[1] += red[(w * MLP_ROWS + tid) * 4 + 0];
      o[2] += red[(w * MLP_ROWS + tid) * 4 + 1];
      o[3] += red[(w * MLP_ROWS + tid) * 4 + 2];
    }
    if (p.activate) {
      o[1] = __frcp_rn(1.f + __expf(-o[1]));
      o[2] = __frcp_rn(1.f + __expf(-o[2]));
      o[3] = __frcp_rn(1.f + __expf(-o[3]));
    }
    *(f32x4*)(p.table + (size_t)(m0 + tid) * 4) = o;
  }
}

extern "C" int coocc_render_heads_h2(const void* x_h2, int V, int Cin, int width, const void* ws0_pack, const float* bs0,
                                     const float* ws1, const float* bs1, const void* const* wr_packs_host,
                                     const float* const* br_host, int n_rgb_hidden, const float* wr_out, const float* br_out,
                                     float* table, int activate, void* stream) {
  COOCC_CHECK_ARG(x_h2 && table && ws0_pack && bs0 && ws1 && bs1 && V > 0, "render_heads_h2: null pointer / empty volume");
  COOCC_CHECK_ARG(Cin == 128 && width == MLP_W, "render_heads_h2: input_dim 128 and net_width 256 (the COOCC_Ray heads); other shapes take the layer-by-layer path");
  COOCC_CHECK_ARG(n_rgb_hidden >= 0 && n_rgb_hidden <= MLP_MAXH, "render_heads_h2: 0 .. 4 hidden rgb layers");
  COOCC_CHECK_ARG(n_rgb_hidden == 0 || (wr_packs_host && br_host && wr_out && br_out), "render_heads_h2: rgb head pointers");
  COOCC_CHECK_ARG(((uintptr_t)x_h2 & 15) == 0 && ((uintptr_t)table & 15) == 0 && ((uintptr_t)ws1 & 15) == 0 && (!wr_out || ((uintptr_t)wr_out & 15) == 0),
                  "render_heads_h2: pointers must be 16-byte aligned");
  HeadsK k;
  memset(&k, 0, sizeof(k));
  k.x = (const char*)x_h2; k.V = V;
  k.ws0 = (const char*)ws0_pack; k.bs0 = bs0; k.ws1 = ws1; k.bs1 = bs1;
  k.nh = n_rgb_hidden;
  for (int l = 0; l < n_rgb_hidden; ++l) {
    COOCC_CHECK_ARG(wr_packs_host[l] && br_host[l] && ((uintptr_t)br_host[l] & 15) == 0, "render_heads_h2: null / misaligned rgb layer");
    k.wr[l] = (const char*)wr_packs_host[l];
    k.br[l] = br_host[l];
  }
  COOCC_CHECK_ARG(((uintptr_t)bs0 & 15) == 0, "render_heads_h2: biases must be 16-byte aligned");
  k.wro = wr_out; k.bro = br_out; k.table = table; k.activate = activate;
  const void* z = nullptr;
  int rc = coocc_zero_row(&z);
  if (rc != COOCC_OK) return rc;
  k.zrow = (const char*)z;
  hipLaunchKernelGGL(k_render_heads_h2<128>, dim3(cdiv(V, MLP_ROWS)), dim3(256), 0, as_stream(stream), k);
  COOCC_LAUNCH_CHECK("k_render_heads_h2");
  return COOCC_OK;
}
