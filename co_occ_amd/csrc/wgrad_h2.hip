// Weight gradients on the split-f16 engine (round 4): dW[c][n] = sum over voxels m of X[m][c] * dY[m][n] with every fp32 operand as
// two f16 halves (hi + lo 2^-11) and three v_mfma_f32_32x32x16_f16 per product (hi hi, hi lo, lo hi), fp32 accumulate -- the
// arithmetic of csrc/gemm_h2.hip with the reduction index on the VOXEL axis.  The reference trains these layers through cuDNN's
// fp32 backward (resnet3d.py:34-64, fpn3d.py:70-106, bifuser_n.py:23-30); csrc/conv_bwd.hip's k_wgrad is the fp32-MFMA form.
//
// The f16 MFMA wants 8 consecutive k (= voxels) per lane for one row (= channel): channels-last rows [voxel][channel] are the wrong
// way round, so both operands are first written "k-major" (KH2):  [voxel / 8][hi | lo][channel][8 voxels as f16] -- 16 bytes per
// (group of 8 voxels, plane, channel); a lane's MFMA fragment is ONE 16-byte read, a workgroup's operand tile (128 channels x 16
// voxels x 2 planes) is four contiguous 2 KB segments.  k_rows_to_kh2 makes it from fp32 rows (scale by a power of two: static, and
// for the gradient operand the device-chosen one of coocc_conv_epilogue_bwd_ex).
//
// k_wgrad_h2<NDZ>: workgroup = 8 waves, tile 128 (Cin) x 128 (Cout) for one slab (a slice of the voxel axis); a step = 16 voxels:
// both operand tiles (8 KB each) go global -> registers -> LDS (double buffered, one barrier per step), every wave reads its one
// A and two B fragments per plane from LDS and owns 32 x 64 outputs.  NDZ = 3 is the Winograd-domain form (conv_bwd.hip:
// dU[p][dz][c][n] = sum_row V[p][row + dz - 1][c] dM[p][row][n], z taps direct): rows are (tile, z) with z fastest and Z in
// {2, 4, 8}, so a group of 8 voxels holds whole z columns and the shifted operand V[row + dz - 1] is the SAME 16-byte fragment
// moved by one f16 lane with the column ends masked off -- A and B are read once for the three taps (3 x 2 x 3 = 18 MFMAs per
// wave and step for 6 LDS reads).  Partial sums go to slabs in the layout of k_wgrad / k_wino_wgrad_reduce, which are reused.
#include "common.h"
#include "conv_k.h"
#include "h2_rows.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// fp32 rows [rows][stride] -> KH2 [rows_pad / 8][2][C][8] (rows past `rows`: zero).  One thread per (group, channel quad).
__global__ __launch_bounds__(256) void k_rows_to_kh2(const float* __restrict__ x, int stride, long long rows, long long rows_pad, int C,
                                                      float scale, const float* __restrict__ scale_dev, char* __restrict__ out,
                                                      int* __restrict__ flag) {
  const int c4 = C >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (rows_pad >> 3) * c4) return;
  const int cq = (int)(i % c4);
  const long long g = i / c4;
  if (scale_dev) scale *= *scale_dev;
  f32x4 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long long r = 8 * g + j;
    v[j] = r < rows ? *(const f32x4*)(x + r * stride + 4 * cq) * scale : f32x4{0.f, 0.f, 0.f, 0.f};
    h2_guard(flag, v[j]);
  }
  char* o = out + ((size_t)g * 2 * C + 4 * cq) * 16;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      _Float16 a, b;
      split_h2(v[j][e], a, b);
      hi[j] = a; lo[j] = b;
    }
    *(f16x8*)(o + e * 16) = hi;
    *(f16x8*)(o + ((size_t)C + e) * 16) = lo;
  }
}

extern "C" int coocc_rows_to_kh2(const float* x, int stride, int64_t rows, int64_t rows_pad, int C, float scale, const float* scale_dev,
                                 void* out_kh2, void* stream) {
  COOCC_CHECK_ARG(x && out_kh2 && rows >= 0 && rows_pad >= rows && rows_pad % 16 == 0 && C > 0 && C % 4 == 0 && stride % 4 == 0 &&
                  stride >= C && scale > 0.f, "rows_to_kh2: bad args (rows_pad % 16, C % 4, stride % 4)");
  COOCC_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)out_kh2 & 15) == 0, "rows_to_kh2: pointers must be 16-byte aligned");
  if (rows_pad == 0) return COOCC_OK;
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  hipLaunchKernelGGL(k_rows_to_kh2, dim3(cdiv((rows_pad / 8) * (C / 4), 256)), dim3(256), 0, as_stream(stream), x, stride,
                     (long long)rows, (long long)rows_pad, C, scale, scale_dev, (char*)out_kh2, flag);
  COOCC_LAUNCH_CHECK("k_rows_to_kh2");
  return COOCC_OK;
}

struct WgradH2 {
  const char* A;            // KH2 of the input rows  [groups][2][Cin][16 B]
  const char* B;            // KH2 of the gradient rows [groups][2][Cout][16 B]
  int Cin, Cout, ntiles;
  long long gpp;            // groups per transform point (NDZ = 1: all groups); even
  int S, gps;               // slabs per point, groups per slab (even; the last slab of a point may be shorter or empty)
  int Z;                    // NDZ = 3: z extent (2 | 4 | 8): the columns inside a group of 8 rows
  float alpha;              // 1 / (static operand scales)
  const float* alpha_dev;   // ... times this device word (1 / the gradient operand's device-chosen scale), or NULL
  float* slabs;             // [slab][NDZ][Cin][Cout]
};

// element j of the fragment <- element j - 1 (dz = 0) / j + 1 (dz = 2), zero where that crosses a z column (mask per dword)
__device__ __forceinline__ f16x8 kh2_shift(f16x8 v, int dz, u32x4 m) {
  const u32x4 d = __builtin_bit_cast(u32x4, v);
  u32x4 o;
  if (dz == 0) {
    o[0] = d[0] << 16;
    o[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
    o[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
    o[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
  } else {
    o[0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
    o[1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
    o[2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
    o[3] = d[3] >> 16;
  }
  o = o & m;
  return __builtin_bit_cast(f16x8, o);
}

template <int NDZ>
__global__ __launch_bounds__(512) void k_wgrad_h2(WgradH2 p) {
  __shared__ __attribute__((aligned(16))) char As[2][8192];        // [buffer][group of the step (2)][plane (2)][128 channels][16 B]
  __shared__ __attribute__((aligned(16))) char Bs[2][8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wc = wave >> 1, wn = wave & 1;                           // this wave: channels 32 wc .. +31 of the tile, outputs 64 wn .. +63
  const int ct = blockIdx.x / p.ntiles, nt = blockIdx.x - ct * p.ntiles;
  const int c0 = ct * 128, n0 = nt * 128;
  const int pt = blockIdx.y / p.S, sl_ = blockIdx.y - pt * p.S;
  const long long g0 = (long long)pt * p.gpp + (long long)sl_ * p.gps;
  const int steps = (int)(max(0ll, min((long long)p.gps, p.gpp - (long long)sl_ * p.gps)) >> 1);
  // staging: thread -> (segment = group * 2 + plane, channel) of both tiles
  const int seg = tid >> 7, sc = tid & 127;
  const bool a_ok = c0 + sc < p.Cin, b_ok = n0 + sc < p.Cout;
  const char* ga = p.A + (((size_t)(g0 + (seg >> 1)) * 2 + (seg & 1)) * p.Cin + c0 + sc) * 16;
  const char* gb = p.B + (((size_t)(g0 + (seg >> 1)) * 2 + (seg & 1)) * p.Cout + n0 + sc) * 16;
  const size_t astep = (size_t)4 * p.Cin * 16, bstep = (size_t)4 * p.Cout * 16;       // two groups x two planes
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 ra = zero4, rb = zero4;
  auto gload = [&](int s) {
    ra = a_ok ? *(const u32x4*)(ga + (size_t)s * astep) : zero4;
    rb = b_ok ? *(const u32x4*)(gb + (size_t)s * bstep) : zero4;
  };
  auto lstore = [&](int buf) {
    *(u32x4*)&As[buf][tid * 16] = ra;
    *(u32x4*)&Bs[buf][tid * 16] = rb;
  };
  // masks of the shifted fragments (per dword = two voxels): the voxels at a z-column end take nothing from the neighbour column
  u32x4 m0 = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, m2 = m0;
  if (NDZ == 3) {
    if (p.Z == 8) { m0[0] = 0xFFFF0000u; m2[3] = 0x0000FFFFu; }
    else if (p.Z == 4) { m0[0] = m0[2] = 0xFFFF0000u; m2[1] = m2[3] = 0x0000FFFFu; }
    else { m0 = u32x4{0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u}; m2 = u32x4{0x0000FFFFu, 0x0000FFFFu, 0x0000FFFFu, 0x0000FFFFu}; }
  }
  f32x16 hh[NDZ][2], xx[NDZ][2];
#pragma unroll
  for (int d = 0; d < NDZ; ++d)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hh[d][j][r] = 0.f; xx[d][j][r] = 0.f; }

  if (steps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  const unsigned aoff = (unsigned)((h * 2) * 128 + wc * 32 + li) * 16u;          // + 2048 for the lo plane
  const unsigned boff = (unsigned)((h * 2) * 128 + wn * 64 + li) * 16u;          // + 512 for the second 32 outputs
  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) gload(s + 1);
    const f16x8 ahi = *(const f16x8*)&As[buf][aoff], alo = *(const f16x8*)&As[buf][aoff + 2048];
    f16x8 bhi[2], blo[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bhi[j] = *(const f16x8*)&Bs[buf][boff + j * 512];
      blo[j] = *(const f16x8*)&Bs[buf][boff + j * 512 + 2048];
    }
#pragma unroll
    for (int d = 0; d < NDZ; ++d) {
      const f16x8 ah = (NDZ == 1 || d == 1) ? ahi : kh2_shift(ahi, d, d == 0 ? m0 : m2);
      const f16x8 al = (NDZ == 1 || d == 1) ? alo : kh2_shift(alo, d, d == 0 ? m0 : m2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        hh[d][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bhi[j], hh[d][j], 0, 0, 0);
        xx[d][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, blo[j], xx[d][j], 0, 0, 0);
        xx[d][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bhi[j], xx[d][j], 0, 0, 0);
      }
    }
    if (s + 1 < steps) lstore(buf ^ 1);
    __syncthreads();
  }
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha, lo = alpha * (1.f / H2_LO_SCALE);
#pragma unroll
  for (int d = 0; d < NDZ; ++d) {
    float* sl = p.slabs + ((size_t)blockIdx.y * NDZ + d) * (size_t)p.Cin * p.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + wn * 64 + j * 32 + li;
        if (c < p.Cin && n < p.Cout) sl[(size_t)c * p.Cout + n] = hh[d][j][r] * alpha + xx[d][j][r] * lo;
      }
  }
}

// reduce kernels of conv_bwd.hip (same slab layouts)
int coocc_wgrad_reduce_launch(const float* slabs, int nslices, int Cin, int Cout, int taps, float* dw, int accumulate, hipStream_t s);
int coocc_wino_wgrad_reduce_launch(const float* slabs, int S, int Cin, int Cout, int tile, float* dw, int accumulate, hipStream_t s);

// dw[Cout][Cin][1] (+)= sum_m x[m][c] dy[m][n] from the two KH2 operands (rows_pad rows each): the 1x1x1 / Linear layers
extern "C" int coocc_conv_wgrad_h2(const void* x_kh2, const void* dy_kh2, int64_t rows_pad, int Cin, int Cout, float alpha,
                                   const float* alpha_dev, float* dw, int accumulate, float* ws, int64_t ws_floats, void* stream) {
  COOCC_CHECK_ARG(x_kh2 && dy_kh2 && dw && ws && rows_pad > 0 && rows_pad % 16 == 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 &&
                  Cout % 4 == 0 && alpha > 0.f, "conv_wgrad_h2: bad args");
  const int ctiles = (Cin + 127) / 128, ntiles = (Cout + 127) / 128;
  const int64_t per = (int64_t)Cin * Cout;
  const long long groups = rows_pad / 8;
  // slices of the voxel axis: enough workgroups for the chip, >= 32 steps each, within the workspace
  static const int wg_target = getenv("COOCC_WGRAD_H2_WGS") ? atoi(getenv("COOCC_WGRAD_H2_WGS")) : 256;
  long long S = 1;
  while (S < 256 && (long long)ctiles * ntiles * S < wg_target && groups / (2 * S) >= 32 && (2 * S) * per <= ws_floats) S *= 2;
  COOCC_CHECK_ARG(S * per <= ws_floats, "conv_wgrad_h2: workspace smaller than one weight slab");
  long long gps = (groups + S - 1) / S;
  gps += gps & 1;
  const int nslices = (int)S;
  WgradH2 p;
  p.A = (const char*)x_kh2; p.B = (const char*)dy_kh2; p.Cin = Cin; p.Cout = Cout; p.ntiles = ntiles; p.gpp = groups;
  p.S = (int)S; p.gps = (int)gps; p.Z = 1; p.alpha = alpha; p.alpha_dev = alpha_dev; p.slabs = ws;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(k_wgrad_h2<1>, dim3((unsigned)(ctiles * ntiles), (unsigned)nslices), dim3(512), 0, s, p);
  COOCC_LAUNCH_CHECK("k_wgrad_h2");
  return coocc_wgrad_reduce_launch(ws, nslices, Cin, Cout, 1, dw, accumulate, s);
}

// Winograd-domain weight gradient (coocc_wino_wgrad) from KH2 operands: V_kh2 / dM_kh2 = (tile+2)^2 points x group_rows rows each
// (group_rows % 16 == 0, rows (tile, z) with z fastest, Z in {2, 4, 8}); dw[Cout][Cin][3][3][3] (+)= G^T dU G.
extern "C" int coocc_wino_wgrad_h2(const void* V_kh2, const void* dM_kh2, int64_t group_rows, int Z, int Cin, int Cout, int tile,
                                   float alpha, const float* alpha_dev, float* dw, int accumulate, float* ws, int64_t ws_floats,
                                   void* stream) {
  COOCC_CHECK_ARG(V_kh2 && dM_kh2 && dw && ws && group_rows > 0 && Cin > 0 && Cout > 0 && Cin % 4 == 0 && Cout % 4 == 0 && alpha > 0.f,
                  "wino_wgrad_h2: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4 && group_rows % 16 == 0 && (Z == 2 || Z == 4 || Z == 8),
                  "wino_wgrad_h2: tile 2..4, group_rows % 16 == 0, Z in {2, 4, 8} (whole z columns inside a group of 8 rows)");
  const int pts = (tile + 2) * (tile + 2);
  const int ctiles = (Cin + 127) / 128, ntiles = (Cout + 127) / 128;
  const int64_t per = (int64_t)3 * Cin * Cout;
  const long long gpp = group_rows / 8;                       // groups per transform point
  static const int wg_target = getenv("COOCC_WGRAD_H2_WGS") ? atoi(getenv("COOCC_WGRAD_H2_WGS")) : 256;
  int S = 1;                                                  // slices per point (the last one of a point may be shorter)
  while (S < 64 && (long long)ctiles * ntiles * pts * S < wg_target && gpp / (2 * S) >= 16 && (int64_t)pts * 2 * S * per <= ws_floats) S *= 2;
  COOCC_CHECK_ARG((int64_t)pts * S * per <= ws_floats, "wino_wgrad_h2: workspace smaller than (tile+2)^2 weight slabs");
  long long gps = (gpp + S - 1) / S;
  gps += gps & 1;
  WgradH2 p;
  p.A = (const char*)V_kh2; p.B = (const char*)dM_kh2; p.Cin = Cin; p.Cout = Cout; p.ntiles = ntiles; p.gpp = gpp;
  p.S = S; p.gps = (int)gps; p.Z = Z; p.alpha = alpha; p.alpha_dev = alpha_dev; p.slabs = ws;
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(k_wgrad_h2<3>, dim3((unsigned)(ctiles * ntiles), (unsigned)(pts * S)), dim3(512), 0, s, p);
  COOCC_LAUNCH_CHECK("k_wgrad_h2");
  return coocc_wino_wgrad_reduce_launch(ws, S, Cin, Cout, tile, dw, accumulate, s);
}
