// Trilinear / bilinear resampling on channels-last rows (F.interpolate, align_corners=False):
// FPN3D top-down add, OccHead multi-level softmax mix, render-map x16 upsample.
// Source index rule (ATen area_pixel_compute_source_index): src = max(0, scale*(dst+0.5)-0.5),
// scale = in/out (fp32), i0 = floor(src), i1 = i0 + (i0 < in-1), lambda = src - i0.
#include <stdlib.h>

#include "conv_k.h"
#include "h2_rows.h"

struct Lin1 { int i0, i1; float w0, w1; };

__device__ __forceinline__ Lin1 lin_src(int dst, int in, int out) {
  Lin1 r;
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

__device__ __forceinline__ f32x4 tri_sample(const float* __restrict__ vol, int b, int C, int X, int Y, int Z,
                                            const Lin1& lx, const Lin1& ly, const Lin1& lz, int c) {
  auto at = [&](int x, int y, int z) {
    return *(const f32x4*)(vol + ((((size_t)b * X + x) * Y + y) * Z + z) * C + c);
  };
  f32x4 v000 = at(lx.i0, ly.i0, lz.i0), v001 = at(lx.i0, ly.i0, lz.i1);
  f32x4 v010 = at(lx.i0, ly.i1, lz.i0), v011 = at(lx.i0, ly.i1, lz.i1);
  f32x4 v100 = at(lx.i1, ly.i0, lz.i0), v101 = at(lx.i1, ly.i0, lz.i1);
  f32x4 v110 = at(lx.i1, ly.i1, lz.i0), v111 = at(lx.i1, ly.i1, lz.i1);
  return lx.w0 * (ly.w0 * (lz.w0 * v000 + lz.w1 * v001) + ly.w1 * (lz.w0 * v010 + lz.w1 * v011)) +
         lx.w1 * (ly.w0 * (lz.w0 * v100 + lz.w1 * v101) + ly.w1 * (lz.w0 * v110 + lz.w1 * v111));
}

// ---- z-column forms (one thread = the ZF voxels of one (x, y) column, four channels) ----------------------------------------
// The per-voxel kernels below issue eight 16-byte loads per output and coarse level; a column of ZF fine voxels only needs the
// four coarse corner columns (4 x ZC loads for ZF outputs).  For ZF = 2^k ZC the z source rule is exact in fp32 (scale a power
// of two), so the corner INDICES are compile-time constants and the columns stay in registers; the WEIGHTS are still computed by
// lin_src at run time and the sample is the same expression as tri_sample, so the results are the same bits as the per-voxel
// kernels' (tests: test_interp_column_forms_equal_the_per_voxel_kernels).
template <int ZF, int ZC> struct ZSrc {
  static_assert(ZF % ZC == 0 && ((ZF / ZC) & (ZF / ZC - 1)) == 0, "power-of-two z ratio");
  // lin_src(z, ZC, ZF) in integers: src = max(0, (2z + 1 - R) / (2R)), R = ZF / ZC
  static constexpr int R = ZF / ZC;
  static constexpr int i0(int z) { return ZF == ZC ? z : (2 * z + 1 - R < 0 ? 0 : (2 * z + 1 - R) / (2 * R)); }
  static constexpr int i1(int z) { return ZF == ZC ? z : i0(z) + (i0(z) < ZC - 1 ? 1 : 0); }
};

template <int ZC>
__device__ __forceinline__ void load_corner_columns(f32x4 (&q)[4][ZC], const float* __restrict__ vol, int b, int C, int X, int Y,
                                                     const Lin1& lx, const Lin1& ly, int c) {
  const float* base = vol + (size_t)b * X * Y * ZC * C + c;
  const float* p[4] = {base + ((size_t)lx.i0 * Y + ly.i0) * ZC * C, base + ((size_t)lx.i0 * Y + ly.i1) * ZC * C,
                       base + ((size_t)lx.i1 * Y + ly.i0) * ZC * C, base + ((size_t)lx.i1 * Y + ly.i1) * ZC * C};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < ZC; ++k) q[j][k] = *(const f32x4*)(p[j] + (size_t)k * C);
}

// tri_sample's expression on the register columns: q[0] = (x0, y0), q[1] = (x0, y1), q[2] = (x1, y0), q[3] = (x1, y1)
template <int ZF, int ZC, int Z>
__device__ __forceinline__ f32x4 column_sample(const f32x4 (&q)[4][ZC], const Lin1& lx, const Lin1& ly, const Lin1& lz) {
  constexpr int a = ZSrc<ZF, ZC>::i0(Z), e = ZSrc<ZF, ZC>::i1(Z);
  const f32x4 v000 = q[0][a], v001 = q[0][e], v010 = q[1][a], v011 = q[1][e];
  const f32x4 v100 = q[2][a], v101 = q[2][e], v110 = q[3][a], v111 = q[3][e];
  return lx.w0 * (ly.w0 * (lz.w0 * v000 + lz.w1 * v001) + ly.w1 * (lz.w0 * v010 + lz.w1 * v011)) +
         lx.w1 * (ly.w0 * (lz.w0 * v100 + lz.w1 * v101) + ly.w1 * (lz.w0 * v110 + lz.w1 * v111));
}

template <int ZF, int ZC, int Z>
struct UpAddZ {
  static __device__ __forceinline__ void run(const f32x4 (&q)[4][ZC], const f32x4 (&f)[ZF], const Lin1& lx, const Lin1& ly, int Zc,
                                             float* __restrict__ o, size_t row0, int C, int c, void* __restrict__ twin,
                                             int* __restrict__ flag) {
    const Lin1 lz = lin_src(Z, Zc, ZF);
    const f32x4 r = f[Z] + column_sample<ZF, ZC, Z>(q, lx, ly, lz);
    *(f32x4*)(o + (size_t)Z * C) = r;
    if (twin) { store_h2_pair(twin, row0 + Z, C, c, r); h2_guard(flag, r); }
    if constexpr (Z + 1 < ZF) UpAddZ<ZF, ZC, Z + 1>::run(q, f, lx, ly, Zc, o, row0, C, c, twin, flag);
  }
};

template <int ZF, int ZC>
__global__ __launch_bounds__(256) void k_upsample_add_col(const float* __restrict__ coarse, float* __restrict__ fine, int B, int C,
                                                           int Xc, int Yc, int Zc, int Xf, int Yf, void* __restrict__ twin,
                                                           int* __restrict__ flag) {
  const int c4 = C >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * Xf * Yf * c4) return;
  const int c = (int)(i % c4) * 4;
  size_t v = i / c4;
  const size_t row0 = v * ZF;
  const int y = (int)(v % Yf); v /= Yf;
  const int x = (int)(v % Xf);
  const int b = (int)(v / Xf);
  const Lin1 lx = lin_src(x, Xc, Xf), ly = lin_src(y, Yc, Yf);
  f32x4 q[4][ZC], f[ZF];
  float* o = fine + row0 * C + c;
#pragma unroll
  for (int z = 0; z < ZF; ++z) f[z] = *(const f32x4*)(o + (size_t)z * C);
  load_corner_columns<ZC>(q, coarse, b, C, Xc, Yc, lx, ly, c);
  UpAddZ<ZF, ZC, 0>::run(q, f, lx, ly, Zc, o, row0, C, c, twin, flag);
}

// COOCC_INTERP_COLUMN: bit 0 = z-column FPN upsample-add, bit 1 = half-z-column OccHead mix (read per call; 0 = the per-voxel kernels).
// History of bit 1: rounds 5-6 kept the mix form off because it was bit-exact alone and wrong next to other streams' split-f16 GEMMs.
// Cause (round 6, profiles/r6_pk_opsel_probe.txt): its `acc * wn[z][0]` compiled to v_pk_fma_f32 ... op_sel:[0,1,0], the packed-fp32
// form gfx950 mis-reads in lanes 48-63 while another wave of the SIMD runs a 128-bit-operand MFMA.  The kernel is COOCC_SCALAR_FP32
// now (no packed fp32) and bit-stable under every co-runner that broke it (tests/test_gpu_corunner.py).
#define INTERP_COLUMN_DEFAULT 3
static int interp_column_mask() {
  const char* e = getenv("COOCC_INTERP_COLUMN");
  return e && e[0] >= '0' && e[0] <= '7' ? e[0] - '0' : INTERP_COLUMN_DEFAULT;
}

// fpn3d.py:88-92  laterals[i-1] += interpolate(laterals[i], size=prev_shape, trilinear)
__global__ __launch_bounds__(256) void k_upsample_add(const float* __restrict__ coarse, float* __restrict__ fine,
                                                       int B, int C, int Xc, int Yc, int Zc, int Xf, int Yf, int Zf,
                                                       void* __restrict__ twin, int* __restrict__ flag) {
  const int c4 = C >> 2;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)B * Xf * Yf * Zf * c4;
  if (i >= total) return;
  int c = (int)(i % c4) * 4;
  size_t v = i / c4;
  const size_t row = v;
  int z = (int)(v % Zf); v /= Zf;
  int y = (int)(v % Yf); v /= Yf;
  int x = (int)(v % Xf); int b = (int)(v / Xf);
  Lin1 lx = lin_src(x, Xc, Xf), ly = lin_src(y, Yc, Yf), lz = lin_src(z, Zc, Zf);
  f32x4 s = tri_sample(coarse, b, C, Xc, Yc, Zc, lx, ly, lz, c);
  f32x4* o = (f32x4*)(fine + ((((size_t)b * Xf + x) * Yf + y) * Zf + z) * C + c);
  const f32x4 r = *o + s;
  *o = r;
  if (twin) { store_h2_pair(twin, row, C, c, r); h2_guard(flag, r); }      // H2 twin of the updated rows (the fpn_conv that reads them next)
}

extern "C" int coocc_upsample_add_trilinear_ex(const float* coarse, float* fine, int B, int C, int Xc, int Yc,
                                               int Zc, int Xf, int Yf, int Zf, void* fine_h2_twin, void* stream) {
  COOCC_CHECK_ARG(coarse && fine && B > 0 && C > 0 && C % 4 == 0, "upsample_add: bad args (C % 4 == 0)");
  COOCC_CHECK_ARG(!fine_h2_twin || (C % 32 == 0 && ((uintptr_t)fine_h2_twin & 15) == 0), "upsample_add: the H2 twin needs C % 32 == 0");
  int* flag = nullptr;
  if (fine_h2_twin && coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  size_t total = (size_t)B * Xf * Yf * Zf * (C / 4);
  // z-column form for the z ratios of the shipped grids (COOCC_INTERP_COLUMN=0: the per-voxel kernel everywhere; read per call)
  const size_t cols = (size_t)B * Xf * Yf * (C / 4);
#define UPADD_COL(ZF_, ZC_)                                                                                                     \
  if (Zf == ZF_ && Zc == ZC_) {                                                                                                 \
    hipLaunchKernelGGL((k_upsample_add_col<ZF_, ZC_>), dim3(cdiv(cols, 256)), dim3(256), 0, as_stream(stream), coarse, fine, B, \
                       C, Xc, Yc, Zc, Xf, Yf, fine_h2_twin, flag);                                                              \
    COOCC_LAUNCH_CHECK("k_upsample_add_col");                                                                                   \
    return COOCC_OK;                                                                                                            \
  }
  if (interp_column_mask() & 1) {
    UPADD_COL(8, 4) UPADD_COL(4, 2) UPADD_COL(2, 1)
  }
#undef UPADD_COL
  hipLaunchKernelGGL(k_upsample_add, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), coarse, fine, B, C, Xc,
                     Yc, Zc, Xf, Yf, Zf, fine_h2_twin, flag);
  COOCC_LAUNCH_CHECK("k_upsample_add");
  return COOCC_OK;
}

extern "C" int coocc_upsample_add_trilinear(const float* coarse, float* fine, int B, int C, int Xc, int Yc,
                                            int Zc, int Xf, int Yf, int Zf, void* stream) {
  return coocc_upsample_add_trilinear_ex(coarse, fine, B, C, Xc, Yc, Zc, Xf, Yf, Zf, nullptr, stream);
}

// occ_head.py:155-166: softmax over the level logits, then
//   out = sum_l interpolate(occ_l, level-0 size) * w_l      (accumulated in level order)
struct MixLevels { const float* p[4]; int X[4], Y[4], Z[4]; int L; };

__global__ __launch_bounds__(256) void k_occhead_mix(MixLevels lv, const float* __restrict__ wlogit,
                                                      float* __restrict__ out, int B, int C, void* __restrict__ twin,
                                                      int* __restrict__ flag) {
  const int c4 = C >> 2;
  const int X0 = lv.X[0], Y0 = lv.Y[0], Z0 = lv.Z[0];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)B * X0 * Y0 * Z0 * c4;
  if (i >= total) return;
  int c = (int)(i % c4) * 4;
  size_t v = i / c4;
  const size_t row = v;
  int z = (int)(v % Z0); v /= Z0;
  int y = (int)(v % Y0); v /= Y0;
  int x = (int)(v % X0); int b = (int)(v / X0);
  float w[4];
  float mx = -INFINITY;
  for (int l = 0; l < lv.L; ++l) { w[l] = wlogit ? wlogit[row * lv.L + l] : 0.f; mx = fmaxf(mx, w[l]); }
  float sum = 0.f;
  for (int l = 0; l < lv.L; ++l) { w[l] = expf(w[l] - mx); sum += w[l]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < lv.L; ++l) {
    f32x4 s;
    if (lv.X[l] == X0 && lv.Y[l] == Y0 && lv.Z[l] == Z0) {
      // a level already on the output grid (level 0): the trilinear sample IS the voxel (weights 1, 0 -- the eight-tap
      // expression below returns exactly this value: 1 * v + 0 * v), one 16-byte load instead of eight
      s = *(const f32x4*)(lv.p[l] + row * C + c);
    } else {
      Lin1 lx = lin_src(x, lv.X[l], X0), ly = lin_src(y, lv.Y[l], Y0), lz = lin_src(z, lv.Z[l], Z0);
      s = tri_sample(lv.p[l], b, C, lv.X[l], lv.Y[l], lv.Z[l], lx, ly, lz, c);
    }
    acc = acc + s * (w[l] / sum);
  }
  *(f32x4*)(out + row * C + c) = acc;
  if (twin) { store_h2_pair(twin, row, C, c, acc); h2_guard(flag, acc); }     // H2 twin for occ_pred_conv's first 1x1x1 layer
}

// z-column form of the mix: four levels, level 0 on the output grid, levels 1-3 coarser in z by powers of two.  A thread takes HALF a
// column (Z0 / 2 voxels, four channels): the whole column needs ~320 registers (one wave per SIMD, measured slower than the
// per-voxel kernel), a half ~120; the halves are grid.y, so the plane ranges below are compile-time per wave.
template <int ZL, int PLO, int NP>
__device__ __forceinline__ void load_corner_planes(f32x4 (&q)[4][NP], const float* __restrict__ vol, int b, int C, int X, int Y,
                                                    const Lin1& lx, const Lin1& ly, int c) {
  const float* base = vol + (size_t)b * X * Y * ZL * C + c + (size_t)PLO * C;
  const float* p[4] = {base + ((size_t)lx.i0 * Y + ly.i0) * ZL * C, base + ((size_t)lx.i0 * Y + ly.i1) * ZL * C,
                       base + ((size_t)lx.i1 * Y + ly.i0) * ZL * C, base + ((size_t)lx.i1 * Y + ly.i1) * ZL * C};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < NP; ++k) q[j][k] = *(const f32x4*)(p[j] + (size_t)k * C);
}

template <int Z0, int ZL, int LV, int ZB, int ZN, int PLO, int NP, int Z>
struct MixZ {
  static __device__ __forceinline__ void run(const f32x4 (&q)[4][NP], const Lin1& lx, const Lin1& ly, int Zl, const float (&wn)[ZN][4],
                                             f32x4 (&acc)[ZN]) {
    const Lin1 lz = lin_src(Z, Zl, Z0);
    constexpr int a = ZSrc<Z0, ZL>::i0(Z) - PLO, e = ZSrc<Z0, ZL>::i1(Z) - PLO;
    static_assert(a >= 0 && e < NP, "plane range");
    const f32x4 v000 = q[0][a], v001 = q[0][e], v010 = q[1][a], v011 = q[1][e];
    const f32x4 v100 = q[2][a], v101 = q[2][e], v110 = q[3][a], v111 = q[3][e];
    const f32x4 s = lx.w0 * (ly.w0 * (lz.w0 * v000 + lz.w1 * v001) + ly.w1 * (lz.w0 * v010 + lz.w1 * v011)) +
                    lx.w1 * (ly.w0 * (lz.w0 * v100 + lz.w1 * v101) + ly.w1 * (lz.w0 * v110 + lz.w1 * v111));
    acc[Z - ZB] = acc[Z - ZB] + s * wn[Z - ZB][LV];
    if constexpr (Z + 1 < ZB + ZN) MixZ<Z0, ZL, LV, ZB, ZN, PLO, NP, Z + 1>::run(q, lx, ly, Zl, wn, acc);
  }
};

template <int Z0, int ZL, int LV, int ZB, int ZN>
__device__ __forceinline__ void mix_level(const MixLevels& lv, int b, int C, int x, int y, int X0, int Y0, int c,
                                          const float (&wn)[ZN][4], f32x4 (&acc)[ZN]) {
  constexpr int PLO = ZSrc<Z0, ZL>::i0(ZB), NP = ZSrc<Z0, ZL>::i1(ZB + ZN - 1) - PLO + 1;
  const Lin1 lx = lin_src(x, lv.X[LV], X0), ly = lin_src(y, lv.Y[LV], Y0);
  f32x4 q[4][NP];
  load_corner_planes<ZL, PLO, NP>(q, lv.p[LV], b, C, lv.X[LV], lv.Y[LV], lx, ly, c);
  MixZ<Z0, ZL, LV, ZB, ZN, PLO, NP, ZB>::run(q, lx, ly, lv.Z[LV], wn, acc);
}

template <int Z0, int Z1, int Z2, int Z3, int H>
__device__ __forceinline__ void mix_half_column(const MixLevels& lv, const float* __restrict__ wlogit, float* __restrict__ out, int b,
                                                int C, int x, int y, int X0, int Y0, int c, size_t col_row0,
                                                void* __restrict__ twin, int* __restrict__ flag) {
  constexpr int ZN = Z0 / 2, ZB = H * ZN;
  const size_t row0 = col_row0 + ZB;
  float wn[ZN][4];                       // w_l / sum per voxel: the per-voxel kernel's softmax weights, same operations
  f32x4 acc[ZN], wl[ZN];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int z = 0; z < ZN; ++z) {
    wl[z] = wlogit ? *(const f32x4*)(wlogit + (row0 + z) * 4) : zero;
    acc[z] = *(const f32x4*)(lv.p[0] + (row0 + z) * C + c);              // level 0: the voxel itself
  }
#pragma unroll
  for (int z = 0; z < ZN; ++z) {
    float w[4] = {wl[z][0], wl[z][1], wl[z][2], wl[z][3]}, mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < 4; ++l) mx = fmaxf(mx, w[l]);
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < 4; ++l) { w[l] = expf(w[l] - mx); sum += w[l]; }
#pragma unroll
    for (int l = 0; l < 4; ++l) wn[z][l] = w[l] / sum;
    acc[z] = zero + acc[z] * wn[z][0];
  }
  mix_level<Z0, Z1, 1, ZB, ZN>(lv, b, C, x, y, X0, Y0, c, wn, acc);
  mix_level<Z0, Z2, 2, ZB, ZN>(lv, b, C, x, y, X0, Y0, c, wn, acc);
  mix_level<Z0, Z3, 3, ZB, ZN>(lv, b, C, x, y, X0, Y0, c, wn, acc);
#pragma unroll
  for (int z = 0; z < ZN; ++z) {
    *(f32x4*)(out + (row0 + z) * C + c) = acc[z];
    if (twin) { store_h2(twin, row0 + z, C, c, acc[z]); h2_guard(flag, acc[z]); }
  }
}

template <int Z0, int Z1, int Z2, int Z3>
__global__ COOCC_SCALAR_FP32 __launch_bounds__(256) void k_occhead_mix_col(MixLevels lv, const float* __restrict__ wlogit, float* __restrict__ out,
                                                          int B, int C, void* __restrict__ twin, int* __restrict__ flag) {
  static_assert(Z0 % 2 == 0, "half columns");
  const int c4 = C >> 2;
  const int X0 = lv.X[0], Y0 = lv.Y[0];
  const unsigned i = blockIdx.x * 256u + threadIdx.x;                     // host: columns * c4 < 2^31
  if (i >= (unsigned)(B * X0 * Y0) * (unsigned)c4) return;
  const int c = (int)(i % (unsigned)c4) * 4;
  unsigned v = i / (unsigned)c4;
  const size_t col_row0 = (size_t)v * Z0;
  const int y = (int)(v % (unsigned)Y0); v /= (unsigned)Y0;
  const int x = (int)(v % (unsigned)X0);
  const int b = (int)(v / (unsigned)X0);
  if (blockIdx.y == 0) mix_half_column<Z0, Z1, Z2, Z3, 0>(lv, wlogit, out, b, C, x, y, X0, Y0, c, col_row0, twin, flag);
  else mix_half_column<Z0, Z1, Z2, Z3, 1>(lv, wlogit, out, b, C, x, y, X0, Y0, c, col_row0, twin, flag);
}

extern "C" int coocc_occhead_mix_ex(const float* const* levels_host, const int* dims_host, int L, const float* wlogit,
                                    float* out, int B, int C, void* out_h2_twin, void* stream) {
  COOCC_CHECK_ARG(levels_host && dims_host && out && L >= 1 && L <= 4 && C % 4 == 0, "occhead_mix: bad args");
  COOCC_CHECK_ARG(!out_h2_twin || (C % 32 == 0 && ((uintptr_t)out_h2_twin & 15) == 0), "occhead_mix: the H2 twin needs C % 32 == 0");
  int* flag = nullptr;
  if (out_h2_twin && coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  MixLevels lv;
  lv.L = L;
  for (int l = 0; l < 4; ++l) {
    lv.p[l] = l < L ? levels_host[l] : nullptr;
    lv.X[l] = l < L ? dims_host[l * 3 + 0] : 1;
    lv.Y[l] = l < L ? dims_host[l * 3 + 1] : 1;
    lv.Z[l] = l < L ? dims_host[l * 3 + 2] : 1;
  }
  size_t total = (size_t)B * lv.X[0] * lv.Y[0] * lv.Z[0] * (C / 4);
  const size_t cols = (size_t)B * lv.X[0] * lv.Y[0] * (C / 4);
#define MIX_COL(Z0_, Z1_, Z2_, Z3_)                                                                                              \
  if (lv.Z[0] == Z0_ && lv.Z[1] == Z1_ && lv.Z[2] == Z2_ && lv.Z[3] == Z3_) {                                                   \
    hipLaunchKernelGGL((k_occhead_mix_col<Z0_, Z1_, Z2_, Z3_>), dim3(cdiv(cols, 256), 2), dim3(256), 0, as_stream(stream), lv,  \
                       wlogit, out, B, C, out_h2_twin, flag);                                                                   \
    COOCC_LAUNCH_CHECK("k_occhead_mix_col");                                                                                    \
    return COOCC_OK;                                                                                                            \
  }
  if (L == 4 && cols < (1u << 31) && (interp_column_mask() & 2)) {
    MIX_COL(8, 4, 2, 1)
  }
#undef MIX_COL
  hipLaunchKernelGGL(k_occhead_mix, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), lv, wlogit, out, B, C, out_h2_twin, flag);
  COOCC_LAUNCH_CHECK("k_occhead_mix");
  return COOCC_OK;
}

extern "C" int coocc_occhead_mix(const float* const* levels_host, const int* dims_host, int L, const float* wlogit,
                                 float* out, int B, int C, void* stream) {
  return coocc_occhead_mix_ex(levels_host, dims_host, L, wlogit, out, B, C, nullptr, stream);
}

// coocc_ray.py:617-622: F.interpolate(scale_factor=16, mode='bilinear') of depth_map and rgb_map.
// maps [N,H,W,4] (r,g,b,depth) -> rgbs [N,sH,sW,3], depths [N,sH,sW]; 4 output pixels per lane.  This kernel is the
// write-bandwidth-bound part of rendering, so (1) an aligned quad of output pixels that shares its source columns
// (always, when the scale is a multiple of 8) fetches its 4 taps once instead of 16 times, and (2) the 12 rgb floats
// of a lane go through LDS so that every store instruction of a wave covers 1 KB of contiguous memory instead of
// 16-byte pieces at a 48-byte stride.
__global__ __launch_bounds__(256) void k_upsample_maps(const float* __restrict__ maps, int N, int H, int W, int scale,
                                                        float* __restrict__ rgbs, float* __restrict__ depths) {
  __shared__ f32x4 stage[4][192];
  const int oH = H * scale, oW = W * scale;
  const int q = oW >> 2;  // quads per row
  const size_t total = (size_t)N * oH * q;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float o[16];
  if (i < total) {
    int xq = (int)(i % q);
    size_t r = i / q;
    int oy = (int)(r % oH);
    int n = (int)(r / oH);
    Lin1 ly = lin_src(oy, H, oH);
    const f32x4* m = (const f32x4*)maps + (size_t)n * H * W;
    Lin1 lx0 = lin_src(xq * 4, W, oW), lx3 = lin_src(xq * 4 + 3, W, oW);
    if (lx0.i0 == lx3.i0 && lx0.i1 == lx3.i1) {
      f32x4 v00 = m[ly.i0 * W + lx0.i0], v01 = m[ly.i0 * W + lx0.i1];
      f32x4 v10 = m[ly.i1 * W + lx0.i0], v11 = m[ly.i1 * W + lx0.i1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Lin1 lx = lin_src(xq * 4 + k, W, oW);
        f32x4 v = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
        o[k * 3 + 0] = v[0]; o[k * 3 + 1] = v[1]; o[k * 3 + 2] = v[2];
        o[12 + k] = v[3];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        Lin1 lx = lin_src(xq * 4 + k, W, oW);
        f32x4 v00 = m[ly.i0 * W + lx.i0], v01 = m[ly.i0 * W + lx.i1];
        f32x4 v10 = m[ly.i1 * W + lx.i0], v11 = m[ly.i1 * W + lx.i1];
        f32x4 v = ly.w0 * (lx.w0 * v00 + lx.w1 * v01) + ly.w1 * (lx.w0 * v10 + lx.w1 * v11);
        o[k * 3 + 0] = v[0]; o[k * 3 + 1] = v[1]; o[k * 3 + 2] = v[2];
        o[12 + k] = v[3];
      }
    }
    *(f32x4*)(depths + i * 4) = f32x4{o[12], o[13], o[14], o[15]};
    if (rgbs == nullptr) return;             // depth-only render branch (coocc_ray.py:436-484): no colour maps
    stage[wave][lane * 3 + 0] = f32x4{o[0], o[1], o[2], o[3]};
    stage[wave][lane * 3 + 1] = f32x4{o[4], o[5], o[6], o[7]};
    stage[wave][lane * 3 + 2] = f32x4{o[8], o[9], o[10], o[11]};
  }
  if (rgbs == nullptr) return;
  // wave-private staging: the lanes of one wave execute in lock step, the LDS round trip only needs the waitcnt
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const size_t wave_i0 = (size_t)blockIdx.x * blockDim.x + wave * 64;  // first quad of this wave
  f32x4* pr = (f32x4*)rgbs + wave_i0 * 3;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int g = j * 64 + lane;
    if (wave_i0 * 3 + g < total * 3) pr[g] = stage[wave][g];
  }
}

extern "C" int coocc_upsample_maps(const float* maps, int N, int H, int W, int scale, float* rgbs, float* depths,
                                   void* stream) {
  COOCC_CHECK_ARG(maps && depths && N > 0 && H > 0 && W > 0 && scale >= 1, "upsample_maps: bad args");
  COOCC_CHECK_ARG((W * scale) % 4 == 0, "upsample_maps: output width must be a multiple of 4");
  size_t total = (size_t)N * H * scale * (W * scale / 4);
  hipLaunchKernelGGL(k_upsample_maps, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), maps, N, H, W, scale,
                     rgbs, depths);
  COOCC_LAUNCH_CHECK("k_upsample_maps");
  return COOCC_OK;
}
