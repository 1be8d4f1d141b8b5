// Winograd F(m x m, 3x3), m = 2, 3 or 4, over the (x, y) axes of the 3x3x3 stride-1 convolutions; the z axis
// stays a direct 3-tap convolution.  Multiplies per output: 27*Cin direct, 12*Cin (m = 2), 8.33*Cin (m = 3),
// 6.75*Cin (m = 4).
//
//   V[p]   = B^T d B        input transform  (this file)     d: (m+2)^2 (x,y) patch at (m*tx-1, m*ty-1), fixed z
//   M[p]   = sum_{dz,ci} V[p][.., z+dz-1, ci] U[p][dz][ci][co]   ONE coocc_conv_fwd launch: rows = (m+2)^2 x
//                                                                 (tiles*Z), kx=ky=1, kz=3, weight pack selected
//                                                                 per transform point (wgroup_rows)
//   Y      = A^T M A        output transform + the conv epilogue (scale, bias, residual, ReLU)  (this file)
//
// with U[p] = G g G^T computed once per weight version on the host in fp64 (Toom-Cook matrices; see Wino<6>).
// fp32 error of F(4x4,3x3) at K = 3*512: ~4e-6 of the output scale, inside the 1e-4 parity bound
// (tests/test_gpu_conv.py checks both tile sizes against torch's direct conv).
// Layout: V / M are [(m+2)^2][Gpad][C], p = a*(m+2) + e (a along x), row = ((b*Tx + tx)*Ty + ty)*Z + z,
// Tx = ceil(X/m), Gpad = roundup(rows, lcm(640, Z)).
#include "conv_k.h"
#include "h2_rows.h"

template <int N> struct Wino;
template <> struct Wino<4> {   // F(2,3)
  static constexpr int M = 2;
  template <typename T> static __device__ __forceinline__ void bt(const T* d, T* t) {
    t[0] = d[0] - d[2]; t[1] = d[1] + d[2]; t[2] = d[2] - d[1]; t[3] = d[1] - d[3];
  }
  template <typename T> static __device__ __forceinline__ void at(const T* m, T* y) {
    y[0] = m[0] + m[1] + m[2]; y[1] = m[1] - m[2] - m[3];
  }
  // adjoint of at(): t = A y (the gradient of the output transform, used by the Winograd-domain wgrad)
  template <typename T> static __device__ __forceinline__ void a(const T* y, T* t) {
    t[0] = y[0]; t[1] = y[0] + y[1]; t[2] = y[0] - y[1]; t[3] = -y[1];
  }
};
template <> struct Wino<5> {   // F(3,3) on the points (0, -1, 2, 1/2, inf): rms error 1.7e-6 of the output scale, between
  static constexpr int M = 3;  // F(2,3) (0.75e-6) and F(4,3) (3.5e-6); 8.33*Cin multiplies per output
  template <typename T> static __device__ __forceinline__ void bt(const T* d, T* t) {
    t[0] = d[0] - 1.5f * (d[1] + d[2]) + d[3];
    t[1] = d[1] - 2.5f * d[2] + d[3];
    t[2] = 0.5f * (d[2] - d[1]) + d[3];
    t[3] = -2.f * d[1] - d[2] + d[3];
    t[4] = d[1] - 1.5f * (d[2] + d[3]) + d[4];
  }
  template <typename T> static __device__ __forceinline__ void at(const T* m, T* y) {
    y[0] = m[0] + m[1] + m[2] + m[3];
    y[1] = 2.f * m[2] - m[1] + 0.5f * m[3];
    y[2] = m[1] + 4.f * m[2] + 0.25f * m[3] + m[4];
  }
  template <typename T> static __device__ __forceinline__ void a(const T* y, T* t) {
    t[0] = y[0]; t[1] = y[0] - y[1] + y[2]; t[2] = y[0] + 2.f * y[1] + 4.f * y[2];
    t[3] = y[0] + 0.5f * y[1] + 0.25f * y[2]; t[4] = y[2];
  }
};
template <> struct Wino<6> {   // F(4,3) on the points (0, 1, -1, 1/2, -2, inf): ~2.2x lower fp32 error than the
  static constexpr int M = 4;  // textbook (0, +-1, +-2, inf) set (measured; cf. Barabasz et al., "Error analysis and
                               // improving the accuracy of Winograd convolution")
  template <typename T> static __device__ __forceinline__ void bt(const T* d, T* t) {
    t[0] = d[0] - 1.5f * d[1] - 2.f * d[2] + 1.5f * d[3] + d[4];
    t[1] = -d[1] + 0.5f * d[2] + 2.5f * d[3] + d[4];
    t[2] = d[1] - 2.5f * d[2] + 0.5f * d[3] + d[4];
    t[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    t[4] = 0.5f * (d[1] - d[3]) - d[2] + d[4];
    t[5] = d[1] - 1.5f * d[2] - 2.f * d[3] + 1.5f * d[4] + d[5];
  }
  template <typename T> static __device__ __forceinline__ void at(const T* m, T* y) {
    const T s12 = m[1] + m[2], d12 = m[1] - m[2];
    y[0] = m[0] + s12 + m[3] + m[4];
    y[1] = d12 + 0.5f * m[3] - 2.f * m[4];
    y[2] = s12 + 0.25f * m[3] + 4.f * m[4];
    y[3] = d12 + 0.125f * m[3] - 8.f * m[4] + m[5];
  }
  template <typename T> static __device__ __forceinline__ void a(const T* y, T* t) {
    const T s02 = y[0] + y[2], s13 = y[1] + y[3];
    t[0] = y[0]; t[1] = s02 + s13; t[2] = s02 - s13;
    t[3] = y[0] + 0.5f * y[1] + 0.25f * y[2] + 0.125f * y[3];
    t[4] = y[0] - 2.f * y[1] + 4.f * y[2] - 8.f * y[3];
    t[5] = y[3];
  }
};

template <int N>
__global__ __launch_bounds__(256) void k_wino_in(const float* __restrict__ in, int in_stride, int B, int X, int Y, int Z,
                                                  int C, int Tx, int Ty, size_t gstride, int vstride, float* __restrict__ V) {
  constexpr int MO = Wino<N>::M;
  const int c4 = C >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long rows = (long long)B * Tx * Ty * Z;
  if (i >= rows * c4) return;
  const int c = (int)(i % c4) * 4;
  const long long row = i / c4;
  long long r = row;
  const int z = (int)(r % Z); r /= Z;
  const int ty = (int)(r % Ty); r /= Ty;
  const int tx = (int)(r % Tx); const int b = (int)(r / Tx);
  f32x4 t[N][N];   // t[a][e]: x-transformed, column e of the patch
#pragma unroll
  for (int e = 0; e < N; ++e) {
    f32x4 d[N], q[N];
    const int y = MO * ty - 1 + e;
#pragma unroll
    for (int a = 0; a < N; ++a) {
      const int x = MO * tx - 1 + a;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y)
        v = *(const f32x4*)(in + ((((size_t)b * X + x) * Y + y) * Z + z) * in_stride + c);
      d[a] = v;
    }
    Wino<N>::bt(d, q);
#pragma unroll
    for (int a = 0; a < N; ++a) t[a][e] = q[a];
  }
  float* o = V + (size_t)row * vstride + c;
#pragma unroll
  for (int a = 0; a < N; ++a) {
    f32x4 q[N];
    Wino<N>::bt(t[a], q);
#pragma unroll
    for (int e = 0; e < N; ++e) *(f32x4*)(o + (size_t)(a * N + e) * gstride) = q[e];
  }
}

// The same transform writing V * scale as "H2 rows" (gemm_h2.hip): per row and 32-channel chunk 64 bytes of f16 hi followed by
// 64 bytes of f16 lo = f16((v - hi) * 2^11).  A thread owns 4 channels: 8-byte stores.
template <int N>
__global__ __launch_bounds__(256) void k_wino_in_h2(const float* __restrict__ in, int in_stride, int B, int X, int Y, int Z,
                                                     int C, int Tx, int Ty, size_t gstride_bytes, int vstride, float scale,
                                                     char* __restrict__ V, int* __restrict__ flag, const float* __restrict__ scale_dev) {
  constexpr int MO = Wino<N>::M;
  if (scale_dev) scale *= *scale_dev;
  const int c4 = C >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long rows = (long long)B * Tx * Ty * Z;
  if (i >= rows * c4) return;
  const int c = (int)(i % c4) * 4;
  const long long row = i / c4;
  long long r = row;
  const int z = (int)(r % Z); r /= Z;
  const int ty = (int)(r % Ty); r /= Ty;
  const int tx = (int)(r % Tx); const int b = (int)(r / Tx);
  f32x4 t[N][N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    f32x4 d[N], q[N];
    const int y = MO * ty - 1 + e;
#pragma unroll
    for (int a = 0; a < N; ++a) {
      const int x = MO * tx - 1 + a;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y)
        v = *(const f32x4*)(in + ((((size_t)b * X + x) * Y + y) * Z + z) * in_stride + c);
      d[a] = v;
    }
    Wino<N>::bt(d, q);
#pragma unroll
    for (int a = 0; a < N; ++a) t[a][e] = q[a];
  }
#pragma unroll
  for (int a = 0; a < N; ++a) {
    f32x4 q[N];
    Wino<N>::bt(t[a], q);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      // lanes l, l ^ 1 hold adjacent channel quads of one row (thread i: quad i % (C / 4) of row i / (C / 4), C % 32 == 0):
      // 16-byte stores through store_h2_pair (the 8-byte hi / lo pairs of this kernel's 92 MB were its slow half)
      const f32x4 sv = q[e] * scale;
      store_h2_pair(V + (size_t)(a * N + e) * gstride_bytes, (size_t)row, vstride, c, sv);
      h2_guard(flag, sv);
    }
  }
}

extern "C" int coocc_wino_input_h2_ex(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, void* V,
                                      int vstride, int64_t group_rows, float scale, const float* scale_dev, void* stream) {
  COOCC_CHECK_ARG(in && V && B > 0 && X > 0 && Y > 0 && Z > 0 && C > 0 && C % 32 == 0 && in_stride % 4 == 0, "wino_input_h2: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4, "wino_input_h2: tile must be 2, 3 or 4");
  COOCC_CHECK_ARG(vstride >= C && vstride % 32 == 0 && ((uintptr_t)V & 127) == 0, "wino_input_h2: V row stride / alignment");
  COOCC_CHECK_ARG(scale > 0.f, "wino_input_h2: scale must be positive");
  const int Tx = (X + tile - 1) / tile, Ty = (Y + tile - 1) / tile;
  const long long rows = (long long)B * Tx * Ty * Z;
  COOCC_CHECK_ARG(group_rows >= rows, "wino_input_h2: group_rows smaller than B*ceil(X/tile)*ceil(Y/tile)*Z");
  const dim3 grid(cdiv(rows * (C / 4), 256));
  const size_t gstride = (size_t)group_rows * vstride * 4;
  hipStream_t s = as_stream(stream);
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  if (tile == 2)
    hipLaunchKernelGGL(k_wino_in_h2<4>, grid, dim3(256), 0, s, in, in_stride, B, X, Y, Z, C, Tx, Ty, gstride, vstride, scale, (char*)V, flag, scale_dev);
  else if (tile == 3)
    hipLaunchKernelGGL(k_wino_in_h2<5>, grid, dim3(256), 0, s, in, in_stride, B, X, Y, Z, C, Tx, Ty, gstride, vstride, scale, (char*)V, flag, scale_dev);
  else
    hipLaunchKernelGGL(k_wino_in_h2<6>, grid, dim3(256), 0, s, in, in_stride, B, X, Y, Z, C, Tx, Ty, gstride, vstride, scale, (char*)V, flag, scale_dev);
  COOCC_LAUNCH_CHECK("k_wino_in_h2");
  return COOCC_OK;
}

extern "C" int coocc_wino_input_h2(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, void* V,
                                   int vstride, int64_t group_rows, float scale, void* stream) {
  return coocc_wino_input_h2_ex(in, in_stride, B, X, Y, Z, C, tile, V, vstride, group_rows, scale, nullptr, stream);
}

static int wino_input_impl(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, float* V, int vstride,
                           int64_t group_rows, void* stream) {
  COOCC_CHECK_ARG(in && V && B > 0 && X > 0 && Y > 0 && Z > 0 && C > 0 && C % 4 == 0 && in_stride % 4 == 0, "wino_input: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4, "wino_input: tile must be 2, 3 or 4");
  COOCC_CHECK_ARG(vstride >= C && vstride % 4 == 0 && ((uintptr_t)V & 15) == 0, "wino_input: V row stride / alignment");
  const int Tx = (X + tile - 1) / tile, Ty = (Y + tile - 1) / tile;
  const long long rows = (long long)B * Tx * Ty * Z;
  COOCC_CHECK_ARG(group_rows >= rows, "wino_input: group_rows smaller than B*ceil(X/tile)*ceil(Y/tile)*Z");
  const dim3 grid(cdiv(rows * (C / 4), 256));
  const size_t gstride = (size_t)group_rows * vstride;
  if (tile == 2)
    hipLaunchKernelGGL(k_wino_in<4>, grid, dim3(256), 0, as_stream(stream), in, in_stride, B, X, Y, Z, C, Tx, Ty, gstride, vstride, V);
  else if (tile == 3)
    hipLaunchKernelGGL(k_wino_in<5>, grid, dim3(256), 0, as_stream(stream), in, in_stride, B, X, Y, Z, C, Tx, Ty, gstride, vstride, V);
  else
    hipLaunchKernelGGL(k_wino_in<6>, grid, dim3(256), 0, as_stream(stream), in, in_stride, B, X, Y, Z, C, Tx, Ty, gstride, vstride, V);
  COOCC_LAUNCH_CHECK("k_wino_in");
  return COOCC_OK;
}

extern "C" int coocc_wino_input(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, float* V,
                                int64_t group_rows, void* stream) {
  return wino_input_impl(in, in_stride, B, X, Y, Z, C, tile, V, C, group_rows, stream);
}

// the same transform writing C channels into rows of `vstride` floats (V already offset to the first of them): the input
// channels of one GEMM may come from several channel ranges of the source rows (the dense half of con_enc.0, fuser.py)
extern "C" int coocc_wino_input_strided(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, float* V,
                                        int vstride, int64_t group_rows, void* stream) {
  return wino_input_impl(in, in_stride, B, X, Y, Z, C, tile, V, vstride, group_rows, stream);
}

template <int N>
__global__ __launch_bounds__(256) void k_wino_out(const float* __restrict__ Mb, size_t gstride, int B, int X, int Y, int Z,
                                                   int C, int Tx, int Ty, float* __restrict__ out, int out_stride,
                                                   const float* __restrict__ scale, const float* __restrict__ bias,
                                                   const float* __restrict__ res, int res_stride, int relu,
                                                   void* __restrict__ twin, int* __restrict__ flag) {
  constexpr int MO = Wino<N>::M;
  const int c4 = (C + 3) >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long rows = (long long)B * Tx * Ty * Z;
  if (i >= rows * c4) return;
  const int c = (int)(i % c4) * 4;
  const long long row = i / c4;
  long long r = row;
  const int z = (int)(r % Z); r /= Z;
  const int ty = (int)(r % Ty); r /= Ty;
  const int tx = (int)(r % Tx); const int b = (int)(r / Tx);
  const int nc = min(4, C - c);
  const bool vec = (C & 3) == 0;
  const float* src = Mb + (size_t)row * C + c;
  f32x4 s[MO][N];   // s[a][e]: x-reduced (A^T M), column e
#pragma unroll
  for (int e = 0; e < N; ++e) {
    f32x4 m[N], q[MO];
#pragma unroll
    for (int a = 0; a < N; ++a) {
      const float* p = src + (size_t)(a * N + e) * gstride;
      if (vec) m[a] = *(const f32x4*)p;
      else m[a] = f32x4{p[0], nc > 1 ? p[1] : 0.f, nc > 2 ? p[2] : 0.f, nc > 3 ? p[3] : 0.f};
    }
    Wino<N>::at(m, q);
#pragma unroll
    for (int a = 0; a < MO; ++a) s[a][e] = q[a];
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (e < nc) {
      if (scale) sc[e] = scale[c + e];
      if (bias) bi[e] = bias[c + e];
    }
#pragma unroll
  for (int a = 0; a < MO; ++a) {
    f32x4 y[MO];
    Wino<N>::at(s[a], y);
    const int x = MO * tx + a;
    if (x >= X) continue;
#pragma unroll
    for (int bb = 0; bb < MO; ++bb) {
      const int yy = MO * ty + bb;
      if (yy >= Y) continue;
      const size_t orow = (((size_t)b * X + x) * Y + yy) * Z + z;
      f32x4 v = y[bb] * sc + bi;
      float* o = out + orow * out_stride + c;
      const float* rr = res ? res + orow * res_stride + c : nullptr;
      if (vec && (out_stride & 3) == 0 && (!res || (res_stride & 3) == 0)) {
        if (rr) v = v + *(const f32x4*)rr;
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *(f32x4*)o = v;
        if (twin) { store_h2_pair(twin, orow, C, c, v); h2_guard(flag, v); } // the next split-f16 layer's operand, written by the producer (lane pairs: same row, same branch)
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < nc) {
            float u = v[e] + (rr ? rr[e] : 0.f);
            o[e] = relu ? fmaxf(u, 0.f) : u;
          }
      }
    }
  }
}

extern "C" int coocc_wino_output_ex(const float* Mb, int64_t group_rows, int B, int X, int Y, int Z, int C, int tile, float* out,
                                    int out_stride, const float* scale, const float* bias, const float* res, int res_stride,
                                    int relu, void* out_h2_twin, void* stream) {
  COOCC_CHECK_ARG(Mb && out && B > 0 && X > 0 && Y > 0 && Z > 0 && C > 0, "wino_output: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4, "wino_output: tile must be 2, 3 or 4");
  COOCC_CHECK_ARG(((uintptr_t)out & 15) == 0 && (!res || ((uintptr_t)res & 15) == 0), "wino_output: out/res must be 16-byte aligned");
  COOCC_CHECK_ARG(!out_h2_twin || (C % 32 == 0 && (out_stride & 3) == 0 && (!res || (res_stride & 3) == 0) && ((uintptr_t)out_h2_twin & 15) == 0),
                  "wino_output: the H2 twin needs C % 32 == 0 and 16-byte aligned rows");
  const int Tx = (X + tile - 1) / tile, Ty = (Y + tile - 1) / tile;
  const long long rows = (long long)B * Tx * Ty * Z;
  COOCC_CHECK_ARG(group_rows >= rows, "wino_output: group_rows too small");
  int* flag = nullptr;
  if (out_h2_twin && coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  const dim3 grid(cdiv(rows * ((C + 3) / 4), 256));
  if (tile == 2)
    hipLaunchKernelGGL(k_wino_out<4>, grid, dim3(256), 0, as_stream(stream), Mb, (size_t)group_rows * C, B, X, Y, Z, C, Tx, Ty,
                       out, out_stride, scale, bias, res, res_stride, relu, out_h2_twin, flag);
  else if (tile == 3)
    hipLaunchKernelGGL(k_wino_out<5>, grid, dim3(256), 0, as_stream(stream), Mb, (size_t)group_rows * C, B, X, Y, Z, C, Tx, Ty,
                       out, out_stride, scale, bias, res, res_stride, relu, out_h2_twin, flag);
  else
    hipLaunchKernelGGL(k_wino_out<6>, grid, dim3(256), 0, as_stream(stream), Mb, (size_t)group_rows * C, B, X, Y, Z, C, Tx, Ty,
                       out, out_stride, scale, bias, res, res_stride, relu, out_h2_twin, flag);
  COOCC_LAUNCH_CHECK("k_wino_out");
  return COOCC_OK;
}

extern "C" int coocc_wino_output(const float* Mb, int64_t group_rows, int B, int X, int Y, int Z, int C, int tile, float* out,
                                 int out_stride, const float* scale, const float* bias, const float* res, int res_stride,
                                 int relu, void* stream) {
  return coocc_wino_output_ex(Mb, group_rows, B, X, Y, Z, C, tile, out, out_stride, scale, bias, res, res_stride, relu, nullptr, stream);
}

// ------------------------------------------------------------------ gradient of the output transform (training)
// dM[p = xi*(m+2) + eta] = sum_{i,j} A^T[i][xi] dY[i][j] A^T[j][eta] for every (tile, z) row: the second operand of the
// Winograd-domain weight gradient dU[p][dz] = sum_rows V[p][row + dz - 1]^T dM[p][row].  Same row layout as V / M;
// outputs outside the volume (ragged last tiles) contribute zeros.
template <int N>
__global__ __launch_bounds__(256) void k_wino_gradout(const float* __restrict__ dy, int dy_stride, int B, int X, int Y, int Z,
                                                       int C, int Tx, int Ty, size_t gstride, float* __restrict__ dM) {
  constexpr int MO = Wino<N>::M;
  const int c4 = C >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long rows = (long long)B * Tx * Ty * Z;
  if (i >= rows * c4) return;
  const int c = (int)(i % c4) * 4;
  const long long row = i / c4;
  long long r = row;
  const int z = (int)(r % Z); r /= Z;
  const int ty = (int)(r % Ty); r /= Ty;
  const int tx = (int)(r % Tx); const int b = (int)(r / Tx);
  f32x4 t[N][MO];   // t[xi][j]: x-transformed, column j of the output patch
#pragma unroll
  for (int j = 0; j < MO; ++j) {
    f32x4 d[MO], q[N];
    const int y = MO * ty + j;
#pragma unroll
    for (int a = 0; a < MO; ++a) {
      const int x = MO * tx + a;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (x < X && y < Y) v = *(const f32x4*)(dy + ((((size_t)b * X + x) * Y + y) * Z + z) * dy_stride + c);
      d[a] = v;
    }
    Wino<N>::a(d, q);
#pragma unroll
    for (int xi = 0; xi < N; ++xi) t[xi][j] = q[xi];
  }
  float* o = dM + (size_t)row * C + c;
#pragma unroll
  for (int xi = 0; xi < N; ++xi) {
    f32x4 q[N];
    Wino<N>::a(t[xi], q);
#pragma unroll
    for (int e = 0; e < N; ++e) *(f32x4*)(o + (size_t)(xi * N + e) * gstride) = q[e];
  }
}

extern "C" int coocc_wino_gradout(const float* dy, int dy_stride, int B, int X, int Y, int Z, int C, int tile, float* dM,
                                  int64_t group_rows, void* stream) {
  COOCC_CHECK_ARG(dy && dM && B > 0 && X > 0 && Y > 0 && Z > 0 && C > 0 && C % 4 == 0 && dy_stride % 4 == 0, "wino_gradout: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4, "wino_gradout: tile must be 2, 3 or 4");
  const int Tx = (X + tile - 1) / tile, Ty = (Y + tile - 1) / tile;
  const long long rows = (long long)B * Tx * Ty * Z;
  COOCC_CHECK_ARG(group_rows >= rows, "wino_gradout: group_rows too small");
  const dim3 grid(cdiv(rows * (C / 4), 256));
  if (tile == 2)
    hipLaunchKernelGGL(k_wino_gradout<4>, grid, dim3(256), 0, as_stream(stream), dy, dy_stride, B, X, Y, Z, C, Tx, Ty,
                       (size_t)group_rows * C, dM);
  else if (tile == 3)
    hipLaunchKernelGGL(k_wino_gradout<5>, grid, dim3(256), 0, as_stream(stream), dy, dy_stride, B, X, Y, Z, C, Tx, Ty,
                       (size_t)group_rows * C, dM);
  else
    hipLaunchKernelGGL(k_wino_gradout<6>, grid, dim3(256), 0, as_stream(stream), dy, dy_stride, B, X, Y, Z, C, Tx, Ty,
                       (size_t)group_rows * C, dM);
  COOCC_LAUNCH_CHECK("k_wino_gradout");
  return COOCC_OK;
}

// ------------------------------------------------------------------ transforms that write the weight-gradient operands (KH2)
// csrc/wgrad_h2.hip reads both operands of the Winograd-domain weight gradient "k-major": [row / 8][hi | lo][C][8 rows as f16].
// A workgroup owns one group of 8 consecutive (tile, z) rows x 128 channels (thread = (row, channel quad), the arithmetic of
// k_wino_in_h2 / k_wino_gradout), and after each sixth of the transform (the N points of one a / xi index) the 8 rows trade places
// through a 24 KB LDS tile so that every (point, plane, channel) leaves as ONE 16-byte store -- the fp32 V / dM and the conversion
// pass over them (2 x 92 MB at 128 channels and 100x100x8, the most expensive part of the h2 weight gradient) are never written.
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void kh2_emit(const f32x4 (&q)[N], float scale, char* lds, int rz, int cql, char* out, size_t pstride_bytes,
                                         long long g, int C, int c0, int* flag) {
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const f32x4 v = q[e] * scale;
    h2_guard(flag, v);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      _Float16 hi, lo;
      split_h2(v[ch], hi, lo);
      const int cs = 4 * cql + ch, slot = cs ^ ((cs >> 3) & 7);      // XOR skew: the 32 channel quads of a wave spread over all banks
      *(_Float16*)(lds + ((e * 2 + 0) * 128 + slot) * 16 + rz * 2) = hi;
      *(_Float16*)(lds + ((e * 2 + 1) * 128 + slot) * 16 + rz * 2) = lo;
    }
  }
  __syncthreads();
  for (int u = threadIdx.x; u < N * 256; u += 256) {
    const int e = u >> 8, plane = (u >> 7) & 1, c = (u & 127) ^ (((u & 127) >> 3) & 7);
    if (c0 + c < C)
      *(f16x8w*)(out + (size_t)e * pstride_bytes + (((size_t)g * 2 + plane) * C + c0 + c) * 16) = *(const f16x8w*)(lds + u * 16);
  }
  __syncthreads();
}

template <int N>
__global__ __launch_bounds__(256) void k_wino_in_kh2(const float* __restrict__ in, int in_stride, int B, int X, int Y, int Z, int C,
                                                      int Tx, int Ty, size_t pstride_bytes, float scale,
                                                      const float* __restrict__ scale_dev, char* __restrict__ Vk, int* __restrict__ flag) {
  constexpr int MO = Wino<N>::M;
  __shared__ __attribute__((aligned(16))) char lds[N * 2 * 128 * 16];
  if (scale_dev) scale *= *scale_dev;
  const int rz = threadIdx.x >> 5, cql = threadIdx.x & 31, c0 = blockIdx.y * 128, c = c0 + 4 * cql;
  const long long g = blockIdx.x, row = 8 * g + rz, rows = (long long)B * Tx * Ty * Z;
  const bool live = row < rows && c < C;
  long long r = row;
  const int z = (int)(r % Z); r /= Z;
  const int ty = (int)(r % Ty); r /= Ty;
  const int tx = (int)(r % Tx); const int b = (int)(r / Tx);
  f32x4 t[N][N];
#pragma unroll
  for (int e = 0; e < N; ++e) {
    f32x4 d[N], q[N];
    const int y = MO * ty - 1 + e;
#pragma unroll
    for (int a = 0; a < N; ++a) {
      const int x = MO * tx - 1 + a;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live && (unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y)
        v = *(const f32x4*)(in + ((((size_t)b * X + x) * Y + y) * Z + z) * in_stride + c);
      d[a] = v;
    }
    Wino<N>::bt(d, q);
#pragma unroll
    for (int a = 0; a < N; ++a) t[a][e] = q[a];
  }
#pragma unroll
  for (int a = 0; a < N; ++a) {
    f32x4 q[N];
    Wino<N>::bt(t[a], q);
    kh2_emit<N>(q, scale, lds, rz, cql, Vk + (size_t)(a * N) * pstride_bytes, pstride_bytes, g, C, c0, flag);
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_wino_gradout_kh2(const float* __restrict__ dy, int dy_stride, int B, int X, int Y, int Z, int C,
                                                           int Tx, int Ty, size_t pstride_bytes, float scale,
                                                           const float* __restrict__ scale_dev, char* __restrict__ Mk,
                                                           int* __restrict__ flag) {
  constexpr int MO = Wino<N>::M;
  __shared__ __attribute__((aligned(16))) char lds[N * 2 * 128 * 16];
  if (scale_dev) scale *= *scale_dev;
  const int rz = threadIdx.x >> 5, cql = threadIdx.x & 31, c0 = blockIdx.y * 128, c = c0 + 4 * cql;
  const long long g = blockIdx.x, row = 8 * g + rz, rows = (long long)B * Tx * Ty * Z;
  const bool live = row < rows && c < C;
  long long r = row;
  const int z = (int)(r % Z); r /= Z;
  const int ty = (int)(r % Ty); r /= Ty;
  const int tx = (int)(r % Tx); const int b = (int)(r / Tx);
  f32x4 t[N][MO];
#pragma unroll
  for (int j = 0; j < MO; ++j) {
    f32x4 d[MO], q[N];
    const int y = MO * ty + j;
#pragma unroll
    for (int a = 0; a < MO; ++a) {
      const int x = MO * tx + a;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (live && x < X && y < Y) v = *(const f32x4*)(dy + ((((size_t)b * X + x) * Y + y) * Z + z) * dy_stride + c);
      d[a] = v;
    }
    Wino<N>::a(d, q);
#pragma unroll
    for (int xi = 0; xi < N; ++xi) t[xi][j] = q[xi];
  }
#pragma unroll
  for (int xi = 0; xi < N; ++xi) {
    f32x4 q[N];
    Wino<N>::a(t[xi], q);
    kh2_emit<N>(q, scale, lds, rz, cql, Mk + (size_t)(xi * N) * pstride_bytes, pstride_bytes, g, C, c0, flag);
  }
}

// which = 0: V (coocc_wino_input's transform), 1: dM (coocc_wino_gradout's) -- as KH2 [(tile+2)^2][group_rows / 8][2][C][8 f16]
extern "C" int coocc_wino_operand_kh2(int which, const float* x, int x_stride, int B, int X, int Y, int Z, int C, int tile, void* out_kh2,
                                      int64_t group_rows, float scale, const float* scale_dev, void* stream) {
  COOCC_CHECK_ARG((which == 0 || which == 1) && x && out_kh2 && B > 0 && X > 0 && Y > 0 && Z > 0 && C > 0 && C % 4 == 0 &&
                  x_stride % 4 == 0 && scale > 0.f, "wino_operand_kh2: bad args");
  COOCC_CHECK_ARG(tile >= 2 && tile <= 4 && group_rows % 16 == 0 && ((uintptr_t)out_kh2 & 15) == 0 && ((uintptr_t)x & 15) == 0,
                  "wino_operand_kh2: tile 2..4, group_rows % 16 == 0, 16-byte aligned pointers");
  const int Tx = (X + tile - 1) / tile, Ty = (Y + tile - 1) / tile;
  COOCC_CHECK_ARG(group_rows >= (long long)B * Tx * Ty * Z, "wino_operand_kh2: group_rows too small");
  const dim3 grid((unsigned)(group_rows / 8), (unsigned)((C + 127) / 128));
  const size_t ps = (size_t)group_rows * C * 4;
  hipStream_t s = as_stream(stream);
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
#define COOCC_KH2_LAUNCH(K, NN) hipLaunchKernelGGL(K<NN>, grid, dim3(256), 0, s, x, x_stride, B, X, Y, Z, C, Tx, Ty, ps, scale, scale_dev, \
                                                   (char*)out_kh2, flag)
  if (which == 0) {
    if (tile == 2) COOCC_KH2_LAUNCH(k_wino_in_kh2, 4); else if (tile == 3) COOCC_KH2_LAUNCH(k_wino_in_kh2, 5); else COOCC_KH2_LAUNCH(k_wino_in_kh2, 6);
  } else {
    if (tile == 2) COOCC_KH2_LAUNCH(k_wino_gradout_kh2, 4); else if (tile == 3) COOCC_KH2_LAUNCH(k_wino_gradout_kh2, 5); else COOCC_KH2_LAUNCH(k_wino_gradout_kh2, 6);
  }
#undef COOCC_KH2_LAUNCH
  COOCC_LAUNCH_CHECK("k_wino_operand_kh2");
  return COOCC_OK;
}

// ------------------------------------------------------------------ device-side weight transform (training)
// U[p = xi*(m+2) + eta][dz] = sum_{a,b} G[xi][a] G[eta][b] g[a][b][dz], written straight into the fragment-major packs
// the grouped GEMM reads (one pack per transform point, conv_layout.h).  Training re-packs every step, so the host
// fp64 einsum of the inference path (core.PackedConv.wino_pack) is replaced by this kernel (fp64 accumulation kept).
// dgrad != 0: packs of the transposed convolution dx = conv(dy, W'), W'[c][n][a][b][dz] = w[n][c][2-a][2-b][2-dz].
__constant__ double c_G4[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
__constant__ double c_G5[5][3] = {{1, 0, 0}, {-2. / 9, 2. / 9, -2. / 9}, {1. / 9, 2. / 9, 4. / 9}, {-8. / 9, -4. / 9, -2. / 9}, {0, 0, 1}};
__constant__ double c_G6[6][3] = {{1, 0, 0}, {1. / 3, 1. / 3, 1. / 3}, {-1. / 3, 1. / 3, -1. / 3}, {-16. / 15, -8. / 15, -4. / 15},
                                  {1. / 15, -2. / 15, 4. / 15}, {0, 0, 1}};

template <int N>
__global__ __launch_bounds__(256) void k_wino_weights(const float* __restrict__ w, int Cout, int Cin, int dgrad, int Npad,
                                                       size_t pack_floats, float* __restrict__ packed) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)Cout * Cin * 3) return;
  const int dz = (int)(i % 3);
  const int c = (int)((i / 3) % Cin), n = (int)(i / (3LL * Cin));
  const double (*G)[3] = N == 4 ? c_G4 : (N == 5 ? c_G5 : c_G6);
  const float* g = w + ((size_t)n * Cin + c) * 27;
  double t[N][3];     // t[xi][b] = sum_a G[xi][a] g[a][b]
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    double col[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) col[a] = dgrad ? (double)g[((2 - a) * 3 + (2 - b)) * 3 + (2 - dz)] : (double)g[(a * 3 + b) * 3 + dz];
#pragma unroll
    for (int xi = 0; xi < N; ++xi) t[xi][b] = G[xi][0] * col[0] + G[xi][1] * col[1] + G[xi][2] * col[2];
  }
  // GEMM roles: forward K = Cin (c), N = Cout (n); dgrad K = Cout (n), N = Cin (c)
  const int kk = dgrad ? n : c, nn = dgrad ? c : n;
  const size_t idx = wfrag_index((size_t)(kk / KC) * 3 + dz, Npad >> 7, nn, kk % KC);
#pragma unroll
  for (int xi = 0; xi < N; ++xi)
#pragma unroll
    for (int eta = 0; eta < N; ++eta)
      packed[(size_t)(xi * N + eta) * pack_floats + idx] =
          (float)(G[eta][0] * t[xi][0] + G[eta][1] * t[xi][1] + G[eta][2] * t[xi][2]);
}

extern "C" int64_t coocc_wino_pack_weights_dev(const float* w, int Cout, int Cin, int tile, int dgrad, float* packed,
                                               void* stream) {
  if (Cout <= 0 || Cin <= 0 || tile < 2 || tile > 4) return coocc_set_error(COOCC_EINVAL, "wino_pack_weights_dev: bad args");
  const int N = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  const int kch = (K + KC - 1) / KC, Npad = (N + NPAD_TO - 1) / NPAD_TO * NPAD_TO;
  const size_t pack_floats = (size_t)3 * kch * Npad * KC;
  const int pts = (tile + 2) * (tile + 2);
  const int64_t total = (int64_t)pts * (int64_t)pack_floats;
  if (!packed) return total;
  if (!w) return coocc_set_error(COOCC_EINVAL, "wino_pack_weights_dev: null weights");
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)total, s) != hipSuccess)
    return coocc_set_error(COOCC_EHIP, "wino_pack_weights_dev: memset failed");
  const dim3 grid(cdiv((long long)Cout * Cin * 3, 256));
  if (tile == 2) hipLaunchKernelGGL(k_wino_weights<4>, grid, dim3(256), 0, s, w, Cout, Cin, dgrad, Npad, pack_floats, packed);
  else if (tile == 3) hipLaunchKernelGGL(k_wino_weights<5>, grid, dim3(256), 0, s, w, Cout, Cin, dgrad, Npad, pack_floats, packed);
  else hipLaunchKernelGGL(k_wino_weights<6>, grid, dim3(256), 0, s, w, Cout, Cin, dgrad, Npad, pack_floats, packed);
  if (hipGetLastError() != hipSuccess) return coocc_set_error(COOCC_EHIP, "wino_pack_weights_dev: launch failed");
  return total;
}

// The same device-side weight transform writing the SPLIT-F16 packs of gemm_h2.hip (mfma_dtype 3): U = G g G^T in fp64, split
// hi = f16(U), lo = f16((U - hi) * 2^11), laid out [(tile+2)^2][(K chunk, dz)][Npad/32][2 k16 steps][hi | lo][64 lanes][8 f16]
// (core.PackedConv._h2_layout).  With it the training path -- which re-packs from the live parameter every step -- runs its
// Winograd forward and dgrad GEMMs on the f16 matrix cores like inference does.  K (Cin forward, Cout dgrad) % 32 == 0.
// (A one-thread-per-16-byte-unit form of this kernel -- coalesced stores, 8 x the fp64 transform work per thread -- measured 76 us
// against 32 us per layer for this one-thread-per-weight form with its scattered 2-byte stores: kept as is.)
template <int N>
__global__ __launch_bounds__(256) void k_wino_weights_h2(const float* __restrict__ w, int Cout, int Cin, int dgrad, int Npad,
                                                          size_t pack_halfs, _Float16* __restrict__ packed, int* __restrict__ flag) {
  // Thread -> (k, n, dz) in the PACK's order (round 6): lanes run along e (8 k values = 16 contiguous bytes) and then li (32 columns),
  // so a wave's 2-byte stores fill two 512-byte runs per transform point instead of 64 scattered halfwords (32 -> 21 us per
  // layer launch under rocprofv3; training re-packs 26 layers per step).
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int Kd = dgrad ? Cout : Cin, Nd = dgrad ? Cin : Cout, ntiles = (Nd + 31) >> 5;
  if (i >= (long long)(Kd >> 5) * 3 * ntiles * 1024) return;
  const int e_ = (int)(i & 7), li_ = (int)((i >> 3) & 31), hf_ = (int)((i >> 8) & 1), s_ = (int)((i >> 9) & 1);
  long long r_ = i >> 10;
  const int nt_ = (int)(r_ % ntiles); r_ /= ntiles;
  const int dz = (int)(r_ % 3);
  const int chunk_ = (int)(r_ / 3);
  const int kk_ = chunk_ * 32 + s_ * 16 + hf_ * 8 + e_, nn_ = nt_ * 32 + li_;
  if (nn_ >= Nd) return;
  const int c = dgrad ? nn_ : kk_, n = dgrad ? kk_ : nn_;
  const double (*G)[3] = N == 4 ? c_G4 : (N == 5 ? c_G5 : c_G6);
  const float* g = w + ((size_t)n * Cin + c) * 27;
  double t[N][3];
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    double col[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) col[a] = dgrad ? (double)g[((2 - a) * 3 + (2 - b)) * 3 + (2 - dz)] : (double)g[(a * 3 + b) * 3 + dz];
#pragma unroll
    for (int xi = 0; xi < N; ++xi) t[xi][b] = G[xi][0] * col[0] + G[xi][1] * col[1] + G[xi][2] * col[2];
  }
  const int kk = dgrad ? n : c, nn = dgrad ? c : n;          // GEMM roles: forward K = Cin, N = Cout; dgrad K = Cout, N = Cin
  const int chunk = kk >> 5, k32 = kk & 31, sidx = k32 >> 4, hf = (k32 >> 3) & 1, e = k32 & 7;
  const int nt = nn >> 5, li = nn & 31;
  // halfs: ((((chunk * 3 + dz) * (Npad / 32) + nt) * 2 + s) * 2 + plane) * 512 + (hf * 32 + li) * 8 + e
  const size_t base = ((((size_t)chunk * 3 + dz) * (Npad >> 5) + nt) * 2 + sidx) * 2 * 512 + (size_t)(hf * 32 + li) * 8 + e;
#pragma unroll
  for (int xi = 0; xi < N; ++xi)
#pragma unroll
    for (int eta = 0; eta < N; ++eta) {
      const double u = G[eta][0] * t[xi][0] + G[eta][1] * t[xi][1] + G[eta][2] * t[xi][2];
      const _Float16 hi = (_Float16)u;
      const _Float16 lo = (_Float16)((u - (double)hi) * 2048.0);
      _Float16* o = packed + (size_t)(xi * N + eta) * pack_halfs + base;
      o[0] = hi;
      o[512] = lo;
      if (flag && !(fabs(u) < 32768.0)) *(volatile int*)flag = 1;
    }
}

extern "C" int64_t coocc_wino_pack_weights_h2_dev(const float* w, int Cout, int Cin, int tile, int dgrad, void* packed, void* stream) {
  if (Cout <= 0 || Cin <= 0 || tile < 2 || tile > 4) return coocc_set_error(COOCC_EINVAL, "wino_pack_weights_h2_dev: bad args");
  const int N = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  if (K % 32) return coocc_set_error(COOCC_EINVAL, "wino_pack_weights_h2_dev: the GEMM's K (Cin forward, Cout dgrad) must be a multiple of 32");
  const int Npad = (N + NPAD_TO - 1) / NPAD_TO * NPAD_TO;
  const size_t pack_halfs = (size_t)3 * (K / 32) * Npad * 32 * 2;          // hi + lo halves per (k, n, dz)
  const int pts = (tile + 2) * (tile + 2);
  const int64_t total_floats = (int64_t)pts * (int64_t)(pack_halfs / 2);   // reported in 4-byte units, like the fp32 variant
  if (!packed) return total_floats;
  if (!w) return coocc_set_error(COOCC_EINVAL, "wino_pack_weights_h2_dev: null weights");
  hipStream_t s = as_stream(stream);
  if (Npad != N && hipMemsetAsync(packed, 0, (size_t)total_floats * 4, s) != hipSuccess)
    return coocc_set_error(COOCC_EHIP, "wino_pack_weights_h2_dev: memset failed");
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  const dim3 grid(cdiv((long long)(K / 32) * 3 * ((N + 31) / 32) * 1024, 256));
  if (tile == 2) hipLaunchKernelGGL(k_wino_weights_h2<4>, grid, dim3(256), 0, s, w, Cout, Cin, dgrad, Npad, pack_halfs, (_Float16*)packed, flag);
  else if (tile == 3) hipLaunchKernelGGL(k_wino_weights_h2<5>, grid, dim3(256), 0, s, w, Cout, Cin, dgrad, Npad, pack_halfs, (_Float16*)packed, flag);
  else hipLaunchKernelGGL(k_wino_weights_h2<6>, grid, dim3(256), 0, s, w, Cout, Cin, dgrad, Npad, pack_halfs, (_Float16*)packed, flag);
  if (hipGetLastError() != hipSuccess) return coocc_set_error(COOCC_EHIP, "wino_pack_weights_h2_dev: launch failed");
  return total_floats;
}
