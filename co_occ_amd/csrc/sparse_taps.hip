// Sparse half of a dense 3x3x3 convolution whose input channels are non-zero on a SMALL set of voxels only.
//
// con_enc.0 of BiFuser_N reads cat[img, pts, fused_img, fused_pts] (bifuser_n.py:169-172): the pts and fused_img slots are
// non-zero exactly on the LiDAR voxels (5-30 % of the grid, 12 % in the bench).  Convolution is linear in its input channels,
// so those 2C channels are convolved in scatter form: one dense GEMM over the Np occupied rows produces, for every occupied
// input voxel u and tap t, the contribution P[u][t][:] = W_t . in[u] that belongs to output voxel v = u - (t - 1)
// (conv3d.hip, row-table mode, N = 27 * Cout), and this kernel sums, per output voxel and in tap order (deterministic), the
// contributions of its occupied neighbours:
//     S[v][n] = scale[n] * sum_t P[ ord(v + t - 1) ][t][n]        ord = voxel -> row of P, or -1
// S is handed to the Winograd output transform of the dense half as its residual (added before the ReLU), so the layer's
// result is relu(bn(dense + sparse)) up to fp32 rounding of the reassociated sum.  34 GFLOP instead of the 126 GFLOP those
// channels cost inside the F(2x2) GEMM at 12 % occupancy.
#include "common.h"

int coocc_h2_flag_ptr(int** out);      // gemm_h2.hip: the process-wide host-mapped flag block (word 0 range guard, word 1 fault)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_sparse_tap_sum(const float* __restrict__ P, const int32_t* __restrict__ map, int X, int Y,
                                                         int Z, int nvox, int Cout, const float* __restrict__ scale,
                                                         float* __restrict__ S, int s_stride, int p_rows, int* fault) {
  const int v = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (v >= nvox) return;
  const int z = v % Z, y = (v / Z) % Y, x = (v / (Z * Y)) % X, b = v / (Z * Y * X);
  const size_t prow = (size_t)27 * Cout;
  for (int c0 = 0; c0 < Cout; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool on = c < Cout;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int t = 0;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dz = -1; dz <= 1; ++dz, ++t) {
          const int ux = x + dx, uy = y + dy, uz = z + dz;
          if ((unsigned)ux >= (unsigned)X || (unsigned)uy >= (unsigned)Y || (unsigned)uz >= (unsigned)Z) continue;
          const int ord = map[((b * X + ux) * Y + uy) * Z + uz];        // wave-uniform
          if (ord < 0) continue;
          if (p_rows > 0 && ord >= p_rows) {                      // an ordinal past the rows P was sized for is never valid: raise the
            if (lane == 0 && fault) *(volatile int*)fault = COOCC_FAULT_SPARSE_ORDINAL;   // sticky fault word (the host's next
            continue;                                             // coocc_device_fault read raises) instead of a GPU memory fault
          }                                                       // (found the memset-node problem of coocc_voxel_index_map_dev, round 5)
          if (on) acc = acc + *(const f32x4*)(P + (size_t)ord * prow + (size_t)t * Cout + c);
        }
    if (on) {
      if (scale) {
        const f32x4 sc = *(const f32x4*)(scale + c);
        acc = acc * sc;
      }
      *(f32x4*)(S + (size_t)v * s_stride + c) = acc;
    }
  }
}

extern "C" int coocc_sparse_tap_sum(const float* P, const int32_t* map, int B, int X, int Y, int Z, int Cout, const float* scale,
                                    float* S, int s_stride, int p_rows, void* stream) {
  COOCC_CHECK_ARG(P && map && S && B > 0 && X > 0 && Y > 0 && Z > 0 && Cout > 0 && Cout % 4 == 0 && s_stride % 4 == 0 &&
                      s_stride >= Cout, "sparse_tap_sum: bad args");
  COOCC_CHECK_ARG(((uintptr_t)P & 15) == 0 && ((uintptr_t)S & 15) == 0 && (!scale || ((uintptr_t)scale & 15) == 0), "sparse_tap_sum: alignment");
  const long long nvox = (long long)B * X * Y * Z;
  COOCC_CHECK_ARG(nvox < (1ll << 31), "sparse_tap_sum: grid too large");
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  hipLaunchKernelGGL(k_sparse_tap_sum, dim3(cdiv(nvox, 4)), dim3(256), 0, as_stream(stream), P, map, X, Y, Z, (int)nvox, Cout, scale, S,
                     s_stride, p_rows, flag + 1);
  COOCC_LAUNCH_CHECK("k_sparse_tap_sum");
  return COOCC_OK;
}
