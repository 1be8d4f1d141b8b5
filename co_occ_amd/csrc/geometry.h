// get_geometry's per-point chain (ViewTransformerLSSBEVDepth.py:117-150) from the 39 per-camera constants: shared by
// k_get_geometry / the fused lift-splat key kernel (pool.hip) and the render kernel that evaluates it in place (render.hip).
#pragma once
#include "common.h"

// the chain for ONE sample of camera block `m` (39 floats) at frustum coordinates (xw, yh, dd) = (xs[w], ys[h], ds[d]).
// Every operation is an EXPLICIT intrinsic (no compiler contraction): the function is inlined into kernels of very different shape
// (k_get_geometry, the pooling key kernel, the two ray kernels, where loop-invariant parts get hoisted), and a sample position that
// differs by one ulp between them can land in the neighbouring voxel.  Round 4 found exactly that: with plain `a*b + c*d + e*f`
// expressions hipcc fused a different product in the LDS-free ray kernel than in k_get_geometry, and one of five scenes rendered
// differently from the geometry-tensor path.  The 3x3 products are accumulated as the matmuls they restate
// (ViewTransformerLSSBEVDepth.py:131-147): first product rounded, the other two fused in order; translations added afterwards.
__device__ __forceinline__ float geo_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return __fmaf_rn(a2, b2, __fmaf_rn(a1, b1, __fmul_rn(a0, b0)));
}
__device__ __forceinline__ void geometry_sample(const float* __restrict__ m, float xw, float yh, float dd, float& gx, float& gy, float& gz) {
  const float px = __fsub_rn(xw, m[9]), py = __fsub_rn(yh, m[10]), pz = __fsub_rn(dd, m[11]);
  float qx = geo_dot3(m[0], px, m[1], py, m[2], pz);
  float qy = geo_dot3(m[3], px, m[4], py, m[5], pz);
  float qz = geo_dot3(m[6], px, m[7], py, m[8], pz);
  qx = __fmul_rn(qx, qz); qy = __fmul_rn(qy, qz);
  qx = __fsub_rn(qx, m[33]); qy = __fsub_rn(qy, m[34]); qz = __fsub_rn(qz, m[35]);
  const float ex = __fadd_rn(geo_dot3(m[12], qx, m[13], qy, m[14], qz), m[21]);
  const float ey = __fadd_rn(geo_dot3(m[15], qx, m[16], qy, m[17], qz), m[22]);
  const float ez = __fadd_rn(geo_dot3(m[18], qx, m[19], qy, m[20], qz), m[23]);
  gx = __fadd_rn(geo_dot3(m[24], ex, m[25], ey, m[26], ez), m[36]);
  gy = __fadd_rn(geo_dot3(m[27], ex, m[28], ey, m[29], ez), m[37]);
  gz = __fadd_rn(geo_dot3(m[30], ex, m[31], ey, m[32], ez), m[38]);
}

// ... for the flat point index i = ((cam * D + d) * fH + h) * fW + w.  32-bit index arithmetic whenever the point count allows it
// (64-bit divisions are emulated); the ray kernels pass (cam, d, h, w) directly and divide nothing.
__device__ __forceinline__ void geometry_point(const float* __restrict__ mats, const float* __restrict__ xs,
                                               const float* __restrict__ ys, const float* __restrict__ ds, size_t i, int D,
                                               int fH, int fW, float& gx, float& gy, float& gz) {
  int w, h, d, cam;
  if (i < 0xFFFFFFFFull) {
    unsigned r = (unsigned)i;
    w = (int)(r % (unsigned)fW); r /= (unsigned)fW;
    h = (int)(r % (unsigned)fH); r /= (unsigned)fH;
    d = (int)(r % (unsigned)D); cam = (int)(r / (unsigned)D);
  } else {
    w = (int)(i % fW); size_t r = i / fW;
    h = (int)(r % fH); r /= fH;
    d = (int)(r % D); cam = (int)(r / D);
  }
  geometry_sample(mats + (size_t)cam * COOCC_CAM_FLOATS, xs[w], ys[h], ds[d], gx, gy, gz);
}
