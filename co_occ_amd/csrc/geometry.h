// get_geometry's per-point chain (ViewTransformerLSSBEVDepth.py:117-150) from the 39 per-camera constants: shared by
// k_get_geometry / the fused lift-splat key kernel (pool.hip) and the render kernel that evaluates it in place (render.hip).
#pragma once
#include "common.h"

// the chain for ONE sample of camera block `m` (39 floats) at frustum coordinates (xw, yh, dd) = (xs[w], ys[h], ds[d])
__device__ __forceinline__ void geometry_sample(const float* __restrict__ m, float xw, float yh, float dd, float& gx, float& gy, float& gz) {
  float px = xw - m[9], py = yh - m[10], pz = dd - m[11];
  float qx = m[0] * px + m[1] * py + m[2] * pz;
  float qy = m[3] * px + m[4] * py + m[5] * pz;
  float qz = m[6] * px + m[7] * py + m[8] * pz;
  qx *= qz; qy *= qz;
  qx -= m[33]; qy -= m[34]; qz -= m[35];
  float ex = m[12] * qx + m[13] * qy + m[14] * qz + m[21];
  float ey = m[15] * qx + m[16] * qy + m[17] * qz + m[22];
  float ez = m[18] * qx + m[19] * qy + m[20] * qz + m[23];
  gx = m[24] * ex + m[25] * ey + m[26] * ez + m[36];
  gy = m[27] * ex + m[28] * ey + m[29] * ez + m[37];
  gz = m[30] * ex + m[31] * ey + m[32] * ez + m[38];
}

// ... for the flat point index i = ((cam * D + d) * fH + h) * fW + w.  32-bit index arithmetic whenever the point count allows it:
// a 64-bit division is ~100 emulated instructions on the GPU, and the three of them were most of the key kernel of the fused
// lift-splat and of the in-kernel-geometry ray kernel (which now passes (cam, d, h, w) directly and divides nothing).
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
