// get_geometry's per-point chain (ViewTransformerLSSBEVDepth.py:117-150) from the 39 per-camera constants: shared by
// k_get_geometry / the fused lift-splat key kernel (pool.hip) and the render kernel that evaluates it in place (render.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ void geometry_point(const float* __restrict__ mats, const float* __restrict__ xs,
                                               const float* __restrict__ ys, const float* __restrict__ ds, size_t i, int D,
                                               int fH, int fW, float& gx, float& gy, float& gz) {
  int w = (int)(i % fW); size_t r = i / fW;
  int h = (int)(r % fH); r /= fH;
  int d = (int)(r % D); int cam = (int)(r / D);
  const float* m = mats + (size_t)cam * COOCC_CAM_FLOATS;
  float px = xs[w] - m[9], py = ys[h] - m[10], pz = ds[d] - m[11];
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

