// Shared helpers for libcoocc_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/coocc_hip.h"

#define COOCC_WAVE 64

int coocc_set_error(int code, const char* fmt, ...);

#define COOCC_CHECK_ARG(cond, ...)                                  \
  do {                                                              \
    if (!(cond)) return coocc_set_error(COOCC_EINVAL, __VA_ARGS__); \
  } while (0)

#define COOCC_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess)                                                                \
      return coocc_set_error(COOCC_EHIP, "%s failed: %s (%s:%d)", #call,                  \
                             hipGetErrorString(e__), __FILE__, __LINE__);                 \
  } while (0)

#define COOCC_LAUNCH_CHECK(name)                                                          \
  do {                                                                                    \
    hipError_t e__ = hipGetLastError();                                                   \
    if (e__ != hipSuccess)                                                                \
      return coocc_set_error(COOCC_EHIP, "launch of %s failed: %s", name,                 \
                             hipGetErrorString(e__));                                     \
  } while (0)

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// squared distance with the contraction nvcc applies to the reference expression
// (furthest_point_sample_cuda.cu:65-66, ball_query_cuda.cu:41-42); see oracle/c/coocc_oracle.c.
__device__ __forceinline__ float sqdist3(float x1, float y1, float z1, float x2, float y2,
                                         float z2) {
  float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}
