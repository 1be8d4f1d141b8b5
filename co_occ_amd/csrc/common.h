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

// Kernel attribute: no packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) in this kernel.
// gfx950 hazard found in round 6 (profiles/r6_pk_opsel_probe.txt, tools/proto/pk_opsel_probe.hip): a packed-fp32 instruction whose op_sel
// routes the HIGH dword of src1 into the LOW result (what hipcc emits for `vec * other[1]`) reads that operand as 0.0 in lanes 48-63
// now and then WHILE a wave of another kernel on the same SIMD runs a 128-bit-operand MFMA (v_mfma_f32_32x32x16_f16 ...): bit-exact
// alone, wrong next to this package's own split-f16 GEMMs.  Every kernel whose ISA held that form carries this attribute, and
// tools/isa_lint.py (tests/test_isa_lint.py) checks the built library for it.
#define COOCC_SCALAR_FP32 __attribute__((target("no-packed-fp32-ops")))

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// squared distance with the contraction nvcc applies to the reference expression
// (furthest_point_sample_cuda.cu:65-66, ball_query_cuda.cu:41-42); see oracle/c/coocc_oracle.c.
__device__ __forceinline__ float sqdist3(float x1, float y1, float z1, float x2, float y2,
                                         float z2) {
  float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}
