// Synthetic co-runners for the "one load beat of another kernel reads as zero next to a split-f16 GEMM" defect
// (profiles/r6_corunner_defect.txt).  tools/debug/mix_trigger.py showed that k_gemm_h2z with its LDS-DMA image, weight loads, fragment
// ds_reads AND stores compiled out (COOCC_H2_ABLATE=15: accumulator zeroing, barriers, the MFMA stream) still breaks the parked
// half-column OccHead mix, while the fp32-MFMA kernels do not.  This file is that skeleton with every property a parameter: the MFMA
// instruction, the VGPR allocation of the wave, the LDS allocation of the workgroup, the workgroup size.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libmfma_co.so tools/proto/mfma_corunner.hip
//   (loaded by tools/debug/mix_trigger2.py through ctypes)
#include <hip/hip_runtime.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_mfma_f32_32x32x16_f16 (the split-f16 engine's instruction)   1: v_mfma_f32_32x32x8_f16   2: v_mfma_f32_32x32x2_f32
//      3: v_mfma_f32_16x16x32_f16    4: v_mfma_f32_32x32x16_bf16    5: no matrix instruction (v_fma_f32 on the same registers)
template <int KIND, int NV>
__global__ __launch_bounds__(256, 2) void k_co(float* sink, int iters) {
  extern __shared__ char dyn[];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float seed = (float)(threadIdx.x & 7) * 0.125f;
  f16x8 a8, b8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(seed + e); b8[e] = (_Float16)(1.f - seed * e); }
  if (NV == 256) asm volatile("" ::: "v255");
  if (NV == 208) asm volatile("" ::: "v207");
  if (NV == 128) asm volatile("" ::: "v127");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 3; ++rep)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
        if constexpr (KIND == 1) {
          f16x4 a4 = {a8[0], a8[1], a8[2], a8[3]}, b4 = {b8[0], b8[1], b8[2], b8[3]};
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i], 0, 0, 0);
        }
        if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a8[0], (float)b8[0], acc[i], 0, 0, 0);
        if constexpr (KIND == 3) {
          f32x4 c = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
          acc[i][0] = c[0]; acc[i][1] = c[1]; acc[i][2] = c[2]; acc[i][3] = c[3];
        }
        if constexpr (KIND == 4) {
          typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
          bf16x8 ab, bb;
#pragma unroll
          for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(float)a8[e]; bb[e] = (__bf16)(float)b8[e]; }
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
        }
        if constexpr (KIND == 5) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = __builtin_fmaf((float)a8[r & 7], (float)b8[r & 7], acc[i][r]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    if ((it & 3) == 3) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) sink[0] = s + (dyn ? 1.f : 0.f);
}

static float* g_sink = nullptr;

template <int KIND>
static int launch_kind(int nv, int blocks, int iters, int lds, hipStream_t s) {
  switch (nv) {
    case 256: hipLaunchKernelGGL((k_co<KIND, 256>), dim3(blocks), dim3(256), lds, s, g_sink, iters); break;
    case 208: hipLaunchKernelGGL((k_co<KIND, 208>), dim3(blocks), dim3(256), lds, s, g_sink, iters); break;
    case 128: hipLaunchKernelGGL((k_co<KIND, 128>), dim3(blocks), dim3(256), lds, s, g_sink, iters); break;
    default: hipLaunchKernelGGL((k_co<KIND, 0>), dim3(blocks), dim3(256), lds, s, g_sink, iters); break;
  }
  return (int)hipGetLastError();
}

extern "C" int mfma_co_launch(int kind, int nv, int blocks, int iters, int lds_bytes, void* stream) {
  if (!g_sink && hipMalloc(&g_sink, 256) != hipSuccess) return -1;
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case 0: return launch_kind<0>(nv, blocks, iters, lds_bytes, s);
    case 1: return launch_kind<1>(nv, blocks, iters, lds_bytes, s);
    case 2: return launch_kind<2>(nv, blocks, iters, lds_bytes, s);
    case 3: return launch_kind<3>(nv, blocks, iters, lds_bytes, s);
    case 4: return launch_kind<4>(nv, blocks, iters, lds_bytes, s);
    default: return launch_kind<5>(nv, blocks, iters, lds_bytes, s);
  }
}
