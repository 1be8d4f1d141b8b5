// Probe for the hazard suspected behind the "wrong only next to split-f16 GEMMs" kernels:
//
//     v_mad_u64_u32 v[a:b], s[N:N+1], ...        ; VALU, carry-out mask -> s[N:N+1]  (dead value)
//     ... a few instructions ...
//     s_and_saveexec_b64 s[N:N+1], s[M:M+1]      ; SALU, saved EXEC -> the SAME pair
//     ...
//     s_or_b64 exec, exec, s[N:N+1]              ; restore
//
// is what the compiler emitted around the failing store of k_occhead_mix_col<8,4,2,1,0>.  If the VALU's SGPR write can land AFTER the
// SALU's (write-after-write on an SGPR pair between the two pipes), the restore ORs in the carry-out (0) instead of the saved EXEC and
// the lanes the guard mask excludes stay off: the following stores are skipped in those lanes.  The high SGPR (lanes 32-63) is
// written last by a wave64 VALU op, which is where every wrong row of the mix kernel sat.
//
// victim<DIST>: per iteration, the sequence above through inline asm with DIST quarter-rate integer multiplies between the VALU
// write and the saveexec (the compiled kernel had two v_mul_lo_u32 + a v_cmp + a v_add3), guard mask = the EVEN lanes, then every lane
// stores its iteration counter.  A lane that is still off after the restore leaves a hole.  Run alone and next to an MFMA co-runner
// (f16 32x32x16, the split-f16 GEMM's instruction; or fp32 32x32x2) on a second stream.
//
//   hipcc --offload-arch=gfx950 -O3 -o sgpr_waw_probe tools/proto/sgpr_waw_probe.hip && ./sgpr_waw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                         \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int DIST, int SAME>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ out, int iters, unsigned mul) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  unsigned a = t * 2654435761u + 1u, acc = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned long long r64;
    unsigned m0 = a, m1 = a ^ 0x9e3779b9u;
    // s[40:41]: VALU carry-out, then the saved EXEC (SAME = 1) or a different pair s[44:45] for the saved EXEC (SAME = 0: control)
    if (SAME) {
      asm volatile(
          "s_mov_b32 s42, 0x55555555\n\t"
          "s_mov_b32 s43, 0x55555555\n\t"
          "v_mad_u64_u32 %0, s[40:41], %2, %3, 0\n\t"
          ".rept %c4\n\t"
          "v_mul_lo_u32 %1, %1, %3\n\t"
          ".endr\n\t"
          "s_and_saveexec_b64 s[40:41], s[42:43]\n\t"
          "v_add_u32 %1, %1, 1\n\t"
          "s_or_b64 exec, exec, s[40:41]\n\t"
          : "=&v"(r64), "+v"(m1)
          : "v"(m0), "s"(mul), "n"(DIST)
          : "s40", "s41", "s42", "s43", "vcc");
    } else {
      asm volatile(
          "s_mov_b32 s42, 0x55555555\n\t"
          "s_mov_b32 s43, 0x55555555\n\t"
          "v_mad_u64_u32 %0, s[40:41], %2, %3, 0\n\t"
          ".rept %c4\n\t"
          "v_mul_lo_u32 %1, %1, %3\n\t"
          ".endr\n\t"
          "s_and_saveexec_b64 s[44:45], s[42:43]\n\t"
          "v_add_u32 %1, %1, 1\n\t"
          "s_or_b64 exec, exec, s[44:45]\n\t"
          : "=&v"(r64), "+v"(m1)
          : "v"(m0), "s"(mul), "n"(DIST)
          : "s40", "s41", "s42", "s43", "s44", "s45", "vcc");
    }
    acc += (unsigned)r64 + (unsigned)(r64 >> 32) + m1;
    a = a * 1664525u + 1013904223u;
    out[(size_t)it * gridDim.x * 256u + t] = (unsigned)it + 1u;        // every lane: a hole = the lane was off
  }
  if (acc == 0x12345678u) out[0] = acc;                                 // keep the arithmetic alive
}

// co-runners: dense MFMA loops, enough waves to sit on every SIMD next to the victim's
template <int F16>
__global__ __launch_bounds__(256) void corunner(float* sink, int iters) {
  f32x16 c0 = {}, c1 = {}, c2 = {};
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(1.0f + e * 0.01f); }
  for (int i = 0; i < iters; ++i) {
    if (F16) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[1], (float)b[1], c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[2], (float)b[2], c2, 0, 0, 0);
    }
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e];
  if (s == 123.456f) sink[0] = s;
}

template <int DIST, int SAME>
static long long run(const char* name, int co, hipStream_t s0, hipStream_t s1, unsigned* out, float* sink) {
  const int blocks = 1024, iters = 64, reps = 20;
  const size_t n = (size_t)iters * blocks * 256;
  std::vector<unsigned> h(n);
  long long holes = 0, high = 0, odd = 0;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipMemsetAsync(out, 0, n * 4, s0));
    CHECK(hipStreamSynchronize(s0));
    if (co == 1) hipLaunchKernelGGL((corunner<1>), dim3(2048), dim3(256), 0, s1, sink, 20000);
    if (co == 2) hipLaunchKernelGGL((corunner<0>), dim3(2048), dim3(256), 0, s1, sink, 20000);
    hipLaunchKernelGGL((victim<DIST, SAME>), dim3(blocks), dim3(256), 0, s0, out, iters, 3u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      const unsigned want = (unsigned)(i / ((size_t)blocks * 256)) + 1u;
      if (h[i] != want) {
        ++holes;
        const unsigned lane = (unsigned)(i % 64);
        high += lane >= 32;
        odd += lane & 1;
      }
    }
  }
  printf("%-44s co-runner %-12s: %lld holes of %zu stores (in lanes 32-63: %lld, in odd lanes: %lld)\n", name,
         co == 0 ? "none" : co == 1 ? "mfma f16" : "mfma f32", holes, n * reps, high, odd);
  fflush(stdout);
  return holes;
}

int main() {
  hipStream_t s0, s1;
  CHECK(hipStreamCreate(&s0));
  CHECK(hipStreamCreate(&s1));
  unsigned* out;
  float* sink;
  CHECK(hipMalloc(&out, (size_t)64 * 1024 * 256 * 4));
  CHECK(hipMalloc(&sink, 64));
  long long total = 0;
  for (int co = 0; co < 3; ++co) {
    total += run<0, 1>("same pair, 0 multiplies between", co, s0, s1, out, sink);
    total += run<2, 1>("same pair, 2 multiplies between", co, s0, s1, out, sink);
    total += run<6, 1>("same pair, 6 multiplies between", co, s0, s1, out, sink);
    total += run<0, 0>("control: saved EXEC in another pair", co, s0, s1, out, sink);
  }
  printf("total holes: %lld\n", total);
  return 0;
}
