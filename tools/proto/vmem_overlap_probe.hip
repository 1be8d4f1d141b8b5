// Probe, second candidate for the "wrong only next to split-f16 GEMMs" kernels (profiles/r6_corunner_defect.txt).
// tools/debug/mix_values.py showed that the differing rows of k_occhead_mix_col<8,4,2,1,0> hold finite values in ONE component (0 or 2) of
// the slot kept in v[0:3], lanes 48-63 only, and that the H2 twin made from the same registers differs too: the REGISTER is wrong,
// not the store.  In the compiled kernel v[0:3] is the destination of a global_load_dwordx4 issued while two earlier loads whose
// 64-bit ADDRESSES live in v[0:1] and v[2:3] are still in flight:
//
//     v_lshl_add_u64 v[0:1], ...            global_load_dwordx4 v[60:63],   v[0:1], off
//     v_lshl_add_u64 v[2:3], ...            global_load_dwordx4 v[100:103], v[2:3], off
//     v_lshl_add_u64 v[4:5], ...            global_load_dwordx4 v[0:3],     v[4:5], off     <- dest overlaps both address pairs
//
// and the wrong components are exactly the LOW dwords of those pairs (v0, v2).  This probe issues that sequence through inline asm
// (OVERLAP = 1) or the same loads with a disjoint destination (OVERLAP = 0), waits with s_waitcnt vmcnt(0), and checks all three
// results -- alone, next to an LDS-DMA co-runner (global_load_lds_dwordx4 + ds_read + f16 MFMA: the split-f16 GEMM's staging), next to
// a plain-load MFMA co-runner.
//
//   hipcc --offload-arch=gfx950 -O3 -o vmem_overlap_probe tools/proto/vmem_overlap_probe.hip && ./vmem_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                         \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// src: n uint4 per region, three regions A | B | C, element k of region r = {r, k, ~k, k * 2654435761}.  out: [iters][threads][3] uint4.
#define VICTIM_ASM(CDST)                                                                                                       \
  asm volatile(                                                                                                                \
      "v_lshl_add_u64 v[0:1], %0, 4, %1\n\t"                                                                                   \
      "global_load_dwordx4 v[8:11], v[0:1], off\n\t"                                                                           \
      "v_lshl_add_u64 v[2:3], %0, 4, %2\n\t"                                                                                   \
      "global_load_dwordx4 v[12:15], v[2:3], off\n\t"                                                                          \
      "v_lshl_add_u64 v[4:5], %0, 4, %3\n\t"                                                                                   \
      "global_load_dwordx4 " CDST ", v[4:5], off\n\t"                                                                          \
      "s_waitcnt vmcnt(0)\n\t"                                                                                                 \
      "global_store_dwordx4 %4, v[8:11], off\n\t"                                                                              \
      "global_store_dwordx4 %4, v[12:15], off offset:16\n\t"                                                                   \
      "global_store_dwordx4 %4, " CDST ", off offset:32\n\t"                                                                   \
      "s_waitcnt vmcnt(0)\n\t"                                                                                                 \
      :                                                                                                                        \
      : "v"(i), "s"(pa), "s"(pb), "s"(pc), "v"(o)                                                                              \
      : "v0", "v1", "v2", "v3", "v4", "v5", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
        "memory")

template <int OVERLAP>
__global__ __launch_bounds__(256) void victim(const u32x4* __restrict__ src, u32x4* __restrict__ out, unsigned n, int iters) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x, nt = gridDim.x * 256u;
  const u32x4 *pa = src, *pb = src + n, *pc = src + 2 * (size_t)n;
  for (int it = 0; it < iters; ++it) {
    const unsigned long long i = (t * 7u + (unsigned)it * 131u) % n;
    u32x4* o = out + ((size_t)it * nt + t) * 3;
    if (OVERLAP) VICTIM_ASM("v[0:3]");
    else VICTIM_ASM("v[16:19]");
  }
}

// co-runners.  LDSDMA = 1: the split-f16 GEMM's staging -- every thread moves 16 B per step straight into LDS with
// global_load_lds_dwordx4, the wave reads fragments back with ds_read_b128 and feeds f16 MFMAs.  LDSDMA = 0: the same traffic through
// VGPRs (global_load_dwordx4 + ds_write_b128).
__device__ __forceinline__ void lds_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int LDSDMA>
__global__ __launch_bounds__(256) void corunner(const char* __restrict__ big, size_t bytes, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[2][8][4096];
  f32x16 c0 = {}, c1 = {};
  const size_t stride = (size_t)gridDim.x * 8 * 4096;
  size_t off = (size_t)blockIdx.x * 8 * 4096;
  for (int i = 0; i < iters; ++i) {
    char* buf = &lds[i & 1][0][0];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const char* g = big + (off + (size_t)k * 4096 + threadIdx.x * 16) % bytes;
      if (LDSDMA) lds_dma16(g, buf + k * 4096 + (threadIdx.x & ~63) * 16);       // wave-uniform LDS base, lane * 16 added by the hardware
      else *(u32x4*)(buf + k * 4096 + threadIdx.x * 16) = *(const u32x4*)g;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f16x8 a = *(const f16x8*)(buf + k * 4096 + threadIdx.x * 16);
      const f16x8 b = *(const f16x8*)(buf + k * 4096 + ((threadIdx.x + 64) & 255) * 16);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    off += stride;
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
  if (s == 123.456f) sink[0] = s;
}

template <int OVERLAP>
static long long run(int co, hipStream_t s0, hipStream_t s1, const u32x4* src, u32x4* out, unsigned n, const char* big, size_t bytes,
                     float* sink) {
  const int blocks = 1024, iters = 32, reps = 8;
  const size_t nt = (size_t)blocks * 256, cnt = (size_t)iters * nt * 3;
  std::vector<u32x4> h(cnt);
  long long bad = 0, bad_c = 0, bad_lo = 0, bad_hi48 = 0;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipMemsetAsync(out, 0xff, cnt * 16, s0));
    CHECK(hipStreamSynchronize(s0));
    if (co == 1) hipLaunchKernelGGL((corunner<1>), dim3(1024), dim3(256), 0, s1, big, bytes, sink, 400);
    if (co == 2) hipLaunchKernelGGL((corunner<0>), dim3(1024), dim3(256), 0, s1, big, bytes, sink, 400);
    hipLaunchKernelGGL((victim<OVERLAP>), dim3(blocks), dim3(256), 0, s0, src, out, n, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), out, cnt * 16, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < cnt; ++k) {
      const size_t it = k / (nt * 3), t = (k / 3) % nt;
      const unsigned reg = (unsigned)(k % 3), i = (unsigned)(((unsigned)t * 7u + (unsigned)it * 131u) % n);
      const unsigned want[4] = {reg, i, ~i, i * 2654435761u};
      for (int e = 0; e < 4; ++e)
        if (h[k][e] != want[e]) {
          ++bad;
          bad_c += reg == 2;
          bad_lo += (e == 0 || e == 2);
          bad_hi48 += (t % 64) >= 48;
          if (bad <= 4) printf("    rep %d it %zu thread %zu (lane %zu) load %c component %d: got %08x want %08x\n", r, it, t, t % 64, 'A' + reg, e, h[k][e], want[e]);
        }
    }
  }
  printf("%-34s co-runner %-26s: %lld wrong dwords of %zu (load C: %lld, components 0/2: %lld, lanes 48-63: %lld)\n",
         OVERLAP ? "dest v[0:3] overlaps the addresses" : "control: dest v[16:19]",
         co == 0 ? "none" : co == 1 ? "global_load_lds + f16 MFMA" : "plain loads + f16 MFMA", bad, cnt * 4 * reps, bad_c, bad_lo, bad_hi48);
  fflush(stdout);
  return bad;
}

int main() {
  hipStream_t s0, s1;
  CHECK(hipStreamCreate(&s0));
  CHECK(hipStreamCreate(&s1));
  const unsigned n = 1u << 20;
  std::vector<u32x4> hs((size_t)3 * n);
  for (unsigned r = 0; r < 3; ++r)
    for (unsigned k = 0; k < n; ++k) hs[(size_t)r * n + k] = u32x4{r, k, ~k, k * 2654435761u};
  u32x4 *src, *out;
  char* big;
  float* sink;
  const size_t bytes = (size_t)1 << 30;
  CHECK(hipMalloc(&src, hs.size() * 16));
  CHECK(hipMemcpy(src, hs.data(), hs.size() * 16, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&out, (size_t)32 * 1024 * 256 * 3 * 16));
  CHECK(hipMalloc(&big, bytes + 65536));
  CHECK(hipMemset(big, 0x3c, bytes + 65536));
  CHECK(hipMalloc(&sink, 64));
  long long total = 0;
  for (int co = 0; co < 3; ++co) {
    total += run<1>(co, s0, s1, src, out, n, big, bytes, sink);
    total += run<0>(co, s0, s1, src, out, n, big, bytes, sink);
  }
  printf("total wrong dwords: %lld\n", total);
  return 0;
}
