// Probe: packed-fp32 VALU instructions whose op_sel routes the HIGH dword of a 64-bit source pair into the LOW result lane, executed
// while another wave of the SIMD runs gfx950's 128-bit-operand matrix instructions (v_mfma_f32_32x32x16_f16 & co.).
//
// How we got here (profiles/r6_corunner_defect.txt): the two kernels of this package that were bit-exact alone and wrong next to the
// split-f16 GEMMs are exactly the two whose ISA holds `v_pk_{fma,mul}_f32 ... op_sel:[..1..]`:
//     k_occhead_mix_col<..., VAR 0 / 2>   v_pk_fma_f32 v[0:1], v[0:1], v[98:99], 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]     (4 per wave)
//     k_fine2_h2<4, IMG_INSIDE = true>    v_pk_mul_f32 v[34:35], v[36:37], v[34:35] op_sel:[1,0] op_sel_hi:[0,1]        (145)
// (the register assignment that "fixed" the mix -- VAR 1 -- has none; the shipped k_fine2_h2<4,false> has none), and a 30-line MFMA-only
// co-runner reproduces the failure as long as its matrix instruction is one of the new 4-VGPR-operand forms (32x32x16 f16 / bf16,
// 16x16x32 f16); 32x32x8 f16, 32x32x2 f32 and plain FMAs do not.  The failing element was always the LOW half of the pair, lanes 48-63.
//
// This probe runs each instruction FORM through inline asm in a loop on lane-dependent operands, compares with the same products
// from plain v_mul_f32 / v_fma_f32, and counts mismatches by result half and by 16-lane quarter -- alone and next to the co-runners.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel_probe tools/proto/pk_opsel_probe.hip && /tmp/pk_opsel_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                         \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

enum { F_FMA_B_HI_HI = 0, F_MUL_SWAP_A, F_MUL_SWAP_A_SGPR, F_FMA_PLAIN, F_ADD_B_HI, F_MUL_B_HI_LO, F_MUL_NOSEL, F_FMA_B_HI_HI_V0, F_FMA_B_HI_HI_ACC, F_MOV_A_HI, F_MOV_B_LO, F_FMA_C_HI, F_FMA_A_HI, F_ADD_A_HI, NFORMS };
static const char* form_name[NFORMS] = {
    "v_pk_fma_f32 d,a,b,0 op_sel:[0,1,0] op_sel_hi:[1,1,0]   (mix: both halves x b.hi)",
    "v_pk_mul_f32 d,a,b   op_sel:[1,0]   op_sel_hi:[0,1]     (fine2: a swapped)",
    "v_pk_mul_f32 d,a,s   op_sel:[1,0]   op_sel_hi:[0,1]     (samplers: a swapped, SGPR b)",
    "v_pk_fma_f32 d,a,b,c op_sel_hi:[1,0,0]                  (control: both halves x b.lo, no op_sel)",
    "v_pk_add_f32 d,a,b   op_sel:[0,1]   op_sel_hi:[1,1]     (both halves + b.hi)",
    "v_pk_mul_f32 d,a,b   op_sel:[0,1]   op_sel_hi:[1,0]     (b swapped)",
    "v_pk_mul_f32 d,a,b                                      (control: no modifiers)",
    "form 0 with d = a = v[0:1]                               (the mix's registers)",
    "v_pk_fma_f32 d,a,b,d op_sel:[0,1,0] op_sel_hi:[1,1,1]   (accumulating, b.hi)",
    "v_pk_mov_b32 d,a,b   op_sel:[1,0]                       (d.lo = a.hi, d.hi = b.lo)",
    "v_pk_mov_b32 d,a,b   op_sel:[0,1]                       (d.lo = a.lo, d.hi = b.hi)",
    "v_pk_fma_f32 d,a,b,c op_sel:[0,0,1] op_sel_hi:[1,1,1]   (low half + c.hi)",
    "v_pk_fma_f32 d,a,b,c op_sel:[1,0,0] op_sel_hi:[1,1,1]   (low half a.hi x b.lo)",
    "v_pk_add_f32 d,a,b   op_sel:[1,0]   op_sel_hi:[1,1]     (low half a.hi + b.lo)",
};

// counts[form][half 0/1][quarter 0..3]
template <int FORM>
__global__ __launch_bounds__(256) void victim(unsigned long long* counts, int iters, float sa, float sb, float* ex, unsigned* nex) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned lane = threadIdx.x & 63;
  unsigned bad[2] = {0, 0};
  f32x2 sg = {sa, sb};
  for (int it = 0; it < iters; ++it) {
    const float u = (float)((t * 7u + (unsigned)it * 13u) & 1023u) * 0.03125f + 1.f;
    f32x2 a = {u, u + 0.5f}, b = {2.f + (float)(lane & 15), 3.f + (float)(it & 31)}, c = {0.25f, -0.75f};
    f32x2 d, want;
    if (FORM == F_FMA_B_HI_HI) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[0] * b[1], a[1] * b[1]};
    } else if (FORM == F_MUL_SWAP_A) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[1] * b[0], a[0] * b[1]};
    } else if (FORM == F_MUL_SWAP_A_SGPR) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "s"(sg));
      want = f32x2{a[1] * sg[0], a[0] * sg[1]};
    } else if (FORM == F_FMA_PLAIN) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
      want = f32x2{__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[0], c[0])};
    } else if (FORM == F_ADD_B_HI) {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[0] + b[1], a[1] + b[1]};
    } else if (FORM == F_MUL_B_HI_LO) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[0] * b[1], a[1] * b[0]};
    } else if (FORM == F_MUL_NOSEL) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[0] * b[0], a[1] * b[1]};
    } else if (FORM == F_FMA_B_HI_HI_V0) {
      float d0, d1;
      asm volatile(
          "v_mov_b32 v0, %2\n\tv_mov_b32 v1, %3\n\ts_nop 1\n\t"
          "v_pk_fma_f32 v[0:1], v[0:1], %4, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\ts_nop 1\n\t"
          "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1"
          : "=v"(d0), "=v"(d1)
          : "v"(a[0]), "v"(a[1]), "v"(b)
          : "v0", "v1");
      d = f32x2{d0, d1};
      want = f32x2{a[0] * b[1], a[1] * b[1]};
    } else if (FORM == F_FMA_B_HI_HI_ACC) {
      d = c;
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(b));
      want = f32x2{__builtin_fmaf(a[0], b[1], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
    } else if (FORM == F_MOV_A_HI) {
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));     // d.lo = src0[op_sel[0]], d.hi = src1[op_sel[1]]
      want = f32x2{a[1], b[0]};
    } else if (FORM == F_MOV_B_LO) {
      asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[0], b[1]};
    } else if (FORM == F_FMA_C_HI) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
      want = f32x2{__builtin_fmaf(a[0], b[0], c[1]), __builtin_fmaf(a[1], b[1], c[1])};
    } else if (FORM == F_FMA_A_HI) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
      want = f32x2{__builtin_fmaf(a[1], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])};
    } else {
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
      want = f32x2{a[1] + b[0], a[1] + b[1]};
    }
    if (__float_as_uint(d[0]) != __float_as_uint(want[0]) || __float_as_uint(d[1]) != __float_as_uint(want[1])) {
      const unsigned k = atomicAdd(nex, 1u);      // a few examples: what did the instruction return?
      if (k < 16) {
        float* e = ex + k * 12;
        e[0] = (float)FORM; e[1] = (float)lane; e[2] = d[0]; e[3] = d[1]; e[4] = want[0]; e[5] = want[1];
        e[6] = a[0]; e[7] = a[1]; e[8] = b[0]; e[9] = b[1]; e[10] = c[0]; e[11] = c[1];
      }
    }
    bad[0] += __float_as_uint(d[0]) != __float_as_uint(want[0]);
    bad[1] += __float_as_uint(d[1]) != __float_as_uint(want[1]);
  }
  if (bad[0]) atomicAdd(&counts[(FORM * 2 + 0) * 4 + (lane >> 4)], (unsigned long long)bad[0]);
  if (bad[1]) atomicAdd(&counts[(FORM * 2 + 1) * 4 + (lane >> 4)], (unsigned long long)bad[1]);
}

// co-runners: KIND 0 v_mfma_f32_32x32x16_f16, 1 v_mfma_f32_32x32x8_f16, 2 plain FMAs
template <int KIND>
__global__ __launch_bounds__(256, 2) void corunner(float* sink, int iters) {
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float seed = (float)(threadIdx.x & 7) * 0.125f;
  f16x8 a8, b8;
#pragma unroll
  for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(seed + e); b8[e] = (_Float16)(1.f - seed * e); }
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
        if constexpr (KIND == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = __builtin_fmaf((float)a8[r & 7], (float)b8[r & 7], acc[i][r]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) sink[0] = s;
}

static float* g_ex; static unsigned* g_nex;
template <int FORM>
static void run_form(int co, hipStream_t s0, hipStream_t s1, unsigned long long* counts, float* sink, float* ex = nullptr, unsigned* nex = nullptr) {
  ex = g_ex; nex = g_nex;
  const int reps = 40, blocks = 4096, iters = 256;
  for (int r = 0; r < reps; ++r) {
    if (co == 1) hipLaunchKernelGGL(corunner<0>, dim3(2048), dim3(256), 0, s1, sink, 160);
    if (co == 2) hipLaunchKernelGGL(corunner<1>, dim3(2048), dim3(256), 0, s1, sink, 160);
    if (co == 3) hipLaunchKernelGGL(corunner<2>, dim3(2048), dim3(256), 0, s1, sink, 160);
    hipLaunchKernelGGL(victim<FORM>, dim3(blocks), dim3(256), 0, s0, counts, iters, 1.5f, 2.5f, ex, nex);
    hipLaunchKernelGGL(victim<FORM>, dim3(blocks), dim3(256), 0, s0, counts, iters, 1.5f, 2.5f, ex, nex);
    CHECK(hipDeviceSynchronize());
  }
}

int main() {
  hipStream_t s0, s1;
  CHECK(hipStreamCreate(&s0));
  CHECK(hipStreamCreate(&s1));
  unsigned long long* counts;
  float* sink;
  CHECK(hipMalloc(&counts, NFORMS * 8 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&g_ex, 16 * 12 * 4));
  CHECK(hipMalloc(&g_nex, 4));
  const char* co_name[4] = {"alone", "next to v_mfma_f32_32x32x16_f16", "next to v_mfma_f32_32x32x8_f16", "next to plain v_fma_f32"};
  const double execs = 40.0 * 2 * 4096 * 256 * 256;
  for (int co = 0; co < 4; ++co) {
    CHECK(hipMemset(counts, 0, NFORMS * 8 * sizeof(unsigned long long)));
    CHECK(hipMemset(g_nex, 0, 4));
    run_form<0>(co, s0, s1, counts, sink); run_form<1>(co, s0, s1, counts, sink); run_form<2>(co, s0, s1, counts, sink);
    run_form<3>(co, s0, s1, counts, sink); run_form<4>(co, s0, s1, counts, sink); run_form<5>(co, s0, s1, counts, sink);
    run_form<6>(co, s0, s1, counts, sink); run_form<7>(co, s0, s1, counts, sink); run_form<8>(co, s0, s1, counts, sink);
    run_form<9>(co, s0, s1, counts, sink); run_form<10>(co, s0, s1, counts, sink); run_form<11>(co, s0, s1, counts, sink);
    run_form<12>(co, s0, s1, counts, sink); run_form<13>(co, s0, s1, counts, sink);
    std::vector<unsigned long long> h(NFORMS * 8);
    CHECK(hipMemcpy(h.data(), counts, h.size() * 8, hipMemcpyDeviceToHost));
    printf("== %s (%.2e lane-executions per form)\n", co_name[co], execs);
    for (int f = 0; f < NFORMS; ++f) {
      unsigned long long lo = h[f * 8] + h[f * 8 + 1] + h[f * 8 + 2] + h[f * 8 + 3], hi = h[f * 8 + 4] + h[f * 8 + 5] + h[f * 8 + 6] + h[f * 8 + 7];
      printf("  %-90s wrong: low half %10llu (lanes 0-15 %llu, 16-31 %llu, 32-47 %llu, 48-63 %llu)  high half %10llu (%llu, %llu, %llu, %llu)\n",
             form_name[f], lo, h[f * 8], h[f * 8 + 1], h[f * 8 + 2], h[f * 8 + 3], hi, h[f * 8 + 4], h[f * 8 + 5], h[f * 8 + 6], h[f * 8 + 7]);
    }
    unsigned n = 0;
    float hex_[16 * 12];
    CHECK(hipMemcpy(&n, g_nex, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hex_, g_ex, sizeof(hex_), hipMemcpyDeviceToHost));
    for (unsigned k = 0; k < (n < 16 ? n : 16); ++k) {
      const float* e = hex_ + k * 12;
      printf("    example: form %d lane %2d got {%g, %g} want {%g, %g}  a {%g, %g} b {%g, %g} c {%g, %g}\n", (int)e[0], (int)e[1], e[2], e[3], e[4], e[5],
             e[6], e[7], e[8], e[9], e[10], e[11]);
    }
    fflush(stdout);
  }
  return 0;
}
