// hipcc's wait counts after the two LDS-DMA forms.  Compile only:
//   /opt/rocm/lib/llvm/bin/clang++ --offload-arch=gfx950 -O3 -S -x hip tools/proto/lds_dma_waitcnt.hip --cuda-device-only -o - | grep -n "s_waitcnt\|_lds\|lds$"
// k<0> (buffer_load_dwordx4 ... lds): the use of `wn`, loaded BEFORE the LDS load, waits with vmcnt(1) -- the LDS load stays in flight.
// k<1> (global_load_lds_dwordx4):     the same use waits with vmcnt(0) -- the FLAT-encoded form counts as touching LDS and memory,
//                                     and the wait-count pass drains the counter at the next vector-load dependency.
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(const char* a, const char* w, float* out, int n) {
  __shared__ __attribute__((aligned(16))) char lds[2][4096];
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0xFFFFFF00, 0x00020000);
  float acc = 0;
  for (int i = 0; i < n; ++i) {
    f16x8 wn = *(const f16x8*)(w + i * 4096 + threadIdx.x * 16);
    if (MODE == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&lds[(i + 1) & 1][0], 16, threadIdx.x * 16 + i * 4096, 0, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + threadIdx.x * 16 + i * 4096), (__attribute__((address_space(3))) void*)&lds[(i + 1) & 1][0], 16, 0, 0);
    acc += (float)wn[0];              // needs wn only: the LDS load issued after it may stay in flight
    asm volatile("; use %0" :: "v"(acc));
  }
  out[threadIdx.x] = acc;
}
template __global__ void k<0>(const char*, const char*, float*, int);
template __global__ void k<1>(const char*, const char*, float*, int);
