// Prototype: fp32-accurate GEMM on the f16 matrix cores by operand splitting (a = hi + lo * 2^-11).
// Checks the error of  hh + (hl + lh) * 2^-11  against fp64, next to the fp32 MFMA and to bf16x3 (6 products).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[M,N] = A[M,K] * B[N,K]^T ; one wave per 32x32 tile, K multiple of 16
template <int MODE>
__global__ void k(const float* A, const float* B, float* C, int M, int N, int K, float sA) {
  const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  f32x16 hh = {}, x = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float* ap = A + (size_t)(m0 + li) * K + k0 + 8 * h;
    const float* bp = B + (size_t)(n0 + li) * K + k0 + 8 * h;
    if (MODE == 0) {            // fp32 MFMA 32x32x2: 8 steps
      for (int s = 0; s < 8; ++s)
        hh = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)(m0 + li) * K + k0 + 2 * s + h], B[(size_t)(n0 + li) * K + k0 + 2 * s + h], hh, 0, 0, 0);
    } else if (MODE == 1 || MODE == 2 || MODE == 4) {   // f16 split: 1 = scaled lo (2 accumulators), 2 = unscaled lo (1 acc), 4 = hi only
      f16x8 ah, al, bh, bl;
      const float ls = MODE == 1 ? 2048.f : 1.f;
      for (int e = 0; e < 8; ++e) {
        float a = ap[e] * sA, b = bp[e];
        ah[e] = (_Float16)a; al[e] = (_Float16)((a - (float)ah[e]) * ls);
        bh[e] = (_Float16)b; bl[e] = (_Float16)((b - (float)bh[e]) * ls);
      }
      hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, hh, 0, 0, 0);
      if (MODE == 1) {
        x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, x, 0, 0, 0);
        x = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, x, 0, 0, 0);
      } else if (MODE == 2) {
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, hh, 0, 0, 0);
        hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, hh, 0, 0, 0);
      }
    } else if (MODE == 3) {     // bf16 x 3, six products, small terms first
      bf16x8 a0, a1, a2, b0, b1, b2;
      for (int e = 0; e < 8; ++e) {
        float a = ap[e], b = bp[e];
        a0[e] = (__bf16)a; float r = a - (float)a0[e]; a1[e] = (__bf16)r; r -= (float)a1[e]; a2[e] = (__bf16)r;
        b0[e] = (__bf16)b; r = b - (float)b0[e]; b1[e] = (__bf16)r; r -= (float)b1[e]; b2[e] = (__bf16)r;
      }
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, x, 0, 0, 0);
      x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, x, 0, 0, 0);
      hh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, hh, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h, n = n0 + li;
    float v = MODE == 1 ? (hh[r] + x[r] * (1.f / 2048.f)) / sA : (MODE == 3 ? hh[r] + x[r] : hh[r] / (MODE == 0 ? 1.f : sA));
    C[(size_t)m * N + n] = v;
  }
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2 * log(urand())) * cos(6.283185307179586 * urand()); }

int main() {
  const int M = 256, N = 128;
  for (int K : {768, 3456, 13824}) for (int dist = 0; dist < 3; ++dist) {
    std::vector<float> A((size_t)M * K), B((size_t)N * K), C((size_t)M * N);
    srand(1 + K + dist);
    const double wscale = dist == 2 ? 1e-4 : 0.02;
    for (auto& v : A) { double g = nrand(); g = g > 0 ? g : 0; v = (float)(g * exp(nrand()) * (dist == 1 ? 30 : 1)); }
    for (auto& v : B) v = (float)(nrand() * wscale);
    std::vector<double> R((size_t)M * N);
    double rms = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
      R[(size_t)m * N + n] = s; rms += s * s;
    }
    rms = sqrt(rms / (M * N));
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("K=%5d dist=%d (act x%s, w %.0e):", K, dist, dist == 1 ? "30" : "1", wscale);
    for (int mode = 0; mode < 5; ++mode) {
      dim3 g(M / 32, N / 32);
      const float sA = 1.f;
      if (mode == 0) k<0><<<g, 64>>>(dA, dB, dC, M, N, K, sA);
      if (mode == 1) k<1><<<g, 64>>>(dA, dB, dC, M, N, K, sA);
      if (mode == 2) k<2><<<g, 64>>>(dA, dB, dC, M, N, K, sA);
      if (mode == 3) k<3><<<g, 64>>>(dA, dB, dC, M, N, K, sA);
      if (mode == 4) k<4><<<g, 64>>>(dA, dB, dC, M, N, K, sA);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double e = 0, mx = 0;
      for (size_t i = 0; i < C.size(); ++i) { double d = C[i] - R[i]; e += d * d; mx = fmax(mx, fabs(d)); }
      const char* names[] = {"f32mfma", "f16x2 scaled-lo", "f16x2 unscaled-lo", "bf16x3(6)", "f16 hi only"};
      printf("  %s rms %.3e max %.3e |", names[mode], sqrt(e / C.size()) / rms, mx / rms);
    }
    printf("\n");
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
