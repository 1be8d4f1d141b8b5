// GSFusion index search (K2..K5): farthest-point sampling, brute-force top-K, ball query,
// deterministic assignment.  All distance math is the exact expression of the reference
// kernels (see common.h sqdist3); tie rules are documented per kernel.
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  for (int m = 32; m > 0; m >>= 1) {
    u64 o = shfl_xor_u64(v, m);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
  for (int m = 32; m > 0; m >>= 1) {
    u64 o = shfl_xor_u64(v, m);
    v = o < v ? o : v;
  }
  return v;
}

// ------------------------------------------------------------------ K2: FPS
// Reference: furthest_point_sampling_kernel<block> (furthest_point_sample_cuda.cu:25-141),
// one block per batch, block = min(2^floor(log2 n), 1024).  Its winner among points of
// equal (maximal) temp is the one with minimal (bitrev_L(k mod block), k), L = log2(block):
// inside a thread the strict '>' (:69-70) keeps the lowest k, and the shared-memory tree
// (:76-136) keeps the LEFT operand on ties, its last level deciding on tid bit 0, the one
// before on bit 1, ...  We therefore reduce the totally ordered 64-bit key
//   (bits(temp) << 32) | ~((bitrev_L(k mod block) << 22) | k)
// with max(), which makes the point->lane assignment irrelevant: every lane keeps its
// points (xyz + running temp) in VGPRs, the remainder streams from L2, the reduction is
// DPP/shuffle inside each wave and one LDS round across the 16 waves.
#define FPS_THREADS 1024
#define FPS_KBITS 22

__device__ __forceinline__ unsigned fps_tiebreak(int k, int L, unsigned blockmask) {
  unsigned r = L ? (__brev((unsigned)k & blockmask) >> (32 - L)) : 0u;
  return (r << FPS_KBITS) | (unsigned)k;
}

template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void k_fps_f32(int n, int m, const float* __restrict__ pts_all,
                                                          float* __restrict__ temp_all,
                                                          int32_t* __restrict__ idx_all, int L) {
  __shared__ u64 wbest[2][FPS_THREADS / 64];
  const int tid = threadIdx.x;
  const float* pts = pts_all + (size_t)blockIdx.x * n * 3;
  float* temp = temp_all + (size_t)blockIdx.x * n;
  int32_t* idx = idx_all + (size_t)blockIdx.x * m;
  const unsigned blockmask = (1u << L) - 1u;

  float px[PPT], py[PPT], pz[PPT], pt[PPT];
  unsigned ntb[PPT];  // ~tiebreak of the cached point, 0 for the slots past n
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    int k = tid + i * FPS_THREADS;
    bool v = k < n;
    px[i] = v ? pts[k * 3 + 0] : 0.f;
    py[i] = v ? pts[k * 3 + 1] : 0.f;
    pz[i] = v ? pts[k * 3 + 2] : 0.f;
    pt[i] = v ? 1e10f : 0.f;
    ntb[i] = v ? ~fps_tiebreak(k, L, blockmask) : 0u;
  }
  for (int k = tid + PPT * FPS_THREADS; k < n; k += FPS_THREADS) temp[k] = 1e10f;

  int old = 0;
  if (tid == 0) idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    u64 best = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      float d = sqdist3(x1, y1, z1, px[i], py[i], pz[i]);
      float t = fminf(d, pt[i]);  // empty slots hold temp 0 and key (0, 0): they never win
      pt[i] = t;
      u64 key = ((u64)__float_as_uint(t) << 32) | (u64)ntb[i];
      best = key > best ? key : best;
    }
#pragma unroll 2
    for (int k = tid + PPT * FPS_THREADS; k < n; k += FPS_THREADS) {
      float d = sqdist3(x1, y1, z1, pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2]);
      float t = fminf(d, temp[k]);
      temp[k] = t;
      u64 key = ((u64)__float_as_uint(t) << 32) | (u64)(~fps_tiebreak(k, L, blockmask));
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) wbest[j & 1][tid >> 6] = best;
    __syncthreads();
    u64 b = wbest[j & 1][tid & 15];  // 16 wave results: one LDS read + 4 butterfly steps
#pragma unroll
    for (int mm = 8; mm > 0; mm >>= 1) {
      u64 o = shfl_xor_u64(b, mm);
      b = o > b ? o : b;
    }
    old = (int)((~(unsigned)b) & ((1u << FPS_KBITS) - 1u));
    if (tid == 0) idx[j] = old;
  }
}

extern "C" int coocc_furthest_point_sampling(int b, int n, int m, const float* points, float* temp,
                                             int32_t* idx, void* stream) {
  COOCC_CHECK_ARG(points && temp && idx && b > 0 && n > 0 && m >= 0, "fps: bad args");
  COOCC_CHECK_ARG(n < (1 << FPS_KBITS), "fps: n must be < 2^22");
  if (m == 0) return COOCC_OK;
  int L = 0;
  while ((2 << L) <= n && L < 10) ++L;  // block = min(2^floor(log2 n), 1024) = 1 << L
  if (n <= 8 * FPS_THREADS)
    hipLaunchKernelGGL(k_fps_f32<8>, dim3(b), dim3(FPS_THREADS), 0, as_stream(stream), n, m, points, temp, idx, L);
  else
    hipLaunchKernelGGL(k_fps_f32<10>, dim3(b), dim3(FPS_THREADS), 0, as_stream(stream), n, m, points, temp, idx, L);
  COOCC_LAUNCH_CHECK("k_fps_f32");
  return COOCC_OK;
}

// ------------------------------------------------------------------ K4: ball query
// ball_query_kernel (ball_query_cuda.cu:11-54): serial scan per centre, keep the first
// `nsample` hits in index order, pad with the first hit, zeros when there is none.
// Here: one wave per centre, 64 candidates per step, ballot + prefix-popcount keep order.
__global__ __launch_bounds__(256) void k_ball_query(int n, int m, float min_r2, float max_r2, int nsample,
                                                     const float* __restrict__ new_xyz_all,
                                                     const float* __restrict__ xyz_all,
                                                     int32_t* __restrict__ idx_all) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= m) return;
  const float* new_xyz = new_xyz_all + ((size_t)blockIdx.y * m + c) * 3;
  const float* xyz = xyz_all + (size_t)blockIdx.y * n * 3;
  int32_t* idx = idx_all + ((size_t)blockIdx.y * m + c) * nsample;
  const float cx = new_xyz[0], cy = new_xyz[1], cz = new_xyz[2];
  int cnt = 0, first = 0;
  for (int base = 0; base < n && cnt < nsample; base += 64) {
    int k = base + lane;
    bool hit = false;
    if (k < n) {
      float d2 = sqdist3(xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2], cx, cy, cz);
      hit = (d2 == 0.f) || (d2 >= min_r2 && d2 < max_r2);
    }
    u64 bal = __ballot(hit);
    if (bal) {
      if (cnt == 0) first = base + (int)__ffsll((long long)bal) - 1;
      int slot = cnt + __popcll(bal & ((1ull << lane) - 1ull));
      if (hit && slot < nsample) idx[slot] = k;
      cnt += __popcll(bal);
    }
  }
  if (cnt > nsample) cnt = nsample;
  for (int l = cnt + lane; l < nsample; l += 64) idx[l] = first;  // first == 0 when no hit
}

extern "C" int coocc_ball_query(int b, int n, int m, float min_radius, float max_radius, int nsample,
                                const float* new_xyz, const float* xyz, int32_t* idx, void* stream) {
  COOCC_CHECK_ARG(new_xyz && xyz && idx && b > 0 && n > 0 && m > 0 && nsample > 0, "ball_query: bad args");
  dim3 grid(cdiv(m, 4), b);
  hipLaunchKernelGGL(k_ball_query, grid, dim3(256), 0, as_stream(stream), n, m, min_radius * min_radius,
                     max_radius * max_radius, nsample, new_xyz, xyz, idx);
  COOCC_LAUNCH_CHECK("k_ball_query");
  return COOCC_OK;
}

// ------------------------------------------------------------------ K3: top-K
// bifuser_n.py:101-103: dist = norm(repr_query - key); topk(K, largest=False).
// One wave per query row; each lane keeps its K best (d^2, idx) keys sorted in VGPRs while
// scanning keys lane-strided, then K rounds of wave-min pop the global order.  Order is the
// total order (d^2, key index), the canonical tie rule.
template <int K>
__global__ __launch_bounds__(256) void k_knn_topk(int nq, int nk, const float* __restrict__ q,
                                                   const float* __restrict__ key, float* __restrict__ val,
                                                   int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nq) return;
  const float qx = q[r * 3 + 0], qy = q[r * 3 + 1], qz = q[r * 3 + 2];
  u64 best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = ~0ull;
  for (int k = lane; k < nk; k += 64) {
    float d2 = sqdist3(qx, qy, qz, key[k * 3 + 0], key[k * 3 + 1], key[k * 3 + 2]);
    u64 kk = ((u64)__float_as_uint(d2) << 32) | (unsigned)k;
    if (kk < best[K - 1]) {
      best[K - 1] = kk;
#pragma unroll
      for (int j = K - 1; j > 0; --j) {
        if (best[j] < best[j - 1]) { u64 t = best[j]; best[j] = best[j - 1]; best[j - 1] = t; }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < K; ++o) {
    u64 mn = wave_min_u64(best[0]);
    if (best[0] == mn) {  // unique owner: keys embed the index
#pragma unroll
      for (int j = 0; j < K - 1; ++j) best[j] = best[j + 1];
      best[K - 1] = ~0ull;
    }
    if (lane == 0) {
      val[(size_t)r * K + o] = __fsqrt_rn(__uint_as_float((unsigned)(mn >> 32)));
      idx[(size_t)r * K + o] = (int32_t)(unsigned)mn;
    }
  }
}

extern "C" int coocc_knn_topk(int nq, int nk, int K, const float* q, const float* key, float* val,
                              int32_t* idx, void* stream) {
  COOCC_CHECK_ARG(q && key && val && idx && nq > 0 && nk > 0, "knn_topk: bad args");
  COOCC_CHECK_ARG(K >= 1 && K <= 8 && K <= nk, "knn_topk: need 1 <= K <= min(8, nk)");
  dim3 grid(cdiv(nq, 4)), block(256);
  hipStream_t s = as_stream(stream);
  switch (K) {
    case 1: hipLaunchKernelGGL(k_knn_topk<1>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    case 2: hipLaunchKernelGGL(k_knn_topk<2>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    case 3: hipLaunchKernelGGL(k_knn_topk<3>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    case 4: hipLaunchKernelGGL(k_knn_topk<4>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    case 5: hipLaunchKernelGGL(k_knn_topk<5>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    case 6: hipLaunchKernelGGL(k_knn_topk<6>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    case 7: hipLaunchKernelGGL(k_knn_topk<7>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
    default: hipLaunchKernelGGL(k_knn_topk<8>, grid, block, 0, s, nq, nk, q, key, val, idx); break;
  }
  COOCC_LAUNCH_CHECK("k_knn_topk");
  return COOCC_OK;
}

// ------------------------------------------------------------------ K5: assignment
// bifuser_n.py:104-125: query_NN_key_idx[k][group[c,:]] = nn[c,k] for valid centres, later
// centres overriding earlier ones.  atomicMax on the centre ordinal makes "last writer wins"
// deterministic; a second pass translates the winning centre into its k-th key.
__global__ void k_assign_winner(int nc, int K, int ns, int nq, float thresh, const float* __restrict__ val,
                                const int32_t* __restrict__ group, int32_t* __restrict__ winner) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc * ns) return;
  int c = i / ns;
  int qi = group[i];
  for (int k = 0; k < K; ++k)
    if (val[c * K + k] < thresh) atomicMax(&winner[(size_t)k * nq + qi], c);
}

__global__ void k_assign_lookup(int K, int nq, const int32_t* __restrict__ nn,
                                const int32_t* __restrict__ winner, int32_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * nq) return;
  int k = i / nq;
  int w = winner[i];
  out[i] = w >= 0 ? nn[w * K + k] : -1;
}

extern "C" int coocc_knn_assign(int nc, int K, int ns, int nq, float dist_thresh, const float* val,
                                const int32_t* nn, const int32_t* group, int32_t* winner, int32_t* out,
                                void* stream) {
  COOCC_CHECK_ARG(val && nn && group && winner && out && nc > 0 && K > 0 && ns > 0 && nq > 0, "knn_assign: bad args");
  hipStream_t s = as_stream(stream);
  COOCC_HIP(hipMemsetAsync(winner, 0xFF, sizeof(int32_t) * (size_t)K * nq, s));
  hipLaunchKernelGGL(k_assign_winner, dim3(cdiv((long long)nc * ns, 256)), dim3(256), 0, s, nc, K, ns, nq,
                     dist_thresh, val, group, winner);
  hipLaunchKernelGGL(k_assign_lookup, dim3(cdiv((long long)K * nq, 256)), dim3(256), 0, s, K, nq, nn, winner, out);
  COOCC_LAUNCH_CHECK("knn_assign");
  return COOCC_OK;
}

__global__ void k_knn_threshold(int nq, float thresh, const float* __restrict__ val,
                                const int32_t* __restrict__ nn, int32_t* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) out[i] = val[i] < thresh ? nn[i] : -1;
}

extern "C" int coocc_knn_threshold(int nq, float dist_thresh, const float* val, const int32_t* nn,
                                   int32_t* out, void* stream) {
  COOCC_CHECK_ARG(val && nn && out && nq > 0, "knn_threshold: bad args");
  hipLaunchKernelGGL(k_knn_threshold, dim3(cdiv(nq, 256)), dim3(256), 0, as_stream(stream), nq, dist_thresh, val, nn, out);
  COOCC_LAUNCH_CHECK("k_knn_threshold");
  return COOCC_OK;
}

__global__ void k_index_rows(const int32_t* __restrict__ base, int nbase, const int32_t* __restrict__ sel, int n,
                             int32_t* __restrict__ rows) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = sel[i];
  if (s < 0) s += nbase;  // Python negative indexing: -1 -> last row (bifuser_n.py:139-144)
  rows[i] = (s >= 0 && s < nbase) ? base[s] : -1;  // out of range (IndexError in the reference) -> zero row
}

extern "C" int coocc_index_rows_i32(const int32_t* base, int nbase, const int32_t* sel, int n, int32_t* rows,
                                    void* stream) {
  COOCC_CHECK_ARG(base && sel && rows && nbase > 0 && n >= 0, "index_rows: bad args");
  if (n == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_index_rows, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), base, nbase, sel, n, rows);
  COOCC_LAUNCH_CHECK("k_index_rows");
  return COOCC_OK;
}
