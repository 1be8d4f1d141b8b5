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
// Wave-wide max/min of a 64-bit key without the LDS crossbar: four DPP steps reduce each row
// of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror), four v_readlane pairs
// combine the rows on the scalar unit.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
  int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int l) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
#define U64MAX(a, b) ((a) > (b) ? (a) : (b))
#define U64MIN(a, b) ((a) < (b) ? (a) : (b))
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
  u64 o;
  o = dpp_u64<0xB1>(v); v = U64MAX(v, o);
  o = dpp_u64<0x4E>(v); v = U64MAX(v, o);
  o = dpp_u64<0x141>(v); v = U64MAX(v, o);
  o = dpp_u64<0x140>(v); v = U64MAX(v, o);
  u64 a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
  a = U64MAX(a, b); c = U64MAX(c, d);
  return U64MAX(a, c);
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
  u64 o;
  o = dpp_u64<0xB1>(v); v = U64MIN(v, o);
  o = dpp_u64<0x4E>(v); v = U64MIN(v, o);
  o = dpp_u64<0x141>(v); v = U64MIN(v, o);
  o = dpp_u64<0x140>(v); v = U64MIN(v, o);
  u64 a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
  a = U64MIN(a, b); c = U64MIN(c, d);
  return U64MIN(a, c);
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

// ------------------------------------------------------------------ K2 on voxel lists (fused path)
// Exact FPS for points that are distinct voxels of one X*Y*Z grid (what BiFuser_N feeds in:
// bifuser_n.py:130-131,97).  Same selections as k_fps_f32 / the reference kernel, but the
// per-iteration work is pruned: voxels are grouped in 4x4x8 tiles ("buckets", 128 positions,
// coordinates implicit); every bucket caches the best key of its points in LDS.  A new sample s
// can only lower temps of points closer than their current temp, so a bucket whose box is at
// squared distance >= its cached maximal temp is skipped untouched (temps are minima, the skipped
// values would not change).  Per iteration: (A) each lane tests its buckets and appends the
// dirty ones to an LDS list (wave ballot compaction), (B) one wave per dirty bucket refreshes
// its 128 positions (2 per lane, coalesced temp/index rows from L2) and its cached key with a
// wave max, (C) block-wide max over the bucket keys.  After a few dozen samples only a handful
// of buckets is dirty, so an iteration costs three barriers instead of a pass over all points.
#define FT_X 4
#define FT_Y 4
#define FT_Z 8
#define FT_P 128

// One voxel position of a bucket: running temp (integer squared distance, saturated at the key's
// maximum) and the tie rank of the point (-1 = empty position).
// Tie rank: the reference's winner among equal temps is min (bitrev_L(k mod block), k) (see K2 above);
// r(k) = bitrev_L(k mod block) * q + (k >> L), q = ceil(n / block), is that order as one integer
// < block * q, so a key (temp << RB) | (rmask - r) is totally ordered like the 64-bit key of k_fps_f32
// and fits 32 bits for the nuScenes grids (temp < 2^15, r < 2^16): half the DPP/compare work of
// every reduction on the per-sample critical path.
struct FpsCell { int temp; int r; };

__global__ __launch_bounds__(256) void k_fpsv_scatter(const int32_t* __restrict__ lin, int n, int Y, int Z, int NBY,
                                                       int NBZ, int L, int q, int tinit, FpsCell* __restrict__ cell) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int l = lin[k];
  int z = l % Z; l /= Z;
  int y = l % Y; int x = l / Y;
  int b = ((x / FT_X) * NBY + y / FT_Y) * NBZ + z / FT_Z;
  int p = ((x % FT_X) * FT_Y + (y % FT_Y)) * FT_Z + (z % FT_Z);
  unsigned rev = L ? (__brev((unsigned)k & ((1u << L) - 1u)) >> (32 - L)) : 0u;
  cell[(size_t)b * FT_P + p] = FpsCell{tinit, (int)(rev * (unsigned)q + ((unsigned)k >> L))};
}

// LDS-only barrier: the refresh stores to `cell` are consumed later by the SAME wave only, so the
// block barrier must not wait for them (a plain __syncthreads() drains vmcnt and would put the L2
// write latency on every iteration's critical path).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned wave_max_key(unsigned v) {
  v = max(v, dpp_u32<0xB1>(v));
  v = max(v, dpp_u32<0x4E>(v));
  v = max(v, dpp_u32<0x141>(v));
  v = max(v, dpp_u32<0x140>(v));
  unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
  unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ u64 wave_max_key(u64 v) { return wave_max_u64(v); }
// maximum over each 16-lane row, left in every lane of the row
__device__ __forceinline__ unsigned row_max_key(unsigned v) {
  v = max(v, dpp_u32<0xB1>(v));
  v = max(v, dpp_u32<0x4E>(v));
  v = max(v, dpp_u32<0x141>(v));
  v = max(v, dpp_u32<0x140>(v));
  return v;
}
__device__ __forceinline__ u64 row_max_key(u64 v) {
  u64 o;
  o = dpp_u64<0xB1>(v); v = U64MAX(v, o);
  o = dpp_u64<0x4E>(v); v = U64MAX(v, o);
  o = dpp_u64<0x141>(v); v = U64MAX(v, o);
  o = dpp_u64<0x140>(v); v = U64MAX(v, o);
  return v;
}
__device__ __forceinline__ unsigned first_lane_key(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 first_lane_key(u64 v) {
  return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

// Bucket b is owned by lane ((b / NW) % 64) of wave (b % NW), slot (b / NW) / 64: its cached best
// key, the position of that point and the bucket origin live in that lane's VGPRs, so the dirty
// test, the refresh bookkeeping and the per-wave maximum need no LDS at all.  One block barrier
// per sample (publishing the NW per-wave maxima, double-buffered by parity).  Dirty buckets are
// refreshed two at a time so that their L2 loads and reductions overlap.
template <int THREADS, int FPS_RMAX, typename KT>
__global__ __launch_bounds__(THREADS) void k_fps_voxels(int n, int m, int Y, int Z, int NBY, int NBZ, int NB,
                                                         const int32_t* __restrict__ lin, FpsCell* __restrict__ cell,
                                                         int32_t* __restrict__ idx, int L, int q, int RB, long long* dbg) {
  constexpr int NW = THREADS / 64;
  // latency-bound serial chain: when convolution waves share the CU, win every issue arbitration
  __builtin_amdgcn_s_setprio(3);
  __shared__ KT wbest[2][NW];
  __shared__ int wloc[2][NW];   // packed sample coordinates x | y << 10 | z << 20 of the wave's best point
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const KT rmask = (KT)(((KT)1 << RB) - 1);
  const int tmax = sizeof(KT) == 4 ? (int)((1u << (32 - RB)) - 1u) : 0x7FFFFFFF;
  long long tA = 0, tM = 0, tB = 0, tT = 0, nd = 0;

  KT key[FPS_RMAX];     // best key of the owned bucket (0 = empty / no bucket)
  int org[FPS_RMAX];    // bucket origin x | y << 10 | z << 20
  int pos[FPS_RMAX];    // position (0..127) of the best point inside the bucket
#pragma unroll
  for (int r = 0; r < FPS_RMAX; ++r) {
    const int b = (lane + 64 * r) * NW + wave;
    key[r] = 0; pos[r] = 0; org[r] = 0;
    if (b < NB) {                                  // the only divisions of the kernel
      int bz = b % NBZ; int qq = b / NBZ;
      org[r] = ((qq / NBY) * FT_X) | (((qq % NBY) * FT_Y) << 10) | ((bz * FT_Z) << 20);
    }
  }
  // initial keys (temp = tmax everywhere): the wave walks its buckets, lanes = positions
#pragma unroll
  for (int r = 0; r < FPS_RMAX; ++r) {
    for (int src = 0; src < 64; ++src) {
      const int b = (src + 64 * r) * NW + wave;    // wave-uniform
      if (b >= NB) break;
      KT best = 0;
      int bp = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int rr = cell[(size_t)b * FT_P + lane + 64 * h].r;
        KT kk = rr >= 0 ? (KT)(((KT)tmax << RB) | (rmask - (KT)rr)) : (KT)0;
        if (kk > best) { best = kk; bp = lane + 64 * h; }
      }
      KT wb = wave_max_key(best);
      u64 own = __ballot(best == wb);
      int wp = __builtin_amdgcn_readlane(bp, (int)__ffsll((long long)own) - 1);
      if (lane == src) { key[r] = wb; pos[r] = wp; }
    }
  }
  if (tid == 0) idx[0] = 0;
  int sx, sy, sz;
  {
    int l = lin[0];                                // sample 0 is list entry 0 (furthest_point_sample_cuda.cu:46-47)
    sz = l % Z; l /= Z;
    sy = l % Y; sx = l / Y;
  }

  // refresh of ONE bucket's two positions in this lane: new temps, lane-best key and its position
  auto refresh = [&](FpsCell* cp, const FpsCell& c0_, const FpsCell& c1_, int o, KT& best, int& bp) {
    const int x0 = o & 1023, y0 = (o >> 10) & 1023, z0 = o >> 20;
    best = 0; bp = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const FpsCell c = h ? c1_ : c0_;
      const int p = lane + 64 * h;
      if (c.r >= 0) {
        int ex = x0 + (p >> 5) - sx, ey = y0 + ((p >> 3) & 3) - sy, ez = z0 + (p & 7) - sz;
        int t = min(__mul24(ex, ex) + __mul24(ey, ey) + __mul24(ez, ez), c.temp);
        cp[p].temp = t;
        KT kk = (KT)(((KT)t << RB) | (rmask - (KT)c.r));
        if (kk > best) { best = kk; bp = p; }
      }
    }
  };

  for (int j = 1; j < m; ++j) {
    long long c0 = dbg ? clock64() : 0;
    // (A) dirty test + refresh of the owned buckets (squares as 24-bit multiplies: v_mul_lo_u32 is quarter rate)
#pragma unroll
    for (int r = 0; r < FPS_RMAX; ++r) {
      if (r * 64 * NW >= NB) break;
      bool d = false;
      if (key[r]) {
        int x0 = org[r] & 1023, y0 = (org[r] >> 10) & 1023, z0 = org[r] >> 20;
        int dx = max(max(x0 - sx, sx - (x0 + FT_X - 1)), 0);
        int dy = max(max(y0 - sy, sy - (y0 + FT_Y - 1)), 0);
        int dz = max(max(z0 - sz, sz - (z0 + FT_Z - 1)), 0);
        d = (__mul24(dx, dx) + __mul24(dy, dy) + __mul24(dz, dz)) < (int)(key[r] >> RB);
      }
      u64 bal = __ballot(d);
      nd += __popcll(bal);
      while (bal) {
        const int s0 = (int)__ffsll((long long)bal) - 1;
        bal &= bal - 1;
        const bool two = bal != 0;                                // wave-uniform
        const int s1 = two ? (int)__ffsll((long long)bal) - 1 : s0;
        bal &= bal - 1;
        FpsCell* cp0 = cell + (size_t)((s0 + 64 * r) * NW + wave) * FT_P;
        FpsCell* cp1 = cell + (size_t)((s1 + 64 * r) * NW + wave) * FT_P;
        const FpsCell a0 = cp0[lane], a1 = cp0[lane + 64];
        FpsCell b0 = a0, b1 = a1;
        if (two) { b0 = cp1[lane]; b1 = cp1[lane + 64]; }
        KT best0, best1 = 0;
        int bp0, bp1 = 0;
        refresh(cp0, a0, a1, __builtin_amdgcn_readlane(org[r], s0), best0, bp0);
        if (two) refresh(cp1, b0, b1, __builtin_amdgcn_readlane(org[r], s1), best1, bp1);
        const KT wb0 = wave_max_key(best0);
        const KT wb1 = two ? wave_max_key(best1) : (KT)0;
        const int wp0 = __builtin_amdgcn_readlane(bp0, (int)__ffsll((long long)__ballot(best0 == wb0)) - 1);
        if (lane == s0) { key[r] = wb0; pos[r] = wp0; }
        if (two) {
          const int wp1 = __builtin_amdgcn_readlane(bp1, (int)__ffsll((long long)__ballot(best1 == wb1)) - 1);
          if (lane == s1) { key[r] = wb1; pos[r] = wp1; }
        }
      }
    }
    long long c1 = dbg ? clock64() : 0;
    // (B) wave maximum over the owned buckets, published with the sample coordinates
    KT best = 0;
    int bo = 0, bpz = 0;
#pragma unroll
    for (int r = 0; r < FPS_RMAX; ++r) {
      if (key[r] > best) { best = key[r]; bo = org[r]; bpz = pos[r]; }
    }
    // origin fields never carry into each other: x0 + 3 < 1024 etc.
    const int loc = bo + (bpz >> 5) + (((bpz >> 3) & 3) << 10) + ((bpz & 7) << 20);
    KT wb = wave_max_key(best);
    if (best == wb && wb) { wbest[j & 1][wave] = wb; wloc[j & 1][wave] = loc; }
    if (!wb && lane == 0) wbest[j & 1][wave] = 0;
    long long c2 = dbg ? clock64() : 0;
    lds_barrier();
    long long c3 = dbg ? clock64() : 0;
    // global winner: lanes 0..NW-1 each fetch one wave's candidate, one 16-lane DPP max + two readlanes
    // (16 waves x a 15-step compare/select chain was ~1/3 of the per-sample critical path)
    KT g;
    int gl;
    if constexpr (NW <= 16) {   // both key widths (round 5: the 64-bit keys of the large grids walked a 15-step compare chain here)
      const int li = lane & 15;
      const KT mine = li < NW ? wbest[j & 1][li] : (KT)0;
      const int ml = li < NW ? wloc[j & 1][li] : 0;
      g = first_lane_key(row_max_key(mine));
      const u64 own = __ballot(mine == g) & 0xFFFFull;
      gl = __builtin_amdgcn_readlane(ml, (int)__ffsll((long long)own) - 1);
    } else {
      g = wbest[j & 1][0];
      int gw = 0;
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        KT o = wbest[j & 1][w];
        if (o > g) { g = o; gw = w; }
      }
      gl = wloc[j & 1][gw];
    }
    sx = gl & 1023; sy = (gl >> 10) & 1023; sz = gl >> 20;
    if (tid == 0) {   // list ordinal of the winner from its tie rank (off the critical path)
      const unsigned rr = (unsigned)(rmask - (g & rmask));
      const unsigned hi = rr / (unsigned)q, lo = rr - hi * (unsigned)q;
      const unsigned rev = L ? (__brev(hi) >> (32 - L)) : 0u;
      idx[j] = (int)((lo << L) | rev);
    }
    // wbest/wloc parity j&1 is rewritten two iterations later, i.e. after one more barrier that
    // every reader of these values has already passed.
    if (dbg) { long long c4 = clock64(); tA += c1 - c0; tM += c2 - c1; tB += c3 - c2; tT += c4 - c3; }
  }
  if (dbg && lane == 0) {
    long long* o = dbg + wave * 8;
    o[0] = tA; o[1] = tM; o[2] = tB; o[3] = tT; o[4] = 0; o[5] = nd;
  }
}

// ------------------------------------------------------------------ K2, register-resident form (round 3)
// k_fps_voxels refreshes a dirty bucket through L2 (load 128 {temp, rank} cells, min, store): one L2 round trip on the critical
// path of every one of the 2047 dependent iterations -- 1.2 us per iteration.  For grids of at most FPSR_SLOTS * 16 buckets
// (100 x 100 x 8: 625) the whole table fits the register file instead: a position's state IS its 32-bit key
// (temp << RB | rmask - rank; 0 = empty), wave w holds buckets w, w + 16, ... in registers (positions lane, lane + 64 of slot s:
// element s of two 32-wide + two 8-wide register vectors), 80 VGPRs of the 128 a 1024-thread workgroup may use.  A refresh is then
// ~60 VALU / DPP instructions with no memory access; the dirty slot is wave-uniform, so its registers are addressed through the
// VGPR index mode (ext-vector element with an SGPR index -> s_set_gpr_idx; a 40-way switch over named registers made hipcc spill
// 127 of them).  Same arithmetic, same selections as k_fps_voxels (and so as the reference kernel).  gridDim.x = number of
// independent problems: the two search directions of BiFuser_N run as ONE launch, one workgroup (= one CU) each.
#define FPSR_SLOTS 40
struct FpsRegProblem { const int32_t* lin; const FpsCell* cell; int32_t* idx; int n, L, q, RB; };
struct FpsRegArgs { FpsRegProblem p[2]; int m, Y, Z, NBY, NBZ, NB; long long* dbg; };

__global__ __launch_bounds__(1024) void k_fps_voxels_reg(FpsRegArgs a) {
  constexpr int NW = 16;
  __builtin_amdgcn_s_setprio(3);
  __shared__ unsigned wbest[2][NW];
  __shared__ int wloc[2][NW];
  const FpsRegProblem pr = a.p[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int RB = pr.RB, Y = a.Y, Z = a.Z, NB = a.NB, m = a.m;
  const unsigned rmask = (1u << RB) - 1u;
  const unsigned tmax = (1u << (32 - RB)) - 1u;
  long long* dbg = a.dbg;
  long long tA = 0, tM = 0, tB = 0, tT = 0, nd = 0;

  typedef unsigned u32x32 __attribute__((ext_vector_type(32)));
  typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
  u32x32 ca0, ca1;        // positions lane / lane + 64 of slots 0..31
  u32x8 cb0, cb1;         // slots 32..39
  // lane s owns the bookkeeping of the wave's slot s (bucket s * NW + wave): best key, its position, the bucket origin
  unsigned key = 0;
  int org = 0, pos = 0;
  {
    const int b = lane * NW + wave;
    if (lane < FPSR_SLOTS && b < NB) {
      int bz = b % a.NBZ; int qq = b / a.NBZ;
      org = ((qq / a.NBY) * FT_X) | (((qq % a.NBY) * FT_Y) << 10) | ((bz * FT_Z) << 20);
    }
  }
#pragma unroll
  for (int i = 0; i < FPSR_SLOTS; ++i) {
    const int b = i * NW + wave;                  // wave-uniform
    unsigned k0 = 0, k1 = 0;
    if (b < NB) {
      const int r0 = pr.cell[(size_t)b * FT_P + lane].r, r1 = pr.cell[(size_t)b * FT_P + lane + 64].r;
      k0 = r0 >= 0 ? ((tmax << RB) | (rmask - (unsigned)r0)) : 0u;
      k1 = r1 >= 0 ? ((tmax << RB) | (rmask - (unsigned)r1)) : 0u;
      const unsigned best = max(k0, k1);
      const int bp = k1 > k0 ? lane + 64 : lane;
      const unsigned wb = wave_max_key(best);
      const u64 own = __ballot(best == wb);
      const int wp = __builtin_amdgcn_readlane(bp, (int)__ffsll((long long)own) - 1);
      if (lane == i) { key = wb; pos = wp; }
    }
    if (i < 32) { ca0[i] = k0; ca1[i] = k1; }
    else { cb0[i - 32] = k0; cb1[i - 32] = k1; }
  }
  if (tid == 0) pr.idx[0] = 0;
  int sx, sy, sz;
  {
    int l = pr.lin[0];                            // sample 0 is list entry 0 (furthest_point_sample_cuda.cu:46-47)
    sz = l % Z; l /= Z;
    sy = l % Y; sx = l / Y;
  }

  for (int j = 1; j < m; ++j) {
    long long c0 = dbg ? clock64() : 0;
    // (A) dirty test of the owned bucket, refresh of the wave's dirty slots straight in registers
    bool d = false;
    if (key) {
      const int x0 = org & 1023, y0 = (org >> 10) & 1023, z0 = org >> 20;
      const int dx = max(max(x0 - sx, sx - (x0 + FT_X - 1)), 0);
      const int dy = max(max(y0 - sy, sy - (y0 + FT_Y - 1)), 0);
      const int dz = max(max(z0 - sz, sz - (z0 + FT_Z - 1)), 0);
      d = (unsigned)(__mul24(dx, dx) + __mul24(dy, dy) + __mul24(dz, dz)) < (key >> RB);   // 24-bit multiplies: full rate
    }
    u64 bal = __ballot(d);
    nd += __popcll(bal);
    const int exb = (lane >> 5) - sx, eyb = ((lane >> 3) & 3) - sy, ezb = (lane & 7) - sz;
    while (bal) {
      const int s0 = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)bal) - 1);
      bal &= bal - 1;
      const int o = __builtin_amdgcn_readlane(org, s0);
      // squared distances of this lane's two positions of the bucket (lane, lane + 64: x + 2) to the new sample
      const int ex = (o & 1023) + exb, ey = ((o >> 10) & 1023) + eyb, ez = (o >> 20) + ezb;
      const int yz = __mul24(ey, ey) + __mul24(ez, ez);
      const unsigned d0 = (unsigned)(__mul24(ex, ex) + yz), d1 = (unsigned)(__mul24(ex + 2, ex + 2) + yz);
      unsigned k0, k1;
      if (s0 < 32) { k0 = ca0[s0]; k1 = ca1[s0]; }
      else { k0 = cb0[s0 - 32]; k1 = cb1[s0 - 32]; }
      if (k0) k0 = (min(d0, k0 >> RB) << RB) | (k0 & rmask);
      if (k1) k1 = (min(d1, k1 >> RB) << RB) | (k1 & rmask);
      if (s0 < 32) { ca0[s0] = k0; ca1[s0] = k1; }
      else { cb0[s0 - 32] = k0; cb1[s0 - 32] = k1; }
      const unsigned best = max(k0, k1);
      const int bp = k1 > k0 ? lane + 64 : lane;
      const unsigned wb = wave_max_key(best);
      const int wp = __builtin_amdgcn_readlane(bp, (int)__ffsll((long long)__ballot(best == wb)) - 1);
      if (lane == s0) { key = wb; pos = wp; }
    }
    long long c1 = dbg ? clock64() : 0;
    // (B) wave maximum over the owned buckets, published with the sample coordinates
    const int loc = org + (pos >> 5) + (((pos >> 3) & 3) << 10) + ((pos & 7) << 20);
    const unsigned wb = wave_max_key(key);
    if (key == wb && wb) { wbest[j & 1][wave] = wb; wloc[j & 1][wave] = loc; }
    if (!wb && lane == 0) wbest[j & 1][wave] = 0;
    long long c2 = dbg ? clock64() : 0;
    lds_barrier();
    long long c3 = dbg ? clock64() : 0;
    // (C) global winner: lanes 0..15 each fetch one wave's candidate, one 16-lane DPP max + two readlanes
    const int li = lane & 15;
    const unsigned mine = wbest[j & 1][li];
    const int ml = wloc[j & 1][li];
    unsigned v = mine;
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x141>(v));
    v = max(v, dpp_u32<0x140>(v));
    const unsigned g = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    const u64 own = __ballot(mine == g) & 0xFFFFull;
    const int gl = __builtin_amdgcn_readlane(ml, (int)__ffsll((long long)own) - 1);
    sx = gl & 1023; sy = (gl >> 10) & 1023; sz = gl >> 20;
    if (tid == 0) pr.idx[j] = (int)(g & rmask);      // the winner's tie rank; turned into its list ordinal after the loop (a division:
                                                     // inside the loop it made wave 0 the one every barrier waits for)
    if (dbg) { long long c4 = clock64(); tA += c1 - c0; tM += c2 - c1; tB += c3 - c2; tT += c4 - c3; }
  }
  __syncthreads();
  for (int j = 1 + tid; j < m; j += 1024) {
    const unsigned rr = rmask - (unsigned)pr.idx[j];
    const unsigned hi = rr / (unsigned)pr.q, lo = rr - hi * (unsigned)pr.q;
    const unsigned rev = pr.L ? (__brev(hi) >> (32 - pr.L)) : 0u;
    pr.idx[j] = (int)((lo << pr.L) | rev);
  }
  if (dbg && lane == 0 && blockIdx.x == 0) {
    long long* o = dbg + wave * 8;
    o[0] = tA; o[1] = tM; o[2] = tB; o[3] = tT; o[4] = 0; o[5] = nd;
  }
}

extern "C" size_t coocc_fps_voxels_ws(int X, int Y, int Z) {
  size_t nb = (size_t)((X + FT_X - 1) / FT_X) * ((Y + FT_Y - 1) / FT_Y) * ((Z + FT_Z - 1) / FT_Z);
  return nb * FT_P * sizeof(FpsCell);
}

static int g_fps_threads = 1024;   // measured on MI355X (100x100x8 grid, 52 k voxels): 256 -> 3.8 ms, 512 -> 3.4 ms, 1024 -> 3.0 ms
static long long* g_fps_dbg = nullptr;
extern "C" void coocc_fps_voxels_set_debug(long long* p) { g_fps_dbg = p; }
extern "C" void coocc_fps_voxels_set_threads(int t) { g_fps_threads = t; }

struct FpsPrep { int NBX, NBY, NBZ, L, q, RB, tinit, threads, R; long long NB; bool key32; };

// argument checks + tie-rank layout + the initial cell table ({tinit, rank} per position, -1 = empty): memset + scatter
static int fps_prepare(const int32_t* lin, int n, int X, int Y, int Z, int m, const int32_t* idx, void* ws, size_t ws_bytes,
                       hipStream_t s, FpsPrep* o) {
  COOCC_CHECK_ARG(lin && idx && ws && n > 0 && m >= 0 && X > 0 && Y > 0 && Z > 0, "fps_voxels: bad args");
  COOCC_CHECK_ARG(n < (1 << FPS_KBITS), "fps_voxels: n must be < 2^22");
  COOCC_CHECK_ARG(X <= 1020 && Y <= 1020 && Z <= 1016, "fps_voxels: grid dims must fit 10-bit packed coordinates");
  o->NBX = (X + FT_X - 1) / FT_X; o->NBY = (Y + FT_Y - 1) / FT_Y; o->NBZ = (Z + FT_Z - 1) / FT_Z;
  o->NB = (long long)o->NBX * o->NBY * o->NBZ;
  // 4 waves (one per SIMD) keep the per-sample instruction stream short; larger bucket tables
  // spread over more waves so that a lane owns at most 8 buckets.
  int threads = g_fps_threads;
  while (o->NB > 8ll * threads && threads < 1024) threads *= 2;
  COOCC_CHECK_ARG(o->NB <= 8ll * threads, "fps_voxels: grid has too many buckets (use coocc_furthest_point_sampling)");
  if (ws_bytes < coocc_fps_voxels_ws(X, Y, Z)) return coocc_set_error(COOCC_ENOMEM, "fps_voxels: workspace too small");
  o->threads = threads;
  o->R = (int)((o->NB + threads - 1) / threads);
  int L = 0;
  while ((2 << L) <= n && L < 10) ++L;            // block = min(2^floor(log2 n), 1024) = 1 << L
  const int q = (n + (1 << L) - 1) >> L;          // tie ranks live in [0, q << L)
  int RB = L;
  while ((1ll << RB) <= ((long long)q << L)) ++RB;   // strict: rank rmask is never used, key 0 stays "empty"
  const long long d2max = (long long)(X - 1) * (X - 1) + (long long)(Y - 1) * (Y - 1) + (long long)(Z - 1) * (Z - 1);
  o->key32 = RB < 31 && d2max + 1 < (1ll << (32 - RB)) - 1;
  o->tinit = o->key32 ? (int)((1u << (32 - RB)) - 1u) : 0x7FFFFFFF;
  o->L = L; o->q = q; o->RB = RB;
  if (m == 0) return COOCC_OK;
  FpsCell* cell = (FpsCell*)ws;
  COOCC_HIP(hipMemsetAsync(cell, 0xFF, sizeof(FpsCell) * (size_t)o->NB * FT_P, s));
  hipLaunchKernelGGL(k_fpsv_scatter, dim3(cdiv(n, 256)), dim3(256), 0, s, lin, n, Y, Z, o->NBY, o->NBZ, L, q, o->tinit, cell);
  COOCC_LAUNCH_CHECK("k_fpsv_scatter");
  return COOCC_OK;
}

static int g_fps_reg = -1;      // COOCC_FPS_REG=0: always the L2-resident kernel
static bool fps_reg_ok(const FpsPrep& o) {
  if (g_fps_reg < 0) g_fps_reg = getenv("COOCC_FPS_REG") ? atoi(getenv("COOCC_FPS_REG")) : 1;
  return g_fps_reg && o.key32 && o.NB <= (long long)FPSR_SLOTS * 16;
}

extern "C" int coocc_fps_voxels(const int32_t* lin, int n, int X, int Y, int Z, int m, int32_t* idx, void* ws,
                                size_t ws_bytes, void* stream) {
  hipStream_t s = as_stream(stream);
  FpsPrep o;
  int rc = fps_prepare(lin, n, X, Y, Z, m, idx, ws, ws_bytes, s, &o);
  if (rc != COOCC_OK || m == 0) return rc;
  FpsCell* cell = (FpsCell*)ws;
  const int NBY = o.NBY, NBZ = o.NBZ, L = o.L, q = o.q, RB = o.RB, R = o.R, threads = o.threads;
  const long long NB = o.NB;
  const bool key32 = o.key32;
  if (fps_reg_ok(o)) {
    FpsRegArgs a = {};
    a.p[0] = FpsRegProblem{lin, cell, idx, n, L, q, RB};
    a.m = m; a.Y = Y; a.Z = Z; a.NBY = NBY; a.NBZ = NBZ; a.NB = (int)NB; a.dbg = g_fps_dbg;
    hipLaunchKernelGGL(k_fps_voxels_reg, dim3(1), dim3(1024), 0, s, a);
    COOCC_LAUNCH_CHECK("k_fps_voxels_reg");
    return COOCC_OK;
  }
#define FPS_LAUNCH(T, RM, KT) \
  hipLaunchKernelGGL((k_fps_voxels<T, RM, KT>), dim3(1), dim3(T), 0, s, n, m, Y, Z, NBY, NBZ, (int)NB, lin, cell, idx, L, q, \
                     key32 ? RB : 32, g_fps_dbg)
#define FPS_PICK_R(T, KT)                     \
  do {                                        \
    if (R <= 2) FPS_LAUNCH(T, 2, KT);         \
    else if (R <= 3) FPS_LAUNCH(T, 3, KT);    \
    else if (R <= 4) FPS_LAUNCH(T, 4, KT);    \
    else if (R <= 5) FPS_LAUNCH(T, 5, KT);    \
    else if (R <= 6) FPS_LAUNCH(T, 6, KT);    \
    else FPS_LAUNCH(T, 8, KT);                \
  } while (0)
#define FPS_PICK(T)                           \
  do {                                        \
    if (key32) FPS_PICK_R(T, unsigned);       \
    else FPS_PICK_R(T, u64);                  \
  } while (0)
  if (threads == 256) FPS_PICK(256);
  else if (threads == 512) FPS_PICK(512);
  else FPS_PICK(1024);
  COOCC_LAUNCH_CHECK("k_fps_voxels");
  return COOCC_OK;
}

// Both search directions of BiFuser_N (bifuser_n.py:132,152: FPS over the pts list and over the img list, same grid, same m) in
// ONE launch of two workgroups.  Returns COOCC_OK, or 2 (nothing launched, no error text) when the grid is too large for the
// register-resident kernel: the caller then issues two coocc_fps_voxels calls on two streams.
extern "C" int coocc_fps_voxels_pair(const int32_t* lin0, int n0, int32_t* idx0, void* ws0, const int32_t* lin1, int n1, int32_t* idx1,
                                     void* ws1, size_t ws_bytes_each, int X, int Y, int Z, int m, void* stream) {
  COOCC_CHECK_ARG(X > 0 && Y > 0 && Z > 0 && n0 > 0 && n1 > 0, "fps_voxels_pair: bad args");
  {
    // eligibility before anything is enqueued (same rules as fps_prepare / fps_reg_ok)
    const long long NB = (long long)((X + FT_X - 1) / FT_X) * ((Y + FT_Y - 1) / FT_Y) * ((Z + FT_Z - 1) / FT_Z);
    if (NB > (long long)FPSR_SLOTS * 16) return 2;
    const long long d2max = (long long)(X - 1) * (X - 1) + (long long)(Y - 1) * (Y - 1) + (long long)(Z - 1) * (Z - 1);
    for (int n : {n0, n1}) {
      int L = 0;
      while ((2 << L) <= n && L < 10) ++L;
      const int q = (n + (1 << L) - 1) >> L;
      int RB = L;
      while ((1ll << RB) <= ((long long)q << L)) ++RB;
      if (!(RB < 31 && d2max + 1 < (1ll << (32 - RB)) - 1)) return 2;
    }
    if (g_fps_reg < 0) g_fps_reg = getenv("COOCC_FPS_REG") ? atoi(getenv("COOCC_FPS_REG")) : 1;
    if (!g_fps_reg) return 2;
  }
  hipStream_t s = as_stream(stream);
  FpsPrep o0, o1;
  int rc = fps_prepare(lin0, n0, X, Y, Z, m, idx0, ws0, ws_bytes_each, s, &o0);
  if (rc != COOCC_OK) return rc;
  rc = fps_prepare(lin1, n1, X, Y, Z, m, idx1, ws1, ws_bytes_each, s, &o1);
  if (rc != COOCC_OK || m == 0) return rc;
  FpsRegArgs a = {};
  a.p[0] = FpsRegProblem{lin0, (const FpsCell*)ws0, idx0, n0, o0.L, o0.q, o0.RB};
  a.p[1] = FpsRegProblem{lin1, (const FpsCell*)ws1, idx1, n1, o1.L, o1.q, o1.RB};
  a.m = m; a.Y = Y; a.Z = Z; a.NBY = o0.NBY; a.NBZ = o0.NBZ; a.NB = (int)o0.NB; a.dbg = g_fps_dbg;
  hipLaunchKernelGGL(k_fps_voxels_reg, dim3(2), dim3(1024), 0, s, a);
  COOCC_LAUNCH_CHECK("k_fps_voxels_reg");
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
      val[(size_t)r * K + o] = (float)__dsqrt_rn((double)__uint_as_float((unsigned)(mn >> 32)));  // correctly rounded
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

// ------------------------------------------------------------------ K3 / K4 on a voxel grid
// In the fuser both point sets are the non-empty voxels of ONE dense grid and the lists are ascending linear voxel ids,
// so "index order" == lexicographic (x,y,z) order and a dense map voxel -> list ordinal (or -1) answers membership in O(1).
// The brute-force kernels above evaluate 2048 x 52 k distances per direction (0.3-1.0 ms each, all CUs); here
//   * the ball query walks only the (2R+1)^2 x Z window of a centre in linear order (<= 1352 voxels at R = 5, Z = 8), and
//   * top-K walks a table of offsets sorted by (d^2, dx, dy, dz) -- exactly the canonical order (d^2, key index) -- and
//     stops after the K-th occupied in-grid voxel, usually inside the first 64 offsets.
// Both are bit-identical to the brute-force kernels (tests/test_gpu_knn.py); representatives whose K-th neighbour lies
// beyond the table radius are finished by the brute-force kernel (k_knn_topk_unresolved).
__global__ __launch_bounds__(256) void k_index_map_scatter(const int32_t* __restrict__ lin, int n, int32_t* __restrict__ map,
                                                            const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min(n, *n_dev);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) map[lin[i]] = i;
}

__global__ __launch_bounds__(256) void k_fill_i32(int32_t* __restrict__ p, int n, int32_t v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// the list length read on the device (n_dev; at most n_cap entries): no host round trip, grid sized for n_cap
extern "C" int coocc_voxel_index_map_dev(const int32_t* lin, int n_cap, const int32_t* n_dev, int nvox, int32_t* map, void* stream) {
  COOCC_CHECK_ARG(map && nvox > 0 && n_cap > 0 && lin && n_dev, "voxel_index_map_dev: bad args");
  // -1 everywhere by a KERNEL: this form is captured into the serving hipGraphs (the scatter-form half of con_enc.0), and with
  // hipMemsetAsync -- a memset NODE in the middle of the captured chain -- k_sparse_tap_sum read stale bytes of the block's
  // previous tenant instead of -1 once three graphs replayed concurrently for a while (ordinals like 610250240 -> a GPU memory
  // fault: bench.py --config openocc / stress200 aborted in their second or third timed window; tools/jobs/gpu_r5_n.sh).
  // COOCC_MAP_MEMSET=1 restores the memset node.
  static const bool fill_kernel = !(getenv("COOCC_MAP_MEMSET") && atoi(getenv("COOCC_MAP_MEMSET")) == 1);
  if (fill_kernel) hipLaunchKernelGGL(k_fill_i32, dim3(cdiv(nvox, 256)), dim3(256), 0, as_stream(stream), map, nvox, -1);
  else COOCC_HIP(hipMemsetAsync(map, 0xFF, (size_t)nvox * 4, as_stream(stream)));
  hipLaunchKernelGGL(k_index_map_scatter, dim3(cdiv(n_cap, 256)), dim3(256), 0, as_stream(stream), lin, n_cap, map, n_dev);
  COOCC_LAUNCH_CHECK("k_index_map_scatter");
  return COOCC_OK;
}

extern "C" int coocc_voxel_index_map(const int32_t* lin, int n, int nvox, int32_t* map, void* stream) {
  COOCC_CHECK_ARG(map && nvox > 0 && n >= 0 && (lin || n == 0), "voxel_index_map: bad args");
  COOCC_HIP(hipMemsetAsync(map, 0xFF, (size_t)nvox * 4, as_stream(stream)));      // -1 everywhere
  if (n) {
    hipLaunchKernelGGL(k_index_map_scatter, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), lin, n, map, (const int32_t*)nullptr);
    COOCC_LAUNCH_CHECK("k_index_map_scatter");
  }
  return COOCC_OK;
}

// centres: list ordinals into the QUERY list (its voxels are lin_q[ord]); map_q: voxel -> query ordinal.
__global__ __launch_bounds__(256) void k_ball_query_vox(int m, float min_r2, float max_r2, int R, int nsample, int X, int Y, int Z,
                                                         const int32_t* __restrict__ centre_ord, const int32_t* __restrict__ lin_q,
                                                         const int32_t* __restrict__ map_q, int32_t* __restrict__ idx_all) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= m) return;
  int32_t* idx = idx_all + (size_t)c * nsample;
  const int v0 = lin_q[centre_ord[c]];
  const int cz = v0 % Z, cy = (v0 / Z) % Y, cx = v0 / (Z * Y);
  const int x0 = max(cx - R, 0), x1 = min(cx + R, X - 1);
  const int y0 = max(cy - R, 0), y1 = min(cy + R, Y - 1);
  const int z0 = max(cz - R, 0), z1 = min(cz + R, Z - 1);
  const int wy = y1 - y0 + 1, wz = z1 - z0 + 1;
  const int total = (x1 - x0 + 1) * wy * wz;
  int cnt = 0, first = 0;
  for (int base = 0; base < total && cnt < nsample; base += 64) {
    const int t = base + lane;
    bool hit = false;
    int ord = -1;
    if (t < total) {
      const int iz = z0 + t % wz, r = t / wz;
      const int iy = y0 + r % wy, ix = x0 + r / wy;
      ord = map_q[(ix * Y + iy) * Z + iz];
      const float d2 = sqdist3((float)ix, (float)iy, (float)iz, (float)cx, (float)cy, (float)cz);
      hit = ord >= 0 && ((d2 == 0.f) || (d2 >= min_r2 && d2 < max_r2));
    }
    const u64 bal = __ballot(hit);
    if (bal) {
      if (cnt == 0) first = __shfl(ord, (int)__ffsll((long long)bal) - 1);
      const int slot = cnt + __popcll(bal & ((1ull << lane) - 1ull));
      if (hit && slot < nsample) idx[slot] = ord;
      cnt += __popcll(bal);
    }
  }
  if (cnt > nsample) cnt = nsample;
  for (int l = cnt + lane; l < nsample; l += 64) idx[l] = first;  // first == 0 when no hit
}

extern "C" int coocc_ball_query_voxels(int m, float min_radius, float max_radius, int nsample, int X, int Y, int Z,
                                       const int32_t* centre_ord, const int32_t* lin_q, const int32_t* map_q, int32_t* idx,
                                       void* stream) {
  COOCC_CHECK_ARG(m > 0 && nsample > 0 && X > 0 && Y > 0 && Z > 0 && centre_ord && lin_q && map_q && idx, "ball_query_voxels: bad args");
  COOCC_CHECK_ARG(max_radius >= 0.f && max_radius < 4096.f, "ball_query_voxels: bad radius");
  const float max_r2 = max_radius * max_radius;
  int R = (int)ceilf(max_radius);
  while (R > 0 && (float)(R * R) >= max_r2) --R;          // largest R with R^2 < max_r2 (hits need d^2 < max_r2, or d^2 == 0)
  hipLaunchKernelGGL(k_ball_query_vox, dim3(cdiv(m, 4)), dim3(256), 0, as_stream(stream), m, min_radius * min_radius, max_r2, R,
                     nsample, X, Y, Z, centre_ord, lin_q, map_q, idx);
  COOCC_LAUNCH_CHECK("k_ball_query_vox");
  return COOCC_OK;
}

// offsets: [noff] packed (dx + 128) | (dy + 128) << 8 | (dz + 128) << 16, sorted by (d^2, dx, dy, dz); every offset with
// d^2 <= table_d2 and |dz| < Z is present.  Representatives that find fewer than K keys inside the table get idx[.][0] = -2.
template <int K>
__global__ __launch_bounds__(256) void k_knn_topk_vox(int nq, int X, int Y, int Z, const int32_t* __restrict__ rep_ord,
                                                       const int32_t* __restrict__ lin_q, const int32_t* __restrict__ map_k,
                                                       const uint32_t* __restrict__ offsets, int noff, float* __restrict__ val,
                                                       int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nq) return;
  const int v0 = lin_q[rep_ord[r]];
  const int cz = v0 % Z, cy = (v0 / Z) % Y, cx = v0 / (Z * Y);
  int found = 0;
  for (int base = 0; base < noff && found < K; base += 64) {
    const int t = base + lane;
    int ord = -1;
    float d2 = 0.f;
    if (t < noff) {
      const uint32_t o = offsets[t];
      const int dx = (int)(o & 255u) - 128, dy = (int)((o >> 8) & 255u) - 128, dz = (int)((o >> 16) & 255u) - 128;
      const int ix = cx + dx, iy = cy + dy, iz = cz + dz;
      if ((unsigned)ix < (unsigned)X && (unsigned)iy < (unsigned)Y && (unsigned)iz < (unsigned)Z) {
        ord = map_k[(ix * Y + iy) * Z + iz];
        d2 = sqdist3((float)cx, (float)cy, (float)cz, (float)ix, (float)iy, (float)iz);
      }
    }
    const u64 bal = __ballot(ord >= 0);
    const int p = found + __popcll(bal & ((1ull << lane) - 1ull));
    if (ord >= 0 && p < K) {
      val[(size_t)r * K + p] = (float)__dsqrt_rn((double)d2);
      idx[(size_t)r * K + p] = ord;
    }
    found += __popcll(bal);
  }
  if (found < K && lane == 0) idx[(size_t)r * K] = -2;       // finished by k_knn_topk_unresolved
}

// brute force (k_knn_topk) for the representatives the offset table could not resolve
template <int K>
__global__ __launch_bounds__(256) void k_knn_topk_unresolved(int nq, int nk, const float* __restrict__ q,
                                                              const float* __restrict__ key, float* __restrict__ val,
                                                              int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nq || idx[(size_t)r * K] != -2) return;
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
    if (best[0] == mn) {
#pragma unroll
      for (int j = 0; j < K - 1; ++j) best[j] = best[j + 1];
      best[K - 1] = ~0ull;
    }
    if (lane == 0) {
      val[(size_t)r * K + o] = (float)__dsqrt_rn((double)__uint_as_float((unsigned)(mn >> 32)));
      idx[(size_t)r * K + o] = (int32_t)(unsigned)mn;
    }
  }
}

template <int K>
static void launch_topk_vox(int nq, int nk, int X, int Y, int Z, const int32_t* rep_ord, const int32_t* lin_q, const int32_t* map_k,
                            const uint32_t* offsets, int noff, const float* q, const float* key, float* val, int32_t* idx,
                            hipStream_t s) {
  hipLaunchKernelGGL(k_knn_topk_vox<K>, dim3(cdiv(nq, 4)), dim3(256), 0, s, nq, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, val, idx);
  hipLaunchKernelGGL(k_knn_topk_unresolved<K>, dim3(cdiv(nq, 4)), dim3(256), 0, s, nq, nk, q, key, val, idx);
}

extern "C" int coocc_knn_topk_voxels(int nq, int nk, int K, int X, int Y, int Z, const int32_t* rep_ord, const int32_t* lin_q,
                                     const int32_t* map_k, const uint32_t* offsets, int noff, const float* q, const float* key,
                                     float* val, int32_t* idx, void* stream) {
  COOCC_CHECK_ARG(nq > 0 && nk > 0 && X > 0 && Y > 0 && Z > 0 && rep_ord && lin_q && map_k && offsets && noff > 0 && q && key &&
                      val && idx, "knn_topk_voxels: bad args");
  COOCC_CHECK_ARG(K >= 1 && K <= 8 && K <= nk, "knn_topk_voxels: need 1 <= K <= min(8, nk)");
  hipStream_t s = as_stream(stream);
  switch (K) {
    case 1: launch_topk_vox<1>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    case 2: launch_topk_vox<2>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    case 3: launch_topk_vox<3>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    case 4: launch_topk_vox<4>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    case 5: launch_topk_vox<5>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    case 6: launch_topk_vox<6>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    case 7: launch_topk_vox<7>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
    default: launch_topk_vox<8>(nq, nk, X, Y, Z, rep_ord, lin_q, map_k, offsets, noff, q, key, val, idx, s); break;
  }
  COOCC_LAUNCH_CHECK("k_knn_topk_vox");
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
  // ball_query pads a short group with its FIRST member (ball_query_cuda.cu:47-52): those entries repeat the atomic entry 0 of the
  // group already issues (atomicMax with the same value is idempotent) -- and at 12 % LiDAR occupancy they are half of all
  // entries, all serialising on one address per centre
  if (i != c * ns && qi == group[c * ns]) return;
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
