// fp32 implicit-GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32), one kernel family
// for: Conv3d k=3/k=1 (+folded eval-mode BN, ReLU, residual) over channels-last voxel rows,
// nn.Linear (+bias, ReLU), and GSFusion's gather -> knn_enc -> gate -> scatter.
//
//   out[orow(m)][n] = epi( sum_{t<taps} sum_{c<Cin} in[src(m,t)][c] * W[n][c][t] )
//
// src(m,t) is either geometric (voxel m's neighbour for tap t, zero padding) or a row table.
// GEMM view: M = output rows, N = Cout, K = taps*Cin.  Block tile BM x BN, K-chunk 32,
// 4 waves; A (gathered activations) and B (packed weights) tiles are staged in LDS with
// K contiguous so one ds_read_b128 feeds 4 MFMA k-steps (lane-half h reads k=8q+4h..+3;
// any permutation of k applied to both operands leaves the sum unchanged).  Global loads
// for chunk i+1 are issued before the MFMAs of chunk i (register prefetch, LDS double buffer).
//
// MFMA 32x32x2 f32 operand maps (MI355X guide): A: lane l holds A[i=l&31][k=l>>5];
// B: B[k=l>>5][j=l&31]; D: 16 regs, col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5).
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#include "conv_k.h"
#include "h2_rows.h"

template <int BM, int BN, int WM, int WN, bool TABLE>
__global__ __launch_bounds__(256, 2) void k_conv(ConvK p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int PA = BM / 32, PB = BN / 32;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  // two LDS buffers when two workgroups still fit a CU (160 KB), else one buffer + a second barrier
  constexpr int NBUF = (2 * 2 * (BM + BN) * LDS_ST * 4 <= 160 * 1024) ? 2 : 1;
  __shared__ float As[NBUF][BM * LDS_ST];
  __shared__ float Bs[NBUF][BN * LDS_ST];

  // XCD-aware tile order: block id -> XCD (id & 7); each XCD walks a contiguous slab of
  // M tiles (neighbouring voxel rows share halo lines in that XCD's L2) and keeps the N
  // tiles of one M tile on the same XCD.
  const int id = blockIdx.x;
  int mtile, nt, slot = id >> 3;
  if (p.mtiles_per_xcd > 0) {
    const int xcd = id & 7;
    const int mt_local = slot / p.ntiles;
    nt = slot - mt_local * p.ntiles;
    mtile = xcd * p.mtiles_per_xcd + mt_local;
    if (mt_local >= p.mtiles_per_xcd || mtile >= p.mtiles) return;
  } else {   // few M tiles: plain order, consecutive ids (= different XCDs) take different tiles
    mtile = id / p.ntiles;
    nt = id - mtile * p.ntiles;
  }
  const int m0 = mtile * BM, n0 = nt * BN;
  if (p.M_dev) {                       // row count on the device (grid sized for the capacity p.M): whole tiles past it leave
    p.M = min(p.M, *p.M_dev);
    if (m0 >= p.M) return;
  }
  const int tid = threadIdx.x;
  const int piece = tid & 7, lrow = tid >> 3;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
  const int li = lane & 31, h = lane >> 5;

  const float* wbase = p.wgroup_rows > 0 ? p.w + (size_t)(m0 / p.wgroup_rows) * p.wgroup_floats : p.w;
  // per-thread A rows
  int rbase[PA];  // GEOM: packed voxel origin; TABLE: m (or -1)
  int rix[PA], riy[PA], riz[PA];
#pragma unroll
  for (int a = 0; a < PA; ++a) {
    int m = m0 + lrow + 32 * a;
    bool ok = m < p.M;
    if (TABLE) {
      rbase[a] = ok ? m : -1;
      rix[a] = riy[a] = riz[a] = 0;
    } else {
      int oz = m % p.Zo; int r = m / p.Zo;
      int oy = r % p.Yo; r /= p.Yo;
      int ox = r % p.Xo; int b = r / p.Xo;
      rbase[a] = ok ? b : -1;
      rix[a] = ox * p.stride - p.px;
      riy[a] = oy * p.stride - p.py;
      riz[a] = oz * p.stride - p.pz;
    }
  }

  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);

  f32x4 ra[PA], rb[PB];
  // TABLE: the source rows of an iteration are fetched one iteration before its feature rows are requested, so
  // the dependent pair (row-table entry -> feature row) costs one memory latency per iteration instead of two
  int nidx[PA];
  auto iload = [&](int it) {
    const int t = it % p.taps;
#pragma unroll
    for (int a = 0; a < PA; ++a) nidx[a] = rbase[a] >= 0 ? p.gather[(size_t)t * p.gstride + rbase[a]] : -1;
  };
  if (TABLE && it0 < it1) iload(it0);
  auto gload = [&](int it) {
    const int kc = it / p.taps, t = it - kc * p.taps;
    int src[PA];
    if (TABLE) {
#pragma unroll
      for (int a = 0; a < PA; ++a) src[a] = nidx[a];
      if (it + 1 < it1) iload(it + 1);
    }
#pragma unroll
    for (int b = 0; b < PB; ++b)
      rb[b] = *(const f32x4*)(wbase + wfrag_index((size_t)it, p.Npad >> 7, n0 + lrow + 32 * b, piece * 4));
    const int cc = kc * KC + piece * 4;
    const bool cok = cc < p.Cin;
    int kd = 0, kh = 0, kw = 0;
    if (!TABLE) {
      kw = t % p.kz; int r = t / p.kz;
      kh = r % p.ky; kd = r / p.ky;
    }
#pragma unroll
    for (int a = 0; a < PA; ++a) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (TABLE) {
        if (src[a] >= 0 && cok) v = *(const f32x4*)(p.in + (size_t)src[a] * p.in_stride + cc);
      } else {
        int ix = rix[a] + kd, iy = riy[a] + kh, iz = riz[a] + kw;
        bool ok = rbase[a] >= 0 && cok && (unsigned)ix < (unsigned)p.Xi && (unsigned)iy < (unsigned)p.Yi &&
                  (unsigned)iz < (unsigned)p.Zi;
        if (ok) {
          size_t row = (((size_t)rbase[a] * p.Xi + ix) * p.Yi + iy) * p.Zi + iz;
          v = *(const f32x4*)(p.in + row * p.in_stride + cc);
        }
      }
      ra[a] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int a = 0; a < PA; ++a) *(f32x4*)&As[buf][(lrow + 32 * a) * LDS_ST + piece * 4] = ra[a];
#pragma unroll
    for (int b = 0; b < PB; ++b) *(f32x4*)&Bs[buf][(lrow + 32 * b) * LDS_ST + piece * 4] = rb[b];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (it0 < it1) {
    gload(it0);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int it = it0; it < it1; ++it) {
    const bool more = it + 1 < it1;
    if (more) gload(it + 1);
    const float* Ab = &As[cur][(wm * WM + li) * LDS_ST + h * 4];
    const float* Bb = &Bs[cur][(wn * WN + li) * LDS_ST + h * 4];
#pragma unroll
    for (int q = 0; q < KC / 8; ++q) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(Ab + i * 32 * LDS_ST + q * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(Bb + j * 32 * LDS_ST + q * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
    }
    if (NBUF == 2) {
      if (more) lstore(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();          // every wave is done reading the tile
      if (more) lstore(0);
      __syncthreads();
    }
  }

  // epilogue: D row = (r&3) + 8*(r>>2) + 4*h, col = li
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splitk > 1) {
          p.ws[((size_t)blockIdx.y * p.M + m) * p.Npad + n] = v;  // n < Npad always
        } else if (n < p.Cout) {
          size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
          p.out[orow * p.out_stride + n] = epilogue(p, v, n, orow);
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_bf16: the same implicit GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32-MFMA rate): operands
// are rounded to bf16 (RNE) when they are staged into LDS, accumulation and the epilogue (folded BN, residual, ReLU) stay
// fp32, activations stay fp32 in HBM.  This is the reduced-precision path of configs[4] (OpenOccupancy, "fp16"): NOT
// bit-comparable with the fp32 reference -- its parity test states its tolerance from an fp64 anchor.  Geometric taps
// only (3x3x3 / 1x1x1, stride 1 / 2), the fp32 fragment-major weight pack is read as is.  K step = 64 (two 32-channel
// chunks of the pack: same channel slice, consecutive taps), 128 x 128 tile, 2 x 2 waves of 64 x 64, LDS rows of 64 bf16
// padded to 144 B so that ds_read_b128 (8 bf16 per lane = one MFMA operand) is conflict-free.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define BF_KC 64
#define BF_ST 72   // LDS row stride in bf16 (144 B)

__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
  return r;
}

__global__ __launch_bounds__(256, 2) void k_conv_bf16(ConvK p) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2, PA = BM / 32, PB = BN / 32;
  __shared__ __bf16 As[2][BM * BF_ST];
  __shared__ __bf16 Bs[2][BN * BF_ST];

  const int id = blockIdx.x;
  int mtile, nt, slot = id >> 3;
  if (p.mtiles_per_xcd > 0) {
    const int xcd = id & 7;
    const int mt_local = slot / p.ntiles;
    nt = slot - mt_local * p.ntiles;
    mtile = xcd * p.mtiles_per_xcd + mt_local;
    if (mt_local >= p.mtiles_per_xcd || mtile >= p.mtiles) return;
  } else {
    mtile = id / p.ntiles;
    nt = id - mtile * p.ntiles;
  }
  const int m0 = mtile * BM, n0 = nt * BN;
  const int tid = threadIdx.x;
  const int piece = tid & 7, lrow = tid >> 3;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;

  int rbase[PA], rix[PA], riy[PA], riz[PA];
#pragma unroll
  for (int a = 0; a < PA; ++a) {
    int m = m0 + lrow + 32 * a;
    bool ok = m < p.M;
    int oz = m % p.Zo; int r = m / p.Zo;
    int oy = r % p.Yo; r /= p.Yo;
    int ox = r % p.Xo; int b = r / p.Xo;
    rbase[a] = ok ? b : -1;
    rix[a] = ox * p.stride - p.px;
    riy[a] = oy * p.stride - p.py;
    riz[a] = oz * p.stride - p.pz;
  }
  // iterations of the fp32 pack are 32-channel chunks ordered (channel chunk, tap); one K step here = two of them
  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);

  f32x4 ra[2][PA], rb[2][PB];
  auto gload = [&](int it2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int it = it2 + half;
      const bool live = it < it1;
      const int kc = it / p.taps, t = it - kc * p.taps;
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (live) v = *(const f32x4*)(p.w + wfrag_index((size_t)it, p.Npad >> 7, n0 + lrow + 32 * b, piece * 4));
        rb[half][b] = v;
      }
      const int cc = kc * KC + piece * 4;
      const bool cok = live && cc < p.Cin;
      const int kw = t % p.kz; int r = t / p.kz;
      const int kh = r % p.ky, kd = r / p.ky;
#pragma unroll
      for (int a = 0; a < PA; ++a) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        int ix = rix[a] + kd, iy = riy[a] + kh, iz = riz[a] + kw;
        bool ok = rbase[a] >= 0 && cok && (unsigned)ix < (unsigned)p.Xi && (unsigned)iy < (unsigned)p.Yi &&
                  (unsigned)iz < (unsigned)p.Zi;
        if (ok) {
          size_t row = (((size_t)rbase[a] * p.Xi + ix) * p.Yi + iy) * p.Zi + iz;
          v = *(const f32x4*)(p.in + row * p.in_stride + cc);
        }
        ra[half][a] = v;
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int a = 0; a < PA; ++a) *(bf16x4*)&As[buf][(lrow + 32 * a) * BF_ST + half * 32 + piece * 4] = to_bf16x4(ra[half][a]);
#pragma unroll
      for (int b = 0; b < PB; ++b) *(bf16x4*)&Bs[buf][(lrow + 32 * b) * BF_ST + half * 32 + piece * 4] = to_bf16x4(rb[half][b]);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (it0 < it1) {
    gload(it0);
    lstore(0);
  }
  __syncthreads();
  int cur = 0;
  for (int it = it0; it < it1; it += 2) {
    const bool more = it + 2 < it1;
    if (more) gload(it + 2);
    const __bf16* Ab = &As[cur][(wm * 64 + li) * BF_ST + h * 8];
    const __bf16* Bb = &Bs[cur][(wn * 64 + li) * BF_ST + h * 8];
#pragma unroll
    for (int s = 0; s < BF_KC / 16; ++s) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(Ab + i * 32 * BF_ST + s * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(Bb + j * 32 * BF_ST + s * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splitk > 1) {
          p.ws[((size_t)blockIdx.y * p.M + m) * p.Npad + n] = v;
        } else if (n < p.Cout) {
          p.out[(size_t)m * p.out_stride + n] = epilogue(p, v, n, (size_t)m);
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_bf16w (mfma_dtype 2): the same arithmetic as k_conv_bf16 with both operands ALREADY bf16 in memory -- activations as
// [rows][Cin] bf16 (coocc_rows_to_bf16: same RNE rounding, done once per layer instead of 27 times per element inside the K
// loop), weights packed once.  k_conv_bf16 moved 64 KB of fp32 operands per 128 x 128 x 64 step through registers (convert,
// ds_write): 128 B/clk per workgroup against ~64 B/clk a CU can pull from L2, which held it at 0.13 of the bf16 peak.
// * The activation tile is staged by global_load_lds (16 B per lane straight into LDS: no staging VGPRs, no conversion VALU
//   work, no ds_write pass).  Its LDS image is lane-linear (a wave's 64 lanes fill 8 rows x 128 B), so the bank-conflict
//   swizzle sits on the SOURCE side: 16-byte slot c of row r holds channel chunk c ^ ((r >> 1) & 7), and the fragment reads
//   apply the same XOR (SQ_LDS_BANK_CONFLICT = 0).  Out-of-range taps / rows read 16 zero bytes from `zrow`.
// * The WEIGHTS stay out of LDS.  A global_load_lds costs 100-185 issue cycles in a phase that also carries ds_reads and MFMAs
//   (MI355X guide); with both tiles staged that way (first version, "k_conv_bf16g": 620-700 TFLOP/s) 8 of them per 16 MFMAs
//   (512 cycles) bounded the K loop -- MFMA pipe busy 36 %, no bank conflicts, little instruction wait, and neither K steps of
//   32 at four workgroups per CU nor a 256 x 128 tile with three LDS stages, counted vmcnt and raw barriers moved it.  Here
//   the pack is fragment-major -- [(chunk, tap)][Npad/32][4 k-steps][64 lanes][8 bf16]: lane l of a wave holds B[k = 16 s +
//   8 (l >> 5) + 0..7][n = 32 nt + (l & 31)], one ordinary 16-byte load per fragment, 1 KB coalesced per wave instruction -- and
//   each wave loads its own 2 x 4 fragments one K step ahead into registers (as k_conv2 does for fp32): 4 global_load_lds + 8
//   plain loads per step, 16 KB of LDS per stage.
// * The (chunk, tap) cursor and the per-row base offsets are advanced incrementally (five integer divisions per K step cost
//   13 %).  One barrier per K step; loads of step i+1 in flight under the MFMAs of step i.  128 x 128 tile, 2 x 2 waves of
//   64 x 64; Cin % 64 == 0.
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_conv_bf16w(ConvK p) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2, BK = 64;
  __shared__ __attribute__((aligned(16))) __bf16 As[2][BM * BK];

  const int id = blockIdx.x;
  int mtile, nt, slot_ = id >> 3;
  if (p.mtiles_per_xcd > 0) {
    const int xcd = id & 7;
    const int mt_local = slot_ / p.ntiles;
    nt = slot_ - mt_local * p.ntiles;
    mtile = xcd * p.mtiles_per_xcd + mt_local;
    if (mt_local >= p.mtiles_per_xcd || mtile >= p.mtiles) return;
  } else {
    mtile = id / p.ntiles;
    nt = id - mtile * p.ntiles;
  }
  const int m0 = mtile * BM, n0 = nt * BN;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, slot = lane & 7;

  int rix[4], riy[4], riz[4];
  long long abase[4];
  const long long rowbytes = (long long)p.in_stride * 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (j * 4 + wave) * 8 + srow;
    const int m = m0 + r;
    int oz = m % p.Zo; int q = m / p.Zo;
    int oy = q % p.Yo; q /= p.Yo;
    int ox = q % p.Xo; int b = q / p.Xo;
    rix[j] = m < p.M ? ox * p.stride - p.px : -(1 << 20);
    riy[j] = oy * p.stride - p.py;
    riz[j] = oz * p.stride - p.pz;
    const unsigned qo = (unsigned)((slot ^ ((r >> 1) & 7)) * 16);
    abase[j] = ((((long long)b * p.Xi + rix[j]) * p.Yi + riy[j]) * p.Zi + riz[j]) * rowbytes + qo;
  }
  const char* inb = (const char*)p.in;
  const char* zrow = (const char*)p.zrow;
  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);
  const int nsteps = it1 - it0;
  int ckc = it0 / p.taps, ct = it0 - ckc * p.taps;
  int ckw = ct % p.kz, ckh = (ct / p.kz) % p.ky, ckd = ct / (p.kz * p.ky);
  // this wave's two 32-column fragments tiles of the pack: 4 KB per (iteration, tile)
  const long long wstep = (long long)(p.Npad >> 5) * 4096;
  const char* wcur = (const char*)p.w + (long long)it0 * wstep + (long long)((n0 >> 5) + wn * 2) * 4096 + lane * 16;

  auto issueA = [&](int buf) {
    const long long tapoff = (((long long)ckd * p.Yi + ckh) * p.Zi + ckw) * rowbytes + (long long)ckc * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = (unsigned)(rix[j] + ckd) < (unsigned)p.Xi && (unsigned)(riy[j] + ckh) < (unsigned)p.Yi &&
                      (unsigned)(riz[j] + ckw) < (unsigned)p.Zi;
      const char* src = ok ? inb + (abase[j] + tapoff) : zrow;
      glds16(src, &As[buf][(j * 4 + wave) * 8 * 64]);
    }
    if (++ckw == p.kz) { ckw = 0; if (++ckh == p.ky) { ckh = 0; if (++ckd == p.kx) { ckd = 0; ++ckc; } } }
  };
  bf16x8 breg[2][4][TN];
  auto loadB = [&](auto bufc) {
    constexpr int B_ = decltype(bufc)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < TN; ++j) breg[B_][s][j] = *(const bf16x8*)(wcur + j * 4096 + s * 1024);
    wcur += wstep;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sw = (li >> 1) & 7;
  auto step = [&](auto curc, bool more) {
    constexpr int C_ = decltype(curc)::value;
    if (more) {
      issueA(C_ ^ 1);
      loadB(std::integral_constant<int, C_ ^ 1>{});
    }
    const __bf16* Ab = &As[C_][(wm * 64 + li) * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = ((s * 2 + h) ^ sw) * 8;
      bf16x8 a[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(Ab + i * 32 * 64 + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], breg[C_][s][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();          // every wave done with this stage; the next step's tile and fragments have landed
  };

  if (nsteps > 0) {
    issueA(0);
    loadB(std::integral_constant<int, 0>{});
  }
  __syncthreads();
  for (int st = 0; st < nsteps; st += 2) {
    step(std::integral_constant<int, 0>{}, st + 1 < nsteps);
    if (st + 1 < nsteps) step(std::integral_constant<int, 1>{}, st + 2 < nsteps);
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splitk > 1) {
          p.ws[((size_t)blockIdx.y * p.M + m) * p.Npad + n] = v;
        } else if (n < p.Cout) {
          p.out[(size_t)m * p.out_stride + n] = epilogue(p, v, n, (size_t)m);
        }
      }
    }
}

// k_conv_bf16z: k_conv_bf16w for stride-1 "same" convolutions (every 3x3x3 layer of C0-C3 but the three strided ones), with the
// activation tile shared by the z taps.  Output rows are linear in (x, y, z), z fastest, so the rows tap (dx, dy, dz) needs for a
// tile of 128 consecutive outputs are the tile's own rows shifted by ((dx-1) Y + (dy-1)) Z + (dz-1): ONE LDS image of 128 + KZ - 1
// rows per (channel chunk, dx, dy) serves all KZ z taps -- the fragments of tap dz are read dz rows further down -- instead of
// KZ images of 128 rows: 4-5 global_load_lds per KZ x 16 MFMAs instead of 4 per 16, and one barrier per KZ taps.  Linear
// neighbours are not always spatial neighbours (a column's first voxel sits next to the previous column's last): a lane zeroes
// its A fragment when its output voxel's tap falls outside the grid (v_cndmask under the MFMAs), which is also exactly the zero
// padding.  The next group's image is issued at the first tap of the current group (three taps of MFMAs to land), the weight
// fragments of the next tap are loaded into the other register set during the current one.
template <int KZ>
__global__ __launch_bounds__(256, 2) void k_conv_bf16z(ConvK p) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
  constexpr int AROWS = 136;                       // 128 + KZ - 1 rounded up to whole 8-row wave instructions
  __shared__ __attribute__((aligned(16))) __bf16 As[2][AROWS * 64];

  const int id = blockIdx.x;
  int mtile, nt, slot_ = id >> 3;
  if (p.mtiles_per_xcd > 0) {
    const int xcd = id & 7;
    const int mt_local = slot_ / p.ntiles;
    nt = slot_ - mt_local * p.ntiles;
    mtile = xcd * p.mtiles_per_xcd + mt_local;
    if (mt_local >= p.mtiles_per_xcd || mtile >= p.mtiles) return;
  } else {
    mtile = id / p.ntiles;
    nt = id - mtile * p.ntiles;
  }
  const int m0 = mtile * BM, n0 = nt * BN;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, slot = lane & 7;

  // staging: LDS row r holds input row  m0 + r - pz + ((dx - px) Yi + (dy - py)) Zi  (any row of the buffer, else zeros)
  const long long rowbytes = (long long)p.in_stride * 2;
  const long long total_rows = (long long)p.Xi * p.Yi * p.Zi * (p.M / ((long long)p.Xo * p.Yo * p.Zo));
  long long arow[5];          // m0 + r - pz for this lane's rows (instruction 4: rows 128..135, issued by wave 0 only)
  unsigned aq[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int r = j < 4 ? (j * 4 + wave) * 8 + srow : 128 + srow;
    arow[j] = (long long)m0 + r - p.pz;
    aq[j] = (unsigned)((slot ^ ((r >> 1) & 7)) * 16);
  }
  // the two output voxels this lane's A fragments belong to (fragment i: row wm*64 + i*32 + li)
  int vx[TM], vy[TM], vz[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 64 + i * 32 + li;
    int oz = m % p.Zo; int q = m / p.Zo;
    int oy = q % p.Yo; q /= p.Yo;
    vx[i] = m < p.M ? q % p.Xo : -(1 << 20);
    vy[i] = oy; vz[i] = oz;
  }
  const char* inb = (const char*)p.in;
  const char* zrow = (const char*)p.zrow;
  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);
  const int ngroups = (it1 - it0) / KZ;                 // the launcher keeps split boundaries on whole (dx, dy) groups
  // cursors: `g*` = the group whose image is issued next, `c*` = the group being computed
  const int g0 = it0 / KZ;                              // group index = (chunk * kx + dx) * ky + dy
  int gkc = g0 / (p.kx * p.ky), gd = (g0 / p.ky) % p.kx, gh = g0 % p.ky;
  int cd = gd, ch_ = gh;
  const long long wstep = (long long)(p.Npad >> 5) * 4096;
  const char* wcur = (const char*)p.w + (long long)it0 * wstep + (long long)((n0 >> 5) + wn * 2) * 4096 + lane * 16;

  auto issueA = [&](int buf) {
    const long long off = ((long long)(gd - p.px) * p.Yi + (gh - p.py)) * p.Zi;
    const long long coff = (long long)gkc * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long L = arow[j] + off;
      const char* src = (L >= 0 && L < total_rows) ? inb + L * rowbytes + coff + aq[j] : zrow;
      glds16(src, &As[buf][(j * 4 + wave) * 8 * 64]);
    }
    if (wave == 0) {
      const long long L = arow[4] + off;
      const char* src = (L >= 0 && L < total_rows) ? inb + L * rowbytes + coff + aq[4] : zrow;
      glds16(src, &As[buf][128 * 64]);
    }
    if (++gh == p.ky) { gh = 0; if (++gd == p.kx) { gd = 0; ++gkc; } }
  };
  bf16x8 breg[2][4][TN];
  auto loadB = [&](auto bufc) {
    constexpr int B_ = decltype(bufc)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < TN; ++j) breg[B_][s][j] = *(const bf16x8*)(wcur + j * 4096 + s * 1024);
    wcur += wstep;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bf16x8 zero8 = {};
  // one tap: MFMAs of tap DZ of the group in stage AB_ with the weight set BB_; loads the next tap's weights into the other set
  auto tap = [&](auto abufc, auto bbufc, auto dzc, bool more_taps, bool xy0, bool xy1) {
    constexpr int AB_ = decltype(abufc)::value, BB_ = decltype(bbufc)::value, DZ = decltype(dzc)::value;
    if (more_taps) loadB(std::integral_constant<int, BB_ ^ 1>{});
    const bool ok0 = xy0 && (unsigned)(vz[0] + DZ - p.pz) < (unsigned)p.Zi;
    const bool ok1 = xy1 && (unsigned)(vz[1] + DZ - p.pz) < (unsigned)p.Zi;
    const int r0 = wm * 64 + li + DZ, r1 = r0 + 32;
    const __bf16* A0 = &As[AB_][r0 * 64];
    const __bf16* A1 = &As[AB_][r1 * 64];
    const int k0 = (r0 >> 1) & 7, k1 = (r1 >> 1) & 7;
    // all eight fragments of the tap first: one LDS latency per tap instead of one per k-step
    bf16x8 a0[4], a1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a0[s] = *(const bf16x8*)(A0 + ((s * 2 + h) ^ k0) * 8);
      a1[s] = *(const bf16x8*)(A1 + ((s * 2 + h) ^ k1) * 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 f0 = ok0 ? a0[s] : zero8, f1 = ok1 ? a1[s] : zero8;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, breg[BB_][s][j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, breg[BB_][s][j], acc[1][j], 0, 0, 0);
      }
    }
  };
  // one (chunk, dx, dy) group in stage GP: KZ taps; weight-set parity of its first tap = (GP * KZ) & 1
  auto group = [&](auto gpc, int g) {
    constexpr int GP = decltype(gpc)::value;
    if (g + 1 < ngroups) issueA(GP ^ 1);
    const bool xy0 = (unsigned)(vx[0] + cd - p.px) < (unsigned)p.Xi && (unsigned)(vy[0] + ch_ - p.py) < (unsigned)p.Yi;
    const bool xy1 = (unsigned)(vx[1] + cd - p.px) < (unsigned)p.Xi && (unsigned)(vy[1] + ch_ - p.py) < (unsigned)p.Yi;
    const bool last = g + 1 >= ngroups;
    tap(gpc, std::integral_constant<int, (GP * KZ) & 1>{}, std::integral_constant<int, 0>{}, KZ > 1 || !last, xy0, xy1);
    if constexpr (KZ > 1)
      tap(gpc, std::integral_constant<int, (GP * KZ + 1) & 1>{}, std::integral_constant<int, 1>{}, KZ > 2 || !last, xy0, xy1);
    if constexpr (KZ > 2)
      tap(gpc, std::integral_constant<int, (GP * KZ + 2) & 1>{}, std::integral_constant<int, 2>{}, !last, xy0, xy1);
    if (++ch_ == p.ky) { ch_ = 0; if (++cd == p.kx) cd = 0; }
    __syncthreads();          // every wave done with this stage; the next group's image has landed
  };

  if (ngroups > 0) {
    issueA(0);
    loadB(std::integral_constant<int, 0>{});
  }
  __syncthreads();
  for (int g = 0; g < ngroups; g += 2) {
    group(std::integral_constant<int, 0>{}, g);
    if (g + 1 < ngroups) group(std::integral_constant<int, 1>{}, g + 1);
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (p.splitk > 1) {
          p.ws[((size_t)blockIdx.y * p.M + m) * p.Npad + n] = v;
        } else if (n < p.Cout) {
          p.out[(size_t)m * p.out_stride + n] = epilogue(p, v, n, (size_t)m);
        }
      }
    }
}

// fp32 rows (row stride in_stride floats, first C columns) -> dense bf16 rows [rows][C], RNE: the operand of k_conv_bf16w
__global__ __launch_bounds__(256) void k_rows_to_bf16(const float* __restrict__ in, int in_stride, long long rows, int C,
                                                       __bf16* __restrict__ out) {
  const int c8 = C >> 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c8) return;
  const long long r = i / c8;
  const int c = (int)(i - r * c8) * 8;
  const f32x4 a = *(const f32x4*)(in + r * in_stride + c), b = *(const f32x4*)(in + r * in_stride + c + 4);
  bf16x8 o;
  o[0] = (__bf16)a[0]; o[1] = (__bf16)a[1]; o[2] = (__bf16)a[2]; o[3] = (__bf16)a[3];
  o[4] = (__bf16)b[0]; o[5] = (__bf16)b[1]; o[6] = (__bf16)b[2]; o[7] = (__bf16)b[3];
  *(bf16x8*)(out + r * C + c) = o;
}

extern "C" int coocc_rows_to_bf16(const float* in, int in_stride, int64_t rows, int C, void* out_bf16, void* stream) {
  COOCC_CHECK_ARG(in && out_bf16 && rows >= 0 && C > 0 && C % 8 == 0 && in_stride % 4 == 0 && in_stride >= C, "rows_to_bf16: bad args");
  COOCC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)out_bf16 & 15) == 0, "rows_to_bf16: pointers must be 16-byte aligned");
  if (rows == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_rows_to_bf16, dim3(cdiv(rows * (C / 8), 256)), dim3(256), 0, as_stream(stream), in, in_stride, (long long)rows, C,
                     (__bf16*)out_bf16);
  COOCC_LAUNCH_CHECK("k_rows_to_bf16");
  return COOCC_OK;
}

// ---------------------------------------------------------------------------------------------
// k_conv2: software-pipelined variant for the large geometric layers (Cout >= 128, M >= 8192).
// Ablation of the phase-structured k_conv on MI355X (profiles/r1_conv_ablation.txt): MFMAs alone
// 142 TFLOP/s, + fragment reads 139, + LDS stores and barrier 125, + global loads 108.  A lone wave
// keeps its SIMD's matrix pipe busy only if the other work of a K-chunk is issued BETWEEN its 80
// MFMAs, and barrier skew is only hidden by a second resident workgroup.  Hence:
//   * B (weights) never touches LDS: the packed layout is fragment-major, each wave loads its own
//     4 x 16 B per lane per chunk with buffer_load, one chunk ahead.  LDS holds only the A tile
//     (160 x 32, double-buffered 46 KB) -> two workgroups per CU (500 tiles = one round of 512 slots
//     for M = 80000, N = 128);
//   * A loads are branch-free: buffer_load with an out-of-range offset returns 0 (zero padding, M and
//     Cin tails), so the loop body is one basic block;
//   * 1 x 4 wave layout (each wave BM x 32), A fragments double-buffered in VGPRs: the LDS reads of
//     k-group q+1 are issued under the MFMAs of group q;
//   * the barrier sits before the last MFMA group: LDS stores of chunk i+1, the barrier and the first
//     fragment reads of chunk i+1 are covered by MFMAs of chunk i.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// PF = chunks in flight between a chunk's global loads and its use.  vmcnt retires in order, so A and
// B loads share one depth: waiting for chunk i+1's A rows must not drain younger loads.  PF = 1 is
// enough while the weights stay cache-resident (many M tiles re-read them); the weight-streaming
// layers (few M tiles, up to 113 MB of weights each) need the HBM latency of 2-3 chunks covered.
// WG = true: the weight-grouped launch of the Winograd path (one weight pack per transform point); a separate
// instantiation so that profilers list it under its own name.
template <int BM, int PF = 1, bool WG = false, int MINW = 2, bool TB = false>
__global__ __launch_bounds__(256, MINW) void k_conv2(ConvK p) {
  constexpr int TM = BM / 32, PA = BM / 32;
  __shared__ float As[2][BM * LDS_ST];
  // TB (row-table mode: sparse convolutions, strided dgrad classes, the GSFusion gather GEMMs): the tile's slice of the
  // [taps][M] table sits in LDS, so the dependent pair (table entry -> feature row) never adds a second global latency
  // and the lookups do not share the in-order vmcnt queue with the prefetched rows
  __shared__ int Ti[TB ? 27 * BM : 1];

  const int id = blockIdx.x;
  int mtile, nt, slot = id >> 3;
  if (p.mtiles_per_xcd > 0) {
    const int xcd = id & 7;
    const int mt_local = slot / p.ntiles;
    nt = slot - mt_local * p.ntiles;
    mtile = xcd * p.mtiles_per_xcd + mt_local;
    if (mt_local >= p.mtiles_per_xcd || mtile >= p.mtiles) return;
  } else {   // few M tiles: plain order, consecutive ids (= different XCDs) take different tiles
    mtile = id / p.ntiles;
    nt = id - mtile * p.ntiles;
  }
  const int m0 = mtile * BM, n0 = nt * 128;
  if (TB && p.M_dev) {                 // row count on the device (grid sized for the capacity p.M): whole tiles past it leave
    p.M = min(p.M, *p.M_dev);
    if (m0 >= p.M) return;
  }

  const int tid = threadIdx.x;
  const int piece = tid & 7, lrow = tid >> 3;
  // wave index as an SGPR: it feeds the scalar offset of the B loads (a VGPR there makes hipcc
  // wrap every buffer_load in a waterfall loop)
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;

  const float* wbase = WG ? p.w + (size_t)(m0 / p.wgroup_rows) * p.wgroup_floats : p.w;
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, p.w_bytes, 0x00020000);

  // per-thread A rows: voxel coordinates of tap (0,0,0) and its row index; rows past M are parked
  // outside the grid so every tap fails the bounds test
  int cx[PA], cy[PA], cz[PA], rrow[PA];
  size_t base_off = 0;
  if (TB) {
    for (int i = tid; i < p.taps * BM; i += 256) {
      const int t = i / BM, m = m0 + (i - t * BM);
      Ti[i] = m < p.M ? p.gather[(size_t)t * p.gstride + m] : -1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < PA; ++a) {
    int m = m0 + lrow + 32 * a;
    int oz = m % p.Zo; int r = m / p.Zo;
    int oy = r % p.Yo; r /= p.Yo;
    int ox = r % p.Xo; int b = r / p.Xo;
    bool ok = m < p.M;
    cx[a] = ok ? ox * p.stride - p.px : -4096;
    cy[a] = oy * p.stride - p.py;
    cz[a] = oz * p.stride - p.pz;
    rrow[a] = ((b * p.Xi + cx[a]) * p.Yi + cy[a]) * p.Zi + cz[a];
  }
  // Tile-relative addressing: input rows are visited in the lexicographic order of the outputs, so every row this
  // tile touches lies in a short window above the row of (m0, tap 0).  The buffer descriptor is based there
  // (64-bit pointer) and the per-lane offsets stay 32-bit for inputs of any size (the Winograd V buffer of a
  // 200x200x16x512 volume is 5.2 GB).
  if (!TB) {
    int oz = m0 % p.Zo; int r = m0 / p.Zo;
    int oy = r % p.Yo; r /= p.Yo;
    int ox = r % p.Xo; int b = r / p.Xo;
    int row0 = ((b * p.Xi + ox * p.stride - p.px) * p.Yi + oy * p.stride - p.py) * p.Zi + oz * p.stride - p.pz;
    row0 = max(row0, 0);
#pragma unroll
    for (int a = 0; a < PA; ++a) rrow[a] -= row0;
    base_off = (size_t)row0 * p.in_stride * 4;
  }
  const size_t remain = p.in_bytes > base_off ? p.in_bytes - base_off : 0;
  __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + base_off), 0,
                                                                    (unsigned)(remain < 0xFFFFFF00ull ? remain : 0xFFFFFF00ull), 0x00020000);
  const unsigned voffB = (unsigned)(lane * 16);
  const int ngroups = p.Npad >> 7;

  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);
  if (it0 >= it1) return;

  f32x4 rq[PF][PA], bq[PF][4], bcur[4];   // slot 0 = chunk it+1 (oldest) ... slot PF-1 = chunk it+PF (just issued)
  // uniform cursor (tap (kd,kh,kw), channel chunk kc, chunk index lc) of the chunk being LOADED
  int lc = it0, lkc = it0 / p.taps;
  const int lt0 = it0 - lkc * p.taps;
  int lkw = lt0 % p.kz, lkh = (lt0 / p.kz) % p.ky, lkd = lt0 / (p.kz * p.ky);
  auto issue_loads = [&](bool live, int slot_) {
    const unsigned soff = (unsigned)(((((size_t)lc * ngroups + nt) * 4 + wn) * 1024) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[slot_][q] = buf_load4(rs_w, live ? voffB + q * 1024u : 0xFFFFFFF0u, soff);
    const int drow = (lkd * p.Yi + lkh) * p.Zi + lkw;
    const int cc = lkc * KC + piece * 4;
    const bool cok = live & (cc < p.Cin);
    if (TB) {
      // lkw is the tap index here (kz = taps, ky = kx = 1 are set by the host for this mode)
#pragma unroll
      for (int a = 0; a < PA; ++a) {
        const int idx = Ti[lkw * BM + lrow + 32 * a];
        const bool ok = cok & (idx >= 0);
        unsigned voff = ok ? (unsigned)(idx * p.in_stride + cc) * 4u : 0xFFFFFFF0u;
        rq[slot_][a] = buf_load4(rs_in, voff, 0);
      }
    } else {
#pragma unroll
    for (int a = 0; a < PA; ++a) {
      // bitwise, not short-circuit: keeps the body free of control flow
      const bool ok = cok & ((unsigned)(cx[a] + lkd) < (unsigned)p.Xi) & ((unsigned)(cy[a] + lkh) < (unsigned)p.Yi) &
                      ((unsigned)(cz[a] + lkw) < (unsigned)p.Zi);
      unsigned voff = ok ? (unsigned)((rrow[a] + drow) * p.in_stride + cc) * 4u : 0xFFFFFFF0u;
      rq[slot_][a] = buf_load4(rs_in, voff, 0);
    }
    }
    // advance the cursor (scalar, branch-free so the loop body stays one basic block)
    lc += 1; lkw += 1;                         // taps innermost, then the next channel chunk
    const int w2 = lkw == p.kz;
    lkw = w2 ? 0 : lkw; lkh += w2;
    const int w3 = lkh == p.ky;
    lkh = w3 ? 0 : lkh; lkd += w3;
    const int w4 = lkd == p.kx;
    lkd = w4 ? 0 : lkd; lkc += w4;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int a = 0; a < PA; ++a) *(f32x4*)&As[buf][(lrow + 32 * a) * LDS_ST + piece * 4] = rq[0][a];
  };

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  f32x4 fa[2][TM];
  auto lfrag = [&](int buf, int q, int slot_) {
    const float* Ab = &As[buf][li * LDS_ST + h * 4 + q * 8];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[slot_][i] = *(const f32x4*)(Ab + i * 32 * LDS_ST);
  };
  auto mma = [&](int slot_, int q) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot_][i][s], bcur[q][s], acc[i], 0, 0, 0);
  };

  // prologue: chunk it0 -> LDS / bcur, chunks it0+1 .. it0+PF-1 -> queue slots 0 .. PF-2
  issue_loads(true, 0);
  lstore(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) bcur[q] = bq[0][q];
#pragma unroll
  for (int s_ = 0; s_ + 1 < PF; ++s_) issue_loads(it0 + 1 + s_ < it1, s_);
  __syncthreads();
  lfrag(0, 0, 0);
  int cur = 0;
  for (int it = it0; it < it1; ++it) {
    lfrag(cur, 1, 1);
    issue_loads(it + PF < it1, PF - 1);   // chunk it+PF -> newest slot (all-zero dummy past the end)
    __builtin_amdgcn_sched_barrier(0);        // every load is in flight before the first MFMA (hipcc sinks them otherwise)
    mma(0, 0);
    lfrag(cur, 2, 0);
    mma(1, 1);
    lfrag(cur, 3, 1);
    mma(0, 2);
    lstore(cur ^ 1);             // A tile of chunk it+1 -> other LDS buffer
    __syncthreads();
    lfrag(cur ^ 1, 0, 0);        // first fragments of chunk it+1, under the last MFMA group
    mma(1, 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) bcur[q] = bq[0][q];
#pragma unroll
    for (int s_ = 0; s_ + 1 < PF; ++s_) {       // rotate the queue (register moves, ~1 % of the chunk)
#pragma unroll
      for (int a = 0; a < PA; ++a) rq[s_][a] = rq[s_ + 1][a];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[s_][q] = bq[s_ + 1][q];
    }
    cur ^= 1;
  }

  // epilogue: D row = (r&3) + 8*(r>>2) + 4*h, col = li
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m >= p.M) continue;
      float v = acc[i][r];
      if (p.splitk > 1) {
        p.ws[((size_t)blockIdx.y * p.M + m) * p.Npad + n] = v;
      } else if (n < p.Cout) {
        size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
        p.out[orow * p.out_stride + n] = epilogue(p, v, n, orow);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_conv2p: persistent form of k_conv2<128, PF = 1, WG> for the Winograd-domain grouped GEMMs with short K (<= 24
// K-chunks per tile; 12 launches per sample, the dominant kernel of the decoder).  With 12-24 chunks per tile the
// per-tile prologue (first loads -> LDS -> barrier -> first fragments) and the workgroup launch weigh as much as a
// few chunks.  Here a workgroup walks tiles id, id + gridDim.x, ...: on the LAST chunk of a tile the prefetch slot that
// k_conv2 fills with dummy zeros takes chunk 0 of the NEXT tile, so that chunk's LDS store, barrier and first fragment
// reads happen under the last MFMA group of the current tile exactly like any other chunk; only the epilogue stores
// and the accumulator reset sit between two tiles.  Specialised to row-linear geometries, which keeps the per-tile state
// to 8 VGPRs: (WG) the grouped launches (rows = (tile, z), taps = the 3 z neighbours: kx = ky = 1, Xi = Yi = 1,
// stride 1, bare products out), and (EPI) 1x1x1 stride-1 layers (input row = output row) with the usual epilogue --
// the per-voxel render heads and the lateral / prediction 1x1 convs, whose K is only 4-8 chunks.
template <bool WG, bool EPI>
__global__ __launch_bounds__(256, 3) void k_conv2p(ConvK p) {
  constexpr int BM = 128, TM = BM / 32, PA = BM / 32;
  __shared__ float As[2][BM * LDS_ST];
  const int tid = threadIdx.x;
  const int piece = tid & 7, lrow = tid >> 3;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  const int nslots = p.mtiles_per_xcd > 0 ? 8 * p.mtiles_per_xcd * p.ntiles : p.mtiles * p.ntiles;
  auto decode = [&](int id, int& mtile, int& nt_) -> bool {
    if (p.mtiles_per_xcd > 0) {
      const int xcd = id & 7, slot = id >> 3;
      const int mt_local = slot / p.ntiles;
      nt_ = slot - mt_local * p.ntiles;
      mtile = xcd * p.mtiles_per_xcd + mt_local;
      return mt_local < p.mtiles_per_xcd && mtile < p.mtiles;
    }
    mtile = id / p.ntiles;
    nt_ = id - mtile * p.ntiles;
    return true;
  };

  // state of the tile being LOADED: z of tap 0 per row (parked far outside for rows past M) and its input row
  int cz[PA], rrow[PA];
  __amdgpu_buffer_rsrc_t rs_w, rs_in;
  int m0 = 0, n0 = 0, nt = 0;
  int lc = 0, lkc = 0, lkw = 0;
  auto setup = [&](int mtile, int nt_) {
    m0 = mtile * BM; n0 = nt_ * 128; nt = nt_;
    const float* wbase = WG ? p.w + (size_t)(m0 / p.wgroup_rows) * p.wgroup_floats : p.w;
    rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, p.w_bytes, 0x00020000);
    const int row0 = max(m0 - p.pz, 0);
#pragma unroll
    for (int a = 0; a < PA; ++a) {
      const int m = m0 + lrow + 32 * a;
      cz[a] = m < p.M ? m % p.Zi - p.pz : -4096;
      rrow[a] = m - p.pz - row0;
    }
    const size_t base_off = (size_t)row0 * p.in_stride * 4;
    const size_t remain = p.in_bytes > base_off ? p.in_bytes - base_off : 0;
    rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.in + base_off), 0,
                                              (unsigned)(remain < 0xFFFFFF00ull ? remain : 0xFFFFFF00ull), 0x00020000);
    lc = 0; lkc = 0; lkw = 0;
  };
  const unsigned voffB = (unsigned)(lane * 16);
  const int ngroups = p.Npad >> 7;

  f32x4 rq[PA], bq[4], bcur[4];
  auto issue_loads = [&](bool live) {
    const unsigned soff = (unsigned)(((((size_t)lc * ngroups + nt) * 4 + wn) * 1024) * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = buf_load4(rs_w, live ? voffB + q * 1024u : 0xFFFFFFF0u, soff);
    const int cc = lkc * KC + piece * 4;
    const bool cok = live & (cc < p.Cin);
#pragma unroll
    for (int a = 0; a < PA; ++a) {
      const bool ok = cok & ((unsigned)(cz[a] + lkw) < (unsigned)p.Zi);
      unsigned voff = ok ? (unsigned)((rrow[a] + lkw) * p.in_stride + cc) * 4u : 0xFFFFFFF0u;
      rq[a] = buf_load4(rs_in, voff, 0);
    }
    lc += 1; lkw += 1;                           // z taps innermost, then the next channel chunk
    const int w2 = lkw == p.kz;
    lkw = w2 ? 0 : lkw; lkc += w2;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int a = 0; a < PA; ++a) *(f32x4*)&As[buf][(lrow + 32 * a) * LDS_ST + piece * 4] = rq[a];
  };
  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f32x4 fa[2][TM];
  auto lfrag = [&](int buf, int q, int slot_) {
    const float* Ab = &As[buf][li * LDS_ST + h * 4 + q * 8];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[slot_][i] = *(const f32x4*)(Ab + i * 32 * LDS_ST);
  };
  auto mma = [&](int slot_, int q) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot_][i][s], bcur[q][s], acc[i], 0, 0, 0);
  };
  int cur = 0;
  auto chunk_tail = [&]() {          // everything of a chunk after its prefetch has been issued
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0);
    lfrag(cur, 2, 0);
    mma(1, 1);
    lfrag(cur, 3, 1);
    mma(0, 2);
    lstore(cur ^ 1);
    __syncthreads();
    lfrag(cur ^ 1, 0, 0);
    mma(1, 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) bcur[q] = bq[q];
    cur ^= 1;
  };

  int id = blockIdx.x, mtile, ntl;
  while (id < nslots && !decode(id, mtile, ntl)) id += gridDim.x;
  if (id >= nslots) return;
  setup(mtile, ntl);
  int m0c = m0, n0c = n0;            // tile being COMPUTED
  issue_loads(true);
  lstore(0);
#pragma unroll
  for (int q = 0; q < 4; ++q) bcur[q] = bq[q];
  __syncthreads();
  lfrag(0, 0, 0);
  const int iters = p.total_iters;
  for (;;) {
    int nid = id + gridDim.x, nmt = 0, nnt = 0;
    while (nid < nslots && !decode(nid, nmt, nnt)) nid += gridDim.x;
    const bool has_next = nid < nslots;
    for (int it = 0; it + 1 < iters; ++it) {
      lfrag(cur, 1, 1);
      issue_loads(true);
      chunk_tail();
    }
    // last chunk of this tile: the prefetch slot takes chunk 0 of the next tile (or dummy zeros at the very end)
    lfrag(cur, 1, 1);
    if (has_next) setup(nmt, nnt);
    issue_loads(has_next);
    chunk_tail();

    // Winograd-domain products: no scale / bias / residual / ReLU here (they are applied by the output transform)
    // buffer stores based at the tile's first row: rows past M fall outside num_records and are dropped by the
    // hardware; one running 32-bit offset per lane (row steps +1,+1,+1,+5 in D-register order) instead of 64 addresses
    {
      const int rows_left = min(p.M - m0c, BM);
      const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(p.out + (size_t)m0c * p.out_stride), 0, (unsigned)rows_left * (unsigned)p.out_stride * 4u, 0x00020000);
      const int col = n0c + wn * 32 + li;
      const unsigned s1 = (unsigned)p.out_stride * 4u, s5 = 5u * s1;
      unsigned voff = (unsigned)(4 * h * p.out_stride + col) * 4u;
      if (col < p.Cout) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][r];
            if (EPI) v = epilogue(p, v, col, (size_t)(m0c + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_o, (int)voff, 0, 0);
            voff += (r & 3) == 3 ? s5 : s1;
          }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    if (!has_next) break;
    id = nid; m0c = m0; n0c = n0;
  }
}

// split-K second pass: sum the partial slabs in a fixed order, then the epilogue
__global__ __launch_bounds__(256) void k_conv_reduce(ConvK p) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.M * p.Cout) return;
  int m = (int)(i / p.Cout), n = (int)(i - (size_t)m * p.Cout);
  float v = 0.f;
  for (int z = 0; z < p.splitk; ++z) v += p.ws[((size_t)z * p.M + m) * p.Npad + n];
  size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
  p.out[orow * p.out_stride + n] = epilogue(p, v, n, orow);
}

// The same pass for the split-f16 / f16 launches, four channels per thread (Cout % 4 == 0): besides the fp32 rows it writes what
// the single-pass epilogue of those kernels can -- H2 rows instead of fp32 (out_h2), an f16 copy (out16), an H2 TWIN of the
// fp32 rows (out_h2t: the next split-f16 layer's operand, which used to be a conversion launch of its own right after this one).
__global__ __launch_bounds__(256) void k_conv_reduce4(ConvK p) {
  const int c4 = p.Cout >> 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.M * c4) return;
  const int m = (int)(i / c4), n = (int)(i - (size_t)m * c4) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const float* src = p.ws + (size_t)m * p.Npad + n;
  const size_t zs = (size_t)p.M * p.Npad;
  // every operand of the epilogue is requested before the slabs are summed, and the slabs four at a time: the launch is a chain
  // of dependent round trips otherwise (one load, one wait per slab).  The additions keep the slab order (same bits).
  const size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
  const f32x4 rs = (p.res_mode && p.res) ? *(const f32x4*)(p.res + orow * p.res_stride + n) : zero4;
  const f32x4 sc = p.scale ? *(const f32x4*)(p.scale + n) : one4;
  const f32x4 bi = p.bias ? *(const f32x4*)(p.bias + n) : zero4;
  int z = 0;
  for (; z + 4 <= p.splitk; z += 4) {
    const f32x4 a0 = *(const f32x4*)(src + (size_t)z * zs), a1 = *(const f32x4*)(src + (size_t)(z + 1) * zs);
    const f32x4 a2 = *(const f32x4*)(src + (size_t)(z + 2) * zs), a3 = *(const f32x4*)(src + (size_t)(z + 3) * zs);
    v = v + a0; v = v + a1; v = v + a2; v = v + a3;
  }
  if (z + 2 <= p.splitk) {
    const f32x4 a0 = *(const f32x4*)(src + (size_t)z * zs), a1 = *(const f32x4*)(src + (size_t)(z + 1) * zs);
    v = v + a0; v = v + a1;
    z += 2;
  }
  if (z < p.splitk) v = v + *(const f32x4*)(src + (size_t)z * zs);
  if (p.res_mode == 3) v = v + rs;
  if (p.scale) v = v * sc;
  if (p.bias) v = v + bi;
  if (p.res_mode == 1) v = v + rs;
  if (p.relu && (p.relu == 1 || n < p.relu)) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
  if (p.res_mode == 2) v = v * rs;
  if (p.out_h2) { store_h2(p.out, orow, p.out_stride, n, v); h2_guard(p.h2_flag, v); }
  else *(f32x4*)(p.out + orow * p.out_stride + n) = v;
  if (p.out16) { store_f16(p.out16, orow, p.out16_stride, n, v); h2_guard(p.h2_flag, v); }
  if (p.out_h2t) { store_h2(p.out_h2t, orow, p.Cout, n, v); h2_guard(p.h2_flag, v); }
}

// ------------------------------------------------------------------ host side
extern "C" int64_t coocc_conv_pack_weights(const float* w_host, int Cout, int Cin, int taps, int tap_major,
                                           float* packed_host) {
  if (Cout <= 0 || Cin <= 0 || taps <= 0) return coocc_set_error(COOCC_EINVAL, "pack_weights: bad dims");
  const int kch = (Cin + KC - 1) / KC;
  const int Npad = (Cout + NPAD_TO - 1) / NPAD_TO * NPAD_TO;
  const int64_t total = (int64_t)taps * kch * Npad * KC;
  if (!packed_host) return total;
  if (!w_host) return coocc_set_error(COOCC_EINVAL, "pack_weights: null weights");
  memset(packed_host, 0, sizeof(float) * (size_t)total);
  for (int t = 0; t < taps; ++t)
    for (int n = 0; n < Cout; ++n)
      for (int c = 0; c < Cin; ++c) {
        float v = tap_major ? w_host[((size_t)n * taps + t) * Cin + c] : w_host[((size_t)n * Cin + c) * taps + t];
        packed_host[wfrag_index((size_t)(c / KC) * taps + t, Npad >> 7, n, c % KC)] = v;
      }
  return total;
}

// 256 zero bytes per device: the source of padded / out-of-range rows for the global_load_lds staged kernels
int coocc_zero_row(const void** out) {
  static void* zero_row[64] = {nullptr};
  int dev = 0;
  COOCC_HIP(hipGetDevice(&dev));
  COOCC_CHECK_ARG(dev >= 0 && dev < 64, "conv_fwd: device index");
  if (!zero_row[dev]) {
    COOCC_HIP(hipMalloc(&zero_row[dev], 256));
    COOCC_HIP(hipMemset(zero_row[dev], 0, 256));
  }
  *out = zero_row[dev];
  return COOCC_OK;
}

template <int BM, int BN, int WM, int WN>
static void launch_cfg(ConvK& k, bool table, hipStream_t s) {
  k.mtiles = (k.M + BM - 1) / BM;
  k.ntiles = (k.Cout + BN - 1) / BN;
  k.mtiles_per_xcd = k.mtiles >= 64 ? (k.mtiles + 7) / 8 : 0;   // XCD slabs only pay with many M tiles
  dim3 grid(k.mtiles_per_xcd ? 8 * k.mtiles_per_xcd * k.ntiles : k.mtiles * k.ntiles, k.splitk);
  if (table)
    hipLaunchKernelGGL((k_conv<BM, BN, WM, WN, true>), grid, dim3(256), 0, s, k);
  else
    hipLaunchKernelGGL((k_conv<BM, BN, WM, WN, false>), grid, dim3(256), 0, s, k);
}

extern "C" int coocc_conv_fwd(const coocc_conv_desc* d, void* stream) {
  COOCC_CHECK_ARG(d && d->in && d->w && d->out, "conv_fwd: null pointer");
  COOCC_CHECK_ARG(d->M > 0 && d->Cin > 0 && d->Cout > 0 && d->taps > 0, "conv_fwd: bad sizes");
  COOCC_CHECK_ARG(d->Cin % 4 == 0 && d->in_stride % 4 == 0, "conv_fwd: Cin and in_stride must be multiples of 4");
  COOCC_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "conv_fwd: in/w must be 16-byte aligned");
  COOCC_CHECK_ARG(d->res_mode == 0 || d->res, "conv_fwd: res_mode set without res");
  if (!d->gather) {
    if (d->kx > 0) COOCC_CHECK_ARG(d->ky > 0 && d->kz > 0 && d->taps == d->kx * d->ky * d->kz, "conv_fwd: taps != kx*ky*kz");
    else COOCC_CHECK_ARG(d->taps == d->ksize * d->ksize * d->ksize, "conv_fwd: taps != ksize^3");
    COOCC_CHECK_ARG((long long)d->B * d->Xo * d->Yo * d->Zo == d->M, "conv_fwd: M != B*Xo*Yo*Zo");
  }
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.in = d->in; k.w = d->w; k.out = d->out; k.scale = d->scale; k.bias = d->bias; k.res = d->res;
  k.gather = d->gather; k.out_rows = d->out_rows; k.ws = d->ws;
  k.M_dev = (d->gather || d->mfma_dtype >= 3) ? d->M_dev : nullptr;
  k.gstride = d->gather_stride > 0 ? d->gather_stride : d->M;
  COOCC_CHECK_ARG(!d->M_dev || ((d->gather || d->mfma_dtype >= 3) && d->splitk == 1), "conv_fwd: M_dev needs a row table (gather) or mfma_dtype 3, and splitk = 1");
  COOCC_CHECK_ARG(k.gstride >= d->M, "conv_fwd: gather_stride smaller than M");
  k.M = d->M; k.Cin = d->Cin; k.Cout = d->Cout; k.taps = d->taps;
  k.kchunks = (d->Cin + KC - 1) / KC;
  k.Npad = (d->Cout + NPAD_TO - 1) / NPAD_TO * NPAD_TO;
  k.in_stride = d->in_stride; k.out_stride = d->out_stride; k.res_stride = d->res_stride;
  k.Xi = d->Xi; k.Yi = d->Yi; k.Zi = d->Zi; k.Xo = d->Xo; k.Yo = d->Yo; k.Zo = d->Zo;
  k.stride = d->stride;
  if (d->kx > 0) { k.kx = d->kx; k.ky = d->ky; k.kz = d->kz; k.px = d->px; k.py = d->py; k.pz = d->pz; }
  else { k.kx = k.ky = k.kz = d->ksize; k.px = k.py = k.pz = d->pad; }
  k.wgroup_rows = d->wgroup_rows;
  k.wgroup_floats = (size_t)k.taps * k.kchunks * k.Npad * KC;
  k.relu = d->relu; k.res_mode = d->res_mode;
  k.total_iters = k.taps * k.kchunks;
  if (d->mfma_dtype == 3 || d->mfma_dtype == 4) {      // H2 operands (fp32 split into two f16 halves) / one-term f16 operands: gemm_h2.hip
    hipStream_t s3 = as_stream(stream);
    const int rc = coocc_launch_h2(k, d, s3);
    if (rc != COOCC_OK) return rc;
    if (k.splitk > 1 && !k.tile_sem) {      // with arrival counters the last workgroup of every tile has reduced in-kernel
      const bool vec4 = (k.Cout & 3) == 0 && (k.out_stride & 3) == 0 && (!k.res || (k.res_stride & 3) == 0) && (((uintptr_t)k.out) & 15) == 0;
      if (vec4) hipLaunchKernelGGL(k_conv_reduce4, dim3(cdiv((long long)k.M * (k.Cout / 4), 256)), dim3(256), 0, s3, k);
      else hipLaunchKernelGGL(k_conv_reduce, dim3(cdiv((long long)k.M * k.Cout, 256)), dim3(256), 0, s3, k);
      COOCC_LAUNCH_CHECK("k_conv_reduce");
    }
    return COOCC_OK;
  }
  if (d->mfma_dtype == 2) {      // bf16 operands in memory: K steps of 64 channels
    COOCC_CHECK_ARG(d->Cin % 64 == 0 && d->in_stride % 8 == 0, "conv_fwd: bf16 operands need Cin % 64 == 0 and in_stride % 8 == 0");
    k.kchunks = d->Cin / 64;
    k.total_iters = k.taps * k.kchunks;
  }

  // tile configuration by problem shape
  int cfg;  // 0: 128x128 (64x64 waves), 1: 64x128 (32x64), 2: 128x64 (32x64), 3: 128x32 (32x32), 4: 160x128 (160x32)
  if (d->Cout <= 32) cfg = 3;
  else if (d->Cout <= 64) cfg = 2;
  else cfg = d->M >= 8192 ? 0 : 1;
  if (cfg == 0 && d->tile_hint != 128) {
    // 256 CUs x 2 resident workgroups: pick the M tile (128 or 160) that fills whole rounds of 512 slots
    // (M = 80000, N = 128: 625 tiles = 1.22 rounds with 128-row tiles, 500 = 0.98 rounds with 160-row tiles)
    auto util = [&](int bm) {
      long long t = (long long)((d->M + bm - 1) / bm) * ((d->Cout + 127) / 128);
      long long rounds = (t + 511) / 512;
      return (double)t * bm / (double)(rounds * 512) ;   // rows of useful work per slot-round
    };
    if (d->tile_hint == 160 || (d->tile_hint == 0 && util(160) > util(128) * 1.02)) cfg = 4;
  }
  const bool short_k = k.total_iters <= 24;
  static const int persist_env = getenv("COOCC_CONV_PERSIST") ? atoi(getenv("COOCC_CONV_PERSIST")) : 1;
  if (short_k && cfg == 4 && d->tile_hint != 160 && !(d->wgroup_rows > 0 && d->wgroup_rows % 128 != 0)) cfg = 0;   // 128-row tiles, 3 per CU
  // mid-size layers (512 <= M < 8192) also run the pipelined kernel with 128-row tiles + split-K;
  // below that the 64-row tile wastes fewer padded rows (M = 169 at the deepest stage)
  const bool v2small = d->M >= 512;
  const int BM = (cfg == 1 && !(v2small && !d->gather)) ? 64 : (cfg == 4 ? 160 : 128), BN = cfg == 3 ? 32 : (cfg == 2 ? 64 : 128);
  COOCC_CHECK_ARG(d->wgroup_rows == 0 || (d->wgroup_rows % BM == 0 && !d->gather),
                  "conv_fwd: wgroup_rows must be a multiple of the M tile (use a multiple of 640)");
  const long long blocks = (long long)((d->M + BM - 1) / BM) * ((d->Cout + BN - 1) / BN);
  int splitk = d->splitk;
  if (splitk <= 0) {
    splitk = 1;
    static const int target = getenv("COOCC_SPLITK_TARGET") ? atoi(getenv("COOCC_SPLITK_TARGET")) : 512;
    if (blocks < target / 2 && k.total_iters >= 16 && d->ws) {
      // fill (at most) one round of 512 resident workgroups: one more would double the time
      splitk = (int)(target / blocks);
      if (splitk > k.total_iters / 8) splitk = k.total_iters / 8;
      if (splitk > 64) splitk = 64;
      while (splitk > 1 && (long long)splitk * d->M * k.Npad > d->ws_floats) --splitk;
      if (splitk < 1) splitk = 1;
    }
  }
  if (splitk > 1) {
    COOCC_CHECK_ARG(d->ws && (long long)splitk * d->M * k.Npad <= d->ws_floats, "conv_fwd: split-K workspace too small");
  }
  k.iters_per_split = (k.total_iters + splitk - 1) / splitk;
  splitk = (k.total_iters + k.iters_per_split - 1) / k.iters_per_split;  // no empty splits
  k.splitk = splitk;

  hipStream_t s = as_stream(stream);
  const bool table = d->gather != nullptr;
  if (d->mfma_dtype == 2) {
    COOCC_CHECK_ARG(!table && d->wgroup_rows == 0 && !d->out_rows, "conv_fwd: the bf16-MFMA path covers geometric convolutions only");
    const int zrc = coocc_zero_row(&k.zrow);
    if (zrc != COOCC_OK) return zrc;
    k.ntiles = (k.Cout + 127) / 128;
    k.mtiles = (k.M + 127) / 128;
    k.mtiles_per_xcd = k.mtiles >= 64 ? (k.mtiles + 7) / 8 : 0;
    dim3 grid(k.mtiles_per_xcd ? 8 * k.mtiles_per_xcd * k.ntiles : k.mtiles * k.ntiles, k.splitk);
    // stride-1 "same" layers share the activation tile between their z taps (split boundaries on whole (dx, dy) groups)
    static const int zshare_env = getenv("COOCC_BF16_ZSHARE") ? atoi(getenv("COOCC_BF16_ZSHARE")) : 1;
    const bool same = k.stride == 1 && k.Xo == k.Xi && k.Yo == k.Yi && k.Zo == k.Zi && k.kz >= 1 && k.kz <= 3 && k.taps > 1;
    if (zshare_env && same) {
      if (k.iters_per_split % k.kz) {
        k.iters_per_split += k.kz - k.iters_per_split % k.kz;
        k.splitk = (k.total_iters + k.iters_per_split - 1) / k.iters_per_split;
      }
      dim3 gridz(grid.x, k.splitk);
      if (k.kz == 3) hipLaunchKernelGGL(k_conv_bf16z<3>, gridz, dim3(256), 0, s, k);
      else if (k.kz == 2) hipLaunchKernelGGL(k_conv_bf16z<2>, gridz, dim3(256), 0, s, k);
      else hipLaunchKernelGGL(k_conv_bf16z<1>, gridz, dim3(256), 0, s, k);
      COOCC_LAUNCH_CHECK("k_conv_bf16z");
    } else {
      hipLaunchKernelGGL(k_conv_bf16w, grid, dim3(256), 0, s, k);
      COOCC_LAUNCH_CHECK("k_conv_bf16w");
    }
    if (k.splitk > 1) {
      hipLaunchKernelGGL(k_conv_reduce, dim3(cdiv((long long)k.M * k.Cout, 256)), dim3(256), 0, s, k);
      COOCC_LAUNCH_CHECK("k_conv_reduce");
    }
    return COOCC_OK;
  }
  if (d->mfma_dtype == 1) {
    COOCC_CHECK_ARG(!table && d->wgroup_rows == 0 && !d->out_rows, "conv_fwd: the bf16-MFMA path covers geometric convolutions only");
    // K steps of two pack chunks: keep the split boundaries even
    if (k.iters_per_split & 1) { k.iters_per_split += 1; k.splitk = (k.total_iters + k.iters_per_split - 1) / k.iters_per_split; }
    COOCC_CHECK_ARG(k.splitk == 1 || (d->ws && (long long)k.splitk * d->M * k.Npad <= d->ws_floats), "conv_fwd: split-K workspace too small");
    k.mtiles = (k.M + 127) / 128;
    k.ntiles = (k.Cout + 127) / 128;
    k.mtiles_per_xcd = k.mtiles >= 64 ? (k.mtiles + 7) / 8 : 0;
    dim3 grid(k.mtiles_per_xcd ? 8 * k.mtiles_per_xcd * k.ntiles : k.mtiles * k.ntiles, k.splitk);
    hipLaunchKernelGGL(k_conv_bf16, grid, dim3(256), 0, s, k);
    COOCC_LAUNCH_CHECK("k_conv_bf16");
    if (k.splitk > 1) {
      hipLaunchKernelGGL(k_conv_reduce, dim3(cdiv((long long)k.M * k.Cout, 256)), dim3(256), 0, s, k);
      COOCC_LAUNCH_CHECK("k_conv_reduce");
    }
    return COOCC_OK;
  }
  // software-pipelined kernel for the large geometric layers (COOCC_CONV_V2=0 switches it off)
  static const int v2mode = getenv("COOCC_CONV_V2") ? atoi(getenv("COOCC_CONV_V2")) : 1;
  const unsigned long long in_bytes = (unsigned long long)d->B * d->Xi * d->Yi * d->Zi * d->in_stride * 4ull;
  const unsigned long long w_bytes = (unsigned long long)k.taps * k.kchunks * k.Npad * KC * 4ull;
  // rows a tile can touch above its first one (bound: stride-2 outputs advance the input rows up to 8x faster)
  const unsigned long long window_rows = 8ull * 160 + (unsigned long long)k.kx * d->Yi * d->Zi + (unsigned long long)k.ky * d->Zi + k.kz + 8;
  const bool window_ok = window_rows * d->in_stride * 4ull < 0xFFFFFF00ull && (long long)d->B * d->Xi * d->Yi * d->Zi < (1ll << 31);
  // row-table mode on the pipelined kernel (128-row tiles, table slice in LDS).  The caller states the number of input
  // rows in B*Xi*Yi*Zi (1 = unknown -> phase-structured kernel); 32-bit row offsets need the input below 4 GB
  const unsigned long long tb_rows = (unsigned long long)d->B * d->Xi * d->Yi * d->Zi;
  static const int v2table = getenv("COOCC_CONV_V2_TABLE") ? atoi(getenv("COOCC_CONV_V2_TABLE")) : 1;
  if (v2mode && v2table && table && cfg == 0 && d->wgroup_rows == 0 && k.taps <= 27 && tb_rows > 1 &&
      tb_rows * d->in_stride * 4ull < 0xFFFFFF00ull && w_bytes < 0xFFFFFF00ull) {
    k.in_bytes = (size_t)(tb_rows * d->in_stride * 4ull);
    k.w_bytes = (unsigned)w_bytes;
    k.kx = k.ky = 1; k.kz = k.taps; k.px = k.py = k.pz = 0;      // the kernel's cursor walks taps innermost
    k.mtiles = (k.M + 127) / 128;
    k.ntiles = (k.Cout + 127) / 128;
    k.mtiles_per_xcd = k.mtiles >= 64 ? (k.mtiles + 7) / 8 : 0;
    dim3 grid(k.mtiles_per_xcd ? 8 * k.mtiles_per_xcd * k.ntiles : k.mtiles * k.ntiles, k.splitk);
    if (short_k && splitk == 1) hipLaunchKernelGGL((k_conv2<128, 1, false, 3, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((k_conv2<128, 3, false, 2, true>), grid, dim3(256), 0, s, k);
    COOCC_LAUNCH_CHECK("k_conv2<table>");
    if (splitk > 1) {
      hipLaunchKernelGGL(k_conv_reduce, dim3(cdiv((long long)k.M * k.Cout, 256)), dim3(256), 0, s, k);
      COOCC_LAUNCH_CHECK("k_conv_reduce");
    }
    return COOCC_OK;
  }
  if (v2mode && !table && (cfg == 0 || cfg == 4 || (cfg == 1 && v2small)) && window_ok && w_bytes < 0xFFFFFF00ull) {
    k.in_bytes = (size_t)in_bytes;
    k.w_bytes = (unsigned)w_bytes;
    const int BMv = cfg == 4 ? 160 : 128;
    k.mtiles = (k.M + BMv - 1) / BMv;
    k.ntiles = (k.Cout + 127) / 128;
    k.mtiles_per_xcd = k.mtiles >= 64 ? (k.mtiles + 7) / 8 : 0;
    dim3 grid(k.mtiles_per_xcd ? 8 * k.mtiles_per_xcd * k.ntiles : k.mtiles * k.ntiles, k.splitk);
    // persistent short-K variant: 3 workgroups per CU resident, every workgroup the same number of tiles
    const unsigned nslots = grid.x;
    const unsigned per_wg = (nslots + 767) / 768;
    const unsigned pg = ((nslots + per_wg - 1) / per_wg + 7) / 8 * 8;
    const dim3 pgrid(persist_env == 2 ? nslots : (pg < nslots ? pg : nslots), 1);   // 2: debugging (one tile per workgroup, every size)
    // k_conv2p's contract: grouped launch, rows = (tile, z) with the z taps only, bare products out
    const bool persist = persist_env && splitk == 1 && (per_wg >= 2 || persist_env == 2) && k.wgroup_rows > 0 && k.kx == 1 && k.ky == 1 && k.Xi == 1 &&
                         k.Yi == 1 && k.stride == 1 && !k.scale && !k.bias && !k.res && !k.relu && !k.out_rows && k.res_mode == 0;
    // ... and the 1x1x1 stride-1 layers (input row = output row; res_mode 3 = split-K slices is excluded by splitk == 1)
    const bool persist1 = persist_env && splitk == 1 && (per_wg >= 2 || persist_env == 2) && k.wgroup_rows == 0 && k.taps == 1 &&
                          k.kx == 1 && k.ky == 1 && k.kz == 1 && k.px == 0 && k.py == 0 && k.pz == 0 && k.stride == 1 && !k.out_rows;
    static const int pf160 = getenv("COOCC_CONV_PF160") ? atoi(getenv("COOCC_CONV_PF160")) : 2;
    static const int pf128 = getenv("COOCC_CONV_PF128") ? atoi(getenv("COOCC_CONV_PF128")) : 3;
    if (k.wgroup_rows > 0) {
      if (cfg == 4) hipLaunchKernelGGL((k_conv2<160, 2, true>), grid, dim3(256), 0, s, k);
      // short K (3*Cin/32 <= 24 chunks per tile): the per-tile prologue/epilogue weighs as much as the loop, so
      // trade prefetch depth for occupancy -- PF = 1 fits 168 VGPRs = 3 workgroups per CU (0.179 -> 0.158 ms)
      else if (short_k && persist) hipLaunchKernelGGL((k_conv2p<true, false>), pgrid, dim3(256), 0, s, k);
      else if (short_k) hipLaunchKernelGGL((k_conv2<128, 1, true, 3>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((k_conv2<128, 3, true>), grid, dim3(256), 0, s, k);
    }
    else if (cfg != 4 && short_k && persist1) hipLaunchKernelGGL((k_conv2p<false, true>), pgrid, dim3(256), 0, s, k);
    else if (cfg != 4 && short_k && splitk == 1) hipLaunchKernelGGL((k_conv2<128, 1, false, 3>), grid, dim3(256), 0, s, k);
    else if (cfg == 4 && pf160 == 2) hipLaunchKernelGGL((k_conv2<160, 2>), grid, dim3(256), 0, s, k);
    else if (cfg == 4) hipLaunchKernelGGL((k_conv2<160, 1>), grid, dim3(256), 0, s, k);
    else if (pf128 == 3) hipLaunchKernelGGL((k_conv2<128, 3>), grid, dim3(256), 0, s, k);
    else if (pf128 == 2) hipLaunchKernelGGL((k_conv2<128, 2>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((k_conv2<128, 1>), grid, dim3(256), 0, s, k);
    COOCC_LAUNCH_CHECK("k_conv2");
    if (splitk > 1) {
      hipLaunchKernelGGL(k_conv_reduce, dim3(cdiv((long long)k.M * k.Cout, 256)), dim3(256), 0, s, k);
      COOCC_LAUNCH_CHECK("k_conv_reduce");
    }
    return COOCC_OK;
  }
  switch (cfg) {
    case 0: launch_cfg<128, 128, 64, 64>(k, table, s); break;
    case 1: launch_cfg<64, 128, 32, 64>(k, table, s); break;
    case 2: launch_cfg<128, 64, 32, 64>(k, table, s); break;
    case 4: launch_cfg<160, 128, 160, 32>(k, table, s); break;
    default: launch_cfg<128, 32, 32, 32>(k, table, s); break;
  }
  COOCC_LAUNCH_CHECK("k_conv");
  if (splitk > 1) {
    hipLaunchKernelGGL(k_conv_reduce, dim3(cdiv((long long)k.M * k.Cout, 256)), dim3(256), 0, s, k);
    COOCC_LAUNCH_CHECK("k_conv_reduce");
  }
  return COOCC_OK;
}
