// fp32-accurate implicit GEMM on the f16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32-MFMA rate).
//
// Every fp32 operand is split into two halves,  a = hi + lo * 2^-11,  hi = f16(a),  lo = f16((a - hi) * 2^11)
// (22 significand bits; lo is scaled so that it stays a normal f16 wherever hi is), and the product keeps the three
// leading terms in two fp32 accumulator sets:
//
//     sum a b  ~=  sum ah bh  +  2^-11 * sum (ah bl + al bh)            (the dropped al bl term is 2^-22 relative)
//
// Three MFMAs per 32x32x16 step instead of eight 32x32x2 fp32 steps at 1/16 of the rate: 5.3x the fp32-MFMA peak.
// Measured error (tools/proto/split_mfma.hip on MI355X, K = 768..13824, against fp64): rms 1.9e-7 .. 7.5e-7 of the output
// scale, HALF the error of the exact-fp32 MFMA chain on the same data (3.5e-7 .. 1.5e-6): the matrix core sums the 16
// products of a step in one go instead of through 16 dependent fp32 roundings, and the operand rounding (2^-22 per
// product, random sign) is an order of magnitude below the accumulation error of a K >= 768 fp32 chain.
//
// Operand layout ("H2 rows"): a row of C channels (C % 32 == 0) is C/32 chunks of 128 bytes, each
// [32 x f16 hi | 32 x f16 lo] -- the size of the fp32 row it replaces, and one 128-byte chunk is exactly one K = 32
// stage of a row in LDS.  Producers write it directly (coocc_wino_input_h2: the Winograd input transform;
// coocc_rows_to_h2: any fp32 rows).  Weights: coocc pack [(chunk, tap)][Npad/32][k16 step s][plane hi|lo][64 lanes][8 f16],
// lane l holds B[k = 32 chunk + 16 s + 8 (l >> 5) + 0..7][n = 32 nt + (l & 31)]: one 16-byte load per fragment, 1 KB coalesced
// per wave instruction, straight into registers (no LDS), made on the host by core.PackedConv.h2_pack.
//
// k_gemm_h2z<KZ>: stride-1 "same" geometry (the Winograd-domain grouped GEMM with its 3 z taps, and direct 3x3x3 / 3x3xKZ
// layers).  128 x 128 tile, 4 waves, wave w owns columns 32 w .. 32 w + 31 of all 128 rows (B never shared between waves,
// A read by every wave from LDS).  Output rows are linear in (x, y, z), so the rows tap (dx, dy, dz) needs are the tile's own
// rows shifted by a constant: ONE LDS image of 128 + KZ - 1 rows per (chunk, dx, dy) serves all KZ z taps (k_conv_bf16z's
// scheme).  The image is staged by global_load_lds (16 B per lane, no staging registers); it is lane-linear, so the
// bank-conflict swizzle sits on the SOURCE address: 16-byte slot c of LDS row r holds slot c ^ ((r >> 1) & 7) of the chunk.
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "conv_k.h"

#include "h2_rows.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16_(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l,
                                   16, 0, 0);
}

// XY = false: kx = ky = 1 (the Winograd-domain grouped GEMM: only the z taps), no (dx, dy) cursor and no x / y range checks.
// Computed TRANSPOSED (weights as the first MFMA operand): a lane ends up with 4-channel runs of its own output rows, so the
// epilogue is 16-byte stores (and 16-byte scale / bias / residual loads).
// Fragments are SINGLE-buffered (32 registers): a k16 step runs  P0 = W_hi x A_hi, P1 = W_lo x A_hi, P2 = W_hi x A_lo  (4 MFMAs each);
// A_lo of step u is read from LDS under P0/P1 of step u (its registers are free once P2 of step u-1 has issued), A_hi of step u+1
// under P2 of step u; sched_barrier(0) between the phases keeps hipcc from hoisting the reads (more live fragment registers ->
// spills -> scratch reloads that wait vmcnt(0), i.e. for the weight loads just issued).  Two workgroups per CU (<= 256 registers)
// cover each other's barriers, prologues and epilogues.
// epilogue option out_h2: the output row is written as an H2 row (the next split-f16 layer's operand) instead of fp32:
// channels n .. n+3 of row `row` (out_stride = channels per row, a multiple of 32)
// the vector epilogue of both kernels (and of the in-kernel split-K reduction): 4 consecutive channels n .. n+3 of output row orow
__device__ __forceinline__ void h2_epilogue_vec(const ConvK& p, size_t orow, int n, f32x4 v) {
  if (p.res_mode == 3) v = v + *(const f32x4*)(p.res + orow * p.res_stride + n);
  if (p.scale) v = v * *(const f32x4*)(p.scale + n);
  if (p.bias) v = v + *(const f32x4*)(p.bias + n);
  if (p.res_mode == 1) v = v + *(const f32x4*)(p.res + orow * p.res_stride + n);
  if (p.relu && (p.relu == 1 || n < p.relu)) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
  if (p.res_mode == 2) v = v * *(const f32x4*)(p.res + orow * p.res_stride + n);
  if (p.out_h2) { store_h2(p.out, orow, p.out_stride, n, v); h2_guard(p.h2_flag, v); }
  else *(f32x4*)(p.out + orow * p.out_stride + n) = v;
  if (p.out16) { store_f16(p.out16, orow, p.out16_stride, n, v); h2_guard(p.h2_flag, v); }
  if (p.out_h2t) { store_h2(p.out_h2t, orow, p.Cout, n, v); h2_guard(p.h2_flag, v); }
}
// In-kernel split-K reduction (p.tile_sem != NULL): every workgroup of an (M tile, N tile) has written its partial slab; the one
// that arrives LAST sums the slabs in slice order 0 .. splitk-1 (the order k_conv_reduce uses: same bits whichever workgroup is
// last) and runs the epilogue.  The 8 XCDs' L2s are not coherent with each other for ordinary stores, and the obvious fix --
// __threadfence() on both sides = an L2 write-back + invalidate per workgroup -- was measured at 4-8x the whole kernel
// (k_gemm_h2z<1,true> 24 -> 200 us: every arriving workgroup flushes an L2 that other streams keep dirty).  Instead the SLABS
// THEMSELVES are moved with device-scope accesses: relaxed agent-scope atomic stores / loads (global_store / global_load with
// sc1: write-through to, and read from, the point of coherence), ordered against the arrival counter by the vmcnt(0) wait
// that __syncthreads() carries.  Nothing else is flushed or invalidated.  The counter is left at zero for the next launch.
__device__ __forceinline__ void st_agent(float* p, f32x4 v) {
  const unsigned long long a = (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32);
  const unsigned long long b = (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32);
  __hip_atomic_store((unsigned long long*)p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long*)p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 ld_agent(const float* p) {
  const unsigned long long a = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f32x4{__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
}
// Returns true for the workgroup that reduces.
__device__ __forceinline__ bool h2_splitk_arrive(const ConvK& p, int tile) {
  __shared__ int s_last;
  __syncthreads();            // workgroup-scope release: every wave's slab stores have been acknowledged (s_waitcnt vmcnt(0))
  if (threadIdx.x == 0) {
    const int old = atomicAdd(&p.tile_sem[tile], 1);
    s_last = old == p.splitk - 1;
    if (s_last) p.tile_sem[tile] = 0;
  }
  __syncthreads();
  return s_last != 0;
}
template <int TM>
__device__ __forceinline__ void h2_splitk_reduce(const ConvK& p, int m0, int nb, int li) {
  const bool vec = (p.Cout & 3) == 0 && (p.out_stride & 3) == 0 && (!p.res || (p.res_stride & 3) == 0);
  const size_t zs = (size_t)p.M * p.Npad;
#pragma unroll 1
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + i * 32 + li;
    if (m >= p.M) continue;
    const size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 8 * j;
      if (n >= p.Cout) continue;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const float* src = p.ws + (size_t)m * p.Npad + n;
      int z = 0;
      for (; z + 4 <= p.splitk; z += 4) {           // four slabs in flight, summed in slice order
        const f32x4 a0 = ld_agent(src + (size_t)z * zs), a1 = ld_agent(src + (size_t)(z + 1) * zs);
        const f32x4 a2 = ld_agent(src + (size_t)(z + 2) * zs), a3 = ld_agent(src + (size_t)(z + 3) * zs);
        v = v + a0; v = v + a1; v = v + a2; v = v + a3;
      }
      for (; z < p.splitk; ++z) v = v + ld_agent(src + (size_t)z * zs);
      if (vec) h2_epilogue_vec(p, orow, n, v);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.Cout) p.out[orow * p.out_stride + n + e] = epilogue(p, v[e], n + e, orow);
      }
    }
  }
}
#define H2_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ const char* inb_(const ConvK& p) { return (const char*)p.in; }
// TERMS = 3: split operands (H2 rows, fp32-accurate); TERMS = 1: plain f16 operands ("H1 rows": [rows][C] f16, 64 channels per
// 128-byte LDS row, four k16 steps per stage, one MFMA per step) -- the reduced-precision path of configs[4].
template <int KZ, bool XY, int ABL = 0, int TERMS = 3>    // ABL (timing ablations, wrong results): 1 no A image, 2 no weight loads, 4 no fragment reads, 8 no stores
__global__ __launch_bounds__(256, 2) void k_gemm_h2z(ConvK p) {
  constexpr int BM = 128, TM = 4;
  constexpr int AROWS = 136;                       // 128 + KZ - 1 rounded up to whole 8-row wave instructions
  // two stages + 256 zero bytes: a masked fragment reads the zero word that sits in the SAME banks as its real address
  // (ZOFF | (address & 255)), so masking adds no bank conflict (a single zero row cost 2.0e9 conflict cycles per 442 launches)
  constexpr unsigned STAGE = AROWS * 128, ZOFF = 2 * STAGE;
  static_assert(STAGE % 256 == 0 && ZOFF % 256 == 0, "zero block must keep the bank of the address it replaces");
  __shared__ __attribute__((aligned(256))) char As[2 * AROWS * 128 + 256];

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
  const int m0 = mtile * BM, n0 = nt * 128;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, slot = lane & 7;
  if (tid < 16) *(f32x4*)&As[ZOFF + tid * 16] = f32x4{0.f, 0.f, 0.f, 0.f};

  // everything the K loop needs in registers (a field of `p` read inside the loop is a scalar load whose lgkmcnt(0) also waits
  // for the LDS reads in flight)
  const int kx = p.kx, ky = p.ky, px = p.px, py = p.py, Xi = p.Xi, Yi = p.Yi, Zi = p.Zi;
  const unsigned rowbytes = (unsigned)p.in_stride * (TERMS == 3 ? 4 : 2);
  const int total_rows = Xi * Yi * Zi * (p.M / (p.Xo * p.Yo * p.Zo));
  const char* zrow = (const char*)p.zrow;
  // staging: LDS row r holds input row  m0 + r - pz + ((dx - px) Yi + (dy - py)) Zi  (any row of the buffer, else zeros).
  // Addresses are a 64-bit tile base (the lowest row any tap of this tile can touch) + 32-bit offsets inside the tile's window
  // (a few thousand rows), so inputs past 4 GB work (the Winograd V of a 200x200x16x512 volume is 5.2 GB).  The swizzle term of a
  // row does not depend on the instruction: ((r >> 1) & 7) with r = (4 j + wave) 8 + srow  is  (4 (wave & 1) + (srow >> 1)) & 7
  const int arow0 = m0 + srow - p.pz;
  const int minoff = XY ? -(px * Yi + py) * Zi : 0;
  const char* tbase = inb_(p) + (long long)(m0 - p.pz + minoff) * (long long)rowbytes;
  const unsigned aq = (unsigned)((slot ^ ((4 * (wave & 1) + (srow >> 1)) & 7)) * 16);
  // the output voxels this lane's A fragments belong to (fragment i: row i*32 + li); zbits: bit (i*3 + dz) = z tap dz in range
  int vx[TM], vy[TM];
  unsigned zbits = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + i * 32 + li;
    const int oz = m % p.Zo; int q = m / p.Zo;
    if (XY) {
      const int oy = q % p.Yo; q /= p.Yo;
      vx[i] = q % p.Xo; vy[i] = oy;
    }
#pragma unroll
    for (int dz = 0; dz < KZ; ++dz)
      if (m < p.M && (unsigned)(oz + dz - p.pz) < (unsigned)Zi) zbits |= 1u << (i * 3 + dz);
  }
  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);
  const int ngroups = (it1 - it0) / KZ;                 // the launcher keeps split boundaries on whole (dx, dy) groups
  const int g0 = it0 / KZ;                              // group index = (chunk * kx + dx) * ky + dy
  int gkc = XY ? g0 / (kx * ky) : g0, gd = XY ? (g0 / ky) % kx : 0, gh = XY ? g0 % ky : 0;
  int cd = gd, ch_ = gh;
  const long long wstep = (long long)(p.Npad >> 5) * 4096;
  const char* wcur = (const char*)p.w + (p.wgroup_rows > 0 ? (size_t)(m0 / p.wgroup_rows) * p.wgroup_floats * 4 : (size_t)0) +
                     (long long)it0 * wstep + (long long)((n0 >> 5) + wave) * 4096 + lane * 16;

  auto issueA = [&](int buf) {
    if (ABL & 1) { if (XY) { if (++gh == ky) { gh = 0; if (++gd == kx) { gd = 0; ++gkc; } } } else ++gkc; return; }
    const int off = XY ? ((gd - px) * Yi + (gh - py)) * Zi : 0;
    const unsigned coff = (unsigned)gkc * 128 + aq;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 4 && wave != 0) break;
      const int r8 = j < 4 ? (j * 4 + wave) * 8 : 128;
      const int L = arow0 + r8 + off;
      const char* src = (unsigned)L < (unsigned)total_rows ? tbase + ((unsigned)(srow + r8 + off - minoff) * rowbytes + coff) : zrow;
      if (ABL & 16) src = zrow;       // ablation: the same instructions, every lane reads the 16 zero bytes (no HBM traffic)
      glds16_(src, &As[buf * STAGE + r8 * 128]);
    }
    if (XY) { if (++gh == ky) { gh = 0; if (++gd == kx) { gd = 0; ++gkc; } } }
    else ++gkc;
  };
  f16x8 breg[2][2][2];        // [register set][k16 step][plane]
  auto loadB = [&](auto bufc) {
    constexpr int B_ = decltype(bufc)::value;
    if (ABL & 2) { if (B_ == 0 && wcur == (const char*)1) breg[0][0][0] = *(const f16x8*)wcur; return; }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) breg[B_][s][pl] = *(const f16x8*)(wcur + (s * 2 + pl) * 1024);
    wcur += wstep;
  };
  // fragment of row i*32 + li + dz, 16-byte slot q = (2 s + h | 4 + 2 s + h): (li + dz) * 128 + ((q ^ (((li + dz) >> 1) & 7)) << 4)
  unsigned fragoff[KZ][4];
#pragma unroll
  for (int dz = 0; dz < KZ; ++dz)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int sl = TERMS == 3 ? (q & 1) * 4 + 2 * (q >> 1) + h : 2 * q + h;            // TERMS 3: q = 2 s + plane; TERMS 1: q = k16 step
      fragoff[dz][q] = (li + dz) * 128 + ((sl ^ (((li + dz) >> 1) & 7)) << 4);
    }
  f16x8 fr[TERMS == 3 ? 2 : 4][TM];          // TERMS 3: A_hi / A_lo of the current step; TERMS 1: the fragments of four k16 steps
  if (ABL & 4) {
#pragma unroll
    for (int i = 0; i < TM; ++i) { fr[0][i] = *(const f16x8*)&As[lane * 16 + i * 64]; fr[1][i] = *(const f16x8*)&As[lane * 16 + i * 64 + 32]; }
  }
  if (ABL & 2) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) breg[a][s2][pl] = *(const f16x8*)(wcur + (a * 4 + s2 * 2 + pl) * 1024);
  }
  // a lane whose output voxel's tap leaves the grid reads the zero row instead: one select on the LDS address, the loads stay
  // unconditional (hipcc turns "ok ? fragment : 0" into a branch around the loads)
  auto load_frag = [&](auto bufc, auto stagec, auto dzc, auto qc, unsigned bits) {
    constexpr int BUF = decltype(bufc)::value, ST = decltype(stagec)::value, DZ = decltype(dzc)::value, Q = decltype(qc)::value;
    if (ABL & 4) return;
    asm volatile("" : "+v"(bits));      // keep the address selects here: hoisted out of the K loop they are dozens of live registers
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const bool ok = (bits >> (i * 3 + DZ)) & 1;
      const unsigned a = ok ? fragoff[DZ][Q] + (ST * STAGE + i * 4096) : (ZOFF | (fragoff[DZ][Q] & 255u));
      fr[BUF][i] = *(const f16x8*)&As[a];
    }
  };
  auto load_plane = [&](auto planec, auto stagec, auto dzc, auto sc, unsigned bits) {
    constexpr int PL = decltype(planec)::value, S_ = decltype(sc)::value;
    load_frag(planec, stagec, dzc, std::integral_constant<int, 2 * S_ + PL>{}, bits);
  };

  f32x16 hh[TM], xx[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hh[i][r] = 0.f; xx[i][r] = 0.f; }
  auto p01 = [&](auto setc, auto sc) {
    constexpr int B_ = decltype(setc)::value, S_ = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) hh[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][0], fr[0][i], hh[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][1], fr[0][i], xx[i], 0, 0, 0);
  };
  auto p2 = [&](auto setc, auto sc) {
    constexpr int B_ = decltype(setc)::value, S_ = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][0], fr[1][i], xx[i], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;

  // one tap (two k16 steps) of the group in stage ST with weight set BB; `nxt`: what the A_hi fragments fetched under the last P2
  // belong to -- the next tap of this group (stage ST, tap DZ + 1), or, for the last tap, step 0 of the next group (other stage,
  // after the barrier that retires this one)
  bool pendingA = false;
  auto tap = [&](auto stc, auto bbc, auto dzc, bool more_taps, unsigned okbits, unsigned okbits_next) {
    constexpr int ST = decltype(stc)::value, BB = decltype(bbc)::value, DZ = decltype(dzc)::value;
    using STC = std::integral_constant<int, ST>; using SNC = std::integral_constant<int, ST ^ 1>;
    using DZC = std::integral_constant<int, DZ>; using DZN = std::integral_constant<int, (DZ + 1 < KZ ? DZ + 1 : 0)>;
    using BBC = std::integral_constant<int, BB>;
    if (more_taps) loadB(std::integral_constant<int, BB ^ 1>{});
    // the next group's A image is issued AFTER this tap's weight loads: vector-memory data returns in issue order, so the wait for
    // these weights (start of the next tap) would otherwise also wait for the image (HBM latency, one tap after its issue)
    if (DZ == 0 && pendingA) issueA(ST ^ 1);
    if constexpr (TERMS == 1) {
      // four k16 steps, two per phase: steps 2, 3 are fetched under steps 0, 1; the first two steps of the next tap (or of the next
      // group, behind the barrier) under steps 2, 3
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      auto pq = [&](auto qc) {
        constexpr int Q = decltype(qc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) hh[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[BB][Q >> 1][Q & 1], fr[Q][i], hh[i], 0, 0, 0);
      };
      load_frag(I2{}, STC{}, DZC{}, I2{}, okbits);
      load_frag(I3{}, STC{}, DZC{}, I3{}, okbits);
      pq(I0{}); pq(I1{});
      H2_FENCE();
      if constexpr (DZ + 1 < KZ) {
        load_frag(I0{}, STC{}, DZN{}, I0{}, okbits);
        load_frag(I1{}, STC{}, DZN{}, I1{}, okbits);
      } else {
        __syncthreads();
        load_frag(I0{}, SNC{}, I0{}, I0{}, okbits_next);
        load_frag(I1{}, SNC{}, I0{}, I1{}, okbits_next);
      }
      pq(I2{}); pq(I3{});
      H2_FENCE();
      return;
    }
    load_plane(I1{}, STC{}, DZC{}, I0{}, okbits);
    p01(BBC{}, I0{});
    H2_FENCE();
    load_plane(I0{}, STC{}, DZC{}, I1{}, okbits);
    p2(BBC{}, I0{});
    H2_FENCE();
    load_plane(I1{}, STC{}, DZC{}, I1{}, okbits);
    p01(BBC{}, I1{});
    H2_FENCE();
    if constexpr (DZ + 1 < KZ) {
      load_plane(I0{}, STC{}, DZN{}, I0{}, okbits);
    } else {
      __syncthreads();          // every wave has read this stage for the last time; the next group's image has landed
      load_plane(I0{}, SNC{}, I0{}, I0{}, okbits_next);
    }
    p2(BBC{}, I1{});
    H2_FENCE();
  };
  auto xybits = [&]() {
    unsigned b = zbits;
    if (XY) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        if (!((unsigned)(vx[i] + cd - px) < (unsigned)Xi && (unsigned)(vy[i] + ch_ - py) < (unsigned)Yi)) b &= ~(7u << (i * 3));
    }
    return b;
  };
  // one (chunk, dx, dy) group in stage GP: KZ taps; weight-set parity of its first tap = (GP * KZ) & 1
  unsigned okbits = xybits();
  auto group = [&](auto gpc, int g) {
    constexpr int GP = decltype(gpc)::value;
    using GPC = std::integral_constant<int, GP>;
    const bool last = g + 1 >= ngroups;
    pendingA = !last;
    if (XY) { if (++ch_ == ky) { ch_ = 0; if (++cd == kx) cd = 0; } }
    const unsigned nextbits = xybits();
    tap(GPC{}, std::integral_constant<int, (GP * KZ) & 1>{}, I0{}, KZ > 1 || !last, okbits, nextbits);
    if constexpr (KZ > 1)
      tap(GPC{}, std::integral_constant<int, (GP * KZ + 1) & 1>{}, I1{}, KZ > 2 || !last, okbits, nextbits);
    if constexpr (KZ > 2)
      tap(GPC{}, std::integral_constant<int, (GP * KZ + 2) & 1>{}, std::integral_constant<int, 2>{}, !last, okbits, nextbits);
    okbits = nextbits;
  };

  if (ngroups > 0) {
    issueA(0);
    loadB(I0{});
  }
  __syncthreads();
  if (ngroups > 0) {
    load_frag(I0{}, I0{}, I0{}, I0{}, okbits);
    if (TERMS == 1) load_frag(I1{}, I0{}, I0{}, I1{}, okbits);
  }
  for (int g = 0; g < ngroups; g += 2) {
    group(I0{}, g);
    if (g + 1 < ngroups) group(I1{}, g + 1);
  }

  // epilogue: lane (li, h) holds, for output row m0 + i*32 + li, the channels n0 + 32 wave + 8 j + 4 h + 0..3 (j = 0..3)
  const int nb = n0 + wave * 32 + 4 * h;
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha, lo = TERMS == 3 ? alpha * (1.f / H2_LO_SCALE) : 0.f;
  if (p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + i * 32 + li;
      if (m >= p.M) continue;
      float* o = p.ws + ((size_t)blockIdx.y * p.M + m) * p.Npad + nb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (TERMS == 3 ? hh[i][4 * j + e] * alpha + xx[i][4 * j + e] * lo : hh[i][4 * j + e] * alpha);
        if (p.tile_sem) st_agent(o + 8 * j, v);
        else *(f32x4*)(o + 8 * j) = v;
      }
    }
    if (p.tile_sem && h2_splitk_arrive(p, mtile * p.ntiles + nt)) h2_splitk_reduce<TM>(p, m0, nb, li);
  } else {
    const bool vec = (p.Cout & 3) == 0 && (p.out_stride & 3) == 0 && (!p.res || (p.res_stride & 3) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + i * 32 + li;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nb + 8 * j;
        if (n >= p.Cout) continue;
        if ((ABL & 8) && hh[i][4 * j] != 12345.f) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (TERMS == 3 ? hh[i][4 * j + e] * alpha + xx[i][4 * j + e] * lo : hh[i][4 * j + e] * alpha);
        if (vec) {
          h2_epilogue_vec(p, (size_t)m, n, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) p.out[(size_t)m * p.out_stride + n + e] = epilogue(p, v[e], n + e, (size_t)m);
        }
      }
    }
  }
}

// k_gemm_h2w: the general form -- any stride / padding / tap set (one LDS image of 128 rows per (chunk, tap), rows fetched by
// per-lane addresses: the strided and 1x1x1 layers) and row tables (TABLE: input row of (tap t, output m) = gather[t * gstride + m];
// row count optionally on the device).  Same tile, transposed MFMA order, phases and epilogue as k_gemm_h2z; rows that fall
// outside the grid (or table entries < 0) are fetched from the zero row, so no fragment needs masking.
template <bool TABLE, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void k_gemm_h2w(ConvK p) {
  constexpr int BM = 128, TM = 4;
  constexpr unsigned STAGE = BM * 128;
  __shared__ __attribute__((aligned(16))) char As[2 * BM * 128];

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
  const int m0 = mtile * BM, n0 = nt * 128;
  if (p.M_dev) {                       // row count on the device (grid sized for the capacity p.M): whole tiles past it leave
    p.M = min(p.M, *p.M_dev);
    if (m0 >= p.M) return;
  }
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, slot = lane & 7;

  const int kx = p.kx, ky = p.ky, kz = p.kz, Xi = p.Xi, Yi = p.Yi, Zi = p.Zi, taps = p.taps;
  const long long rowbytes = (long long)p.in_stride * (TERMS == 3 ? 4 : 2);
  const char* inb = (const char*)p.in;
  const char* zrow = (const char*)p.zrow;
  const unsigned aq = (unsigned)((slot ^ ((4 * (wave & 1) + (srow >> 1)) & 7)) * 16);
  // this lane's four staging rows r = (4 j + wave) 8 + srow: input coordinates of tap (0,0,0) and its row index
  int rix[4], riy[4], riz[4], rrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + (j * 4 + wave) * 8 + srow;
    if (TABLE) {
      rix[j] = m < p.M ? m : -1; riy[j] = riz[j] = rrow[j] = 0;
    } else {
      int oz = m % p.Zo; int q = m / p.Zo;
      int oy = q % p.Yo; q /= p.Yo;
      int ox = q % p.Xo; const int b = q / p.Xo;
      rix[j] = m < p.M ? ox * p.stride - p.px : -(1 << 20);
      riy[j] = oy * p.stride - p.py;
      riz[j] = oz * p.stride - p.pz;
      rrow[j] = ((b * Xi + rix[j]) * Yi + riy[j]) * Zi + riz[j];
    }
  }
  const int it0 = blockIdx.y * p.iters_per_split;
  const int it1 = min(it0 + p.iters_per_split, p.total_iters);
  const int nsteps = it1 - it0;
  int ckc = it0 / taps, ct = it0 - ckc * taps;
  int ckw = ct % kz, ckh = (ct / kz) % ky, ckd = ct / (kz * ky);
  const long long wstep = (long long)(p.Npad >> 5) * 4096;
  const char* wcur = (const char*)p.w + (long long)it0 * wstep + (long long)((n0 >> 5) + wave) * 4096 + lane * 16;
  const int gstride = p.gstride;
  int tnext[4];                    // TABLE: the source rows of the NEXT iteration (fetched one iteration ahead)
  auto tload = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) tnext[j] = rix[j] >= 0 ? p.gather[(size_t)ct * gstride + rix[j]] : -1;
  };
  if (TABLE && nsteps > 0) tload();

  auto issueA = [&](int buf) {
    const long long coff = (long long)ckc * 128 + aq;
    if (TABLE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char* src = tnext[j] >= 0 ? inb + ((long long)tnext[j] * rowbytes + coff) : zrow;
        glds16_(src, &As[buf * STAGE + (j * 4 + wave) * 8 * 128]);
      }
      if (++ct == taps) { ct = 0; ++ckc; }
      tload();                     // harmless past the last iteration: ct < taps always
    } else {
      const int tapoff = (ckd * Yi + ckh) * Zi + ckw;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = (unsigned)(rix[j] + ckd) < (unsigned)Xi && (unsigned)(riy[j] + ckh) < (unsigned)Yi &&
                        (unsigned)(riz[j] + ckw) < (unsigned)Zi;
        const char* src = ok ? inb + ((long long)(rrow[j] + tapoff) * rowbytes + coff) : zrow;
        glds16_(src, &As[buf * STAGE + (j * 4 + wave) * 8 * 128]);
      }
      if (++ckw == kz) { ckw = 0; if (++ckh == ky) { ckh = 0; if (++ckd == kx) { ckd = 0; ++ckc; } } }
    }
  };
  f16x8 breg[2][2][2];        // [register set][k16 step][plane]
  auto loadB = [&](auto bufc) {
    constexpr int B_ = decltype(bufc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) breg[B_][s][pl] = *(const f16x8*)(wcur + (s * 2 + pl) * 1024);
    wcur += wstep;
  };
  unsigned fragoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int sl = TERMS == 3 ? (q & 1) * 4 + 2 * (q >> 1) + h : 2 * q + h;            // TERMS 3: q = 2 s + plane; TERMS 1: q = k16 step
    fragoff[q] = li * 128 + ((sl ^ ((li >> 1) & 7)) << 4);
  }
  f16x8 fr[TERMS == 3 ? 2 : 4][TM];
  auto load_frag = [&](auto bufc, auto stagec, auto qc) {
    constexpr int BUF = decltype(bufc)::value, ST = decltype(stagec)::value, Q = decltype(qc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) fr[BUF][i] = *(const f16x8*)&As[fragoff[Q] + (ST * STAGE + i * 4096)];
  };
  auto load_plane = [&](auto planec, auto stagec, auto sc) {
    constexpr int PL = decltype(planec)::value, S_ = decltype(sc)::value;
    load_frag(planec, stagec, std::integral_constant<int, 2 * S_ + PL>{});
  };
  f32x16 hh[TM], xx[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hh[i][r] = 0.f; xx[i][r] = 0.f; }
  auto p01 = [&](auto setc, auto sc) {
    constexpr int B_ = decltype(setc)::value, S_ = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) hh[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][0], fr[0][i], hh[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][1], fr[0][i], xx[i], 0, 0, 0);
  };
  auto p2 = [&](auto setc, auto sc) {
    constexpr int B_ = decltype(setc)::value, S_ = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][0], fr[1][i], xx[i], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  // one (chunk, tap) iteration in stage ST (= weight set ST); A_hi of its first step is already in registers
  auto step = [&](auto stc, bool more) {
    constexpr int ST = decltype(stc)::value;
    using STC = std::integral_constant<int, ST>; using SNC = std::integral_constant<int, ST ^ 1>;
    if (more) { issueA(ST ^ 1); loadB(SNC{}); }
    if constexpr (TERMS == 1) {
      using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      auto pq = [&](auto qc) {
        constexpr int Q = decltype(qc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) hh[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[ST][Q >> 1][Q & 1], fr[Q][i], hh[i], 0, 0, 0);
      };
      load_frag(I2{}, STC{}, I2{});
      load_frag(I3{}, STC{}, I3{});
      pq(I0{}); pq(I1{});
      H2_FENCE();
      __syncthreads();
      if (more) { load_frag(I0{}, SNC{}, I0{}); load_frag(I1{}, SNC{}, I1{}); }
      pq(I2{}); pq(I3{});
      H2_FENCE();
      return;
    }
    load_plane(I1{}, STC{}, I0{});
    p01(STC{}, I0{});
    H2_FENCE();
    load_plane(I0{}, STC{}, I1{});
    p2(STC{}, I0{});
    H2_FENCE();
    load_plane(I1{}, STC{}, I1{});
    p01(STC{}, I1{});
    H2_FENCE();
    __syncthreads();            // every wave has read this stage for the last time; the next image has landed
    if (more) load_plane(I0{}, SNC{}, I0{});
    p2(STC{}, I1{});
    H2_FENCE();
  };
  if (nsteps > 0) {
    issueA(0);
    loadB(I0{});
  }
  __syncthreads();
  if (nsteps > 0) {
    load_frag(I0{}, I0{}, I0{});
    if (TERMS == 1) load_frag(I1{}, I0{}, I1{});
  }
  for (int st = 0; st < nsteps; st += 2) {
    step(I0{}, st + 1 < nsteps);
    if (st + 1 < nsteps) step(I1{}, st + 2 < nsteps);
  }

  // epilogue: lane (li, h) holds, for output row m0 + i*32 + li, the channels n0 + 32 wave + 8 j + 4 h + 0..3 (j = 0..3)
  const int nb = n0 + wave * 32 + 4 * h;
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha, lo = TERMS == 3 ? alpha * (1.f / H2_LO_SCALE) : 0.f;
  if (p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + i * 32 + li;
      if (m >= p.M) continue;
      float* o = p.ws + ((size_t)blockIdx.y * p.M + m) * p.Npad + nb;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (TERMS == 3 ? hh[i][4 * j + e] * alpha + xx[i][4 * j + e] * lo : hh[i][4 * j + e] * alpha);
        if (p.tile_sem) st_agent(o + 8 * j, v);
        else *(f32x4*)(o + 8 * j) = v;
      }
    }
    if (p.tile_sem && h2_splitk_arrive(p, mtile * p.ntiles + nt)) h2_splitk_reduce<TM>(p, m0, nb, li);
  } else {
    const bool vec = (p.Cout & 3) == 0 && (p.out_stride & 3) == 0 && (!p.res || (p.res_stride & 3) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + i * 32 + li;
      if (m >= p.M) continue;
      const size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nb + 8 * j;
        if (n >= p.Cout) continue;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (TERMS == 3 ? hh[i][4 * j + e] * alpha + xx[i][4 * j + e] * lo : hh[i][4 * j + e] * alpha);
        if (vec) {
          h2_epilogue_vec(p, orow, n, v);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) p.out[orow * p.out_stride + n + e] = epilogue(p, v[e], n + e, orow);
        }
      }
    }
  }
}

// k_gemm_h2n<NCW>: row-table GEMMs with 32 (NCW = 1) or 64 (NCW = 2) output channels -- the 32 -> 32 / 64 -> 64 / 32 -> 64 rule-book
// layers of the sparse LiDAR encoder.  In k_gemm_h2w every wave owns 32 columns of a 128-column tile, so with Cout = 32 / 64 three /
// two of the four waves multiply zero-padded weight columns, and the ablations of round 5 (profiles/r5_lidar_h2t_ablate.txt: a three-taps-per-round variant, k_gemm_h2t, since removed) put 63 % of
// these launches in the MFMA / LDS instruction stream, not in the gathers.  Here the tile is 256 rows x 32 NCW columns: wave w owns
// column block w % NCW of rows 256 / (4 / NCW) * (w / NCW) ..., i.e. every MFMA is a real one, per tile-row a quarter / half of the
// MFMAs, weight loads and fragment reads.  Same stages, fragments, MFMA order per wave, tap order and epilogue expressions as
// k_gemm_h2w<true>: a column's sum is formed in the same order -> the same bits.  No split-K (these layers have >= 256 tiles).
// NCW = 4: the same wave layout for FULL-width (128-column) layers with few row tiles -- 64-row tiles, wave w on column block w, two
// row blocks per wave, 16 KB of LDS and <= 168 registers, so three workgroups fit a CU: the 128 -> 128 stage of the encoder (45 k rows)
// is 352 tiles of 128 rows for 512 workgroup slots -- 96 CUs run two of them, 160 run one -- and 704 tiles of 64 rows for 768 slots.
template <int NCW>
__global__ __launch_bounds__(256, NCW == 4 ? 3 : 2) void k_gemm_h2n(ConvK p) {
  constexpr int BM = NCW == 4 ? 64 : 256, TM = NCW == 2 ? 4 : 2, NJ = BM / 32;        // NJ staging rows per lane and stage
  constexpr unsigned STAGE = BM * 128;
  __shared__ __attribute__((aligned(16))) char As[2 * BM * 128];

  const int mtiles = (p.M + BM - 1) / BM;                               // p.M = capacity when the row count is on the device
  const int mtile = blockIdx.x;
  if (mtile >= mtiles) return;
  const int m0 = mtile * BM;
  if (p.M_dev) {
    p.M = min(p.M, *p.M_dev);
    if (m0 >= p.M) return;
  }
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, slot = lane & 7;
  const int colblk = NCW == 4 ? wave : (NCW == 2 ? (wave & 1) : 0);
  const int rowbase = NCW == 4 ? 0 : (NCW == 2 ? (wave >> 1) * 128 : wave * 64);         // first tile row of this wave's accumulators
  const int taps = p.taps;
  const long long rowbytes = (long long)p.in_stride * 4;
  const char* inb = (const char*)p.in;
  const char* zrow = (const char*)p.zrow;
  const unsigned aq = (unsigned)((slot ^ ((4 * (wave & 1) + (srow >> 1)) & 7)) * 16);
  int rix[NJ];                      // this lane's staging rows r = (4 j + wave) 8 + srow
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + (j * 4 + wave) * 8 + srow;
    rix[j] = m < p.M ? m : -1;
  }
  const int nsteps = p.total_iters;
  int ckc = 0, ct = 0;
  const long long wstep = (long long)(p.Npad >> 5) * 4096;
  const char* wcur = (const char*)p.w + (long long)colblk * 4096 + lane * 16;
  const int gstride = p.gstride;
  int tnext[NJ];                    // the source rows of the NEXT iteration (fetched one iteration ahead)
  auto tload = [&]() {
#pragma unroll
    for (int j = 0; j < NJ; ++j) tnext[j] = rix[j] >= 0 ? p.gather[(size_t)ct * gstride + rix[j]] : -1;
  };
  if (nsteps > 0) tload();
  auto issueA = [&](int buf) {
    const long long coff = (long long)ckc * 128 + aq;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const char* src = tnext[j] >= 0 ? inb + ((long long)tnext[j] * rowbytes + coff) : zrow;
      glds16_(src, &As[buf * STAGE + (j * 4 + wave) * 8 * 128]);
    }
    if (++ct == taps) { ct = 0; ++ckc; }
    tload();                       // harmless past the last iteration: ct < taps always
  };
  f16x8 breg[2][2][2];        // [register set][k16 step][plane]
  auto loadB = [&](auto bufc) {
    constexpr int B_ = decltype(bufc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) breg[B_][s][pl] = *(const f16x8*)(wcur + (s * 2 + pl) * 1024);
    wcur += wstep;
  };
  unsigned fragoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int sl = (q & 1) * 4 + 2 * (q >> 1) + h;            // q = 2 s + plane
    fragoff[q] = (rowbase + li) * 128 + ((sl ^ ((li >> 1) & 7)) << 4);
  }
  f16x8 fr[2][TM];
  auto load_frag = [&](auto bufc, auto stagec, auto qc) {
    constexpr int BUF = decltype(bufc)::value, ST = decltype(stagec)::value, Q = decltype(qc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) fr[BUF][i] = *(const f16x8*)&As[fragoff[Q] + (ST * STAGE + i * 4096)];
  };
  auto load_plane = [&](auto planec, auto stagec, auto sc) {
    constexpr int PL = decltype(planec)::value, S_ = decltype(sc)::value;
    load_frag(planec, stagec, std::integral_constant<int, 2 * S_ + PL>{});
  };
  f32x16 hh[TM], xx[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hh[i][r] = 0.f; xx[i][r] = 0.f; }
  auto p01 = [&](auto setc, auto sc) {
    constexpr int B_ = decltype(setc)::value, S_ = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) hh[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][0], fr[0][i], hh[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][1], fr[0][i], xx[i], 0, 0, 0);
  };
  auto p2 = [&](auto setc, auto sc) {
    constexpr int B_ = decltype(setc)::value, S_ = decltype(sc)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[B_][S_][0], fr[1][i], xx[i], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  auto step = [&](auto stc, bool more) {
    constexpr int ST = decltype(stc)::value;
    using STC = std::integral_constant<int, ST>; using SNC = std::integral_constant<int, ST ^ 1>;
    if (more) { issueA(ST ^ 1); loadB(SNC{}); }
    load_plane(I1{}, STC{}, I0{});
    p01(STC{}, I0{});
    H2_FENCE();
    load_plane(I0{}, STC{}, I1{});
    p2(STC{}, I0{});
    H2_FENCE();
    load_plane(I1{}, STC{}, I1{});
    p01(STC{}, I1{});
    H2_FENCE();
    __syncthreads();            // every wave has read this stage for the last time; the next image has landed
    if (more) load_plane(I0{}, SNC{}, I0{});
    p2(STC{}, I1{});
    H2_FENCE();
  };
  if (nsteps > 0) {
    issueA(0);
    loadB(I0{});
  }
  __syncthreads();
  if (nsteps > 0) load_frag(I0{}, I0{}, I0{});
  for (int st = 0; st < nsteps; st += 2) {
    step(I0{}, st + 1 < nsteps);
    if (st + 1 < nsteps) step(I1{}, st + 2 < nsteps);
  }

  // epilogue: lane (li, h) holds, for output row m0 + rowbase + i*32 + li, the channels 32 colblk + 8 j + 4 h + 0..3 (j = 0..3)
  const int nb = colblk * 32 + 4 * h;
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha, lo = alpha * (1.f / H2_LO_SCALE);
  const bool vec = (p.Cout & 3) == 0 && (p.out_stride & 3) == 0 && (!p.res || (p.res_stride & 3) == 0);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + rowbase + i * 32 + li;
    if (m >= p.M) continue;
    const size_t orow = p.out_rows ? (size_t)p.out_rows[m] : (size_t)m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 8 * j;
      if (n >= p.Cout) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = hh[i][4 * j + e] * alpha + xx[i][4 * j + e] * lo;
      if (vec) {
        h2_epilogue_vec(p, orow, n, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.Cout) p.out[orow * p.out_stride + n + e] = epilogue(p, v[e], n + e, orow);
      }
    }
  }
}

// k_gemm_h2p: the pointwise layers (1x1x1, stride 1: input row = output row) with K <= 128 -- occ_pred_conv[0] + Q, input_proj,
// voxel_soft_weights[0], the level-0 FPN lateral: 80 000 rows x 128 channels each.  They are HBM-bound (41 MB in, 41-123 MB out)
// and ran at half that bound in k_gemm_h2w (40-65 us against 18-36): with one 16 KB stage in flight per workgroup and two
// workgroups per CU there are 32 KB of reads outstanding per CU -- not enough to cover HBM latency at 8 TB/s -- and the K loop of
// four iterations is all prologue.  Here the WHOLE A tile (<= 4 chunks x 128 rows x 128 B = 64 KB) is requested up front by
// global_load_lds, the weights of all chunks (<= 64 registers per lane: a wave's 32 columns) are loaded once, and the MFMAs run
// from LDS without further waits; two workgroups per CU = 128 KB in flight.  Same tile, fragments, transposed MFMA order and
// epilogue as k_gemm_h2w.
template <int NCH>
__global__ __launch_bounds__(256, 2) void k_gemm_h2p(ConvK p) {
  constexpr int BM = 128, TM = 4;
  constexpr unsigned STAGE = BM * 128;
  __shared__ __attribute__((aligned(16))) char As[NCH * BM * 128];
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
  const int m0 = mtile * BM, n0 = nt * 128;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  const int srow = lane >> 3, slot = lane & 7;
  const long long rowbytes = (long long)p.in_stride * 4;
  const char* inb = (const char*)p.in;
  const char* zrow = (const char*)p.zrow;
  const unsigned aq = (unsigned)((slot ^ ((4 * (wave & 1) + (srow >> 1)) & 7)) * 16);
  // the whole A tile: chunk c of row r = (4 j + wave) 8 + srow -> stage c, lane-linear (the swizzle sits on the source address)
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + (j * 4 + wave) * 8 + srow;
      const char* src = m < p.M ? inb + ((long long)m * rowbytes + c * 128 + aq) : zrow;
      glds16_(src, &As[c * STAGE + (j * 4 + wave) * 8 * 128]);
    }
  // this wave's weights of every chunk: [chunk][k16 step][plane]
  f16x8 breg[NCH][2][2];
  const long long wstep = (long long)(p.Npad >> 5) * 4096;
  const char* wcur = (const char*)p.w + (long long)((n0 >> 5) + wave) * 4096 + lane * 16;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) breg[c][s2][pl] = *(const f16x8*)(wcur + c * wstep + (s2 * 2 + pl) * 1024);
  unsigned fragoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int sl = (q & 1) * 4 + 2 * (q >> 1) + h;                                     // q = 2 s + plane
    fragoff[q] = li * 128 + ((sl ^ ((li >> 1) & 7)) << 4);
  }
  f32x16 hh[TM], xx[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { hh[i][r] = 0.f; xx[i][r] = 0.f; }
  __syncthreads();                       // (carries vmcnt(0): the tile has landed)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f16x8 fhi[TM], flo[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fhi[i] = *(const f16x8*)&As[fragoff[2 * s2 + 0] + (c * STAGE + i * 4096)];
        flo[i] = *(const f16x8*)&As[fragoff[2 * s2 + 1] + (c * STAGE + i * 4096)];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) hh[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[c][s2][0], fhi[i], hh[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[c][s2][1], fhi[i], xx[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i) xx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(breg[c][s2][0], flo[i], xx[i], 0, 0, 0);
      H2_FENCE();
    }
  }
  // epilogue: lane (li, h) holds, for output row m0 + i*32 + li, the channels n0 + 32 wave + 8 j + 4 h + 0..3 (j = 0..3)
  const int nb = n0 + wave * 32 + 4 * h;
  const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha, lo = alpha * (1.f / H2_LO_SCALE);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + i * 32 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 8 * j;
      if (n >= p.Cout) continue;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = hh[i][4 * j + e] * alpha + xx[i][4 * j + e] * lo;
      h2_epilogue_vec(p, (size_t)m, n, v);
    }
  }
}

// One host-mapped word for the whole process (every device sees host-pinned memory at its host address): raised by the H2 / f16
// writers (h2_guard), read by coocc_h2_overflow without any device synchronisation of its own.
static int* g_h2_flag = nullptr;
int coocc_h2_flag_ptr(int** out) {
  if (!g_h2_flag) {
    void* p = nullptr;
    COOCC_HIP(hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(p, 0, 64);
    g_h2_flag = (int*)p;
  }
  *out = g_h2_flag;
  return COOCC_OK;
}
extern "C" int coocc_h2_overflow(int reset) {
  if (!g_h2_flag) return 0;
  const int v = *(volatile int*)g_h2_flag != 0;
  if (reset) *(volatile int*)g_h2_flag = 0;
  return v;
}
// Word 1 of the same mapped block: a sticky DEVICE FAULT code (COOCC_FAULT_*) stored by kernels that meet data no valid caller can
// produce (an index past the buffer it addresses) -- they skip the access instead of faulting the GPU and the host turns the code
// into an error at its next synchronisation point.
extern "C" int coocc_device_fault(int reset) {
  if (!g_h2_flag) return 0;
  const int v = ((volatile int*)g_h2_flag)[1];
  if (reset) ((volatile int*)g_h2_flag)[1] = 0;
  return v;
}

int coocc_launch_h2(ConvK& k, const coocc_conv_desc* d, hipStream_t s) {
  const bool one = d->mfma_dtype == 4;          // one-term f16 operands ([rows][C] f16), 64 channels per stage
  const int kc = one ? 64 : 32;
  COOCC_CHECK_ARG(d->Cin % kc == 0 && d->in_stride % kc == 0, "conv_fwd: H2 operands need Cin and in_stride % 32 == 0 (f16 operands: % 64)");
  COOCC_CHECK_ARG(!one || (d->wgroup_rows == 0 && !d->out_h2), "conv_fwd: the one-term f16 path has no weight groups / H2 output");
  COOCC_CHECK_ARG(d->wgroup_rows == 0 || d->wgroup_rows % 128 == 0, "conv_fwd: wgroup_rows must be a multiple of 128");
  const bool table = d->gather != nullptr;
  // stride-1 "same" geometry with <= 3 z taps: one LDS image per (chunk, dx, dy) serves the z taps (k_gemm_h2z); everything else
  // (strided, 1x1x1, row tables) one image per (chunk, tap) (k_gemm_h2w)
  const bool zshare = !table && !d->out_rows && !d->M_dev && k.stride == 1 && k.Xo == k.Xi && k.Yo == k.Yi && k.Zo == k.Zi && k.kz >= 1 && k.kz <= 3 &&
                      k.taps > 1;
  COOCC_CHECK_ARG(zshare || d->wgroup_rows == 0, "conv_fwd: weight groups need the stride-1 same geometry");
  k.kchunks = d->Cin / kc;
  k.total_iters = k.taps * k.kchunks;
  k.wgroup_floats = (size_t)k.taps * k.kchunks * k.Npad * 32;       // 128 bytes per (chunk, tap, column)
  k.out16 = d->out16;
  k.out16_stride = d->out16_stride;
  COOCC_CHECK_ARG(!d->out16 || (d->Cout % 4 == 0 && d->out16_stride % 4 == 0 && !d->out_rows && ((uintptr_t)d->out16 & 7) == 0),
                  "conv_fwd: out16 needs Cout % 4 == 0, out16_stride % 4 == 0, no row scatter");
  k.alpha = d->alpha != 0.f ? d->alpha : 1.f;
  k.alpha_dev = d->alpha_dev;
  k.M_dev = d->M_dev;
  k.out_h2 = d->out_h2;
  COOCC_CHECK_ARG(!d->out_h2 || (d->Cout % 4 == 0 && d->out_stride % 32 == 0 && !d->out_rows),
                  "conv_fwd: out_h2 needs Cout % 4 == 0, out_stride % 32 == 0, no row scatter");
  k.out_h2t = d->out_h2_twin;
  COOCC_CHECK_ARG(!d->out_h2_twin || (!one && d->Cout % 32 == 0 && !d->out_rows && ((uintptr_t)d->out_h2_twin & 15) == 0 && (d->out_stride & 3) == 0 &&
                                      (!d->res || (d->res_stride & 3) == 0)),
                  "conv_fwd: out_h2_twin needs mfma_dtype 3, Cout % 32 == 0, 16-byte aligned rows, no row scatter");
  k.tile_sem = d->tile_sem;
  int rc = coocc_zero_row(&k.zrow);
  if (rc != COOCC_OK) return rc;
  rc = coocc_h2_flag_ptr(&k.h2_flag);
  if (rc != COOCC_OK) return rc;
  k.ntiles = (k.Cout + 127) / 128;
  k.mtiles = (k.M + 127) / 128;
  k.mtiles_per_xcd = k.mtiles >= 64 ? (k.mtiles + 7) / 8 : 0;
  // split-K on whole groups (k_gemm_h2z: (dx, dy) groups of kz taps; k_gemm_h2w: single iterations)
  const int gsz = zshare ? k.kz : 1;
  int splitk = d->splitk;
  const long long blocks = (long long)k.mtiles * k.ntiles;
  const int ngroups = k.total_iters / gsz;
  if (splitk <= 0) {
    splitk = 1;
    // without arrival counters the reduction is a second launch; its vector form (k_conv_reduce4) writes the H2 / f16 / twin outputs
    const bool vec4 = (d->Cout & 3) == 0 && (d->out_stride & 3) == 0 && (!d->res || (d->res_stride & 3) == 0) && (((uintptr_t)d->out) & 15) == 0;
    const bool second_pass_ok = vec4 || (!d->out_h2 && !d->out16 && !d->out_h2_twin);
    // COOCC_SPLITK_TARGET: workgroups a split layer aims for (512 = two per CU: tuned with the layer alone on the chip)
    static const int target = getenv("COOCC_SPLITK_TARGET") ? atoi(getenv("COOCC_SPLITK_TARGET")) : 512;
    if (blocks < target / 2 && ngroups >= 8 && d->ws && !d->M_dev && (d->tile_sem ? true : (!d->out_rows && second_pass_ok))) {
      splitk = (int)(target / blocks);
      if (splitk > ngroups / 4) splitk = ngroups / 4;
      if (splitk > 64) splitk = 64;
      while (splitk > 1 && (long long)splitk * d->M * k.Npad > d->ws_floats) --splitk;
      if (splitk < 1) splitk = 1;
    }
  }
  int gps = (ngroups + splitk - 1) / splitk;
  k.iters_per_split = gps * gsz;
  k.splitk = (k.total_iters + k.iters_per_split - 1) / k.iters_per_split;
  COOCC_CHECK_ARG(k.splitk == 1 || (d->ws && !d->M_dev && (long long)k.splitk * d->M * k.Npad <= d->ws_floats),
                  "conv_fwd: split-K workspace too small (or split-K with a device row count)");
  COOCC_CHECK_ARG(k.splitk == 1 || d->tile_sem || (!d->out_h2 && !d->out16 && !d->out_h2_twin) ||
                      ((d->Cout & 3) == 0 && (d->out_stride & 3) == 0 && (!d->res || (d->res_stride & 3) == 0) && (((uintptr_t)d->out) & 15) == 0),
                  "conv_fwd: split-K with an H2 / f16 output needs 16-byte aligned rows (Cout, strides % 4 == 0)");
  COOCC_CHECK_ARG(k.splitk == 1 || !d->tile_sem || (long long)k.mtiles * k.ntiles <= d->tile_sem_ints,
                  "conv_fwd: tile_sem holds fewer counters than the launch has output tiles");
  dim3 grid(k.mtiles_per_xcd ? 8 * k.mtiles_per_xcd * k.ntiles : k.mtiles * k.ntiles, k.splitk);
  // pointwise layers with the whole A tile in flight (k_gemm_h2p): 1x1x1 stride 1, K <= 128, one pass, vector epilogue
  static const bool pointwise_on = !(getenv("COOCC_H2_POINTWISE") && atoi(getenv("COOCC_H2_POINTWISE")) == 0);
  if (pointwise_on && !zshare && !table && !one && k.taps == 1 && k.stride == 1 && k.px == 0 && k.py == 0 && k.pz == 0 && k.Xo == k.Xi &&
      k.Yo == k.Yi && k.Zo == k.Zi && k.kchunks <= 4 && k.splitk == 1 && !d->M_dev && !d->out_rows && blocks >= 256 &&
      (d->Cout & 3) == 0 && (d->out_stride & 3) == 0 && (!d->res || (d->res_stride & 3) == 0)) {
    switch (k.kchunks) {
      case 1: hipLaunchKernelGGL(k_gemm_h2p<1>, grid, dim3(256), 0, s, k); break;
      case 2: hipLaunchKernelGGL(k_gemm_h2p<2>, grid, dim3(256), 0, s, k); break;
      case 3: hipLaunchKernelGGL(k_gemm_h2p<3>, grid, dim3(256), 0, s, k); break;
      default: hipLaunchKernelGGL(k_gemm_h2p<4>, grid, dim3(256), 0, s, k); break;
    }
    COOCC_LAUNCH_CHECK("k_gemm_h2p");
    return COOCC_OK;
  }
  if (!zshare) {
    if (one) {
      if (table) hipLaunchKernelGGL((k_gemm_h2w<true, 1>), grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL((k_gemm_h2w<false, 1>), grid, dim3(256), 0, s, k);
    } else {
      // narrow outputs (Cout <= 64) through a row table: 256-row tiles, every wave on real columns (k_gemm_h2n)
      const char* narrow_env = getenv("COOCC_H2_NARROW");          // read per call: tests compare the two kernels in one process
      const bool narrow_on = !(narrow_env && atoi(narrow_env) == 0);
      if (narrow_on && table && k.Cout <= 64 && k.ntiles == 1 && k.splitk == 1 && !d->tile_sem && (long long)k.mtiles >= 512) {
        const dim3 gridn((unsigned)((k.M + 255) / 256), 1);
        if (k.Cout <= 32) hipLaunchKernelGGL(k_gemm_h2n<1>, gridn, dim3(256), 0, s, k);
        else hipLaunchKernelGGL(k_gemm_h2n<2>, gridn, dim3(256), 0, s, k);
        COOCC_LAUNCH_CHECK("k_gemm_h2n");
        return COOCC_OK;
      }
      // full-width row-table layers whose 128-row tiles do not fill the chip evenly: 64-row tiles, three workgroups per CU
      if (narrow_on && table && k.Cout > 96 && k.ntiles == 1 && k.splitk == 1 && !d->tile_sem && (long long)k.mtiles >= 128 && (long long)k.mtiles < 1024) {
        const dim3 gridn((unsigned)((k.M + 63) / 64), 1);
        hipLaunchKernelGGL(k_gemm_h2n<4>, gridn, dim3(256), 0, s, k);
        COOCC_LAUNCH_CHECK("k_gemm_h2n<4>");
        return COOCC_OK;
      }
      if (table) hipLaunchKernelGGL(k_gemm_h2w<true>, grid, dim3(256), 0, s, k);
      else hipLaunchKernelGGL(k_gemm_h2w<false>, grid, dim3(256), 0, s, k);
    }
    COOCC_LAUNCH_CHECK("k_gemm_h2w");
    return COOCC_OK;
  }
  COOCC_CHECK_ARG((long long)d->M < (1ll << 30) && (136ull + 2ull * ((unsigned long long)k.Yi + 2) * k.Zi) * d->in_stride * 4ull < 0xFFFFFF00ull,
                  "conv_fwd: the split-f16 kernel addresses a tile's window with 32-bit byte offsets");
  const bool xy = !(k.kx == 1 && k.ky == 1 && k.px == 0 && k.py == 0);
  if (one) {
    if (k.kz == 3) { if (xy) hipLaunchKernelGGL((k_gemm_h2z<3, true, 0, 1>), grid, dim3(256), 0, s, k); else hipLaunchKernelGGL((k_gemm_h2z<3, false, 0, 1>), grid, dim3(256), 0, s, k); }
    else if (k.kz == 2) { if (xy) hipLaunchKernelGGL((k_gemm_h2z<2, true, 0, 1>), grid, dim3(256), 0, s, k); else hipLaunchKernelGGL((k_gemm_h2z<2, false, 0, 1>), grid, dim3(256), 0, s, k); }
    else { if (xy) hipLaunchKernelGGL((k_gemm_h2z<1, true, 0, 1>), grid, dim3(256), 0, s, k); else hipLaunchKernelGGL((k_gemm_h2z<1, false, 0, 1>), grid, dim3(256), 0, s, k); }
    COOCC_LAUNCH_CHECK("k_gemm_h2z<f16>");
    return COOCC_OK;
  }
  static const int abl = getenv("COOCC_H2_ABLATE") ? atoi(getenv("COOCC_H2_ABLATE")) : 0;     // timing ablations (wrong results)
  if (abl && !xy && k.kz == 3) {
    switch (abl) {
      case 1: hipLaunchKernelGGL((k_gemm_h2z<3, false, 1>), grid, dim3(256), 0, s, k); break;
      case 2: hipLaunchKernelGGL((k_gemm_h2z<3, false, 2>), grid, dim3(256), 0, s, k); break;
      case 4: hipLaunchKernelGGL((k_gemm_h2z<3, false, 4>), grid, dim3(256), 0, s, k); break;
      case 8: hipLaunchKernelGGL((k_gemm_h2z<3, false, 8>), grid, dim3(256), 0, s, k); break;
      case 7: hipLaunchKernelGGL((k_gemm_h2z<3, false, 7>), grid, dim3(256), 0, s, k); break;
      case 16: hipLaunchKernelGGL((k_gemm_h2z<3, false, 16>), grid, dim3(256), 0, s, k); break;
      case 24: hipLaunchKernelGGL((k_gemm_h2z<3, false, 24>), grid, dim3(256), 0, s, k); break;
      default: hipLaunchKernelGGL((k_gemm_h2z<3, false, 15>), grid, dim3(256), 0, s, k); break;
    }
    COOCC_LAUNCH_CHECK("k_gemm_h2z<ablation>");
    return COOCC_OK;
  }
  if (!xy) {
    if (k.kz == 3) hipLaunchKernelGGL((k_gemm_h2z<3, false>), grid, dim3(256), 0, s, k);
    else if (k.kz == 2) hipLaunchKernelGGL((k_gemm_h2z<2, false>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((k_gemm_h2z<1, false>), grid, dim3(256), 0, s, k);
  } else {
    if (k.kz == 3) hipLaunchKernelGGL((k_gemm_h2z<3, true>), grid, dim3(256), 0, s, k);
    else if (k.kz == 2) hipLaunchKernelGGL((k_gemm_h2z<2, true>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((k_gemm_h2z<1, true>), grid, dim3(256), 0, s, k);
  }
  COOCC_LAUNCH_CHECK("k_gemm_h2z");
  return COOCC_OK;
}

// fp32 rows (row stride in_stride floats, first C columns, C % 32 == 0) * scale -> H2 rows [rows][C/32][hi 32 | lo 32] (4 C bytes per row)
__global__ __launch_bounds__(256) void k_rows_to_h2(const float* __restrict__ in, int in_stride, long long rows, int C, float scale,
                                                     char* __restrict__ out, const int32_t* __restrict__ row_ids,
                                                     const int32_t* __restrict__ n_dev, int* __restrict__ flag,
                                                     const float* __restrict__ scale_dev) {
  if (n_dev) rows = min(rows, (long long)*n_dev);
  if (scale_dev) scale *= *scale_dev;
  const int c8 = C >> 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c8) return;
  const long long r = i / c8;
  const int c = (int)(i - r * c8) * 8;
  const long long sr = row_ids ? (long long)row_ids[r] : r;
  const f32x4 a = *(const f32x4*)(in + sr * in_stride + c), b = *(const f32x4*)(in + sr * in_stride + c + 4);
  f16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    _Float16 u, v;
    split_h2(a[e] * scale, u, v); hi[e] = u; lo[e] = v;
    split_h2(b[e] * scale, u, v); hi[4 + e] = u; lo[4 + e] = v;
  }
  char* o = out + r * (long long)C * 4 + (c >> 5) * 128 + (c & 31) * 2;
  *(f16x8*)o = hi;
  *(f16x8*)(o + 64) = lo;
  h2_guard(flag, a * scale);
  h2_guard(flag, b * scale);
}

extern "C" int coocc_rows_to_h2_ex(const float* in, int in_stride, int64_t rows, int C, float scale, const float* scale_dev,
                                   void* out_h2, void* stream) {
  COOCC_CHECK_ARG(in && out_h2 && rows >= 0 && C > 0 && C % 32 == 0 && in_stride % 4 == 0 && in_stride >= C, "rows_to_h2: bad args");
  COOCC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)out_h2 & 15) == 0, "rows_to_h2: pointers must be 16-byte aligned");
  if (rows == 0) return COOCC_OK;
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  hipLaunchKernelGGL(k_rows_to_h2, dim3(cdiv(rows * (C / 8), 256)), dim3(256), 0, as_stream(stream), in, in_stride, (long long)rows, C,
                     scale, (char*)out_h2, (const int32_t*)nullptr, (const int32_t*)nullptr, flag, scale_dev);
  COOCC_LAUNCH_CHECK("k_rows_to_h2");
  return COOCC_OK;
}

extern "C" int coocc_rows_to_h2(const float* in, int in_stride, int64_t rows, int C, float scale, void* out_h2, void* stream) {
  return coocc_rows_to_h2_ex(in, in_stride, rows, C, scale, nullptr, out_h2, stream);
}

// out row j = H2(in[row_ids[j]]) for j < n (n_dev != NULL: n = min(n_cap, *n_dev) read on the device): the compact operand of a
// GEMM over a voxel list (the scatter-form half of con_enc.0)
extern "C" int coocc_rows_to_h2_gather(const float* in, int in_stride, const int32_t* row_ids, int64_t n_cap, const int32_t* n_dev,
                                       int C, float scale, void* out_h2, void* stream) {
  COOCC_CHECK_ARG(in && out_h2 && row_ids && n_cap >= 0 && C > 0 && C % 32 == 0 && in_stride % 4 == 0, "rows_to_h2_gather: bad args");
  COOCC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)out_h2 & 15) == 0, "rows_to_h2_gather: pointers must be 16-byte aligned");
  if (n_cap == 0) return COOCC_OK;
  int* flag = nullptr;
  if (coocc_h2_flag_ptr(&flag) != COOCC_OK) return COOCC_EHIP;
  hipLaunchKernelGGL(k_rows_to_h2, dim3(cdiv(n_cap * (C / 8), 256)), dim3(256), 0, as_stream(stream), in, in_stride, (long long)n_cap, C,
                     scale, (char*)out_h2, row_ids, n_dev, flag, (const float*)nullptr);
  COOCC_LAUNCH_CHECK("k_rows_to_h2");
  return COOCC_OK;
}

// fp32 rows -> f16 rows [rows][C] (RNE): the activation operand of the one-term f16 path when the producer did not write it (out16)
__global__ __launch_bounds__(256) void k_rows_to_f16(const float* __restrict__ in, int in_stride, long long rows, int C, _Float16* __restrict__ out) {
  const int c8 = C >> 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c8) return;
  const long long r = i / c8;
  const int c = (int)(i - r * c8) * 8;
  const f32x4 a = *(const f32x4*)(in + r * in_stride + c), b = *(const f32x4*)(in + r * in_stride + c + 4);
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = (_Float16)a[e]; o[4 + e] = (_Float16)b[e]; }
  *(f16x8*)(out + r * C + c) = o;
}

extern "C" int coocc_rows_to_f16(const float* in, int in_stride, int64_t rows, int C, void* out_f16, void* stream) {
  COOCC_CHECK_ARG(in && out_f16 && rows >= 0 && C > 0 && C % 8 == 0 && in_stride % 4 == 0 && in_stride >= C, "rows_to_f16: bad args");
  COOCC_CHECK_ARG(((uintptr_t)in & 15) == 0 && ((uintptr_t)out_f16 & 15) == 0, "rows_to_f16: pointers must be 16-byte aligned");
  if (rows == 0) return COOCC_OK;
  hipLaunchKernelGGL(k_rows_to_f16, dim3(cdiv(rows * (C / 8), 256)), dim3(256), 0, as_stream(stream), in, in_stride, (long long)rows, C,
                     (_Float16*)out_f16);
  COOCC_LAUNCH_CHECK("k_rows_to_f16");
  return COOCC_OK;
}
