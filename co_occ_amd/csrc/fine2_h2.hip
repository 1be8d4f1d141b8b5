// OccHead fine branch, cascade ratio 2, in ONE launch on the split-f16 engine (round 5; occ_head.py:205-233, eval, B == 1).
//
// Round 4 ran the headline configuration's branch as three launches -- k_fine_sample_voxel_r2 (88 us) -> k_fine_sample_img_grp
// (84 us) -> k_fine_mlp<pre> (154 us, fp32 MFMA) -- with both [8 n, 64] sample matrices (2 x 154 MB at 600 k points) written and
// read back.  The ratio-4 one-launch kernel (fine_fused.hip) was measured SLOWER than that at ratio 2 (383 us): lanes = channels
// (one 256-byte row per load instruction), one coarse voxel after the other, every weight / tap offset broadcast by readlane.
// A first lanes = points version of this file (each lane fetching 16-byte pieces of ITS point's rows: 32 cache lines touched per
// load instruction) took 260 us.  This version:
//
//   * a wave owns 32 fine points = the 8 children of 4 consecutive foreground coarse voxels.  SAMPLING layout: lane = (child pt8,
//     piece): the 8 lanes of a child fetch one 128-byte half row (channels 32 i + 4 piece + 0..3) with ONE coalesced line per
//     8 lanes, 8 lines per instruction; the four coarse voxels (j) live in registers.  A GroupNorm group (4 consecutive channels)
//     is one register quad of one lane, so bias + GroupNorm + ReLU of img_mlp run in place;
//   * the projections are evaluated ONCE per (point, camera) with lanes = points (two cameras per pass, one per half-wave) and
//     left in the wave's LDS tile as 32-byte records (4 row offsets + 4 bilinear weights); a ballot per pass gives the wave-uniform
//     visibility masks that skip whole (camera, coarse voxel) pairs;
//   * MFMA layout: lane (li, h) = point li as the column of v_mfma_f32_32x32x16_f16.  The wave's 8.5 KB LDS tile transposes between
//     the layouts: the trilinear voxel sum goes through it as fp32 and comes back as the INITIAL ACCUMULATOR of fine_mlp[0]
//     (Linear commutes with the resampling: Q = W_f0[:, :128] . voxel features, as before); the image activation goes through it as
//     f16 hi / lo halves (16-byte slots XOR-swizzled by the row, f2_swz) and comes back as B fragments; the logits leave through it as
//     contiguous runs of rows o n + i .. i + 3;
//   * both GEMMs of the chain (64 x 64 and ncls x 64 per point) are three MFMAs per k16 step (hi hi, lo hi, hi lo, fp32 accumulate:
//     gemm_h2.hip's scheme) instead of eight v_mfma_f32_32x32x2_f32; the 24 KB of packed weights sit in LDS, loaded once per
//     (persistent) workgroup; the second GEMM takes its B operand straight from the registers of the first (k-slot order = register
//     order, the weight pack carries the permutation).
//
// Arithmetic per point: the resampling sums run in the order of the three-kernel path (cameras ascending, taps (y, x) ascending;
// voxel taps x, y, z ascending, weight (wx wy) wz; out-of-range taps contribute + 0 instead of being skipped); the MLP differs from
// k_fine_mlp<pre> by the split-f16 products (2^-22 relative per product) -- tests/test_gpu_modules.py checks both against the oracle.
#include <stdlib.h>
#include <string.h>

#include "conv_k.h"
#include "h2_rows.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define F2_CAM_STRIDE 27
#define F2_HDR 17
#define F2_TP 68                // floats per row of the wave's fp32 tile (32 rows x 64 channels + 4: conflict-free C-layout reads)
#define F2_OT_PITCH 36          // floats per staged logits row (32 classes max + 4)
#define F2_WBYTES 24576         // fine_mlp[0] image half: 2 m-tiles x 4 k16 x {hi, lo} x 1 KB; fine_mlp[3]: 1 x 4 x 2 x 1 KB
#define F2_NCONST 416           // b_img g_img be_img b_f0 g_f0 be_f0 [64 each] | b_f3 [32]
#define F2_MAXCAM 8

struct Fine2K {
  const float* Q;            // [X*Y*Z, 64]  W_f0[:, :128] . voxel features
  const float* P;            // [ncam*Hf*Wf, 64]  W_img . image features
  const float* samp;         // IMG == false: the image samples [8 n, 64] (row f = o n + i) made by coocc_fine_sample_img_lin
  const float* prm;          // coocc_projection_params
  const int32_t* lin;        // foreground coarse voxels (linear ids)
  const int32_t* n_dev;      // optional device-side count
  int64_t* fine_xyz;         // [3][8 n]
  float* out;                // [8 n, ncls], row f = o * n + i
  const char* wpack;         // coocc_fine2_pack
  const float* consts;
  int* h2_flag;
  int n, X, Y, Z, ncam, Hf, Wf, ncls, q_stride, dbg;
  float fx1, fy1, fz1, eps_img, eps_f0;
};

// fine_mlp[3] takes its B operand from the registers fine_mlp[0] left: k-slot j (0..7) of k16 step s held by lane half hh <->
// hidden channel (register r = 4 g + t of channel tile i holds channel 32 i + 8 g + 4 hh + t; step s takes registers
// 8 (s & 1) .. + 7 of tile s >> 1)
__host__ __device__ inline int f2_chan(int s, int hh, int j) { return 32 * (s >> 1) + 8 * (2 * (s & 1) + (j >> 2)) + 4 * hh + (j & 3); }

// Weights and constants of the chain -> the kernel's operand order.  w_f0: [64][192] (columns 128..191 = the image half),
// w_f3: [ncls][64].  wpack: [m tile][k16 step][hi | lo][lane = (m, hh)][8 f16]; A operand W[32 mt + m][k]: natural k =
// 16 s + 8 hh + j for fine_mlp[0] (its B operand comes from the LDS tile), f2_chan(s, hh, j) for fine_mlp[3].
__global__ void k_fine2_pack(const float* __restrict__ w_f0, const float* __restrict__ w_f3, int ncls, const float* b_img,
                             const float* g_img, const float* be_img, const float* b_f0, const float* g_f0, const float* be_f0,
                             const float* b_f3, _Float16* __restrict__ wpack, float* __restrict__ consts) {
  for (int e = threadIdx.x; e < 3 * 4 * 64 * 8; e += blockDim.x) {
    const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 3, mt = e >> 11;         // mt 0,1: fine_mlp[0]; 2: fine_mlp[3]
    const int m = lane & 31, hh = lane >> 5;
    float w;
    if (mt < 2) w = w_f0[(32 * mt + m) * 192 + 128 + 16 * s + 8 * hh + j];
    else w = m < ncls ? w_f3[m * 64 + f2_chan(s, hh, j)] : 0.f;
    _Float16 hi, lo;
    split_h2(w, hi, lo);
    const size_t base = ((size_t)(mt * 4 + s) * 2) * 512 + lane * 8 + j;               // f16 elements; 512 per (plane) KB
    wpack[base] = hi;
    wpack[base + 512] = lo;
  }
  for (int c = threadIdx.x; c < 64; c += blockDim.x) {
    consts[c] = b_img[c]; consts[64 + c] = g_img[c]; consts[128 + c] = be_img[c];
    consts[192 + c] = b_f0[c]; consts[256 + c] = g_f0[c]; consts[320 + c] = be_f0[c];
    if (c < 32) consts[384 + c] = c < ncls ? b_f3[c] : 0.f;
  }
}

// bias + GroupNorm over the 4 channels of one register quad + ReLU: the arithmetic of k_groupnorm_rows / k_fine_mlp
__device__ __forceinline__ f32x4 f2_gn4(f32x4 v, f32x4 bi, f32x4 ga, f32x4 be, float eps) {
  float x[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) x[s] = v[s] + bi[s];
  float mean = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) mean += x[s];
  mean /= 4.0f;
  float var = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) { float d = x[s] - mean; var += d * d; }
  var /= 4.0f;
  const float rstd = 1.f / sqrtf(var + eps);
  f32x4 y;
#pragma unroll
  for (int s = 0; s < 4; ++s) y[s] = fmaxf((x[s] - mean) * rstd * ga[s] + be[s], 0.f);
  return y;
}

__device__ __forceinline__ void f2_split8(const f32x16& y, int r0, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 a, b;
    split_h2(y[r0 + j], a, b);
    hi[j] = a; lo[j] = b;
  }
}

// 16-byte-slot swizzle of the f16 activation tile (32 rows x 128 B hi, + 4096: lo): slot' = slot ^ f2_swz(row).  Round 5's slot ^ (row & 7)
// left every access 2-way conflicted (99.9 M SQ_LDS_BANK_CONFLICT cycles per launch, profiles/r5_bench_pmc_sq.txt): a ds_read_b128 lane
// group holds rows {0-3, 12-15, 20-27} (or {4-11, 16-19, 28-31}), whose row & 7 repeats, and the two rows of a ds_write_b64 group differ in
// bit 0 only, i.e. in one slot, while a write's banks are taken mod 32 (both rows of a 256-byte bank row).  With
//     f2_swz(r) = (r >> 1 & 3) | ((r >> 4 ^ r) & 1) << 2
// the 8 even (odd) rows of a read group get 8 distinct slots of their half of the bank row, and rows r, r + 1 of a write group use
// opposite halves of the 32 banks.  A pure layout change: same values, same bits.
__device__ __forceinline__ unsigned f2_swz(int r) { return (unsigned)((r >> 1) & 3) | ((unsigned)(((r >> 4) ^ r) & 1) << 2); }

#define F2_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")      // the tile is private to the wave (LDS is in order per wave)

// IMG: the image samples are made here (phases P and B below); false: they arrive from the grouped sampler's own launch
// (head.FINE2_IMG_INSIDE, default 1 since round 6.  The IMG form was off in round 5 -- a few hundred rows wrong next to other streams'
// split-f16 GEMMs: its sampling code compiled to 132 packed-fp32 instructions with op_sel[src1] = 1, see COOCC_SCALAR_FP32 in common.h)
template <int NW, bool IMG = true>            // waves per workgroup: 4 (two waves per SIMD at two workgroups per CU) | 6 (three; <= 168 registers)
__global__ COOCC_SCALAR_FP32 __launch_bounds__(64 * NW, NW == 6 ? 3 : 2) void k_fine2_h2(Fine2K p) {       // (threads, min waves per SIMD)
  __shared__ __attribute__((aligned(16))) char Wl[F2_WBYTES];
  __shared__ __attribute__((aligned(16))) float Cn[F2_NCONST];
  __shared__ __attribute__((aligned(16))) float Tl[NW][32 * F2_TP];
  const int tid = threadIdx.x;
  if (p.dbg & 1) asm volatile("buffer_inv sc1\n\ts_dcache_inv\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // debug: drop L1 / scalar cache lines
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;             // MFMA layout: point column li, k / row half h
  const int pt8 = lane >> 3, piece = lane & 7;         // sampling layout: child pt8 of coarse voxel j, channels 32 i + 4 piece + 0..3
  for (int i = tid; i < F2_WBYTES / 16; i += 64 * NW) ((f32x4*)Wl)[i] = ((const f32x4*)p.wpack)[i];
  for (int i = tid; i < F2_NCONST; i += 64 * NW) Cn[i] = p.consts[i];
  __syncthreads();
  int n = p.n;
  if (p.n_dev) n = min(n, *p.n_dev);
  const int ntiles = (n + 4 * NW - 1) / (4 * NW);      // 4 coarse voxels = 32 fine points per wave
  const long long nf = (long long)n * 8;
  const int X = p.X, Y = p.Y, Z = p.Z, Hf = p.Hf, Wf = p.Wf, ncls = p.ncls, ncam = p.ncam;
  const float* __restrict__ prm = p.prm;
  float* T = Tl[wave];
  char* Tb = (char*)T;
  float gmax = 0.f;                                    // range guard of the two 16-bit operand conversions

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ci0 = (tile * NW + wave) * 4;
    if (ci0 >= n) continue;                            // wave-uniform; no workgroup barrier inside the loop
    const int nc = min(4, n - ci0);
    if (p.dbg & 6) {                                   // debug: the tile starts from zeros (2) / from 0xFF bytes (4)
      for (int i = lane; i < 32 * F2_TP; i += 64) ((unsigned*)T)[i] = (p.dbg & 4) ? 0xFFFFFFFFu : 0u;
      F2_LDS_FENCE();
    }
    int cx[4], cy[4], cz[4];                           // the wave's coarse voxels (wave-uniform values)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int l = p.lin[ci0 + min(j, nc - 1)];
      cz[j] = l % Z; l /= Z;
      cy[j] = l % Y; cx[j] = l / Y;                    // B == 1
    }

    // ---- A. voxel samples, sampling layout: trilinear sample of Q at the fine voxel centre (k_fine_sample_voxel_r2's arithmetic).
    // Two coarse voxels at a time: 32 independent 16-byte loads per lane in flight before the first is consumed (the branch is
    // bound by latency x bytes in flight, not by arithmetic); voxels past the end of the list repeat the last one (rows unused).
    {
      const int oa = pt8 >> 2, ob = (pt8 >> 1) & 1, oc = pt8 & 1;
      const char* vol = (const char*)p.Q + 16 * piece;              // wave-uniform base + 32-bit byte offsets
      const unsigned rb = (unsigned)p.q_stride * 4u;                // bytes per row of Q
      const float f1[3] = {p.fx1, p.fy1, p.fz1}; const int S[3] = {X, Y, Z};
      const unsigned ms[3] = {(unsigned)(Y * Z) * rb, (unsigned)Z * rb, rb};
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        f32x4 v[2][8][2]; float w[2][8];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          const int q3[3] = {cx[j] * 2 + oa, cy[j] * 2 + ob, cz[j] * 2 + oc};
          unsigned ao[3][2]; float aw[3][2];                        // per axis: byte offset and weight of the two taps (out of range: 0, 0)
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float g = ((float)q3[a] / f1[a] - 0.5f) * 2.f;
            const float pp = ((g + 1.f) * (float)S[a] - 1.f) / 2.f;
            const float fl = floorf(pp);
            const int b0 = (int)fl; const float t = pp - fl;
            const bool in0 = (unsigned)b0 < (unsigned)S[a], in1 = (unsigned)(b0 + 1) < (unsigned)S[a];
            ao[a][0] = in0 ? (unsigned)b0 * ms[a] : 0u; ao[a][1] = in1 ? (unsigned)(b0 + 1) * ms[a] : 0u;
            aw[a][0] = in0 ? 1.f - t : 0.f; aw[a][1] = in1 ? t : 0.f;
          }
#pragma unroll
          for (int tp = 0; tp < 8; ++tp) {
            const int dx = tp >> 2, dy = (tp >> 1) & 1, dz = tp & 1;
            w[jj][tp] = (aw[0][dx] * aw[1][dy]) * aw[2][dz];
            const char* rp = vol + (ao[0][dx] + ao[1][dy] + ao[2][dz]);
            v[jj][tp][0] = *(const f32x4*)rp; v[jj][tp][1] = *(const f32x4*)(rp + 128);
          }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int tp = 0; tp < 8; ++tp) { acc[0] = acc[0] + v[jj][tp][0] * w[jj][tp]; acc[1] = acc[1] + v[jj][tp][1] * w[jj][tp]; }
          *(f32x4*)(T + (8 * (2 * jp + jj) + pt8) * F2_TP + 4 * piece) = acc[0];
          *(f32x4*)(T + (8 * (2 * jp + jj) + pt8) * F2_TP + 32 + 4 * piece) = acc[1];
        }
      }
    }
    F2_LDS_FENCE();
    // ... back in the MFMA layout: the initial accumulator of fine_mlp[0] (row = hidden channel 32 mt + 8 g + 4 h + t, column = point li)
    f32x16 hh[2], xx2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(T + li * F2_TP + 32 * mt + 8 * g + 4 * h);
#pragma unroll
        for (int t = 0; t < 4; ++t) { hh[mt][4 * g + t] = v[t]; xx2[mt][4 * g + t] = 0.f; }
      }
    F2_LDS_FENCE();

    // ---- P. projections, lanes = points: camera 2 q + h of point li per pass -> 32-byte records in the tile + visibility ballots
    unsigned long long vis[F2_MAXCAM / 2];
    {
      const int kk = li >> 3, oo = li & 7;
      const bool valid = kk < nc;
      const int pcx = kk == 0 ? cx[0] : kk == 1 ? cx[1] : kk == 2 ? cx[2] : cx[3];
      const int pcy = kk == 0 ? cy[0] : kk == 1 ? cy[1] : kk == 2 ? cy[2] : cy[3];
      const int pcz = kk == 0 ? cz[0] : kk == 1 ? cz[1] : kk == 2 ? cz[2] : cz[3];
      const int fxi = pcx * 2 + (oo >> 2), fyi = pcy * 2 + ((oo >> 1) & 1), fzi = pcz * 2 + (oo & 1);
      if (valid && h == 0) {
        const long long f = (long long)oo * n + (ci0 + kk);
        p.fine_xyz[f] = fxi; p.fine_xyz[nf + f] = fyi; p.fine_xyz[2 * nf + f] = fzi;
      }
      const float p0 = (float)fxi * prm[9] + prm[12];
      const float p1 = (float)fyi * prm[10] + prm[13];
      const float p2 = (float)fzi * prm[11] + prm[14];
      const float bx = prm[0] * p0 + prm[1] * p1 + prm[2] * p2;
      const float by = prm[3] * p0 + prm[4] * p1 + prm[5] * p2;
      const float bz = prm[6] * p0 + prm[7] * p1 + prm[8] * p2;
      const float wimg1 = prm[15], himg1 = prm[16];
#pragma unroll
      for (int qq = 0; qq < F2_MAXCAM / 2; ++qq) {
        vis[qq] = 0ull;
        if (!IMG || 2 * qq >= ncam) continue;          // wave-uniform
        const int cam = 2 * qq + h;
        const bool camok = cam < ncam;
        const float* q = prm + F2_HDR + (camok ? cam : 0) * F2_CAM_STRIDE;
        const float tx = bx - q[9], ty = by - q[10], tz = bz - q[11];
        const float ccx = q[0] * tx + q[1] * ty + q[2] * tz;
        const float ccy = q[3] * tx + q[4] * ty + q[5] * tz;
        const float ccz = q[6] * tx + q[7] * ty + q[8] * tz;
        const float ix = q[12] * ccx + q[13] * ccy + q[14] * ccz;
        const float iy = q[15] * ccx + q[16] * ccy + q[17] * ccz;
        const float d = q[18] * ccx + q[19] * ccy + q[20] * ccz;
        const float u = ix / (d + 1e-5f), v = iy / (d + 1e-5f);
        float u2 = q[21] * u + q[22] * v + q[25];
        float v2 = q[23] * u + q[24] * v + q[26];
        u2 = (u2 / wimg1 - 0.5f) * 2.f;
        v2 = (v2 / himg1 - 0.5f) * 2.f;
        const bool m = valid && camok && d > 1e-5f && u2 > -1.f && u2 < 1.f && v2 > -1.f && v2 < 1.f;
        const float px = (u2 + 1.f) / 2.f * (float)(Wf - 1), py = (v2 + 1.f) / 2.f * (float)(Hf - 1);
        const float flx = floorf(px), fly = floorf(py);
        const int x0 = m ? (int)flx : 0, y0 = m ? (int)fly : 0;
        const float ax = px - flx, ay = py - fly;
        int off[4]; float w[4];
#pragma unroll
        for (int yy = 0; yy < 2; ++yy)
#pragma unroll
          for (int xx = 0; xx < 2; ++xx) {
            const int x = x0 + xx, y = y0 + yy;
            const bool in = m && (unsigned)x < (unsigned)Wf && (unsigned)y < (unsigned)Hf;
            off[yy * 2 + xx] = ((camok ? cam : 0) * Hf * Wf + (in ? y * Wf + x : 0)) * 256;      // byte offset of the row in P
            w[yy * 2 + xx] = in ? (xx ? ax : 1.f - ax) * (yy ? ay : 1.f - ay) : 0.f;
          }
        vis[qq] = __ballot(m);
        if (camok) {
          int* rec = (int*)(Tb + ((size_t)cam * 32 + li) * 32);
          *(int4*)rec = int4{off[0], off[1], off[2], off[3]};
          *(f32x4*)(rec + 4) = f32x4{w[0], w[1], w[2], w[3]};
        }
      }
    }
    F2_LDS_FENCE();

    // ---- B. image samples, sampling layout: sum over the cameras that see the point of the bilinear sample of P.  Per camera two
    // coarse voxels at a time (16 loads per lane in flight); a point the camera does not see has weight 0 and row 0 in its record
    f32x4 ai[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ai[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ai[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (!IMG) {
      // the finished samples of this lane's four children: rows (pt8 n + ci0 + j), channels 32 i + 4 piece + 0..3
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* sp = p.samp + ((size_t)pt8 * n + ci0 + min(j, nc - 1)) * 64 + 4 * piece;
        ai[j][0] = *(const f32x4*)sp; ai[j][1] = *(const f32x4*)(sp + 32);
      }
    } else {
      const char* img = (const char*)p.P + 16 * piece;
#pragma unroll
      for (int cam = 0; cam < F2_MAXCAM; ++cam) {
        if (cam >= ncam) break;                          // wave-uniform
        const unsigned vm = (unsigned)(vis[cam >> 1] >> (32 * (cam & 1)));
        if (vm == 0u) continue;                          // nobody in this wave is seen by the camera
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          if (((vm >> (16 * jp)) & 0xffffu) == 0u) continue;       // wave-uniform
          f32x4 v[2][4][2]; f32x4 w[2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int* rec = (const int*)(Tb + ((size_t)cam * 32 + 8 * (2 * jp + jj) + pt8) * 32);
            const int4 off = *(const int4*)rec;
            w[jj] = *(const f32x4*)(rec + 4);
            const int o4[4] = {off.x, off.y, off.z, off.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const char* rp = img + (unsigned)o4[t];
              if (p.dbg & 8) {                         // debug: 8-byte loads instead of 16-byte ones
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 a0 = *(const volatile f32x2*)rp, a1 = *(const volatile f32x2*)(rp + 8);
                const f32x2 b0 = *(const volatile f32x2*)(rp + 128), b1 = *(const volatile f32x2*)(rp + 136);
                v[jj][t][0] = f32x4{a0[0], a0[1], a1[0], a1[1]}; v[jj][t][1] = f32x4{b0[0], b0[1], b1[0], b1[1]};
              } else {
                v[jj][t][0] = *(const f32x4*)rp; v[jj][t][1] = *(const f32x4*)(rp + 128);
              }
            }
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              ai[2 * jp + jj][0] = ai[2 * jp + jj][0] + v[jj][t][0] * w[jj][t];
              ai[2 * jp + jj][1] = ai[2 * jp + jj][1] + v[jj][t][1] * w[jj][t];
            }
        }
      }
    }
    F2_LDS_FENCE();                                      // every record has been read: the tile becomes the B operand of fine_mlp[0]
    // img_mlp's bias + GroupNorm + ReLU in place, split into f16 halves -> tile rows [hi 128 B | lo 128 B at + 4096], slot ^ f2_swz(row)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const f32x4 bi = *(const f32x4*)(Cn + 32 * i + 4 * piece), ga = *(const f32x4*)(Cn + 64 + 32 * i + 4 * piece),
                  be = *(const f32x4*)(Cn + 128 + 32 * i + 4 * piece);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 y = f2_gn4(ai[j][i], bi, ga, be, p.eps_img);
        gmax = fmaxf(gmax, fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])));
        f16x4 yh, yl;
#pragma unroll
        for (int t = 0; t < 4; ++t) { _Float16 a, b; split_h2(y[t], a, b); yh[t] = a; yl[t] = b; }
        const unsigned slot = (unsigned)(4 * i + (piece >> 1)) ^ f2_swz(8 * j + pt8);
        char* dst = Tb + (8 * j + pt8) * 128 + (slot << 4) + (piece & 1) * 8;
        *(f16x4*)dst = yh;
        *(f16x4*)(dst + 4096) = yl;
      }
    }
    F2_LDS_FENCE();

    // ---- C. fine_mlp[0]: h = Q sample + W_f0[:, 128:] . y1, three MFMAs per k16 step
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const char* src = Tb + li * 128 + (((unsigned)(2 * s + h) ^ f2_swz(li)) << 4);
      const f16x8 bhi = *(const f16x8*)src, blo = *(const f16x8*)(src + 4096);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const f16x8 ahi = *(const f16x8*)&Wl[((mt * 4 + s) * 2 + 0) * 1024 + lane * 16];
        const f16x8 alo = *(const f16x8*)&Wl[((mt * 4 + s) * 2 + 1) * 1024 + lane * 16];
        hh[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, hh[mt], 0, 0, 0);
        xx2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, xx2[mt], 0, 0, 0);
        xx2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, xx2[mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 32 * i + 8 * g + 4 * h;
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = hh[i][4 * g + t] + xx2[i][4 * g + t] * (1.f / H2_LO_SCALE);
        v = f2_gn4(v, *(const f32x4*)(Cn + 192 + c0), *(const f32x4*)(Cn + 256 + c0), *(const f32x4*)(Cn + 320 + c0), p.eps_f0);
        gmax = fmaxf(gmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
#pragma unroll
        for (int t = 0; t < 4; ++t) hh[i][4 * g + t] = v[t];
      }

    // ---- D. fine_mlp[3]: Linear(64 -> ncls <= 32), B operand from the registers; weight rows >= ncls are zero in the pack
    f32x16 o, ox;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[r] = 0.f; ox[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f16x8 bhi, blo;
      f2_split8(hh[s >> 1], 8 * (s & 1), bhi, blo);
      const f16x8 ahi = *(const f16x8*)&Wl[((8 + s) * 2 + 0) * 1024 + lane * 16];
      const f16x8 alo = *(const f16x8*)&Wl[((8 + s) * 2 + 1) * 1024 + lane * 16];
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, o, 0, 0, 0);
      ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, ox, 0, 0, 0);
      ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, ox, 0, 0, 0);
    }
    // ---- E. logits of point li, classes 8 j + 4 h + t -> the tile, then out rows (o n + ci0 .. + nc - 1) as contiguous runs
    F2_LDS_FENCE();                                      // the B fragments of step C have been read
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bi = *(const f32x4*)(Cn + 384 + 8 * j + 4 * h);
      f32x4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = o[4 * j + t] + ox[4 * j + t] * (1.f / H2_LO_SCALE) + bi[t];
      *(f32x4*)(T + li * F2_OT_PITCH + 8 * j + 4 * h) = v;
    }
    F2_LDS_FENCE();
    {
      const int seg = nc * ncls, total = 8 * seg;                // <= 128, <= 1024
      const unsigned inv_seg = ((1u << 20) + seg - 1) / seg, inv_cls = ((1u << 20) + ncls - 1) / ncls;
#pragma unroll 1
      for (int idx = lane; idx < total; idx += 64) {
        const int oq = (int)(((unsigned)idx * inv_seg) >> 20), rem = idx - oq * seg;
        const int kq = (int)(((unsigned)rem * inv_cls) >> 20), c = rem - kq * ncls;
        p.out[((long long)oq * n + ci0) * ncls + rem] = T[(kq * 8 + oq) * F2_OT_PITCH + c];
      }
    }
    F2_LDS_FENCE();                                      // ... read before the next tile overwrites it
  }
  if (p.h2_flag && !(gmax < H2_GUARD)) *(volatile int*)p.h2_flag = 1;
}

static bool f2_aligned16(const void* a) { return ((uintptr_t)a & 15) == 0; }

// Packs the chain's weights / constants for coocc_fine2_h2 (once per weight version; host side: head.OccHead._packed).
// wpack: 24576 bytes, consts: 416 floats.
extern "C" int coocc_fine2_pack(const float* w_f0, const float* w_f3, int ncls, const float* b_img, const float* gn_img_w,
                                const float* gn_img_b, const float* b_f0, const float* gn_f0_w, const float* gn_f0_b, const float* b_f3,
                                void* wpack, float* consts, void* stream) {
  COOCC_CHECK_ARG(w_f0 && w_f3 && b_img && gn_img_w && gn_img_b && b_f0 && gn_f0_w && gn_f0_b && b_f3 && wpack && consts, "fine2_pack: null pointer");
  COOCC_CHECK_ARG(ncls >= 1 && ncls <= 32, "fine2_pack: 1 <= ncls <= 32");
  hipLaunchKernelGGL(k_fine2_pack, dim3(1), dim3(256), 0, as_stream(stream), w_f0, w_f3, ncls, b_img, gn_img_w, gn_img_b, b_f0, gn_f0_w,
                     gn_f0_b, b_f3, (_Float16*)wpack, consts);
  COOCC_LAUNCH_CHECK("k_fine2_pack");
  return COOCC_OK;
}

// Q: [X*Y*Z, 64] at q_stride floats per row, P: [ncam*Hf*Wf, 64] (the two Linear layers applied BEFORE the resamplings, as for
// coocc_fine_fused); lin: the n
// (with n_dev: at most n_cap) foreground coarse voxels; final_size == 2 * (X, Y, Z); wpack / consts from coocc_fine2_pack.
// Outputs as coocc_fine_fused: fine_xyz [3][8 n], logits [8 n, ncls] (row o * n + i).
// img_samples != NULL: [8 n, 64] image samples from coocc_fine_sample_img_lin (P and params are then unused and may be NULL).
extern "C" int coocc_fine2_h2(const float* Q, int q_stride, int X, int Y, int Z, const float* P, int ncam, int Hf, int Wf, const float* params,
                              const int32_t* coarse_lin, int n_cap, const int32_t* n_dev, const int* final_size_host, const void* wpack,
                              const float* consts, float eps_img, float eps_f0, int ncls, int64_t* fine_xyz, float* out,
                              const float* img_samples, void* stream) {
  COOCC_CHECK_ARG(Q && (img_samples || (P && params)) && coarse_lin && final_size_host && wpack && consts && fine_xyz && out, "fine2_h2: null pointer");
  COOCC_CHECK_ARG(n_cap >= 0 && ncls >= 1 && ncls <= 32 && ncam >= 1 && ncam <= F2_MAXCAM && X > 0 && Y > 0 && Z > 0 && Hf > 0 && Wf > 0,
                  "fine2_h2: ncls <= 32, <= 8 cameras");
  COOCC_CHECK_ARG(q_stride >= 64 && q_stride % 4 == 0, "fine2_h2: q_stride (floats per row of Q) >= 64, a multiple of 4");
  COOCC_CHECK_ARG(final_size_host[0] == 2 * X && final_size_host[1] == 2 * Y && final_size_host[2] == 2 * Z,
                  "fine2_h2: final_occ_size must be 2 x the coarse grid");
  COOCC_CHECK_ARG(f2_aligned16(Q) && f2_aligned16(P) && f2_aligned16(wpack) && f2_aligned16(consts) && f2_aligned16(img_samples),
                  "fine2_h2: arrays must be 16-byte aligned");
  if (n_cap == 0) return COOCC_OK;
  Fine2K p;
  memset(&p, 0, sizeof(p));
  p.Q = Q; p.P = P; p.samp = img_samples; p.prm = params ? params : consts; p.lin = coarse_lin; p.n_dev = n_dev; p.fine_xyz = fine_xyz; p.out = out;
  p.wpack = (const char*)wpack; p.consts = consts;
  if (coocc_h2_flag_ptr(&p.h2_flag) != COOCC_OK) return COOCC_EHIP;
  p.q_stride = q_stride;
  p.n = n_cap; p.X = X; p.Y = Y; p.Z = Z; p.ncam = ncam; p.Hf = Hf; p.Wf = Wf; p.ncls = ncls;
  p.fx1 = (float)(final_size_host[0] - 1); p.fy1 = (float)(final_size_host[1] - 1); p.fz1 = (float)(final_size_host[2] - 1);
  p.eps_img = eps_img; p.eps_f0 = eps_f0;
  // persistent: two workgroups per CU walk the tiles.  COOCC_FINE2_WAVES = 4 | 6 waves per workgroup
  static const int nw = getenv("COOCC_FINE2_WAVES") ? atoi(getenv("COOCC_FINE2_WAVES")) : 4;
  const int per = 4 * (nw == 6 ? 6 : 4);
  const long long tiles = ((long long)n_cap + per - 1) / per;
  // COOCC_FINE2_GRID=full: one tile per workgroup (debug: tools/debug/fine2_concurrent.py -- the run-to-run differences of the image
  // samples show up on every call with this grid, even on one stream)
  p.dbg = getenv("COOCC_FINE2_DBG") ? atoi(getenv("COOCC_FINE2_DBG")) : 0;
  const char* ge = getenv("COOCC_FINE2_GRID");
  const int grid = (ge && ge[0] == 'f') ? (int)tiles : (int)(tiles < 512 ? tiles : 512);
  // COOCC_FINE2_PADLDS: extra (unused) dynamic LDS per workgroup -- debug: keeps other kernels' workgroups off the CU
  const size_t pad = getenv("COOCC_FINE2_PADLDS") ? (size_t)atoi(getenv("COOCC_FINE2_PADLDS")) : 0;
  if (pad) {
    hipFuncSetAttribute((const void*)k_fine2_h2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
  }
  if (img_samples) hipLaunchKernelGGL((k_fine2_h2<4, false>), dim3(grid), dim3(256), pad, as_stream(stream), p);
  else if (nw == 6) hipLaunchKernelGGL(k_fine2_h2<6>, dim3(grid), dim3(384), pad, as_stream(stream), p);
  else hipLaunchKernelGGL(k_fine2_h2<4>, dim3(grid), dim3(256), pad, as_stream(stream), p);
  COOCC_LAUNCH_CHECK("k_fine2_h2");
  return COOCC_OK;
}
