// OccHead fine branch, cascade ratio 2, in ONE launch on the split-f16 engine (round 5; occ_head.py:205-233, eval, B == 1).
//
// Round 4 ran the headline configuration's branch as three launches -- k_fine_sample_voxel_r2 (88 us) -> k_fine_sample_img_grp
// (84 us) -> k_fine_mlp<pre> (154 us, fp32 MFMA) -- with both [8 n, 64] sample matrices (2 x 154 MB at 600 k points) written and
// read back.  The ratio-4 one-launch kernel (fine_fused.hip) was measured SLOWER than that at ratio 2 (383 us): lanes = channels,
// one coarse voxel after the other, every weight / tap offset broadcast by readlane, two waves per SIMD.  This kernel turns the
// problem round:
//
//   * lanes = POINTS.  A wave owns 32 fine points (the 8 children of 4 consecutive foreground coarse voxels); lane (li, h) holds,
//     for point li, the 32 channels {32 i + 8 g + 4 h + t} -- exactly the C/D layout of v_mfma_f32_32x32x16_f16 with the point as
//     the column.  Projection, visibility and interpolation weights are per-lane scalars (no broadcasts); a tap row of 64 channels
//     (256 B) is fetched as 8 x 16-byte loads by each of the two lanes of a point, every load independent of every other: the
//     branch is bound by L1/L2 row traffic (~2 GB), not by the latency of a serial gather.
//   * the samples never leave the register file: the bilinear image sum gets bias + GroupNorm + ReLU in place (a GroupNorm group is
//     4 consecutive channels = 4 registers of one lane), is split into f16 hi / lo halves in registers (the B operand of the next
//     MFMA, k-slot order chosen to match the registers: the weight pack carries the permutation), the trilinear voxel sum IS the
//     initial accumulator of fine_mlp[0] (Linear commutes with the resampling: Q = W_f0[:, :128] . voxel features, as before);
//   * both GEMMs of the chain (64 x 64 and ncls x 64 per point) are three v_mfma_f32_32x32x16_f16 per k16 step (hi hi, lo hi, hi lo,
//     fp32 accumulate: gemm_h2.hip's scheme) instead of eight v_mfma_f32_32x32x2_f32: 72 matrix instructions per 32 points; the
//     24 KB of packed weights sit in LDS, loaded once per (persistent) workgroup;
//   * the logits go through a 2.5 KB LDS tile per wave so that the rows o * n + i .. i + 3 of child o leave as contiguous runs.
//
// Arithmetic per point: the resampling sums run in the order of the three-kernel path (cameras ascending, taps (y, x) ascending;
// voxel taps x, y, z ascending, weight (wx wy) wz), so the samples are the same bits; the MLP differs from k_fine_mlp<pre> by the
// split-f16 products (2^-22 relative per product) -- tests/test_gpu_modules.py checks both against the oracle.
#include <stdlib.h>
#include <string.h>

#include "conv_k.h"
#include "h2_rows.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define F2_CAM_STRIDE 27
#define F2_HDR 17
#define F2_OT_PITCH 36          // floats per staged logits row (32 classes max + 4)
#define F2_WBYTES 24576         // fine_mlp[0] image half: 2 m-tiles x 4 k16 x {hi, lo} x 1 KB; fine_mlp[3]: 1 x 4 x 2 x 1 KB
#define F2_NCONST 416           // b_img g_img be_img b_f0 g_f0 be_f0 [64 each] | b_f3 [32]

struct Fine2K {
  const float* Q;            // [X*Y*Z, 64]  W_f0[:, :128] . voxel features
  const float* P;            // [ncam*Hf*Wf, 64]  W_img . image features
  const float* prm;          // coocc_projection_params
  const int32_t* lin;        // foreground coarse voxels (linear ids)
  const int32_t* n_dev;      // optional device-side count
  int64_t* fine_xyz;         // [3][8 n]
  float* out;                // [8 n, ncls], row f = o * n + i
  const char* wpack;         // coocc_fine2_pack
  const float* consts;
  int* h2_flag;
  int n, X, Y, Z, ncam, Hf, Wf, ncls;
  float fx1, fy1, fz1, eps_img, eps_f0;
};

// k-slot j (0..7) of k16 step s held by lane half hh  <->  input channel (the registers a lane owns after a layer: r = 4 g + t
// of channel tile i holds channel 32 i + 8 g + 4 hh + t; step s takes registers 8 (s & 1) .. + 7 of tile s >> 1)
__host__ __device__ inline int f2_chan(int s, int hh, int j) { return 32 * (s >> 1) + 8 * (2 * (s & 1) + (j >> 2)) + 4 * hh + (j & 3); }

// Weights and constants of the chain -> the kernel's operand order.  w_f0: [64][192] (columns 128..191 = the image half),
// w_f3: [ncls][64].  wpack: [m tile][k16 step][hi | lo][lane = (m, hh)][8 f16] (A operand: W[32 mt + m][f2_chan(s, hh, j)]).
__global__ void k_fine2_pack(const float* __restrict__ w_f0, const float* __restrict__ w_f3, int ncls, const float* b_img,
                             const float* g_img, const float* be_img, const float* b_f0, const float* g_f0, const float* be_f0,
                             const float* b_f3, _Float16* __restrict__ wpack, float* __restrict__ consts) {
  for (int e = threadIdx.x; e < 3 * 4 * 64 * 8; e += blockDim.x) {
    const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 3, mt = e >> 11;         // mt 0,1: fine_mlp[0]; 2: fine_mlp[3]
    const int m = lane & 31, hh = lane >> 5, k = f2_chan(s, hh, j);
    float w;
    if (mt < 2) w = w_f0[(32 * mt + m) * 192 + 128 + k];
    else w = m < ncls ? w_f3[m * 64 + k] : 0.f;
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

// bias + GroupNorm (groups of 4 consecutive channels = registers 4 g .. 4 g + 3) + ReLU: the arithmetic of k_groupnorm_rows /
// k_fine_mlp / ff_bias_gn_relu; constants from LDS
__device__ __forceinline__ void f2_bias_gn_relu(f32x16& v, const float* bias, const float* gamma, const float* beta, float eps, int c0) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 bi = *(const f32x4*)(bias + c0 + 8 * g), ga = *(const f32x4*)(gamma + c0 + 8 * g), be = *(const f32x4*)(beta + c0 + 8 * g);
    float x[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = v[4 * g + s] + bi[s];
    float mean = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) mean += x[s];
    mean /= 4.0f;
    float var = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) { float d = x[s] - mean; var += d * d; }
    var /= 4.0f;
    const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
    for (int s = 0; s < 4; ++s) v[4 * g + s] = fmaxf((x[s] - mean) * rstd * ga[s] + be[s], 0.f);
  }
}

__device__ __forceinline__ void f2_split8(const f32x16& y, int r0, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 a, b;
    split_h2(y[r0 + j], a, b);
    hi[j] = a; lo[j] = b;
  }
}

// acc[i][4 g + t] += w * row[32 i + 8 g + 4 h + t]: the lane's half of one 64-channel tap row (rp already points at channel 4 h)
__device__ __forceinline__ void f2_tap(const float* __restrict__ rp, float w, f32x16 (&acc)[2]) {
  f32x4 v[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) v[i][g] = *(const f32x4*)(rp + 32 * i + 8 * g);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[i][4 * g + t] = acc[i][4 * g + t] + v[i][g][t] * w;
}

template <int OCC>
__global__ __launch_bounds__(256, OCC) void k_fine2_h2(Fine2K p) {
  __shared__ __attribute__((aligned(16))) char Wl[F2_WBYTES];
  __shared__ __attribute__((aligned(16))) float Cn[F2_NCONST];
  __shared__ __attribute__((aligned(16))) float Ot[4][32 * F2_OT_PITCH];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 31, h = lane >> 5;
  for (int i = tid; i < F2_WBYTES / 16; i += 256) ((f32x4*)Wl)[i] = ((const f32x4*)p.wpack)[i];
  for (int i = tid; i < F2_NCONST; i += 256) Cn[i] = p.consts[i];
  __syncthreads();
  int n = p.n;
  if (p.n_dev) n = min(n, *p.n_dev);
  const int ntiles = (n + 15) >> 4;                    // 16 coarse voxels = 128 fine points per workgroup tile
  const long long nf = (long long)n * 8;
  const int X = p.X, Y = p.Y, Z = p.Z, Hf = p.Hf, Wf = p.Wf, ncls = p.ncls;
  const float* __restrict__ prm = p.prm;
  float* ot = Ot[wave];
  float gmax = 0.f;                                    // range guard of the two 16-bit operand conversions

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int ci0 = tile * 16 + wave * 4;
    if (ci0 >= n) continue;                            // wave-uniform; no workgroup barrier inside the loop
    const int nc = min(4, n - ci0);
    const int kk = li >> 3, oo = li & 7;
    const bool valid = kk < nc;
    const int ci = ci0 + (valid ? kk : 0);
    int l = p.lin[ci];
    const int cz = l % Z; l /= Z;
    const int cy = l % Y; const int cx = l / Y;         // B == 1
    const int oa = oo >> 2, ob = (oo >> 1) & 1, oc = oo & 1;
    const int fxi = cx * 2 + oa, fyi = cy * 2 + ob, fzi = cz * 2 + oc;
    if (valid && h == 0) {
      const long long f = (long long)oo * n + ci;
      p.fine_xyz[f] = fxi; p.fine_xyz[nf + f] = fyi; p.fine_xyz[2 * nf + f] = fzi;
    }

    // ---- image samples: sum over the cameras that see the point of the bilinear sample of P (k_fine_sample_img_grp's arithmetic)
    f32x16 ai[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) ai[i][r] = 0.f;
    {
      const float p0 = (float)fxi * prm[9] + prm[12];
      const float p1 = (float)fyi * prm[10] + prm[13];
      const float p2 = (float)fzi * prm[11] + prm[14];
      const float bx = prm[0] * p0 + prm[1] * p1 + prm[2] * p2;
      const float by = prm[3] * p0 + prm[4] * p1 + prm[5] * p2;
      const float bz = prm[6] * p0 + prm[7] * p1 + prm[8] * p2;
      const float wimg1 = prm[15], himg1 = prm[16];
#pragma unroll 1
      for (int cam = 0; cam < p.ncam; ++cam) {
        const float* q = prm + F2_HDR + cam * F2_CAM_STRIDE;
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
        const bool m = valid && d > 1e-5f && u2 > -1.f && u2 < 1.f && v2 > -1.f && v2 < 1.f;
        if (__ballot(m) == 0ull) continue;               // nobody in this wave is seen by the camera
        if (m) {
          const float px = (u2 + 1.f) / 2.f * (float)(Wf - 1), py = (v2 + 1.f) / 2.f * (float)(Hf - 1);
          const float flx = floorf(px), fly = floorf(py);
          const int x0 = (int)flx, y0 = (int)fly;
          const float ax = px - flx, ay = py - fly;
          const float* base = p.P + (size_t)cam * Hf * Wf * 64 + 4 * h;
#pragma unroll
          for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int xx = 0; xx < 2; ++xx) {
              const int x = x0 + xx, y = y0 + yy;
              const bool in = (unsigned)x < (unsigned)Wf && (unsigned)y < (unsigned)Hf;
              const float w = in ? (xx ? ax : 1.f - ax) * (yy ? ay : 1.f - ay) : 0.f;
              f2_tap(base + (size_t)(in ? y * Wf + x : 0) * 64, w, ai);
            }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) f2_bias_gn_relu(ai[i], Cn, Cn + 64, Cn + 128, p.eps_img, 32 * i + 4 * h);

    // ---- voxel samples: trilinear sample of Q at the fine voxel centre (k_fine_sample_voxel_r2's arithmetic) = initial accumulator
    f32x16 hh[2], xx2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { hh[i][r] = 0.f; xx2[i][r] = 0.f; }
    if (valid) {
      int i0[3]; float tt[3];
      const int q3[3] = {fxi, fyi, fzi}; const float f1[3] = {p.fx1, p.fy1, p.fz1}; const int S[3] = {X, Y, Z};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float g = ((float)q3[a] / f1[a] - 0.5f) * 2.f;
        const float pp = ((g + 1.f) * (float)S[a] - 1.f) / 2.f;
        const float fl = floorf(pp);
        i0[a] = (int)fl; tt[a] = pp - fl;
      }
      const float* vol = p.Q + 4 * h;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int x = i0[0] + dx;
        if ((unsigned)x >= (unsigned)X) continue;
        const float wx = dx ? tt[0] : 1.f - tt[0];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const int y = i0[1] + dy;
          if ((unsigned)y >= (unsigned)Y) continue;
          const float wxy = wx * (dy ? tt[1] : 1.f - tt[1]);
#pragma unroll
          for (int dz = 0; dz < 2; ++dz) {
            const int z = i0[2] + dz;
            if ((unsigned)z >= (unsigned)Z) continue;
            f2_tap(vol + (((size_t)x * Y + y) * Z + z) * 64, wxy * (dz ? tt[2] : 1.f - tt[2]), hh);
          }
        }
      }
    }

    // ---- fine_mlp[0]: h = Q sample + W_f0[:, 128:] . y1  (y1 = ai), three MFMAs per k16 step
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) gmax = fmaxf(gmax, ai[i][r]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f16x8 bhi, blo;
      f2_split8(ai[s >> 1], 8 * (s & 1), bhi, blo);
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
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) hh[i][r] = hh[i][r] + xx2[i][r] * (1.f / H2_LO_SCALE);
      f2_bias_gn_relu(hh[i], Cn + 192, Cn + 256, Cn + 320, p.eps_f0, 32 * i + 4 * h);
#pragma unroll
      for (int r = 0; r < 16; ++r) gmax = fmaxf(gmax, hh[i][r]);
    }

    // ---- fine_mlp[3]: Linear(64 -> ncls <= 32); weight rows >= ncls are zero in the pack
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
    // logits of point li, classes 8 j + 4 h + t -> the wave's staging tile, then out rows (o n + ci0 .. + nc - 1) as contiguous runs
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bi = *(const f32x4*)(Cn + 384 + 8 * j + 4 * h);
      f32x4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = o[4 * j + t] + ox[4 * j + t] * (1.f / H2_LO_SCALE) + bi[t];
      *(f32x4*)(ot + li * F2_OT_PITCH + 8 * j + 4 * h) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is private to the wave
    {
      const int seg = nc * ncls, total = 8 * seg;                // <= 128, <= 1024
      const unsigned inv_seg = ((1u << 20) + seg - 1) / seg, inv_cls = ((1u << 20) + ncls - 1) / ncls;
#pragma unroll 1
      for (int idx = lane; idx < total; idx += 64) {
        const int oq = (int)(((unsigned)idx * inv_seg) >> 20), rem = idx - oq * seg;
        const int kq = (int)(((unsigned)rem * inv_cls) >> 20), c = rem - kq * ncls;
        p.out[((long long)oq * n + ci0) * ncls + rem] = ot[(kq * 8 + oq) * F2_OT_PITCH + c];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // ... read before the next tile overwrites it
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

// Q: [X*Y*Z, 64], P: [ncam*Hf*Wf, 64] (the two Linear layers applied BEFORE the resamplings, as for coocc_fine_fused); lin: the n
// (with n_dev: at most n_cap) foreground coarse voxels; final_size == 2 * (X, Y, Z); wpack / consts from coocc_fine2_pack.
// Outputs as coocc_fine_fused: fine_xyz [3][8 n], logits [8 n, ncls] (row o * n + i).
extern "C" int coocc_fine2_h2(const float* Q, int X, int Y, int Z, const float* P, int ncam, int Hf, int Wf, const float* params,
                              const int32_t* coarse_lin, int n_cap, const int32_t* n_dev, const int* final_size_host, const void* wpack,
                              const float* consts, float eps_img, float eps_f0, int ncls, int64_t* fine_xyz, float* out, void* stream) {
  COOCC_CHECK_ARG(Q && P && params && coarse_lin && final_size_host && wpack && consts && fine_xyz && out, "fine2_h2: null pointer");
  COOCC_CHECK_ARG(n_cap >= 0 && ncls >= 1 && ncls <= 32 && ncam >= 1 && X > 0 && Y > 0 && Z > 0 && Hf > 0 && Wf > 0, "fine2_h2: ncls <= 32");
  COOCC_CHECK_ARG(final_size_host[0] == 2 * X && final_size_host[1] == 2 * Y && final_size_host[2] == 2 * Z,
                  "fine2_h2: final_occ_size must be 2 x the coarse grid");
  COOCC_CHECK_ARG(f2_aligned16(Q) && f2_aligned16(P) && f2_aligned16(wpack) && f2_aligned16(consts), "fine2_h2: arrays must be 16-byte aligned");
  if (n_cap == 0) return COOCC_OK;
  Fine2K p;
  memset(&p, 0, sizeof(p));
  p.Q = Q; p.P = P; p.prm = params; p.lin = coarse_lin; p.n_dev = n_dev; p.fine_xyz = fine_xyz; p.out = out;
  p.wpack = (const char*)wpack; p.consts = consts;
  if (coocc_h2_flag_ptr(&p.h2_flag) != COOCC_OK) return COOCC_EHIP;
  p.n = n_cap; p.X = X; p.Y = Y; p.Z = Z; p.ncam = ncam; p.Hf = Hf; p.Wf = Wf; p.ncls = ncls;
  p.fx1 = (float)(final_size_host[0] - 1); p.fy1 = (float)(final_size_host[1] - 1); p.fz1 = (float)(final_size_host[2] - 1);
  p.eps_img = eps_img; p.eps_f0 = eps_f0;
  const long long tiles = ((long long)n_cap + 15) / 16;
  // persistent: OCC workgroups per CU walk the tiles.  COOCC_FINE2_OCC = 2 (256 registers: every tap row of a point in flight at
  // once) | 3 (168 registers, three waves per SIMD)
  static const int occ = getenv("COOCC_FINE2_OCC") ? atoi(getenv("COOCC_FINE2_OCC")) : 2;
  const long long cap = 256ll * (occ == 3 ? 3 : 2);
  const int grid = (int)(tiles < cap ? tiles : cap);
  if (occ == 3) hipLaunchKernelGGL(k_fine2_h2<3>, dim3(grid), dim3(256), 0, as_stream(stream), p);
  else hipLaunchKernelGGL(k_fine2_h2<2>, dim3(grid), dim3(256), 0, as_stream(stream), p);
  COOCC_LAUNCH_CHECK("k_fine2_h2");
  return COOCC_OK;
}
