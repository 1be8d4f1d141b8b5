// OccHead fine branch in ONE kernel (round 3): the two resamplings and the MLP chain of occ_head.py:205-233 for 64 fine points
// per wave, nothing but the logits and the fine coordinates written to HBM.
//
// Round 2 ran the branch as k_fine_sample_voxel_r2 -> k_fine_sample_img_grp -> k_fine_mlp<pre>: every fine point's two 64-channel
// samples (512 B) were written and read back -- 616 MB of the branch's 750 MB at configs[1] (600 k points), 5 GB at the
// OpenOccupancy cascade (9.8 M points), all three kernels at ~2 TB/s.  Here a wave owns the R^3 children of 64 / R^3 coarse voxels
// (ratio 2: 8 coarse voxels x 8 children; ratio 4: one coarse voxel x 64 children), resamples into a private 64 x 64 LDS tile
// (image samples first, voxel samples after the first GroupNorm has consumed them) and runs k_fine_mlp<pre>'s register chain from
// that tile.  The resampling code is the grouped kernels' (same expressions, same accumulation order: csrc/fine.hip) and the MLP is
// k_fine_mlp<true>'s (csrc/fine_mlp.hip), so the logits are bit-identical to the three-kernel path
// (tests/test_gpu_modules.py::test_fused_fine_branch_equals_three_kernel_path).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FINE_CAM_STRIDE 27
#define FINE_HDR 17
#define FF_PITCH 68          // floats per LDS row (64 channels + 4): 16-byte rows, lanes li = 0..31 spread over the banks

struct FineFused {
  const float* Q;            // [X*Y*Z, 64]  W_f0[:, :128] . voxel features (Linear applied before the trilinear resampling)
  const float* P;            // [ncam*Hf*Wf, 64]  W_img . image features (Linear applied before the bilinear resampling)
  const float* prm;          // coocc_projection_params
  const int32_t* lin;        // foreground coarse voxels (linear ids)
  const int32_t* n_dev;      // optional device-side count
  int64_t* fine_xyz;         // [3][n * R^3]
  float* out;                // [n * R^3, ncls], row f = o * n + i
  const float* b_img; const float* g_img; const float* be_img;
  const float* w_f0; const float* b_f0; const float* g_f0; const float* be_f0;
  const float* w_f3; const float* b_f3;
  int n, X, Y, Z, ncam, Hf, Wf, ncls;
  float fx1, fy1, fz1, eps_img, eps_f0;
};

__device__ __forceinline__ f32x4 ff_bl4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

// bias + GroupNorm (groups of 4 consecutive channels = registers 4g..4g+3) + ReLU, arithmetic of k_groupnorm_rows / k_fine_mlp
__device__ __forceinline__ void ff_bias_gn_relu(f32x16& v, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps, int c0) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 bi = *(const f32x4*)(bias + c0 + 8 * g), ga = *(const f32x4*)(gamma + c0 + 8 * g),
                be = *(const f32x4*)(beta + c0 + 8 * g);
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
  __builtin_amdgcn_sched_barrier(0);
}

// trilinear samples of Q for the R^3 children of coarse voxel (cx, cy, cz) -> tile rows row0 + o, channels 2 * lane32, + 1
// (one HALF-wave per coarse voxel: 32 lanes x 2 channels; the code of k_fine_sample_voxel_r2 / _rn with C = 64)
template <int R>
__device__ __forceinline__ void ff_sample_voxel(const FineFused& p, int cx, int cy, int cz, int lane32, float* __restrict__ T, int row0,
                                                int a_begin, int a_end) {
  const int X = p.X, Y = p.Y, Z = p.Z;
  constexpr int C = 64;
  int i0[3][R]; float t[3][R];
  const int cc[3] = {cx, cy, cz}; const float f1[3] = {p.fx1, p.fy1, p.fz1}; const int S[3] = {X, Y, Z};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax)
#pragma unroll
    for (int a = 0; a < R; ++a) {
      const int q = cc[ax] * R + a;
      const float g = ((float)q / f1[ax] - 0.5f) * 2.f;
      const float pp = ((g + 1.f) * (float)S[ax] - 1.f) / 2.f;
      const float fl = floorf(pp);
      i0[ax][a] = (int)fl; t[ax][a] = pp - fl;
    }
  auto tapw = [](int b0, float tt, int x) { return (x == b0 ? 1.f - tt : 0.f) + (x == b0 + 1 ? tt : 0.f); };
  const float* vol = p.Q;
  const int c = lane32 * 2;
  if constexpr (R == 2) {
    const int wx0 = min(i0[0][0], i0[0][1]), wy0 = min(i0[1][0], i0[1][1]), wz0 = min(i0[2][0], i0[2][1]);
    f32x2 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = f32x2{0.f, 0.f};
#pragma unroll 1
    for (int kx = 0; kx < 3; ++kx) {
      const int x = wx0 + kx;
      if ((unsigned)x >= (unsigned)X) continue;
      const float ax0 = tapw(i0[0][0], t[0][0], x), ax1 = tapw(i0[0][1], t[0][1], x);
#pragma unroll 1
      for (int ky = 0; ky < 3; ++ky) {
        const int y = wy0 + ky;
        if ((unsigned)y >= (unsigned)Y) continue;
        const float by0 = tapw(i0[1][0], t[1][0], y), by1 = tapw(i0[1][1], t[1][1], y);
        const float w00 = ax0 * by0, w01 = ax0 * by1, w10 = ax1 * by0, w11 = ax1 * by1;
        const float* rowp = vol + (((size_t)x * Y + y) * Z) * C + c;
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          const int z = wz0 + kz;
          if ((unsigned)z >= (unsigned)Z) continue;
          const float cz0 = tapw(i0[2][0], t[2][0], z), cz1 = tapw(i0[2][1], t[2][1], z);
          const f32x2 v = *(const f32x2*)(rowp + (size_t)z * C);
          acc[0] = acc[0] + v * (w00 * cz0); acc[1] = acc[1] + v * (w00 * cz1);
          acc[2] = acc[2] + v * (w01 * cz0); acc[3] = acc[3] + v * (w01 * cz1);
          acc[4] = acc[4] + v * (w10 * cz0); acc[5] = acc[5] + v * (w10 * cz1);
          acc[6] = acc[6] + v * (w11 * cz0); acc[7] = acc[7] + v * (w11 * cz1);
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) *(f32x2*)(T + (row0 + o) * FF_PITCH + c) = acc[o];
  } else {
    int wy0 = i0[1][0], wz0 = i0[2][0];
#pragma unroll
    for (int a = 1; a < R; ++a) { wy0 = min(wy0, i0[1][a]); wz0 = min(wz0, i0[2][a]); }
#pragma unroll 1
    for (int a = a_begin; a < a_end; ++a) {        // the x-children this half-wave takes
      int bx = i0[0][0]; float txa = t[0][0];
#pragma unroll
      for (int k = 1; k < R; ++k) if (a == k) { bx = i0[0][k]; txa = t[0][k]; }
      f32x2 acc[R * R];
#pragma unroll
      for (int o = 0; o < R * R; ++o) acc[o] = f32x2{0.f, 0.f};
#pragma unroll 1
      for (int kx = 0; kx < 2; ++kx) {
        const int x = bx + kx;
        if ((unsigned)x >= (unsigned)X) continue;
        const float wxa = kx ? txa : 1.f - txa;
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
          const int y = wy0 + ky;
          if ((unsigned)y >= (unsigned)Y) continue;
          float wxy[R];
#pragma unroll
          for (int b = 0; b < R; ++b) wxy[b] = wxa * tapw(i0[1][b], t[1][b], y);
          const float* rowp = vol + (((size_t)x * Y + y) * Z) * C + c;
#pragma unroll
          for (int kz = 0; kz < 3; ++kz) {
            const int z = wz0 + kz;
            if ((unsigned)z >= (unsigned)Z) continue;
            const f32x2 v = *(const f32x2*)(rowp + (size_t)z * C);
#pragma unroll
            for (int d = 0; d < R; ++d) {
              const float czd = tapw(i0[2][d], t[2][d], z);
#pragma unroll
              for (int b = 0; b < R; ++b) acc[b * R + d] = acc[b * R + d] + v * (wxy[b] * czd);
            }
          }
        }
      }
#pragma unroll
      for (int o = 0; o < R * R; ++o) *(f32x2*)(T + (row0 + a * R * R + o) * FF_PITCH + c) = acc[o];
    }
  }
}

// bilinear samples of P (all cameras that see the point, summed) for 8 children (g*8 .. g*8+7) of the coarse voxel whose child 0
// has fine coordinates (x00, y00, z00): whole wave, lanes = the 64 channels; tile rows row0 + 0..7
// (the code of k_fine_sample_img_grp with Ci = 64)
template <int R>
__device__ __forceinline__ void ff_sample_img8(const FineFused& p, long long x00, long long y00, long long z00, int g, int lane,
                                               float* __restrict__ T, int row0) {
  const int ncam = p.ncam, Hf = p.Hf, Wf = p.Wf;
  constexpr int Ci = 64;
  const float* prm = p.prm;
  int m = 0, x0 = 0, y0 = 0;
  float ax = 0.f, ay = 0.f;
  if (lane < 8 * ncam) {
    const int o8 = lane / ncam, cam = lane - o8 * ncam;
    const int o = g * 8 + o8;
    const long long fx = x00 + o / (R * R), fy = y00 + (o / R) % R, fz = z00 + o % R;
    float p0 = (float)fx * prm[9] + prm[12];
    float p1 = (float)fy * prm[10] + prm[13];
    float p2 = (float)fz * prm[11] + prm[14];
    float bx = prm[0] * p0 + prm[1] * p1 + prm[2] * p2;
    float by = prm[3] * p0 + prm[4] * p1 + prm[5] * p2;
    float bz = prm[6] * p0 + prm[7] * p1 + prm[8] * p2;
    const float wimg1 = prm[15], himg1 = prm[16];
    const float* q = prm + FINE_HDR + cam * FINE_CAM_STRIDE;
    float tx = bx - q[9], ty = by - q[10], tz = bz - q[11];
    float cx = q[0] * tx + q[1] * ty + q[2] * tz;
    float cy = q[3] * tx + q[4] * ty + q[5] * tz;
    float cz = q[6] * tx + q[7] * ty + q[8] * tz;
    float ix = q[12] * cx + q[13] * cy + q[14] * cz;
    float iy = q[15] * cx + q[16] * cy + q[17] * cz;
    float d = q[18] * cx + q[19] * cy + q[20] * cz;
    float u = ix / (d + 1e-5f), v = iy / (d + 1e-5f);
    float u2 = q[21] * u + q[22] * v + q[25];
    float v2 = q[23] * u + q[24] * v + q[26];
    u2 = (u2 / wimg1 - 0.5f) * 2.f;
    v2 = (v2 / himg1 - 0.5f) * 2.f;
    m = (d > 1e-5f && u2 > -1.f && u2 < 1.f && v2 > -1.f && v2 < 1.f) ? 1 : 0;
    float px = (u2 + 1.f) / 2.f * (float)(Wf - 1), py = (v2 + 1.f) / 2.f * (float)(Hf - 1);
    float flx = floorf(px), fly = floorf(py);
    x0 = (int)flx; y0 = (int)fly;
    ax = px - flx; ay = py - fly;
  }
  int toff[4]; float tw[4];
  const int cbase = lane < 8 * ncam ? (lane % ncam) * Hf * Wf : 0;
#pragma unroll
  for (int yy = 0; yy < 2; ++yy)
#pragma unroll
    for (int xx = 0; xx < 2; ++xx) {
      const int x = x0 + xx, y = y0 + yy;
      const bool in = (unsigned)x < (unsigned)Wf && (unsigned)y < (unsigned)Hf;
      toff[yy * 2 + xx] = cbase + (in ? y * Wf + x : 0);
      tw[yy * 2 + xx] = (in && m) ? (xx ? ax : 1.f - ax) * (yy ? ay : 1.f - ay) : 0.f;
    }
  const unsigned long long seen = __ballot(m != 0);
  unsigned sub[8];
  int rounds = 0;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    sub[o] = (unsigned)(seen >> (o * ncam)) & ((1u << ncam) - 1u);
    rounds = max(rounds, __popc(sub[o]));
  }
  float acc[8];
  unsigned left[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) { acc[o] = 0.f; left[o] = sub[o]; }
  const float* base = p.P + lane;
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
    float v[8][4], w[8][4];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const bool has = left[o] != 0u;                                  // wave-uniform
      const int k = has ? o * ncam + (__ffs((int)left[o]) - 1) : 0;
      left[o] &= left[o] - 1u;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int off = __builtin_amdgcn_readlane(toff[tt], k);
        const float wt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tw[tt]), k));
        w[o][tt] = has ? wt : 0.f;
        v[o][tt] = base[(size_t)off * Ci];
      }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[o] = acc[o] + v[o][tt] * w[o][tt];
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) T[(row0 + o) * FF_PITCH + lane] = acc[o];
}

template <int R>
// (built with -fno-slp-vectorize, co_occ_amd/build.py FILE_FLAGS: the projection block below must not become packed fp32 with op_sel swaps)
__global__ __launch_bounds__(256, 2) void k_fine_fused(FineFused p) {
  constexpr int R3 = R * R * R, CPW = 64 / R3;          // coarse voxels per wave: 8 (ratio 2) | 1 (ratio 4)
  __shared__ __attribute__((aligned(16))) float tiles[4][64 * FF_PITCH];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int li = lane & 31, h = lane >> 5;
  int n = p.n;
  if (p.n_dev) n = min(n, *p.n_dev);
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long ci0 = gw * CPW;                         // first coarse voxel of this wave
  if (ci0 >= n) return;
  const int ncoarse = (int)min((long long)CPW, n - ci0);
  const long long nf = (long long)n * R3;
  float* T = tiles[wave];
  const int Y = p.Y, Z = p.Z;

  // ---- fine coordinates + image samples, one coarse voxel after the other (wave-uniform loop)
#pragma unroll 1
  for (int k = 0; k < ncoarse; ++k) {
    const long long i = ci0 + k;
    int l = p.lin[i];
    const int cz = l % Z; l /= Z;
    const int cy = l % Y; const int cx = l / Y;   // B == 1
    for (int o = lane; o < R3; o += 64) {
      const int oa = o / (R * R), ob = (o / R) % R, oc = o % R;
      const long long f = (long long)o * n + i;
      p.fine_xyz[f] = cx * R + oa; p.fine_xyz[nf + f] = cy * R + ob; p.fine_xyz[2 * nf + f] = cz * R + oc;
    }
#pragma unroll 1
    for (int g = 0; g < R3 / 8; ++g)
      ff_sample_img8<R>(p, (long long)cx * R, (long long)cy * R, (long long)cz * R, g, lane, T, k * R3 + g * 8);
  }
  // the tile is private to the wave: its own LDS writes only have to land before its own reads
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // point of lane (li, h), MFMA tile t: pp = 32 t + li -> coarse voxel pp / R3, child pp % R3
  auto load_tile = [&](f32x16 (&dst)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *(const f32x4*)(T + (32 * t + li) * FF_PITCH + 32 * j + 8 * g + 4 * h);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) dst[j][t][4 * g + s_] = v[s_];
        }
  };
  f32x16 y[2][2], acc[2][2];
  load_tile(acc);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ff_bias_gn_relu(acc[i][t], p.b_img, p.g_img, p.be_img, p.eps_img, 32 * i + 4 * h);
      y[i][t] = acc[i][t];
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- voxel samples into the same tile: a half-wave per coarse voxel
  {
    const int half = lane >> 5, lane32 = lane & 31;
#pragma unroll 1
    for (int k = 0; k < CPW; k += (CPW > 1 ? 2 : 1)) {
      // ratio 2: the two half-waves take two coarse voxels; ratio 4 (one coarse voxel per wave): two x-children each
      const int kk = CPW > 1 ? k + half : 0;
      if (kk < ncoarse) {
        int l = p.lin[ci0 + kk];
        const int cz = l % Z; l /= Z;
        const int cy = l % Y; const int cx = l / Y;
        ff_sample_voxel<R>(p, cx, cy, cz, lane32, T, kk * R3, CPW > 1 ? 0 : half * (R / 2), CPW > 1 ? R : (half + 1) * (R / 2));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  load_tile(acc);          // the accumulators of fine_mlp[0] start from the voxel term

  const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_f0, 0, 64u * 192u * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w_f3, 0, (unsigned)p.ncls * 64u * 4u, 0x00020000);
  const unsigned w0a = (unsigned)(li * 192 + 4 * h) * 4u, w0b = (unsigned)((32 + li) * 192 + 4 * h) * 4u;
  {
    f32x4 an0 = ff_bl4(rw0, w0a + 128u * 4u), an1 = ff_bl4(rw0, w0b + 128u * 4u);
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // k-group q of y1: channel tile j = q >> 2, run g = q & 3
      const int j = q >> 2, g = q & 3;
      const f32x4 a0 = an0, a1 = an1;
      if (q < 7) {
        const unsigned d = (unsigned)(128 + 8 * (q + 1)) * 4u;
        an0 = ff_bl4(rw0, w0a + d); an1 = ff_bl4(rw0, w0b + d);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], y[j][t][4 * g + s], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], y[j][t][4 * g + s], acc[1][t], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ff_bias_gn_relu(acc[i][t], p.b_f0, p.g_f0, p.be_f0, p.eps_f0, 32 * i + 4 * h);
      y[i][t] = acc[i][t];
    }

  // ---- fine_mlp[3]: Linear(64 -> ncls <= 32); weight rows >= ncls read 0
  f32x16 o[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  const unsigned w3 = (unsigned)(li * 64 + 4 * h) * 4u;
  {
    f32x4 an = ff_bl4(rw3, w3);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = q >> 2, g = q & 3;
      const f32x4 a = an;
      if (q < 7) an = ff_bl4(rw3, w3 + (unsigned)(8 * (q + 1)) * 4u);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], y[j][t][4 * g + s], o[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int pp = 32 * t + li;
    const int kk = pp / R3, oo = pp % R3;
    if (kk >= ncoarse) continue;
    float* dst = p.out + ((long long)oo * n + (ci0 + kk)) * p.ncls;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < p.ncls) dst[c] = o[t][r] + p.b_f3[c];
    }
  }
}

// Q: [X*Y*Z, 64] and P: [ncam*Hf*Wf, 64] rows (the two Linear layers applied BEFORE the resamplings: coocc_fine_mlp_pre's
// contract); lin: the n (or, with n_dev, at most n_cap) foreground coarse voxels; final_size == ratio * (X, Y, Z);
// outputs as coocc_fine_sample_voxel (fine_xyz [3][n ratio^3]) and coocc_fine_mlp_pre (logits [n ratio^3, ncls], row o * n + i).
extern "C" int coocc_fine_fused(const float* Q, int X, int Y, int Z, const float* P, int ncam, int Hf, int Wf, const float* params,
                                const int32_t* coarse_lin, int n_cap, const int32_t* n_dev, int ratio, const int* final_size_host,
                                const float* b_img, const float* gn_img_w, const float* gn_img_b, float eps_img, const float* w_f0,
                                const float* b_f0, const float* gn_f0_w, const float* gn_f0_b, float eps_f0, const float* w_f3,
                                const float* b_f3, int ncls, int64_t* fine_xyz, float* out, void* stream) {
  COOCC_CHECK_ARG(Q && P && params && coarse_lin && final_size_host && fine_xyz && out && b_img && gn_img_w && gn_img_b && w_f0 && b_f0 &&
                      gn_f0_w && gn_f0_b && w_f3 && b_f3, "fine_fused: null pointer");
  COOCC_CHECK_ARG((ratio == 2 || ratio == 4) && n_cap >= 0 && ncls >= 1 && ncls <= 32 && ncam >= 1 && ncam <= 8 && X > 0 && Y > 0 && Z > 0 &&
                      Hf > 0 && Wf > 0, "fine_fused: ratio 2 | 4, ncls <= 32, <= 8 cameras");
  COOCC_CHECK_ARG(final_size_host[0] == ratio * X && final_size_host[1] == ratio * Y && final_size_host[2] == ratio * Z,
                  "fine_fused: final_occ_size must be ratio x the coarse grid (the 3-wide window of the grouped resampling)");
  COOCC_CHECK_ARG(((uintptr_t)Q | (uintptr_t)P | (uintptr_t)w_f0 | (uintptr_t)w_f3 | (uintptr_t)b_img | (uintptr_t)b_f0 | (uintptr_t)gn_img_w |
                   (uintptr_t)gn_img_b | (uintptr_t)gn_f0_w | (uintptr_t)gn_f0_b) % 16 == 0, "fine_fused: arrays must be 16-byte aligned");
  if (n_cap == 0) return COOCC_OK;
  FineFused p;
  p.Q = Q; p.P = P; p.prm = params; p.lin = coarse_lin; p.n_dev = n_dev; p.fine_xyz = fine_xyz; p.out = out;
  p.b_img = b_img; p.g_img = gn_img_w; p.be_img = gn_img_b;
  p.w_f0 = w_f0; p.b_f0 = b_f0; p.g_f0 = gn_f0_w; p.be_f0 = gn_f0_b; p.w_f3 = w_f3; p.b_f3 = b_f3;
  p.n = n_cap; p.X = X; p.Y = Y; p.Z = Z; p.ncam = ncam; p.Hf = Hf; p.Wf = Wf; p.ncls = ncls;
  p.fx1 = (float)(final_size_host[0] - 1); p.fy1 = (float)(final_size_host[1] - 1); p.fz1 = (float)(final_size_host[2] - 1);
  p.eps_img = eps_img; p.eps_f0 = eps_f0;
  const int cpw = ratio == 2 ? 8 : 1;
  const long long waves = ((long long)n_cap + cpw - 1) / cpw;
  if (ratio == 2) hipLaunchKernelGGL(k_fine_fused<2>, dim3(cdiv(waves, 4)), dim3(256), 0, as_stream(stream), p);
  else hipLaunchKernelGGL(k_fine_fused<4>, dim3(cdiv(waves, 4)), dim3(256), 0, as_stream(stream), p);
  COOCC_LAUNCH_CHECK("k_fine_fused");
  return COOCC_OK;
}
