/*
 * coocc_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity checker).
 *
 * Plain-C restatement of the three native (CUDA) ops on Co-Occ's fused-voxel hot
 * path plus the canonical forms of the index-search steps around them.  Nothing
 * in the product package imports, links or executes this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Each function cites the reference file:line it follows (paths relative to the
 * reference checkout, M/ = mmdetection3d/mmdet3d/, P/ = projects/mmdet3d_plugin/).
 *
 * Pinned by (tests/test_oracle.py):
 *   - M/../tests/test_models/test_common_modules/test_pointnet_ops.py:10-24  (FPS KAT)
 *   - same file :27-74                                                       (ball query KATs)
 *   - golden vectors produced by running the unmodified reference BiFuser_N /
 *     voxel_pooling with these functions injected as mmdet3d.ops (oracle/gen_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; FMAs are written explicitly)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* squared distance exactly as nvcc contracts
 *   (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)
 * (left-assoc adds, each add fused with the following product).  For the
 * integer voxel coordinates BiFuser_N feeds in, every form is exact. */
static inline float sqdist3(float x1, float y1, float z1, float x2, float y2, float z2) {
  float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

/* M/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:11-15 */
static int opt_n_threads(int work_size) {
  int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  return t;
}

/*
 * furthest_point_sampling_kernel<block_size>, furthest_point_sample_cuda.cu:25-141.
 * Emulates the block exactly: per-thread strided scan with strict '>' (:56-71),
 * then the shared-memory tree where the LEFT operand survives ties (:17-23,:76-136).
 * temp is initialised to 1e10 by the Python wrapper (furthest_point_sample.py:29).
 * xyz: [b,n,3] f32, idx out: [b,m] i32.  Returns 0.
 */
int oracle_fps(int b, int n, int m, const float* xyz, int32_t* idx) {
  if (m <= 0) return 0;
  int block = opt_n_threads(n);
  float* temp = (float*)malloc(sizeof(float) * (size_t)n);
  float* dists = (float*)malloc(sizeof(float) * (size_t)block);
  int* dists_i = (int*)malloc(sizeof(int) * (size_t)block);
  for (int bi = 0; bi < b; ++bi) {
    const float* ds = xyz + (size_t)bi * n * 3;
    int32_t* out = idx + (size_t)bi * m;
    for (int k = 0; k < n; ++k) temp[k] = 1e10f;
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < block; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += block) {
          float d = sqdist3(x1, y1, z1, ds[k * 3 + 0], ds[k * 3 + 1], ds[k * 3 + 2]);
          float d2 = d < temp[k] ? d : temp[k];
          temp[k] = d2;
          if (d2 > best) { besti = k; best = d2; }
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = block / 2; s >= 1; s >>= 1) {
        for (int tid = 0; tid < s; ++tid) {
          float v1 = dists[tid], v2 = dists[tid + s];
          int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(temp); free(dists); free(dists_i);
  return 0;
}

/*
 * ball_query_kernel, M/ops/ball_query/src/ball_query_cuda.cu:11-54.
 * idx must be zero-initialised by the caller (ball_query.py:35).
 * new_xyz: [b,m,3] centres, xyz: [b,n,3], idx: [b,m,nsample].
 */
int oracle_ball_query(int b, int n, int m, float min_radius, float max_radius, int nsample,
                      const float* new_xyz, const float* xyz, int32_t* idx) {
  float max_r2 = max_radius * max_radius, min_r2 = min_radius * min_radius;
  for (int bi = 0; bi < b; ++bi)
    for (int p = 0; p < m; ++p) {
      const float* c = new_xyz + ((size_t)bi * m + p) * 3;
      const float* pts = xyz + (size_t)bi * n * 3;
      int32_t* o = idx + ((size_t)bi * m + p) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        /* (new_x-x)^2 + (new_y-y)^2 + (new_z-z)^2, contracted like sqdist3 */
        float d2 = sqdist3(pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2], c[0], c[1], c[2]);
        if (d2 == 0 || (d2 >= min_r2 && d2 < max_r2)) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
  return 0;
}

/*
 * bev_pool_kernel, M/ops/bev_pool/src/bev_pool_cuda.cu:20-42 (interval sums over
 * rank-sorted rows).  x:[n,c], geom:[n,4] (x,y,z,b) i32, out:[b,d,h,w,c] zeroed here
 * (bev_pool.cpp:41).  Index arithmetic exactly as :33-35: geom[0] strides w*c,
 * geom[1] strides c, geom[2] strides h*w*c.
 */
int oracle_bev_pool_forward(int b, int d, int h, int w, int n, int c, int n_intervals,
                            const float* x, const int32_t* geom, const int32_t* starts,
                            const int32_t* lengths, float* out) {
  (void)n;
  memset(out, 0, sizeof(float) * (size_t)b * d * h * w * c);
  for (int it = 0; it < n_intervals; ++it) {
    int s = starts[it], len = lengths[it];
    const int32_t* g = geom + (size_t)s * 4;
    float* o = out + (size_t)g[3] * d * h * w * c + (size_t)g[2] * h * w * c +
               (size_t)g[0] * w * c + (size_t)g[1] * c;
    for (int cc = 0; cc < c; ++cc) {
      float psum = 0;
      for (int i = 0; i < len; ++i) psum += x[((size_t)s + i) * c + cc];
      o[cc] = psum;
    }
  }
  return 0;
}

/*
 * Brute-force K smallest distances of each representative query to all keys,
 * P/coocc/fuser/bifuser_n.py:101-103 (norm + topk(largest=False)), with the
 * canonical tie order (d^2, key index) ascending (SURVEY.md section 7 item 1;
 * torch.topk leaves tie order unspecified).  val = sqrtf(d^2) like torch.norm.
 * q:[nq,3], key:[nk,3] f32; val:[nq,K] f32; idx:[nq,K] i64.
 */
int oracle_knn_topk(int nq, int nk, int K, const float* q, const float* key, float* val,
                    int64_t* idx) {
  if (K > nk) return -1;
  float* bd = (float*)malloc(sizeof(float) * (size_t)K);
  int64_t* bi = (int64_t*)malloc(sizeof(int64_t) * (size_t)K);
  for (int r = 0; r < nq; ++r) {
    int cnt = 0;
    for (int k = 0; k < nk; ++k) {
      float d2 = sqdist3(q[r * 3], q[r * 3 + 1], q[r * 3 + 2], key[k * 3], key[k * 3 + 1],
                         key[k * 3 + 2]);
      /* insert if (d2,k) < worst kept; keys arrive in ascending k so ties keep earlier k */
      if (cnt < K || d2 < bd[cnt - 1]) {
        int p = cnt < K ? cnt : K - 1;
        while (p > 0 && bd[p - 1] > d2) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
        bd[p] = d2; bi[p] = k;
        if (cnt < K) ++cnt;
      }
    }
    for (int j = 0; j < K; ++j) { val[(size_t)r * K + j] = sqrtf(bd[j]); idx[(size_t)r * K + j] = bi[j]; }
  }
  free(bd); free(bi);
  return 0;
}

/*
 * Assignment step of fps_NN_fast, bifuser_n.py:104-125 (K>1) / :73-85 (K==1):
 * for k in 0..K-1, every query listed in a valid centre's ball row gets that
 * centre's k-th NN key; later (higher centre ordinal) writes win -- the sequential
 * single-thread semantics of index_put_ with duplicate indices.
 * val,nn:[nc,K]; group:[nc,ns] i32; out:[K,nq] i64 (pre-filled with -1 here).
 */
int oracle_knn_assign(int nc, int K, int ns, int nq, float dist_thresh, const float* val,
                      const int64_t* nn, const int32_t* group, int64_t* out) {
  for (size_t i = 0; i < (size_t)K * nq; ++i) out[i] = -1;
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < nc; ++c) {
      if (!(val[(size_t)c * K + k] < dist_thresh)) continue;
      for (int s = 0; s < ns; ++s) out[(size_t)k * nq + group[(size_t)c * ns + s]] = nn[(size_t)c * K + k];
    }
  return 0;
}
