// One C-ABI call for the whole index-search stage of BiFuser_N (K1-K5, bifuser_n.py:129-162): the ~45 launches, the fork /
// join of the two search directions on two streams and the ONE device->host read of the stage (the two non-empty voxel
// counts) are issued from C++.  A Python caller holds no GIL while this runs (ctypes releases it for the call), so the
// helper threads that prefetch the search of the next samples no longer compete with the thread that issues the dense
// stage -- and the host cost of the stage drops from ~3 ms of Python to ~0.15 ms.
//
// Same kernels, same order and same results as co_occ_amd.fuser.BiFuser_N.search (grid forms of top-K / ball query,
// bucket-pruned FPS); B == 1, both voxel lists longer than fps_num (the reference's other branch, bifuser_n.py:54-60 /
// 88-94, stays in Python: the call returns COOCC_SEARCH_SMALL and has launched nothing after the count read).
#include <stdlib.h>
#include <string.h>

#include "common.h"

__global__ __launch_bounds__(256) void k_gather_xyz(const float* __restrict__ xyz, const int32_t* __restrict__ idx, int m,
                                                     float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int s = idx[i];
  out[i * 3 + 0] = xyz[s * 3 + 0]; out[i * 3 + 1] = xyz[s * 3 + 1]; out[i * 3 + 2] = xyz[s * 3 + 2];
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct SearchWs {
  uint8_t* flags;        // [2][V]
  int32_t* cws;          // [2][V/1024 + 2]
  float* xyz;            // [2][V][3]   (img, pts)
  int32_t* maps;         // [2][V]      voxel -> ordinal in (img, pts) list
  // per direction d (0: pts queries <- img keys, main stream; 1: img queries <- pts keys, side stream)
  int32_t* rep[2];       // [fps_num]
  float* rep_xyz[2];     // [fps_num][3]
  float* val[2];         // [fps_num][K]
  int32_t* nn[2];        // [fps_num][K]
  int32_t* group[2];     // [fps_num][max_cluster]
  int32_t* winner[2];    // [K][V]
  void* fps[2];
  size_t fps_bytes;
  size_t total;
};

static SearchWs carve(char* base, int V, int K, int fps_num, int max_cluster, int X, int Y, int Z) {
  SearchWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += align256(bytes); return p; };
  w.flags = (uint8_t*)take((size_t)2 * V);
  w.cws = (int32_t*)take((size_t)2 * (V / 1024 + 2) * 4);
  w.xyz = (float*)take((size_t)2 * V * 3 * 4);
  w.maps = (int32_t*)take((size_t)2 * V * 4);
  w.fps_bytes = coocc_fps_voxels_ws(X, Y, Z);
  for (int d = 0; d < 2; ++d) {
    w.rep[d] = (int32_t*)take((size_t)fps_num * 4);
    w.rep_xyz[d] = (float*)take((size_t)fps_num * 3 * 4);
    w.val[d] = (float*)take((size_t)fps_num * K * 4);
    w.nn[d] = (int32_t*)take((size_t)fps_num * K * 4);
    w.group[d] = (int32_t*)take((size_t)fps_num * max_cluster * 4);
    w.winner[d] = (int32_t*)take((size_t)K * V * 4);
    w.fps[d] = take(w.fps_bytes);
  }
  w.total = o;
  return w;
}

extern "C" size_t coocc_fuser_search_ws(const coocc_search_desc* d) {
  if (!d || d->X <= 0 || d->Y <= 0 || d->Z <= 0 || d->K <= 0) return 0;
  return carve(nullptr, d->X * d->Y * d->Z, d->K, d->fps_num, d->max_cluster, d->X, d->Y, d->Z).total;
}

#define SRC(call)                    \
  do {                               \
    int rc__ = (call);               \
    if (rc__ != COOCC_OK) return rc__; \
  } while (0)

extern "C" int coocc_fuser_search(coocc_search_desc* d, void* stream, void* side_stream) {
  COOCC_CHECK_ARG(d && d->cat4 && d->pts && d->lin && d->counts && d->rows && d->rows_p && d->near_img && d->near_pts && d->ws &&
                      d->offsets && d->counts_host, "fuser_search: null pointer");
  COOCC_CHECK_ARG(d->C > 0 && d->X > 0 && d->Y > 0 && d->Z > 0 && d->K >= 1 && d->K <= 8 && d->fps_num > 0 && d->max_cluster > 0 &&
                      d->noff > 0, "fuser_search: bad sizes");
  COOCC_CHECK_ARG(stream != side_stream && side_stream, "fuser_search: needs a second stream for the img <- pts direction");
  const int V = d->X * d->Y * d->Z, K = d->K, C = d->C;
  SearchWs w = carve((char*)d->ws, V, K, d->fps_num, d->max_cluster, d->X, d->Y, d->Z);
  COOCC_CHECK_ARG(d->ws_bytes >= w.total, "fuser_search: workspace smaller than coocc_fuser_search_ws()");
  hipStream_t s0 = as_stream(stream), s1 = as_stream(side_stream);
  // DIAGNOSTIC ONLY (profiles/r6_serving_probe_events.txt: what of the search costs the serving loop its last 8 %): parts of the stage
  // left out -- 1: the FPS launch, 2: K3-K5 + row tables of both directions, 4: the count read (the host keeps the previous call's
  // counts).  The outputs are then NOT this frame's.
  static const int sdiag = [] { const char* e = getenv("COOCC_SEARCH_DIAG_SKIP"); return e ? atoi(e) : 0; }();
  int32_t* lin_img = d->lin;
  int32_t* lin_pts = d->lin + V;

  // K1: concat rows (img slot already in place) + non-empty flags, stream compaction, counts
  SRC(coocc_fuser_prepare_rows(d->cat4, 1, 4 * C, d->pts, d->pts_rows, d->pts_stride, d->cat4, w.flags, w.flags + V, 1, C, V, stream));
  SRC(coocc_compact_flags(w.flags, V, lin_img, d->counts, w.cws, (size_t)(V / 1024 + 2) * 4, stream));
  SRC(coocc_compact_flags(w.flags + V, V, lin_pts, d->counts + 1, w.cws + (V / 1024 + 2), (size_t)(V / 1024 + 2) * 4, stream));
  if (!((sdiag & 4) && d->counts_host[0] > 0)) COOCC_HIP(hipMemcpyAsync(d->counts_host, d->counts, 8, hipMemcpyDeviceToHost, s0));
  if (!((sdiag & 4) && d->counts_host[0] > 0)) {
    // the one host sync of the stage.  A BLOCKING event: the calling thread sleeps instead of spinning (hipStreamSynchronize may
    // spin), so several prefetch threads per rank -- and 8 ranks per node -- do not burn the host cores the issuing threads need
    // No system-scope fence on it (layout.hip, "device-scope events"): the two counts reach the host through the memcpy command above,
    // which is complete -- staged into the caller's host buffer -- before the event fires; nothing a kernel wrote is read by the host
    // here.  COOCC_SEARCH_COUNT_FENCE=1 restores the default event (one cache writeback / invalidate per sample).
    static const bool fence = [] { const char* e = getenv("COOCC_SEARCH_COUNT_FENCE"); return e && e[0] == '1'; }();
    hipEvent_t got;
    COOCC_HIP(hipEventCreateWithFlags(&got, hipEventBlockingSync | hipEventDisableTiming | (fence ? 0u : hipEventDisableSystemFence)));
    const hipError_t e1 = hipEventRecord(got, s0);
    const hipError_t e2 = e1 == hipSuccess ? hipEventSynchronize(got) : e1;
    (void)hipEventDestroy(got);
    if (e2 != hipSuccess) return coocc_set_error(COOCC_EHIP, "fuser_search: waiting for the voxel counts failed: %s", hipGetErrorString(e2));
  }
  const int Ni = d->counts_host[0], Np = d->counts_host[1];
  if (Ni <= d->fps_num || Np <= d->fps_num) return COOCC_SEARCH_SMALL;

  float* xyz_img = w.xyz;
  float* xyz_pts = w.xyz + (size_t)V * 3;
  SRC(coocc_lin_to_coords(lin_img, Ni, d->X, d->Y, d->Z, xyz_img, nullptr, stream));
  SRC(coocc_lin_to_coords(lin_pts, Np, d->X, d->Y, d->Z, xyz_pts, nullptr, stream));
  int32_t* map_img = w.maps;
  int32_t* map_pts = w.maps + V;
  SRC(coocc_voxel_index_map(lin_img, Ni, V, map_img, stream));
  SRC(coocc_voxel_index_map(lin_pts, Np, V, map_pts, stream));

  // K2 of BOTH directions as one launch of two workgroups when the grid fits the register-resident kernel (csrc/knn.hip); the
  // rest of each direction then forks onto its own stream.  Otherwise each direction runs its own FPS on its stream.
  const int pair_rc = (sdiag & 1) ? COOCC_OK
                                  : coocc_fps_voxels_pair(lin_pts, Np, w.rep[0], w.fps[0], lin_img, Ni, w.rep[1], w.fps[1], w.fps_bytes, d->X, d->Y,
                                                          d->Z, d->fps_num, stream);
  if (pair_rc != COOCC_OK && pair_rc != 2) return pair_rc;
  const bool paired = pair_rc == COOCC_OK;
  // fork / join of the two search directions.  Whatever happens after the fork, the side stream is joined back into `stream` and
  // the events are destroyed before this call returns (a failure in the middle must not leave the caller's side stream forked
  // behind an unjoined event, nor leak two events per failed call).
  hipEvent_t fork = nullptr, join = nullptr;
  COOCC_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming | hipEventDisableSystemFence));   // device-side ordering only
  if (hipEventCreateWithFlags(&join, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
    (void)hipEventDestroy(fork);
    return coocc_set_error(COOCC_EHIP, "fuser_search: hipEventCreate failed");
  }
  bool forked = false;
  auto finish = [&](int rc) -> int {
    if (forked) {
      if (hipEventRecord(join, s1) != hipSuccess || hipStreamWaitEvent(s0, join, 0) != hipSuccess) {
        if (rc == COOCC_OK) rc = coocc_set_error(COOCC_EHIP, "fuser_search: joining the side stream failed");
      }
    }
    (void)hipEventDestroy(fork);
    (void)hipEventDestroy(join);
    return rc;
  };
#define SRF(call)                                \
  do {                                           \
    int rc__ = (call);                           \
    if (rc__ != COOCC_OK) return finish(rc__);   \
  } while (0)
  // Paired FPS (one launch on `stream` for both directions): the img <- pts direction could only start on the side stream once that
  // launch has finished, i.e. the side QUEUE would sit on an unsatisfied cross-queue wait for the whole 2-8 ms chain -- and a queue
  // whose head is a pending wait slows the dispatch of every other queue of the device (the serving loop lost 15-20 % to one such
  // wait per sample, profiles/r6_serving_probe_events.txt).  Both directions then run on `stream`, one after the other (+ ~0.2 ms of
  // search latency, which is prefetched).  Unpaired (large grids): each direction's own FPS chain runs on its stream, forked first.
  void* side_q = paired ? stream : side_stream;
  if (!paired) {
    if (hipEventRecord(fork, s0) != hipSuccess || hipStreamWaitEvent(s1, fork, 0) != hipSuccess)
      return finish(coocc_set_error(COOCC_EHIP, "fuser_search: forking the side stream failed"));
    forked = true;
  }

  // one direction: queries (lin_q, Q, xyz_q, map_q) <- keys (Nk, xyz_k, map_k); near: [K][Q] key ordinals (-1 = none)
  auto direction = [&](int dd, void* st, const int32_t* lin_q, int Q, const float* xyz_q, const int32_t* map_q, int Nk,
                       const float* xyz_k, const int32_t* map_k, int32_t* near) -> int {
    if (sdiag & 2) return COOCC_OK;
    if (!paired) SRC(coocc_fps_voxels(lin_q, Q, d->X, d->Y, d->Z, d->fps_num, w.rep[dd], w.fps[dd], w.fps_bytes, st));
    hipLaunchKernelGGL(k_gather_xyz, dim3(cdiv(d->fps_num, 256)), dim3(256), 0, as_stream(st), xyz_q, w.rep[dd], d->fps_num, w.rep_xyz[dd]);
    COOCC_LAUNCH_CHECK("k_gather_xyz");
    SRC(coocc_knn_topk_voxels(d->fps_num, Nk, K, d->X, d->Y, d->Z, w.rep[dd], lin_q, map_k, d->offsets, d->noff, w.rep_xyz[dd], xyz_k,
                              w.val[dd], w.nn[dd], st));
    SRC(coocc_ball_query_voxels(d->fps_num, 0.f, d->radius, d->max_cluster, d->X, d->Y, d->Z, w.rep[dd], lin_q, map_q, w.group[dd], st));
    SRC(coocc_knn_assign(d->fps_num, K, d->max_cluster, Q, d->dist_thresh, w.val[dd], w.nn[dd], w.group[dd], w.winner[dd], near, st));
    return COOCC_OK;
  };
  // img queries <- nearest pts keys (bifuser_n.py:150-162) on the side stream; for knum > 1 the reference indexes inds_img
  // with the pts ordinals (:158) -- kept
  SRF(direction(1, side_q, lin_img, Ni, xyz_img, map_img, Np, xyz_pts, map_pts, d->near_pts));
  for (int k = 0; k < K && !(sdiag & 2); ++k)
    SRF(coocc_index_rows_i32(K == 1 ? lin_pts : lin_img, K == 1 ? Np : Ni, d->near_pts + (size_t)k * Ni, Ni, d->rows_p + (size_t)k * V, side_q));
  // pts queries <- nearest img keys (bifuser_n.py:137-148)
  SRF(direction(0, stream, lin_pts, Np, xyz_pts, map_pts, Ni, xyz_img, map_img, d->near_img));
  for (int k = 0; k < K && !(sdiag & 2); ++k)
    SRF(coocc_index_rows_i32(lin_img, Ni, d->near_img + (size_t)k * Np, Np, d->rows + (size_t)k * V, stream));
#undef SRF
  return finish(COOCC_OK);
}
