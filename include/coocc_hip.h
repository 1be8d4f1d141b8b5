/*
 * coocc_hip.h -- C ABI of libcoocc_hip.so, the MI355X (gfx950) implementation of
 * Co-Occ's fused-voxel hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all
 *     work is enqueued on it and nothing synchronises unless stated;
 *   - functions return 0 on success, a negative COOCC_E* code otherwise and never
 *     exit() (the reference launchers do: ball_query_cuda.cu:73-77);
 *     coocc_last_error() returns the message of the last failure on this thread;
 *   - dense voxel volumes inside the library are channels-last ("NDHWC": rows of
 *     `stride` floats, one row per voxel, voxel order (b,x,y,z)); the reference's
 *     [B,C,X,Y,Z] layout is converted once at the boundary;
 *   - "P/" = projects/mmdet3d_plugin/, "M/" = mmdetection3d/mmdet3d/ in the reference.
 *
 * The three reference pybind11 extension entry points this library replaces one to
 * one are marked [EXT]; the remaining entry points replace ATen op sequences of the
 * reference's Python modules (file:line given per function).
 */
#ifndef COOCC_HIP_H
#define COOCC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COOCC_OK 0
#define COOCC_EINVAL (-1)  /* bad argument */
#define COOCC_EHIP (-2)    /* HIP runtime / launch failure */
#define COOCC_ENOMEM (-3)  /* workspace too small */

const char* coocc_last_error(void);
int coocc_abi_version(void);

/* ---------------------------------------------------------------- CU-partitioned streams */
/* hipExtStreamCreateWithCUMask wrappers (mask_host: bit i = CU i enabled, nwords 32-bit words).
 * Used to give the single-workgroup FPS chains private CUs next to the convolution stream. */
int coocc_device_cu_count(int* n);
int coocc_stream_create_cu_mask(const uint32_t* mask_host, int nwords, void** stream_out);
int coocc_stream_destroy(void* stream);

/* ---------------------------------------------------------------- device-scope events */
/* hipEvent wrappers created with hipEventDisableSystemFence (+ hipEventDisableTiming unless flags & 1): ordering between streams of
 * ONE device without the system-scope cache writeback / invalidate a default event performs at every record (24 % of the serving
 * loop's throughput with three dense stages in flight).  Not for host-visible or cross-device hand-over: synchronise the stream
 * (or copy device -> host on it) for that.  The reference has no counterpart: it runs one stream. */
int coocc_event_create(int flags, void** event_out);   /* flags: 1 = keep timestamps, 2 = blocking (sleeping) synchronize */
int coocc_event_destroy(void* event);
int coocc_event_record(void* event, void* stream);
int coocc_stream_wait_event(void* stream, void* event);
int coocc_event_synchronize(void* event);
int coocc_event_query(void* event);                 /* 1 complete, 0 not yet, < 0 error */
int coocc_event_elapsed_ms(void* start, void* stop, float* ms);

/* ---------------------------------------------------------------- layout / K1 */

/* [B,C,V] (reference NCDHW, V = X*Y*Z) -> rows of `dst_stride` floats at channel
 * offset `dst_coff` (NDHWC).  Replaces the permute() views of bifuser_n.py:133. */
int coocc_ncdhw_to_ndhwc(const float* src, float* dst, int B, int C, int V, int dst_stride,
                         int dst_coff, void* stream);
/* inverse: NDHWC rows -> [B,C,V] */
int coocc_ndhwc_to_ncdhw(const float* src, float* dst, int B, int C, int V, int src_stride,
                         int src_coff, void* stream);

/* BiFuser_N.forward prologue (bifuser_n.py:129-135) in one pass: reads both
 * [B,C,V] volumes, writes the 4C concat rows [img | pts | 0 | 0] (the zero halves are
 * the fused_feats_* zero-fills of :164-168) and per-voxel non-empty flags
 * (feats.sum(1) != 0, channels summed in ascending order in fp32). */
int coocc_fuser_prepare(const float* img, const float* pts, float* cat4, uint8_t* flag_img,
                        uint8_t* flag_pts, int B, int C, int V, void* stream);
/* The same prologue when a producer hands over channels-last rows [B*V, *_stride] (*_rows != 0) instead of NCDHW: the
 * fused lift-splat and the sparse LiDAR encoder do.  A row source that IS its destination slot (img == cat4,
 * img_stride == 4*C: lift-splat wrote straight into the concat buffer) is only read for its flag. */
int coocc_fuser_prepare_rows(const float* img, int img_rows, int img_stride, const float* pts, int pts_rows,
                             int pts_stride, float* cat4, uint8_t* flag_img, uint8_t* flag_pts, int B, int C, int V,
                             void* stream);

/* torch.nonzero on a flag volume (bifuser_n.py:130-131): ascending linear voxel ids
 * (== lexicographic (b,x,y,z)).  lin:[n<=total] i32, count: 1 i32.  ws: >= 4*(total/1024+2) bytes. */
int coocc_compact_flags(const uint8_t* flags, int total, int32_t* lin, int32_t* count, void* ws,
                        size_t ws_bytes, void* stream);
/* ... and, with map != NULL, the inverse table map[element] = its ordinal in lin, or -1 (written by the same last pass: what
 * coocc_scatter_fine_grouped looks the foreground voxels up in) */
int coocc_compact_flags_ex(const uint8_t* flags, int total, int32_t* lin, int32_t* count, int32_t* map, void* ws,
                           size_t ws_bytes, void* stream);
/* lin -> float xyz rows [n,3] and/or int64 (b,x,y,z) rows [n,4] (either may be NULL) */
int coocc_lin_to_coords(const int32_t* lin, int n, int X, int Y, int Z, float* xyz, int64_t* bxyz,
                        void* stream);

/* ---------------------------------------------------------------- K2..K5 */

/* [EXT] furthest_point_sampling_wrapper (M/ops/furthest_point_sample/src/
 * furthest_point_sample.cpp:35-46; kernel furthest_point_sample_cuda.cu:25-141).
 * points:[b,n,3] f32, temp:[b,n] f32 scratch (initialised to 1e10 here, as
 * furthest_point_sample.py:29 does), idx:[b,m] i32.  Ties resolve exactly as the
 * reference block reduction does (block = min(2^floor(log2 n),1024)). */
int coocc_furthest_point_sampling(int b, int n, int m, const float* points, float* temp,
                                  int32_t* idx, void* stream);

/* Same selection as coocc_furthest_point_sampling for a list of DISTINCT voxels of one X*Y*Z
 * grid (lin:[n] ascending linear ids, batch 0) -- what bifuser_n.py:97 passes in -- with
 * bucket pruning (4x4x8 voxel tiles cached in LDS).  ws >= coocc_fps_voxels_ws(X,Y,Z) bytes. */
size_t coocc_fps_voxels_ws(int X, int Y, int Z);
int coocc_fps_voxels(const int32_t* lin, int n, int X, int Y, int Z, int m, int32_t* idx, void* ws,
                     size_t ws_bytes, void* stream);
/* The two FPS problems of BiFuser_N's search (bifuser_n.py:132 over the pts list, :152 over the img list; same grid, same m) as
 * ONE launch of two workgroups.  Taken when the grid has at most 640 buckets of 4x4x8 voxels (100x100x8: 625), where the whole
 * distance table lives in the register file (no memory access inside the 2047 dependent iterations).  Returns COOCC_OK, or 2
 * with nothing enqueued when the grid is too large (call coocc_fps_voxels twice instead).  Same selections as coocc_fps_voxels. */
int coocc_fps_voxels_pair(const int32_t* lin0, int n0, int32_t* idx0, void* ws0, const int32_t* lin1, int n1, int32_t* idx1,
                          void* ws1, size_t ws_bytes_each, int X, int Y, int Z, int m, void* stream);

/* [EXT] ball_query_wrapper (M/ops/ball_query/src/ball_query.cpp:32-45; kernel
 * ball_query_cuda.cu:11-54).  new_xyz:[b,m,3] centres, xyz:[b,n,3], idx:[b,m,nsample]
 * i32 (zeroed here, as ball_query.py:35 does). */
int coocc_ball_query(int b, int n, int m, float min_radius, float max_radius, int nsample,
                     const float* new_xyz, const float* xyz, int32_t* idx, void* stream);

/* ---- K3 / K4 when both point sets are the non-empty voxels of ONE dense X x Y x Z grid (the fuser's case: the lists
 * are ascending linear voxel ids v = (x*Y + y)*Z + z, so "index order" == lexicographic voxel order).  Bit-identical to
 * coocc_ball_query / coocc_knn_topk on the same points, 20-100x less work (no 2048 x N distance sweep). */
/* The whole index-search stage of BiFuser_N.forward (bifuser_n.py:129-162: non-empty voxel lists, and for both directions FPS ->
 * top-K -> ball query -> assignment -> neighbour row tables) as ONE call: every launch, the fork / join of the two directions
 * over `stream` / `side_stream` and the stage's single device->host read (the two counts) are issued from C++ (csrc/search.hip).
 * B == 1; grid small enough for coocc_fps_voxels.  Returns COOCC_OK, COOCC_SEARCH_SMALL (a list has <= fps_num voxels: the
 * reference takes its other branch, bifuser_n.py:54-60 -- nothing was launched after the count read, use the generic entry
 * points), or a negative error.  On return everything is enqueued and `stream` has joined `side_stream`. */
#define COOCC_SEARCH_SMALL 1
typedef struct coocc_search_desc {
  float* cat4;             /* [V,4C] concat rows; slot 0 (camera rows) already written, slots 1..3 written here / by G1 */
  const float* pts;        /* LiDAR volume: NCDHW [C,V] (pts_rows = 0) or channels-last rows [V,pts_stride] (pts_rows = 1) */
  int pts_rows, pts_stride;
  int C, X, Y, Z, K;       /* channels per slot, grid, knum */
  int fps_num, max_cluster;/* 2048, 200 (bifuser_n.py:137) */
  float radius, dist_thresh;   /* 6, 13.3 */
  const uint32_t* offsets; /* coocc_knn_topk_voxels' sorted offset table */
  int noff;
  int32_t* lin;            /* out [2][V]: ascending non-empty voxel ids of (img, pts) */
  int32_t* counts;         /* out [2] (device) */
  int32_t* near_img;       /* out [K][Np] dense: img ordinal assigned to each pts voxel (-1 none)   (capacity K*V) */
  int32_t* near_pts;       /* out [K][Ni] dense                                                    (capacity K*V) */
  int32_t* rows;           /* out [K][V]: concat-buffer row of the k-th img neighbour of pts voxel j (first Np entries) */
  int32_t* rows_p;         /* out [K][V]: ... of img voxel j (first Ni entries); K > 1 indexes inds_img (bifuser_n.py:158) */
  void* ws;                /* >= coocc_fuser_search_ws(desc) bytes (device) */
  size_t ws_bytes;
  int32_t* counts_host;    /* out [2] (host): Ni, Np */
} coocc_search_desc;
size_t coocc_fuser_search_ws(const coocc_search_desc* d);
int coocc_fuser_search(coocc_search_desc* d, void* stream, void* side_stream);

/* map[v] = ordinal of voxel v in lin[0..n) or -1.  map:[nvox] i32. */
int coocc_voxel_index_map(const int32_t* lin, int n, int nvox, int32_t* map, void* stream);
/* Device-side counts (no host round trip; grids sized for the capacity, surplus workgroups leave at once): the *_dev forms of
 * the entry points whose work list is a data-dependent voxel / point list -- what lets the whole dense stage replay as one
 * captured hipGraph (co_occ_amd/graph.py).  n_dev: int32 on the device, the list length (clamped to the capacity). */
int coocc_voxel_index_map_dev(const int32_t* lin, int n_cap, const int32_t* n_dev, int nvox, int32_t* map, void* stream);
/* ball_query (bifuser_n.py:109) for m centres given as ordinals into the query list lin_q; map_q = its index map.
 * idx:[m,nsample] query ordinals, first-hit padded, zeros when a centre has no hit (ball_query_cuda.cu:11-54). */
int coocc_ball_query_voxels(int m, float min_radius, float max_radius, int nsample, int X, int Y, int Z,
                            const int32_t* centre_ord, const int32_t* lin_q, const int32_t* map_q, int32_t* idx,
                            void* stream);
/* top-K (bifuser_n.py:101-103) for nq representatives given as ordinals into lin_q against the keys of map_k.
 * offsets:[noff] u32 = (dx+128) | (dy+128)<<8 | (dz+128)<<16 sorted by (d^2, dx, dy, dz) and complete up to some radius
 * (co_occ_amd.fuser.offset_table); representatives not resolved inside the table are finished by brute force over
 * q:[nq,3] (the representatives' xyz) / key:[nk,3].  val:[nq,K], idx:[nq,K] as coocc_knn_topk. */
int coocc_knn_topk_voxels(int nq, int nk, int K, int X, int Y, int Z, const int32_t* rep_ord, const int32_t* lin_q,
                          const int32_t* map_k, const uint32_t* offsets, int noff, const float* q, const float* key,
                          float* val, int32_t* idx, void* stream);

/* norm + topk(largest=False) of bifuser_n.py:101-103 without the [nq,nk,3] temporary;
 * ties ordered by (d^2, key index).  q:[nq,3], key:[nk,3]; val:[nq,K] f32 (= sqrt d^2),
 * idx:[nq,K] i32.  1 <= K <= 8, K <= nk. */
int coocc_knn_topk(int nq, int nk, int K, const float* q, const float* key, float* val,
                   int32_t* idx, void* stream);

/* assignment of bifuser_n.py:104-125 (K==1: :73-85): highest valid centre ordinal wins.
 * val,nn:[nc,K]; group:[nc,ns]; winner:[K,nq] i32 scratch; out:[K,nq] i32 (-1 = none). */
int coocc_knn_assign(int nc, int K, int ns, int nq, float dist_thresh, const float* val,
                     const int32_t* nn, const int32_t* group, int32_t* winner, int32_t* out,
                     void* stream);
/* small path (Q <= fps_num, K == 1; bifuser_n.py:54-60): out[q] = val<thresh ? nn : -1 */
int coocc_knn_threshold(int nq, float dist_thresh, const float* val, const int32_t* nn,
                        int32_t* out, void* stream);
/* rows = base[sel[i]] (sel < 0 wraps like Python indexing: bifuser_n.py:139-144) */
int coocc_index_rows_i32(const int32_t* base, int nbase, const int32_t* sel, int n, int32_t* rows,
                         void* stream);

/* ---------------------------------------------------------------- G1, C0..C3 (implicit GEMM) */

/* Pack a conv / linear weight for coocc_conv_fwd.  w_host layout [Cout][Cin][taps]
 * (nn.Conv3d weight flattened; nn.Linear(C*K,C) viewed as [Cout][K][C] must be passed
 * as taps-major via `tap_major=1`: [Cout][taps][Cin]).  Returns required float count
 * when packed == NULL. */
int64_t coocc_conv_pack_weights(const float* w_host, int Cout, int Cin, int taps, int tap_major,
                                float* packed_host);

typedef struct coocc_conv_desc {
  const float* in;        /* input rows (NDHWC or a row table), already offset to channel 0 */
  const float* w;         /* packed weights (device) */
  float* out;
  const float* scale;     /* [Cout] or NULL (folded eval-mode BN scale) */
  const float* bias;      /* [Cout] or NULL */
  const float* res;       /* res_mode 1: residual added before ReLU; 2: gate multiplied after ReLU;
                             3: raw partial sums of an earlier K-slice pass, added before scale/bias */
  const int32_t* gather;  /* NULL: geometric taps; else [taps][M] input row ids */
  const int32_t* out_rows;/* NULL: identity; else [M] output (and res) row ids */
  float* ws;              /* split-K workspace or NULL */
  int64_t ws_floats;      /* capacity of ws in floats (split-K needs splitk*M*roundup(Cout,128)) */
  int M, Cin, Cout, taps;
  int in_stride, out_stride, res_stride;
  int B, Xi, Yi, Zi, Xo, Yo, Zo, ksize, stride, pad;
  int relu, res_mode;
  int splitk;             /* 0 = choose automatically */
  int tile_hint;          /* 0 = choose automatically; 128 / 160 force the M tile of the large configuration */
  int kx, ky, kz;         /* kx > 0: per-axis kernel extents (taps = kx*ky*kz, tap t = (dx*ky + dy)*kz + dz) */
  int px, py, pz;         /*         and paddings, overriding ksize / pad */
  int wgroup_rows;        /* > 0: output rows [g*wgroup_rows, (g+1)*wgroup_rows) use the g-th weight pack of `w`
                             (consecutive packs of taps*ceil(Cin/32)*roundup(Cout,128)*32 floats); must be a
                             multiple of 640.  Used by the Winograd path: one launch, 16 transform points. */
  int mfma_dtype;         /* 0: fp32 operands (v_mfma_f32_32x32x2_f32, exact fp32 -- the parity path);
                             1: operands rounded to bf16 in LDS (v_mfma_f32_32x32x16_bf16), fp32 accumulate / epilogue /
                                storage: the reduced-precision path of the OpenOccupancy config; geometric taps only;
                             2: the same arithmetic with the operands already bf16 in memory: `in` = [rows][in_stride] bf16
                                (coocc_rows_to_bf16), `w` = bf16 pack [(Cin/64 chunk, tap)][roundup(Cout,128)/32][4][64 lanes][8] (fragment-major:
                                lane l of k-step s holds k = 16 s + 8 (l >> 5) + 0..7 of column 32 nt + (l & 31)); Cin % 64 == 0;
                             3: fp32-ACCURATE arithmetic on the f16 matrix cores (csrc/gemm_h2.hip): every operand split into two
                                f16 halves (hi + lo * 2^-11, 22 significand bits), three v_mfma_f32_32x32x16_f16 per step, fp32
                                accumulation -- measured error vs fp64 is half that of mfma_dtype 0.  `in` = H2 rows
                                (coocc_rows_to_h2 / coocc_wino_input_h2), `w` = H2 pack [(Cin/32 chunk, tap)][roundup(Cout,128)/32]
                                [2 k16 steps][hi | lo][64 lanes][8 f16]; Cin % 32 == 0; any geometry or row table (stride-1 "same"
                                layers with <= 3 z taps share one LDS image between the z taps); wgroup_rows a multiple of 128;
                             4: ONE-term f16 (the reduced-precision path of configs[4], coocc_multi_r101_openoccupancy.py + the fp16
                                hooks coocc_ray.py:136,142,265 / fpn3d.py:69): `in` = f16 rows [rows][in_stride] (coocc_rows_to_f16 or a
                                producer's out16), `w` = f16 pack [(Cin/64 chunk, tap)][roundup(Cout,128)/32][4 k16 steps][64 lanes][8 f16];
                                fp32 accumulate / epilogue; Cin % 64 == 0; same kernels as 3 (csrc/gemm_h2.hip, TERMS = 1) */
  float alpha;            /* mfma_dtype 3: the accumulators are multiplied by alpha (1 / operand scale) before the epilogue; 0 = 1 */
  const int32_t* M_dev;   /* row-table launches (gather != NULL; mfma_dtype 3: any launch), splitk = 1: the number of rows read on the DEVICE (<= M; the grid
                             is sized for M = capacity and tiles past *M_dev leave at once) -- lets a captured hipGraph run over
                             voxel lists whose length is only known on the device; NULL: M rows */
  int gather_stride;      /* entries per tap of `gather` (0 = M) */
  void* out16;            /* mfma_dtype 3 / 4: a second, f16 copy of the output rows ([rows][out16_stride] f16, after the epilogue) --
                             16-bit activations written by the PRODUCER's epilogue: the next f16 layer reads 2 bytes per element and no
                             conversion pass runs in between; NULL = none */
  int out16_stride;
  int out_h2;             /* mfma_dtype 3: write the output as H2 rows (out_stride = channels per row, % 32 == 0) -- the producer's
                             epilogue emits the next split-f16 layer's operand, no conversion pass in between */
  void* out_h2_twin;      /* mfma_dtype 3: an H2 copy [M][Cout] (Cout % 32 == 0) of the fp32 output rows, written next to them by the
                             epilogue: the operand of a following split-f16 layer that is NOT on the Winograd path (strided, 1x1x1,
                             small grids, the render MLPs) -- replaces that layer's coocc_rows_to_h2 pass; NULL = none */
  int32_t* tile_sem;      /* mfma_dtype 3 / 4, split-K: arrival counters, one per 128 x 128 output tile, zero on entry and left zero.
                             When given, the LAST workgroup of a tile sums the partial slabs in slice order (the order of the
                             second-pass kernel: same bits) and runs the epilogue itself -- no reduction launch, and the H2 / f16
                             outputs above become available to split-K layers; NULL = the two-launch form */
  int tile_sem_ints;      /* capacity of tile_sem */
  const float* alpha_dev; /* mfma_dtype 3 / 4: the accumulators are also multiplied by *alpha_dev, read on the device -- the inverse of an
                             operand scale that a kernel chose (coocc_conv_epilogue_bwd_ex: the gradient operand of the training
                             path's dgrad GEMMs); NULL = none */
} coocc_conv_desc;

/* nn.Conv3d(k=3|1)+BN(eval)+ReLU(+residual) (bifuser_n.py:23-30, resnet3d.py:34-64,
 * fpn3d.py:46-64, occ_head.py:102-132), nn.Linear (+ReLU) and the gather->knn_enc->gate
 * ->scatter of bifuser_n.py:138-169 as one fp32-MFMA implicit-GEMM kernel family. */
int coocc_conv_fwd(const coocc_conv_desc* d, void* stream);

/* fp32 rows (row stride in_stride floats, first C columns, C % 8 == 0) -> dense bf16 rows [rows][C], round to nearest even:
 * the activation operand of coocc_conv_fwd with mfma_dtype 2 (the fp16 / autocast hooks of coocc_ray.py:136,142,265 and
 * fpn3d.py:69 cast the same tensors once per layer). */
int coocc_rows_to_bf16(const float* in, int in_stride, int64_t rows, int C, void* out_bf16, void* stream);

/* fp32 rows * scale -> "H2 rows": per row C/32 chunks of [32 x f16 hi | 32 x f16 lo], hi = f16(v), lo = f16((v - hi) * 2^11)
 * (4 C bytes per row, the size of the fp32 row): the activation operand of coocc_conv_fwd with mfma_dtype 3.  C % 32 == 0.
 * Replaces nothing in the reference (its convolutions are cuDNN fp32, resnet3d.py:34-64): it is how the same fp32 sums are
 * evaluated on the 16-bit matrix pipe. */
int coocc_rows_to_h2(const float* in, int in_stride, int64_t rows, int C, float scale, void* out_h2, void* stream);
/* ... with the scale also multiplied by *scale_dev (device word, NULL = 1): see coocc_conv_epilogue_bwd_ex */
int coocc_rows_to_h2_ex(const float* in, int in_stride, int64_t rows, int C, float scale, const float* scale_dev, void* out_h2,
                        void* stream);
/* fp32 rows -> f16 rows [rows][C], round to nearest even (C % 8 == 0): the operand of mfma_dtype 4 when no producer wrote it */
int coocc_rows_to_f16(const float* in, int in_stride, int64_t rows, int C, void* out_f16, void* stream);
/* out row j = H2(in[row_ids[j]] * scale), j < n_cap (n_dev != NULL: j < min(n_cap, *n_dev), the count read on the device) */
int coocc_rows_to_h2_gather(const float* in, int in_stride, const int32_t* row_ids, int64_t n_cap, const int32_t* n_dev,
                            int C, float scale, void* out_h2, void* stream);

/* Winograd F(m x m, 3x3), m = tile = 2 or 4, over (x,y) for 3x3x3 stride-1 pad-1 convs (z stays a direct 3-tap
 * conv): input transform, then ONE coocc_conv_fwd launch over (m+2)^2 x group_rows rows (kx=ky=1, kz=3, pz=1,
 * wgroup_rows=group_rows, weights = (m+2)^2 packs of G g G^T), then output transform + epilogue.  V / Mb:
 * [(m+2)^2][group_rows][C]; row = ((b*Tx+tx)*Ty+ty)*Z+z, Tx = ceil(X/m); group_rows a multiple of lcm(640, Z)
 * >= B*Tx*Ty*Z.  12*Cin (m=2) / 6.75*Cin (m=4) multiplies per output instead of 27*Cin. */
int coocc_wino_input(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, float* V,
                     int64_t group_rows, void* stream);
/* coocc_wino_input writing its C channels into V rows of `vstride` floats (V already offset to the first of them): one
 * GEMM's input channels gathered from several channel ranges of the source rows. */
int coocc_wino_input_strided(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, float* V,
                             int vstride, int64_t group_rows, void* stream);
/* coocc_wino_input_strided writing V * scale as H2 rows (mfma_dtype 3): C % 32 == 0, vstride (channels of a V row) % 32 == 0,
 * V offset to a 32-channel chunk boundary (128 bytes per chunk). */
int coocc_wino_input_h2(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, void* V,
                        int vstride, int64_t group_rows, float scale, void* stream);
/* ... with the scale also multiplied by *scale_dev (device word, NULL = 1): see coocc_conv_epilogue_bwd_ex */
int coocc_wino_input_h2_ex(const float* in, int in_stride, int B, int X, int Y, int Z, int C, int tile, void* V,
                           int vstride, int64_t group_rows, float scale, const float* scale_dev, void* stream);
/* Scatter-form sparse half of a dense 3x3x3 convolution (csrc/sparse_taps.hip): P:[Np][27][Cout] = per occupied input voxel
 * and tap the contribution W_t . in[u] (a row-table coocc_conv_fwd with N = 27*Cout), map: voxel -> row of P or -1
 * (coocc_voxel_index_map); S[v][n] = scale[n] * sum_t P[map[v + t - 1]][t][n], taps in order (deterministic).  p_rows > 0: rows of
 * P; a map entry >= p_rows is reported (device printf) and skipped instead of dereferenced. */
int coocc_sparse_tap_sum(const float* P, const int32_t* map, int B, int X, int Y, int Z, int Cout, const float* scale,
                         float* S, int s_stride, int p_rows, void* stream);
int coocc_wino_output(const float* Mb, int64_t group_rows, int B, int X, int Y, int Z, int C, int tile, float* out,
                      int out_stride, const float* scale, const float* bias, const float* res, int res_stride,
                      int relu, void* stream);
/* ... and, with out_h2_twin != NULL (C % 32 == 0), an H2 copy [B*X*Y*Z][C] of the finished rows next to them: the operand of a
 * following split-f16 layer outside the Winograd path (see coocc_conv_desc.out_h2_twin). */
int coocc_wino_output_ex(const float* Mb, int64_t group_rows, int B, int X, int Y, int Z, int C, int tile, float* out,
                         int out_stride, const float* scale, const float* bias, const float* res, int res_stride,
                         int relu, void* out_h2_twin, void* stream);
/* R1, both render heads in ONE launch (csrc/mlp_h2.hip): the per-voxel table [V][4] = (sigma, r, g, b) of
 *   sigma = w_s1 . relu(W_s0 x + b_s0) + b_s1,   rgb = W_ro . relu(... relu(W_r0 x + b_r0) ...) + b_ro
 * (P/utils/nerf_mlp.py:14-105 with COOCC_Ray's hyper-parameters, coocc_ray.py:111-113: input_dim 128, net_width 256, skip_layer
 * None; evaluated at :583-590) on the split-f16 engine.  x_h2: H2 rows [V][128] (coocc_rows_to_h2 or a producer's twin);
 * ws0_pack / wr_packs_host[l]: H2 weight packs of the hidden Linear layers (the mfma_dtype 3 layout of coocc_conv_desc with
 * taps = 1); biases and the output layers' weights (ws1 [256], wr_out [3][256]) plain fp32.  n_rgb_hidden = 0: sigma head only
 * (the depth-only branch :436-484; rgb columns are written as 0).  activate != 0: rgb columns hold sigmoid(logit), as
 * coocc_render_activate_table would leave them.  Replaces six coocc_conv_fwd launches and their 330 MB of hidden activations. */
int coocc_render_heads_h2(const void* x_h2, int V, int Cin, int width, const void* ws0_pack, const float* bs0,
                          const float* ws1, const float* bs1, const void* const* wr_packs_host,
                          const float* const* br_host, int n_rgb_hidden, const float* wr_out, const float* br_out,
                          float* table, int activate, void* stream);
/* coocc_wino_pack_weights_dev for the split-f16 engine (mfma_dtype 3): U = G g G^T in fp64, split into f16 hi / lo, in the H2 pack
 * layout [(tile+2)^2][(K chunk, dz)][roundup(N,128)/32][2 k16 steps][hi | lo][64 lanes][8 f16].  The GEMM's K (Cin forward, Cout
 * dgrad) must be a multiple of 32.  packed == NULL: returns the size in 4-byte units.  Lets the TRAINING path (which re-packs from
 * the live parameter every step; the reference trains these layers through cuDNN, resnet3d.py:34-64) run its Winograd forward and
 * dgrad GEMMs on the f16 matrix cores. */
int64_t coocc_wino_pack_weights_h2_dev(const float* w, int Cout, int Cin, int tile, int dgrad, void* packed, void* stream);
/* coocc_conv_pack_weights_dev (same `mode`s: 0 forward, 1 tap-major Linear, 2 dgrad with flipped taps, 3 dgrad unflipped) for the
 * split-f16 engine: the direct-form H2 pack [(K chunk, tap)][roundup(N,128)/32][2 k16 steps][hi | lo][64 lanes][8 f16] made on the
 * device from the live parameter -- the strided / 1x1x1 / small-grid layers of the TRAINING path (resnet3d.py:34-64, fpn3d.py:70-106
 * under cuDNN upstream) on the f16 matrix cores.  K (Cin forward, Cout dgrad) % 32 == 0.  packed == NULL: size in 4-byte units. */
int64_t coocc_conv_pack_weights_h2_dev(const float* w, int Cout, int Cin, int taps, int mode, void* packed, void* stream);
/* Range guard of the split-f16 engine (resnet3d.py / fpn3d.py / occ_head.py convolutions are fp32 upstream and have no such
 * limit): every kernel that writes a 16-bit operand (H2 rows, f16 twins, the Winograd-domain V) raises a host-visible flag when a
 * value reaches half the f16 range (|v| >= 32768 after the writer's own scale; NaN counts).  Returns the flag (0 / 1) and clears it
 * when reset != 0; no synchronisation of its own -- call it after the stream(s) of interest have been synchronised. */
int coocc_h2_overflow(int reset);
/* Sticky device-fault word, same rules (no synchronisation of its own; cleared when reset != 0): kernels that meet an index no
 * valid caller can produce skip the access and store a COOCC_FAULT_* code here instead of faulting the GPU.  0 = none. */
#define COOCC_FAULT_SPARSE_ORDINAL 1 /* coocc_sparse_tap_sum: a voxel -> ordinal map entry >= p_rows (a corrupted / stale map) */
int coocc_device_fault(int reset);
/* Winograd weight packs made on the device (training re-packs every step): w:[Cout,Cin,3,3,3] -> (tile+2)^2 packs
 * U[p] = G g G^T in the layout coocc_conv_fwd reads with wgroup_rows (taps = 3, the z taps).  dgrad != 0: packs of the
 * transposed convolution (W'[c][n] = w[n][c] with all three tap axes flipped; GEMM N = Cin, K = Cout).
 * packed == NULL: returns the number of floats needed.  Returns that count, or a negative COOCC_E* code. */
int64_t coocc_wino_pack_weights_dev(const float* w, int Cout, int Cin, int tile, int dgrad, float* packed, void* stream);
/* Winograd-domain weight gradient of a 3x3x3 stride-1 pad-1 convolution (training; 4x / 2.25x fewer multiplies than
 * coocc_conv_wgrad).  coocc_wino_gradout: dM[p] = A dY A^T per (tile, z) row, same layout as coocc_wino_input's V
 * ([(tile+2)^2][group_rows][C]; rows past the valid ones must be zero in dM or V).  coocc_wino_ztap_table: the z-tap
 * row table [3][rows_total] (row + dz - 1, or -1 outside 0..Z-1), rows_total = (tile+2)^2 * group_rows.
 * coocc_wino_wgrad: dU[p][dz] = sum_rows V[p][row+dz-1]^T dM[p][row], dw[Cout,Cin,3,3,3] (=|+=) G^T dU G;
 * ws: >= (tile+2)^2 * 3 * Cin * Cout floats (more = more M slices in flight). */
int coocc_wino_gradout(const float* dy, int dy_stride, int B, int X, int Y, int Z, int C, int tile, float* dM,
                       int64_t group_rows, void* stream);
int coocc_wino_ztap_table(int64_t rows_total, int Z, int32_t* table, void* stream);
int coocc_wino_wgrad(const float* V, const float* dM, int64_t group_rows, int Z, int Cin, int Cout, int tile,
                     const int32_t* ztap_table, float* dw, int accumulate, float* ws, int64_t ws_floats,
                     void* stream);
/* Weight gradients on the split-f16 engine (csrc/wgrad_h2.hip; the reference's cuDNN fp32 backward of resnet3d.py:34-64,
 * fpn3d.py:70-106): both operands "k-major" (KH2: [rows_pad / 8][hi | lo][C][8 rows as f16], rows past `rows` zero, rows_pad % 16
 * == 0) so that a lane's 32x32x16 MFMA fragment -- 8 consecutive voxels of one channel -- is one 16-byte read.
 * coocc_rows_to_kh2: fp32 rows * scale (* *scale_dev when given: the device-chosen gradient scale of coocc_conv_epilogue_bwd_ex).
 * coocc_conv_wgrad_h2: dw[Cout][Cin] (=|+=) sum_m x[m][c] dy[m][n] * alpha (* *alpha_dev) -- the 1x1x1 / Linear layers.
 * coocc_wino_wgrad_h2: coocc_wino_wgrad from the KH2 forms of V and dM, Z in {2, 4, 8} (a group of 8 rows holds whole z columns:
 * the three z taps reuse one fragment, shifted by an f16 lane).  ws as for the fp32 entry points. */
int coocc_rows_to_kh2(const float* x, int stride, int64_t rows, int64_t rows_pad, int C, float scale, const float* scale_dev,
                      void* out_kh2, void* stream);
/* The two operands of coocc_wino_wgrad_h2 straight from the activations (which = 0: V, the transform of coocc_wino_input) / the
 * gradients (which = 1: dM, the transform of coocc_wino_gradout) in KH2 form -- no fp32 V / dM, no conversion pass; every row of
 * the group_rows per point is written (rows past the valid ones: zero). */
int coocc_wino_operand_kh2(int which, const float* x, int x_stride, int B, int X, int Y, int Z, int C, int tile, void* out_kh2,
                           int64_t group_rows, float scale, const float* scale_dev, void* stream);
int coocc_conv_wgrad_h2(const void* x_kh2, const void* dy_kh2, int64_t rows_pad, int Cin, int Cout, float alpha,
                        const float* alpha_dev, float* dw, int accumulate, float* ws, int64_t ws_floats, void* stream);
int coocc_wino_wgrad_h2(const void* V_kh2, const void* dM_kh2, int64_t group_rows, int Z, int Cin, int Cout, int tile,
                        float alpha, const float* alpha_dev, float* dw, int accumulate, float* ws, int64_t ws_floats,
                        void* stream);

/* ---------------------------------------------------------------- backward of the conv family
 * Frozen-statistics BN (scale/shift constants), as the forward.  torch.autograd computes these through
 * cudnn/MIOpen conv backward upstream (the reference has no kernel of its own for them). */
/* Device-side packing into the layout coocc_conv_fwd consumes.  mode 0: w[Cout][Cin][taps]; 1: w[Cout][taps][Cin];
 * 2: data-gradient pack with taps flipped (stride-1 convs: run coocc_conv_fwd on dacc with pad' = k-1-pad);
 * 3: data-gradient pack, taps unflipped (use with a dgrad row table).  packed == NULL returns the float count. */
int64_t coocc_conv_pack_weights_dev(const float* w, int Cout, int Cin, int taps, int mode, float* packed,
                                    void* stream);
/* Row tables of a conv geometry: dgrad == 0: [taps][Mo] input row read by output o for tap t (-1 = padding);
 * dgrad != 0: [taps][Mi] output row o with o*stride - pad + t == i (-1 = none). */
int coocc_conv_tap_table(int B, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo, int ksize, int stride, int pad,
                         int dgrad, int32_t* table, void* stream);
/* dpre = dout * (out > 0 if relu);  dres (+)= dpre;  dacc = dpre * scale;  dbias (+)= column sums of dpre
 * (deterministic two-pass; ws >= ceil(M/256)*C floats).  Any of dacc / dres / dbias may be NULL. */
int coocc_conv_epilogue_bwd(const float* dout, int dout_stride, const float* out, int out_stride,
                            const float* scale, int M, int C, int relu, float* dacc, int dacc_stride,
                            float* dres, int dres_stride, int dres_accumulate, float* dbias,
                            int dbias_accumulate, float* ws, int64_t ws_floats, void* stream);
/* ... and the operand scale of the split-f16 engine for dacc, chosen on the device: amax_word (COOCC_AMAX_WORDS zeroed 32-bit
 * words, left zero: the workgroups spread their atomics over 64 of them, 128 bytes apart) collects max |dacc| during the pass, then scale2[0] = the power of two that brings it into [target / 2, target) (1 when the
 * gradient is all zero), scale2[1] = its inverse.  The gradient operand of a dgrad / wgrad GEMM is written as H2 rows with
 * scale_dev = scale2 (coocc_rows_to_h2_ex, coocc_wino_input_h2_ex) and the GEMM undoes it with coocc_conv_desc.alpha_dev =
 * scale2 + 1: whatever the loss scale, the f16 halves see values of magnitude <= target (an f16 is subnormal below 6.1e-5; the
 * reference's cuDNN backward has no such concern, resnet3d.py:34-64).  amax_word / scale2 NULL: coocc_conv_epilogue_bwd. */
#define COOCC_AMAX_WORDS 2048
int coocc_conv_epilogue_bwd_ex(const float* dout, int dout_stride, const float* out, int out_stride,
                               const float* scale, int M, int C, int relu, float* dacc, int dacc_stride,
                               float* dres, int dres_stride, int dres_accumulate, float* dbias,
                               int dbias_accumulate, float* ws, int64_t ws_floats, uint32_t* amax_word, float* scale2,
                               float target, void* stream);
/* dw[Cout][Cin][taps] (+)= sum_m in[table[t][m]][c] * dacc[m][n]  (table NULL: identity rows, taps == 1).
 * in has in_rows rows; fp32 MFMA with the voxel index as K; M-slices reduced in slice order (deterministic). */
int coocc_conv_wgrad(const float* in, int in_rows, int in_stride, const float* dacc, int dacc_stride,
                     const int32_t* table, int M, int Cin, int Cout, int taps, float* dw, int accumulate,
                     float* ws, int64_t ws_floats, void* stream);

/* ---- backward of the HBM-bound ops (torch autograd over eager index_put / cumprod / interpolate upstream) */
/* dst[r] = src[idx[r]] (zeros for idx < 0) and its adjoint dst[idx[r]] += src[r] (fp32 atomics): the G1 row
 * gather of bifuser_n.py:138-169 and its feature gradient. */
int coocc_gather_rows(const float* src, int src_stride, const int32_t* idx, int n, int C, float* dst,
                      int dst_stride, void* stream);
int coocc_scatter_add_rows(const float* src, int src_stride, const int32_t* idx, int n, int C, float* dst,
                           int dst_stride, void* stream);
/* voxel_pooling backward (bev_pool_grad_kernel, bev_pool_cuda.cu:61-84, without the sort):
 * dx[p] = dout[voxel(p)] for kept points, 0 otherwise.  dout: NDHWC rows; dx:[npts,C]. */
int coocc_voxel_pool_bwd(const float* dout, int dout_stride, const float* geom, int npts, int pts_per_batch,
                         int C, const float* lo_dx_host, int B, int X, int Y, int Z, float* dx, void* stream);
/* coocc_lift_splat backward: d_depth[N,D,H,W] and d_feat_nhwc[N,H,W,C] in one deterministic pass. */
int coocc_lift_splat_bwd(const float* dout, int dout_stride, const float* depth, const float* feat_nhwc,
                         const float* geom, int N, int D, int H, int W, int C, int pts_per_batch,
                         const float* lo_dx_host, int B, int X, int Y, int Z, float* d_depth,
                         float* d_feat_nhwc, void* stream);
/* coocc_render_nearest backward: dmaps:[N,H,W,4] (d rgb, d depth) -> dtable:[X*Y*Z,4] (zeroed here, then
 * accumulated with fp32 atomics).  Same arguments as the forward. */
int coocc_render_nearest_bwd(const float* table, int X, int Y, int Z, const float* geom, const float* zvals,
                             int N, int D, int H, int W, const float* bounds_host, const float* dmaps,
                             float* dtable, void* stream);
/* coocc_upsample_maps backward (bilinear adjoint, deterministic gather): drgbs / ddepths may be NULL. */
int coocc_upsample_maps_bwd(const float* drgbs, const float* ddepths, int N, int H, int W, int scale,
                            float* dmaps, void* stream);
/* coocc_render_losses backward: losses_out = the forward's out[3]; gl:[2] (device) upstream gradients. */
int coocc_render_losses_bwd(const float* rgbs, const float* depths, const float* rgb_gt, const float* depth_gt,
                            int64_t npix, int D, const float* losses_out, const float* gl, float* drgbs,
                            float* ddepths, void* stream);

/* coocc_upsample_add_trilinear backward w.r.t. the coarse volume (the fine gradient passes through):
 * dcoarse (+)= adjoint-trilinear(dfine).  Upsampling factors up to 4 per axis. */
int coocc_upsample_trilinear_bwd(const float* dfine, float* dcoarse, int B, int C, int Xc, int Yc, int Zc, int Xf,
                                 int Yf, int Zf, int accumulate, void* stream);

/* coocc_occhead_mix backward: dout:[V0,C].  glevels_host[l]: scratch rows [V0,C] receiving softmax(w)_l * dout (level 0:
 * this IS d level_0; levels >= 1 are pulled down with coocc_upsample_trilinear_bwd); dwlogit:[V0,L] (may be NULL). */
int coocc_occhead_mix_bwd(const float* const* levels_host, const int* dims_host, int L, const float* wlogit,
                          const float* dout, float* const* glevels_host, float* dwlogit, int B, int C,
                          void* stream);

/* coocc_fine_sample_voxel backward: dvol:[X*Y*Z,C] (zeroed here) += trilinear weights * dfeat rows (fp32 atomics). */
int coocc_fine_sample_voxel_bwd(const float* dfeat, int dfeat_stride, int C, int X, int Y, int Z,
                                const int64_t* fine_xyz, int64_t nfine, const int* final_size_host, float* dvol,
                                void* stream);
/* coocc_groupnorm_rows backward.  x: rows BEFORE the (in-place) forward, y: rows after it, dy: upstream gradient, all
 * [n, stride] with C normalised channels; dx same layout; dgamma/dbeta:[C] (zeroed here; may be NULL). */
int coocc_groupnorm_rows_bwd(const float* x, const float* y, const float* dy, int64_t n, int C, int stride, int groups,
                             const float* gamma, float eps, int relu, float* dx, float* dgamma, float* dbeta,
                             void* stream);

/* coocc_fine_sample_img backward: dimg:[ncam,Hf,Wf,Ci] (zeroed here) += bilinear weights * mask * dfeat rows. */
int coocc_fine_sample_img_bwd(const float* dfeat, int dfeat_stride, int ncam, int Ci, int Hf, int Wf,
                              const float* params, const int64_t* fine_xyz, int64_t nfine, float* dimg,
                              void* stream);
/* coocc_groupnorm_nhwc backward (x before / y after the in-place forward, [N,HW,C]); dgamma/dbeta:[C] zeroed here. */
int coocc_groupnorm_nhwc_bwd(const float* x, const float* y, const float* dy, int N, int HW, int C, int groups,
                             const float* gamma, float eps, int relu, float* dx, float* dgamma, float* dbeta,
                             void* stream);

/* BatchNorm3d / SyncBN in training mode on channels-last rows [M,C] (the reference trains with batch statistics):
 * per-channel mean and biased variance (deterministic two-pass, fp64 partials; ws >= 16*ceil(M/256)*C bytes),
 * y = relu((x - mean) * rsqrt(var + eps) * gamma + beta (+ res)), and the backward (dx, dres = dy*[y>0], dgamma, dbeta). */
int coocc_bn_stats(const float* x, int stride, int M, int C, float* mean, float* var, void* ws, size_t ws_bytes,
                   void* stream);
int coocc_bn_apply(const float* x, int M, int C, const float* mean, const float* var, const float* gamma,
                   const float* beta, float eps, const float* res, int relu, float* y, void* stream);
int coocc_bn_backward(const float* x, const float* y, const float* dy, int M, int C, const float* mean, const float* var,
                      const float* gamma, float eps, int relu, float* dx, float* dres, float* dgamma, float* dbeta,
                      void* ws, size_t ws_bytes, void* stream);
/* The same backward in two halves for SyncBN (torch.nn.SyncBatchNorm semantics): `sums` gives this rank's
 * dgamma = sum dy'*xhat and dbeta = sum dy'; the caller all-reduces them; `dx` uses the reduced sums and the
 * cross-rank row count.  mean / var are then the cross-rank batch statistics. */
int coocc_bn_backward_sums(const float* x, const float* y, const float* dy, int M, int C, const float* mean,
                           const float* var, float eps, int relu, float* dgamma, float* dbeta, void* ws,
                           size_t ws_bytes, void* stream);
int coocc_bn_backward_dx(const float* x, const float* y, const float* dy, int M, int C, const float* mean,
                         const float* var, const float* gamma, float eps, int relu, const float* sum_dgamma,
                         const float* sum_dbeta, double count, float* dx, float* dres, void* stream);

/* FPN3D top-down step (fpn3d.py:88-92): fine += trilinear(coarse -> fine size),
 * align_corners=False.  Rows NDHWC with C channels. */
int coocc_upsample_add_trilinear(const float* coarse, float* fine, int B, int C, int Xc, int Yc,
                                 int Zc, int Xf, int Yf, int Zf, void* stream);
/* ... and, with fine_h2_twin != NULL (C % 32 == 0), an H2 copy of the updated fine rows (the fpn_conv that reads them next) */
int coocc_upsample_add_trilinear_ex(const float* coarse, float* fine, int B, int C, int Xc, int Yc,
                                    int Zc, int Xf, int Yf, int Zf, void* fine_h2_twin, void* stream);

/* OccHead.forward_coarse_voxel mix (occ_head.py:155-166): out = sum_l softmax(wlogit)[l] *
 * trilinear(level_l -> level-0 size).  levels: up to 4 NDHWC volumes. */
int coocc_occhead_mix(const float* const* levels_host, const int* dims_host /*[L][3]*/, int L,
                      const float* wlogit /*[V0,L]*/, float* out, int B, int C, void* stream);
/* ... and, with out_h2_twin != NULL (C % 32 == 0), an H2 copy of `out` (occ_pred_conv's first layer reads it next) */
int coocc_occhead_mix_ex(const float* const* levels_host, const int* dims_host /*[L][3]*/, int L,
                         const float* wlogit /*[V0,L]*/, float* out, int B, int C, void* out_h2_twin, void* stream);

/* ---------------------------------------------------------------- C4 fine branch */
/* coarse_occ.argmax(1) != empty_idx over [V] rows of `stride` floats (occ_head.py:182) */
int coocc_argmax_flags(const float* logits, int V, int ncls, int stride, int empty_idx,
                       uint8_t* flags, void* stream);
/* coarse_to_fine_coordinates (P/utils/coordinate_transform.py:3-21, eval branch) + trilinear
 * grid_sample(align_corners=False, zeros) of out_voxel_feats (occ_head.py:205-214), B == 1.
 * coarse_lin:[n] voxel rows; fine_xyz:[3, r^3*n] i64 (offset-major); feat rows
 * [r^3*n, out_stride] (first C columns written). */
int coocc_fine_sample_voxel(const float* vol, int C, int X, int Y, int Z, const int32_t* coarse_lin,
                            int n, int ratio, const int* final_size_host, int64_t* fine_xyz,
                            float* feat, int out_stride, void* stream);
/* project_points_on_img + per-camera bilinear grid_sample(align_corners=True, zeros) * mask,
 * summed over cameras (coordinate_transform.py:25-65, occ_head.py:217-234).  img_nhwc:
 * [ncam,Hf,Wf,Ci]; params (device): [0:9] inv(bda), [9:12] voxel_size, [12:15] range_lo,
 * [15] W_img-1, [16] H_img-1, then per camera 27 floats: inv(rots)[9], trans[3], intrins[9],
 * post_rots[:2,:2][4], post_trans[:2][2].  group = R in {2, 4} (1 is read as 2): fine_xyz is the offset-major list of
 * coocc_fine_sample_voxel with ratio R (f = o*n + i, child o = (a*R+b)*R+c at (child 0) + (a,b,c), R^3 children per
 * coarse voxel): one wave per coarse voxel, child coordinates derived from child 0.  Other values: point by point. */
int coocc_fine_sample_img(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf,
                          const float* params, const int64_t* fine_xyz, int64_t nfine, float* feat,
                          int out_stride, int group, void* stream);
/* Builds the `params` block of coocc_fine_sample_img on the device (3x3 inverses included, no host sync):
 * rots/intrins/post_rots:[ncam,3,3], trans/post_trans:[ncam,3], bda:[3,3] (device); hdr_host:[8] = voxel_size(3),
 * range_lo(3), W_img-1, H_img-1; params:[17 + 27*ncam] (device). */
int coocc_projection_params(const float* rots, const float* trans, const float* intrins, const float* post_rots,
                            const float* post_trans, const float* bda, int ncam, const float* hdr_host,
                            float* params, void* stream);
/* The fine-branch MLP chain of OccHead in one launch (occ_head.py:70-83 modules, 224-233 use):
 *   y1 = ReLU(GN16(samp . w_img^T + b_img));  h = ReLU(GN16(cat[vox, y1] . w_f0^T + b_f0));  out = h . w_f3^T + b_f3
 * samp:[nfine, >=128] image samples, vox:[nfine, >=128] voxel samples (row strides in floats, multiples of 4),
 * w_img:[64,128], w_f0:[64,192], w_f3:[ncls,64] (nn.Linear layout), GroupNorm(16, 64) affine arrays [64],
 * out:[nfine, ncls], ncls <= 32.  Same bits as coocc_conv_fwd (taps = 1) + coocc_groupnorm_rows applied in turn. */
int coocc_fine_mlp(const float* samp, int samp_stride, const float* vox, int vox_stride, int64_t nfine,
                   const float* w_img, const float* b_img, const float* gn_img_w, const float* gn_img_b,
                   float eps_img, const float* w_f0, const float* b_f0, const float* gn_f0_w,
                   const float* gn_f0_b, float eps_f0, const float* w_f3, const float* b_f3, int ncls,
                   float* out, void* stream);
/* Same chain with the two Linear layers that precede a resampling applied BEFORE it (a Linear commutes with the bi- /
 * trilinear interpolation; biases stay after it): samp64:[nfine, >=64] = coocc_fine_sample_img of (img features . w_img^T),
 * vox64:[nfine, >=64] = coocc_fine_sample_voxel of (voxel features . w_f0[:, :128]^T).  Then
 *   y1 = ReLU(GN16(samp64 + b_img));  h = ReLU(GN16(vox64 + y1 . w_f0[:, 128:192]^T + b_f0));  out = h . w_f3^T + b_f3.
 * w_f0 is still the whole [64,192] matrix (only its last 64 columns are read).  Half the sampling traffic and a quarter of
 * the per-point multiplies of coocc_fine_mlp; results differ from it by fp32 rounding only. */
int coocc_fine_mlp_pre(const float* samp64, int samp_stride, const float* vox64, int vox_stride, int64_t nfine,
                       const float* b_img, const float* gn_img_w, const float* gn_img_b, float eps_img,
                       const float* w_f0, const float* b_f0, const float* gn_f0_w, const float* gn_f0_b,
                       float eps_f0, const float* w_f3, const float* b_f3, int ncls, float* out, void* stream);
/* Device-count forms of the fine-branch entry points (see coocc_voxel_index_map_dev): n_dev = number of foreground COARSE
 * voxels (the count coocc_compact_flags leaves on the device), capacities sized for the worst case (every voxel foreground).
 * The packed layouts (fine point f = o*n + i, fine_xyz planes of n*r^3 entries) follow the ACTUAL count, exactly as the
 * host-count forms; n_mul = r^3 converts the coarse count to fine points. */
int coocc_fine_sample_voxel_dev(const float* vol, int C, int X, int Y, int Z, const int32_t* coarse_lin, int n_cap,
                                const int32_t* n_dev, int ratio, const int* final_size_host, int64_t* fine_xyz,
                                float* feat, int out_stride, void* stream);
int coocc_fine_sample_img_dev(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params,
                              const int64_t* fine_xyz, int64_t nfine_cap, const int32_t* n_dev, float* feat,
                              int out_stride, int group, void* stream);
int coocc_fine_mlp_pre_dev(const float* samp64, int samp_stride, const float* vox64, int vox_stride, int64_t nfine_cap,
                           const int32_t* n_dev, int n_mul, const float* b_img, const float* gn_img_w,
                           const float* gn_img_b, float eps_img, const float* w_f0, const float* b_f0,
                           const float* gn_f0_w, const float* gn_f0_b, float eps_f0, const float* w_f3, const float* b_f3,
                           int ncls, float* out, void* stream);
int coocc_scatter_fine_dev(const float* fine_logits, int64_t nfine_cap, const int32_t* n_dev, int n_mul, int ncls, int stride,
                           const int64_t* fine_xyz, float* grid, int Xf, int Yf, int Zf, float empty_val, void* stream);
/* pred_f for the head's own fine points -- the R^3 children of the coarse voxels coarse_lin[0..n) (row f = o n + i, offset
 * o = (a R + b) R + c; final grid = R x coarse grid) -- in ONE output-major pass: no fill, no scattered stores.  Same grid as
 * coocc_scatter_fine on the coordinates coocc_fine_sample_voxel produced for that list.  n_dev != NULL: n = min(n_cap, *n_dev) read
 * on the device (hipGraph form).  map_ws: Xc*Yc*Zc int32 of scratch -- or, with coarse_lin == NULL, the finished voxel -> ordinal
 * table (coocc_compact_flags_ex), which is then used as it is. */
int coocc_scatter_fine_grouped(const float* fine_logits, int ncls, int stride, const int32_t* coarse_lin, int n_cap,
                               const int32_t* n_dev, int R, int Xc, int Yc, int Zc, float* grid, float empty_val,
                               int32_t* map_ws, void* stream);
/* nn.GroupNorm on 2-D rows [n,C] (+ReLU), in place (occ_head.py:70-83) */
int coocc_groupnorm_rows(float* x, int64_t n, int C, int stride, int groups, const float* gamma,
                         const float* beta, float eps, int relu, void* stream);
/* nn.GroupNorm on an NHWC image batch [N,HW,C] (+ReLU), in place (occ_head.py:64-68) */
int coocc_groupnorm_nhwc(float* x, int N, int HW, int C, int groups, const float* gamma,
                         const float* beta, float eps, int relu, void* stream);
/* dense fine grid of simple_test (P/coocc/detectors/coocc_ray.py:546-550):
 * grid [ncls,Xf,Yf,Zf] = empty_val, then fine logits scattered at fine_xyz. */
int coocc_scatter_fine(const float* fine_logits, int64_t nfine, int ncls, int stride,
                       const int64_t* fine_xyz, float* grid, int Xf, int Yf, int Zf, float empty_val,
                       void* stream);

/* ---------------------------------------------------------------- P1, P2 */
/* get_geometry (P/coocc/image2bev/ViewTransformerLSSBEVDepth.py:117-150).
 * and the detector's module-level get_frustum (P/coocc/detectors/coocc_ray.py:732-776; same chain, scale instead of
 * downsample).  mats:[B*N][COOCC_CAM_FLOATS] = inv(post_rots)(9), post_trans(3), rots@inv(intrins[:3,:3])(9), trans(3),
 * bda[:3,:3](9), intrins[:3,3] of a KITTI 3x4/4x4 intrinsic else 0 (3), bda[:3,3] of a 4x4 bda else 0 (3);
 * xs:[fW], ys:[fH], ds:[D] frustum axes of create_frustum (:104-115).  geom:[B*N,D,fH,fW,3]. */
#define COOCC_CAM_FLOATS 39
/* the constants above from the raw calibration tensors (rots/post_rots [B,N,3,3], trans/post_trans [B,N,3], intrins
 * [B,N,k,k] k = intrin_dim 3|4, bda [B,b,b] b = bda_dim 3|4), 3x3 inverses in fp64: one launch, no host synchronisation. */
int coocc_camera_mats(const float* rots, const float* trans, const float* intrins, const float* post_rots,
                      const float* post_trans, const float* bda, int B, int N, int intrin_dim, int bda_dim,
                      float* mats, void* stream);
int coocc_get_geometry(const float* mats, const float* xs, const float* ys, const float* ds, int BN,
                       int D, int fH, int fW, float* geom, void* stream);

/* [EXT] bev_pool_forward (M/ops/bev_pool/src/bev_pool.cpp:22-47; kernel
 * bev_pool_cuda.cu:20-42).  x:[n,c] rank-sorted rows, geom:[n,4] (x,y,z,b) i32,
 * out:[b,d,h,w,c] (zeroed here). */
int coocc_bev_pool_forward(const float* x, const int32_t* geom, const int32_t* interval_lengths,
                           const int32_t* interval_starts, int b, int d, int h, int w, int n, int c,
                           int n_intervals, float* out, void* stream);
/* [EXT] bev_pool_backward (bev_pool.cpp:60-87; kernel bev_pool_cuda.cu:61-84) */
int coocc_bev_pool_backward(const float* out_grad, const int32_t* geom,
                            const int32_t* interval_lengths, const int32_t* interval_starts, int b,
                            int d, int h, int w, int n, int c, int n_intervals, float* x_grad,
                            void* stream);
/* voxel_pooling (P/coocc/image2bev/ViewTransformerLSSVoxel.py:100-123) without argsort and without sorting the points:
 * FOUR launches, none of which waits for another workgroup -- (1) quantise (truncate, then range filter) + per-voxel histogram
 * with per-point slots, (2) scan inside 1024-voxel chunks, (3) CSR fill (every workgroup re-scans the few chunk totals) + global
 * starts + histogram zeroed again, (4) each voxel sums its rows in ascending point id.  x:[npts,C]; geom:[npts,3];
 * lo_dx_host = {bx-dx/2 (3), dx (3)}; out: NDHWC rows [B*X*Y*Z, out_stride].  ws >= coocc_voxel_pool_ws(npts, B*X*Y*Z).
 * ws_clean (all pooling entry points): non-zero = the caller vouches that the LAST write to `ws` was a pooling call with the
 * same (npts, nvox) that returned COOCC_OK -- such a call leaves its histogram zeroed, so the memset (a fifth launch) is
 * skipped; pass 0 for a fresh / reused-for-something-else / differently-sized workspace. */
size_t coocc_voxel_pool_ws(int npts, int nvox);
int coocc_voxel_pool(const float* x, const float* geom, int npts, int pts_per_batch, int C,
                     const float* lo_dx_host, int B, int X, int Y, int Z, float* out, int out_stride,
                     void* ws, size_t ws_bytes, int ws_clean, void* stream);
/* Fused lift (x) splat (ViewTransformerLSSVoxel.py:135-143 + voxel_pooling): the lifted
 * volume depth_prob[n,d,h,w] * img_feat[n,c,h,w] is never materialised.  depth:[N,D,H,W]; feat_nhwc:
 * [N,H,W,C]; geom:[N*D*H*W,3]; out NDHWC rows; same workspace as coocc_voxel_pool(N*D*H*W, B*X*Y*Z).
 * Bit-equal to pooling the materialised volume (products rounded before the add, ascending point id). */
int coocc_lift_splat(const float* depth, const float* feat_nhwc, const float* geom, int N, int D, int H, int W,
                     int C, int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                     int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream);
/* Same, with the geometry of coocc_get_geometry computed inside the key kernel (no [npts,3] tensor in HBM). */
int coocc_lift_splat_cams(const float* depth, const float* feat_nhwc, const float* mats, const float* xs,
                          const float* ys, const float* ds, int N, int D, int H, int W, int C,
                          int pts_per_batch, const float* lo_dx_host, int B, int X, int Y, int Z, float* out,
                          int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream);
/* The per-voxel sums alone, over the CSR binning a previous coocc_lift_splat[_cams] call left in `ws` (same shapes, same
 * geometry): what a fixed camera rig needs per frame -- one launch, bit-equal to the full call.  The reference has the idea as
 * voxel_pooling_accelerated (ViewTransformerLSSBEVDepth.py:242-300: geometry / sort cached on the first call; that variant also
 * caps a voxel at 300 points, this one keeps every point). */
int coocc_lift_splat_reuse(const float* depth, const float* feat_nhwc, int N, int D, int H, int W, int C, int B, int X,
                           int Y, int Z, float* out, int out_stride, void* ws, size_t ws_bytes, void* stream);
/* bev_pool(feats, coords, ...) drop-in (M/ops/bev_pool/bev_pool.py:83-97): coords:[n,4]
 * (x,y,z,b) i64; same sort-and-sum, out NDHWC rows. */
int coocc_bev_pool_coords(const float* x, const int64_t* coords, int n, int C, int B, int X, int Y,
                          int Z, float* out, int out_stride, void* ws, size_t ws_bytes, int ws_clean, void* stream);

/* OccHead fine branch in ONE launch (occ_head.py:205-233, eval, B == 1): fine coordinates + trilinear resampling of Q + bilinear
 * resampling of P over the cameras that see each point + GroupNorm / ReLU / Linear chain -> logits, for cascade ratio 2 | 4.
 * Q: [X*Y*Z, 64] = W_f0[:, :128] . voxel features, P: [ncam*Hf*Wf, 64] = W_img . image features (both Linear layers applied
 * BEFORE the resampling they commute with, as for coocc_fine_mlp_pre); coarse_lin: the foreground coarse voxels (n_dev != NULL:
 * at most n_cap of them, count read on the device); final_size_host = ratio * (X, Y, Z).  Outputs as coocc_fine_sample_voxel +
 * coocc_fine_mlp_pre (fine_xyz [3][n ratio^3], logits [n ratio^3, ncls], row o * n + i), bit-identical to that three-kernel
 * path, without its two [n ratio^3, 64] intermediates in HBM. */
int coocc_fine_fused(const float* Q, int X, int Y, int Z, const float* P, int ncam, int Hf, int Wf, const float* params,
                     const int32_t* coarse_lin, int n_cap, const int32_t* n_dev, int ratio, const int* final_size_host,
                     const float* b_img, const float* gn_img_w, const float* gn_img_b, float eps_img, const float* w_f0,
                     const float* b_f0, const float* gn_f0_w, const float* gn_f0_b, float eps_f0, const float* w_f3,
                     const float* b_f3, int ncls, int64_t* fine_xyz, float* out, void* stream);

/* The same branch for cascade ratio 2 on the split-f16 engine, ONE launch with lanes = fine points (csrc/fine2_h2.hip): samples
 * in registers (same resampling arithmetic as coocc_fine_fused), img_mlp GroupNorm + ReLU in place, fine_mlp[0] / fine_mlp[3]
 * as v_mfma_f32_32x32x16_f16 on f16 hi / lo operand halves with fp32 accumulation (fp32-accurate, not bit-identical to the
 * fp32-MFMA chain).  Q rows are q_stride floats apart (64, or 128 when Q is the upper half of a merged head GEMM).  coocc_fine2_pack: w_f0 [64][192], w_f3 [ncls][64] and the bias / GroupNorm vectors -> wpack (24576 bytes)
 * + consts (416 floats), once per weight version.  Replaces occ_head.py:205-233 for cascade_ratio == 2. */
int coocc_fine2_pack(const float* w_f0, const float* w_f3, int ncls, const float* b_img, const float* gn_img_w,
                     const float* gn_img_b, const float* b_f0, const float* gn_f0_w, const float* gn_f0_b, const float* b_f3,
                     void* wpack, float* consts, void* stream);
/* img_samples != NULL: the image samples [8 n, 64] (row f = o n + i) come from coocc_fine_sample_img_lin's own launch and the
 * kernel runs the voxel resampling + the MLP chain only (P / params may then be NULL).  That is the default composition: with
 * split-f16 GEMMs of ANOTHER stream on the chip the in-kernel image samples differ from run to run. */
int coocc_fine2_h2(const float* Q, int q_stride, int X, int Y, int Z, const float* P, int ncam, int Hf, int Wf, const float* params,
                   const int32_t* coarse_lin, int n_cap, const int32_t* n_dev, const int* final_size_host, const void* wpack,
                   const float* consts, float eps_img, float eps_f0, int ncls, int64_t* fine_xyz, float* out,
                   const float* img_samples, void* stream);
/* The grouped image sampler of the fine branch (project_points_on_img + bilinear samples summed over the seeing cameras,
 * occ_head.py:211-224, coordinate_transform.py:25-65) straight from the foreground list: child 0 of coarse voxel (x, y, z) is
 * fine voxel ratio x (x, y, z) (B == 1, final grid = ratio x coarse grid), so the fine coordinates need not exist yet.
 * feat: [ratio^3 n, Ci] at out_stride floats per row, row f = o n + i; n_dev: optional device-side count (n_cap = capacity). */
int coocc_fine_sample_img_lin(const float* img_nhwc, int ncam, int Ci, int Hf, int Wf, const float* params,
                              const int32_t* coarse_lin, int Yc, int Zc, int n_cap, const int32_t* n_dev, float* feat,
                              int out_stride, int ratio, void* stream);

/* ---------------------------------------------------------------- R1..R3, L1 */
/* inline render block, one launch for all cameras (P/coocc/detectors/coocc_ray.py:575-616).
 * table:[X*Y*Z,4] = raw (sigma, r, g, b) head outputs per voxel -- the heads are pointwise,
 * so per-voxel evaluation equals the reference's per-sample evaluation; geom:[N,D,H,W,3];
 * zvals:[D] = linspace(0,D,D) (:614); bounds_host = xbound,ybound,zbound (lo,hi,step) of
 * :577.  maps:[N,H,W,4] = (r,g,b,depth) before upsampling.  activated != 0: columns 1..3 of the table
 * already hold sigmoid(rgb) (coocc_render_activate_table). */
int coocc_render_nearest(const float* table, int X, int Y, int Z, const float* geom,
                         const float* zvals, int N, int D, int H, int W, const float* bounds_host,
                         int activated, float* maps, void* stream);
/* The same composite with the sample positions evaluated IN the kernel from the camera constants (coocc_camera_mats) and the
 * frustum axes -- get_geometry's chain (ViewTransformerLSSBEVDepth.py:117-150), bit for bit -- instead of read from the
 * [N,D,H,W,3] geometry tensor: 12 bytes per sample less HBM traffic ("if geometry is computed in-kernel the geom
 * term is dropped"). */
int coocc_render_nearest_cams(const float* table, int X, int Y, int Z, const float* mats, const float* xs, const float* ys,
                              const float* ds, const float* zvals, int N, int D, int H, int W,
                              const float* bounds_host, int activated, float* maps, void* stream);
/* sigmoid of the rgb logits once per voxel, in place on columns 1..3 of table:[V,4] (3 exp + 3 rcp per voxel
 * instead of per ray sample). */
int coocc_render_activate_table(float* table, int V, void* stream);
/* x`scale` bilinear upsample, align_corners=False (coocc_ray.py:617-622):
 * maps [N,H,W,4] -> rgbs [N,sH,sW,3], depths [N,sH,sW] */
int coocc_upsample_maps(const float* maps, int N, int H, int W, int scale, float* rgbs,
                        float* depths, void* stream);
/* volume_sampling (P/utils/render_ray.py:28-48): trilinear, align_corners=True, border.
 * vol: channels-last rows [d0*d1*d2, C] of the reference's [1,C,d0,d1,d2] volume;
 * pts:[n,3]; aabb_host:[6] = min xyz, max xyz; feat:[n,C]; mask:[n] u8 (may be NULL). */
int coocc_volume_sampling(const float* vol, int C, int d0, int d1, int d2, const float* pts, int n,
                          const float* aabb_host, float* feat, uint8_t* mask, void* stream);
/* raw2outputs (render_ray.py:198-249) == COOCC_Ray.get_weights (coocc_ray.py:199-213) for
 * the weights: raw:[R,S,4] (rgb, sigma), z:[R,S] -> rgb:[R,3], depth:[R], weights:[R,S]
 * (may be NULL); zmin/zmax = z_vals.min()/max(). */
int coocc_raw2outputs(const float* raw, const float* z, int R, int S, int white_bkgd, float zmin,
                      float zmax, float* rgb, float* depth, float* weights, void* stream);
/* render losses (coocc_ray.py:423-433): out[0]=loss_depth_render, out[1]=loss_rgb, out[2]=number of
 * foreground pixels (kept for the backward).  rgbs/rgb_gt:[npix,3]; depths/depth_gt:[npix]; ws: >= 24 KB of device
 * scratch (fp64 partials of the deterministic two-pass reduction). */
int coocc_render_losses(const float* rgbs, const float* depths, const float* rgb_gt,
                        const float* depth_gt, int64_t npix, int D, float* out, void* ws, size_t ws_bytes,
                        void* stream);

/* On-device evaluation: COOCC_Ray.evaluation_semantic coocc_ray.py:659-684 + fast_hist
 * :726-730 without the device->host copy.  pred: class logits on an h x w x d grid addressed by element strides
 * (so both the channels-last pred_c rows and the NCDHW pred_f grid work); gt:[H,W,D] u8 labels (255 = noise);
 * visible:[H,W,D] u8 or NULL.  hist (int64, device) = [ SC 2x2 | SSC CxC | OCC CxC ], each indexed
 * [label][pred]; SC bins are (label != empty_idx, pred != empty_idx).  accumulate != 0 adds to hist instead
 * of overwriting it (whole-dataset accumulation with one read-back).  C <= 32. */
int coocc_eval_semantic(const float* pred, int64_t stride_c, int64_t stride_x, int64_t stride_y,
                        int64_t stride_z, int C, int h, int w, int d, const uint8_t* gt,
                        const uint8_t* visible, int H, int W, int D, int empty_idx, int accumulate,
                        int64_t* hist, void* stream);
/* Prediction labels of the dump formats: F.interpolate(pred, size=[H,W,D], trilinear, align_corners=False) +
 * argmax(dim=1) (P/coocc/apis/test.py:67-68,198-201), narrowed to u8 as save_output_nuscenes does
 * (P/coocc/apis/utils.py:65).  pred addressed by element strides as in coocc_eval_semantic; labels:[H,W,D] u8. */
int coocc_predict_labels(const float* pred, int64_t stride_c, int64_t stride_x, int64_t stride_y,
                         int64_t stride_z, int C, int h, int w, int d, int H, int W, int D, uint8_t* labels,
                         void* stream);

/* ---------------------------------------------------------------- LiDAR producer */
/* Hard voxelisation (mmdet3d/ops/voxel/src/voxelization_cpu.cpp:44-104 = the deterministic CUDA path of
 * voxelization_cuda.cu): points:[n,F] (xyz first); range_host:[6] xyzxyz min/max; voxel_size_host:[3].
 * voxels:[max_voxels,max_points,F] (zero padded), coors:[max_voxels,3] (z,y,x), num_points:[max_voxels], count:[1]
 * (device) = number of voxels produced.  Voxels are numbered in order of first appearance, each keeps its first
 * max_points points; voxels first seen after max_voxels are dropped. */
size_t coocc_voxelize_ws(int n);
int coocc_voxelize_hard(const float* points, int n, int F, const float* range_host, const float* voxel_size_host,
                        int max_points, int max_voxels, float* voxels, int32_t* coors, int32_t* num_points,
                        int32_t* count, void* ws, size_t ws_bytes, void* stream);
/* HardSimpleVFE (mmdet3d/models/voxel_encoders/voxel_encoder.py:43-45): mean of the first nf features -> [M,out_stride] */
int coocc_vfe_mean(const float* voxels, const int32_t* num_points, int M, int max_points, int F, int nf, float* out,
                   int out_stride, void* stream);
/* Rule books of the sparse encoder (P/coocc/voxel_encoder/sparse_lidar_enc.py; spconv SubMConv3d / SparseConv3d
 * semantics restated).  coors:[M,3] (z,y,x); spatial shape (D,H,W).  A rule book is the [taps][Mo] row table that
 * coocc_conv_fwd consumes through coocc_conv_desc.gather (tap t = (kd*k + kh)*k + kw, spconv's KRSC weight order). */
int coocc_sparse_index_map(const int32_t* coors, int M, int D, int H, int W, int32_t* map /*[D*H*W], -1 = inactive*/,
                           void* stream);
int coocc_sparse_conv_table(const int32_t* out_coors, int Mo, int Di, int Hi, int Wi, int ksize, int stride, int pad,
                            const int32_t* in_map, int32_t* table /*[k^3][Mo]*/, void* stream);
int coocc_sparse_down_flags(const int32_t* coors, int M, int ksize, int stride, int pad, int Do, int Ho, int Wo,
                            uint8_t* flags /*[Do*Ho*Wo]*/, void* stream);
/* linear ids (z*H + y)*W + x -> coors:[n,3] and (optionally) the channels-last rows (x*H + y)*D + z of the dense volume */
int coocc_sparse_lin_to_coors(const int32_t* lin, int n, int D, int H, int W, int32_t* coors, int32_t* dense_rows,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COOCC_HIP_H */
