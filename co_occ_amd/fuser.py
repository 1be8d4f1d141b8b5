"""``BiFuser_N`` (GSFusion) -- MI355X-native mirror of P/coocc/fuser/bifuser_n.py.

Same registry name, constructor, ``forward`` / ``fps_NN_fast`` signatures and ``state_dict``
keys (``con_enc.{0,1,3,4}.*``, ``knn_enc.0.*``) as the reference; all compute runs in
libcoocc_hip.so.  Determinism rules the reference leaves open:
top-K ties ordered by (d^2, key index); duplicate assignment = highest centre ordinal wins;
FPS ties exactly as the reference block reduction.
"""
import torch
from torch import nn

from . import _lib, streams
from ._lib import call, ptr
from . import core as _core_mod
from .core import PackCache, PackedConv, Rows, conv_rows, gather_conv_rows
from .registry import FUSION_LAYERS

_I32, _F32, _I64 = torch.int32, torch.float32, torch.int64


_fps_ws = {}
_side = {}




def _side_stream(dev, cur, which=0):
    key = (dev.index, cur.cuda_stream, which)
    if key not in _side:
        # same scheduling priority as its parent: a search stage prefetched on a high-priority stream keeps it
        _side[key] = streams.side_stream_for(cur, which) or torch.cuda.Stream(device=dev, priority=getattr(cur, "priority", 0))
    return _side[key]


FPS_HIGH_PRIORITY = __import__("os").environ.get("COOCC_FPS_HIGH_PRIORITY", "0") != "0"     # measured: any high-priority stream slows the dense graphs (bench: 140 -> 122 samples/s)
_fps_hi = {}
FPS_MAX_BUCKETS = 8 * 1024     # FPS_RMAX * threads of csrc/knn.hip k_fps_voxels


def _fps_voxels(q_lin, grid, fps_num, which=0, home=None):
    """FPS on a voxel list with the bucket-pruned kernel (same result as the generic one).
    When the pipeline runs on CU-partitioned streams (co_occ_amd.streams) the kernel is issued on
    the reserved-CU stream paired with `home` (the pipeline's main stream) and joined back."""
    dev = q_lin.device
    cur = torch.cuda.current_stream(dev)
    fstream = streams.fps_stream_for(home if home is not None else cur, which)
    if fstream is None and FPS_HIGH_PRIORITY:
        # the 2047-step chain alone on a high-priority stream (one workgroup: it costs the other kernels nothing, but queued
        # at normal priority behind a saturated GPU it waits for a CU slot at every launch); the all-CU kernels of the
        # search stay at their stream's priority -- at high priority they slow the dense stage by 40 % (measured)
        key = (dev.index, cur.cuda_stream, which)
        if key not in _fps_hi:
            _fps_hi[key] = torch.cuda.Stream(device=dev, priority=-1)
        fstream = _fps_hi[key]
    if fstream is not None:
        fstream.wait_stream(cur)
        with torch.cuda.stream(fstream):
            out = _fps_voxels_on_current(q_lin, grid, fps_num)
        cur.wait_stream(fstream)
        out.record_stream(cur)
        q_lin.record_stream(fstream)
        return out
    return _fps_voxels_on_current(q_lin, grid, fps_num)


def _fps_voxels_on_current(q_lin, grid, fps_num):
    dev = q_lin.device
    X, Y, Z = grid
    need = int(_lib.load().coocc_fps_voxels_ws(X, Y, Z))
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _fps_ws or _fps_ws[key].numel() < need:
        _fps_ws[key] = torch.empty(need, device=dev, dtype=torch.uint8)
    ws = _fps_ws[key]
    out = torch.empty(1, fps_num, device=dev, dtype=_I32)
    call("coocc_fps_voxels", ptr(q_lin), q_lin.numel(), X, Y, Z, fps_num, ptr(out), ptr(ws), ws.numel())
    return out


GRID_SEARCH = __import__("os").environ.get("COOCC_GRID_SEARCH", "1") != "0"   # 0: brute-force top-K / ball query
# con_enc.0 split by input-channel support (csrc/sparse_taps.hip): the pts / fused_img slots are non-zero on the LiDAR voxels
# only, so they are convolved in scatter form over those rows while img / fused_pts go through the Winograd GEMM.
# Round 3: under the split-f16 engine the whole layer runs F(4x4) at 2.5x the GEMM rate and the ONE dense 4C-channel GEMM is
# the cheaper form at every occupancy the hipGraph is sized for (measured at 12 % occupancy: dense 0.60 ms against 0.36 + 0.46 ms
# for the split -- the scatter form writes and re-reads a [Np, 27, Cout] fp32 tensor, 265 MB): default off there (+4 % samples/s),
# COOCC_SPLIT_C0=1 restores it; default on for the fp32-MFMA engine.
SPLIT_C0 = __import__("os").environ.get("COOCC_SPLIT_C0", "0" if _core_mod.CONV_ENGINE == "h2" else "1") != "0"
SPLIT_C0_MAX_DENSITY = 0.30      # above this share of LiDAR voxels the dense 4C-channel GEMM is the cheaper form
# ... and only where it is validated end to end: with all 4C channels in F(4x4) the configs[1] scenes hold north_star's 1e-4
# (tests/test_gpu_parity_full.py), the OpenOccupancy scene (128x128x10, 6x896x1600 maps) renders 1.25e-4 / 2.3e-4 from the CPU
# oracle (tests/test_gpu_openocc.py) -- grids above this many voxels keep the split (its LiDAR half is a direct-form convolution)
DENSE_C0_MAX_VOXELS = 100000
_offset_tables = {}


def offset_table(Z, device, radius=16):
    """Offsets (dx,dy,dz) with d^2 <= radius^2 and |dz| < Z, sorted by (d^2, dx, dy, dz) -- the canonical neighbour order
    (d^2, key index) of a dense grid whose key list is ascending in the linear voxel id -- packed for
    coocc_knn_topk_voxels.  radius 16 > the assignment threshold 13.3 (bifuser_n.py:104), so every neighbour that can matter
    is inside the table; the kernel falls back to brute force for a representative with fewer than K keys in range."""
    key = (int(Z), device.index, radius)
    if key not in _offset_tables:
        import numpy as np
        r = np.arange(-radius, radius + 1)
        zr = np.arange(-min(radius, Z - 1), min(radius, Z - 1) + 1)
        dx, dy, dz = np.meshgrid(r, r, zr, indexing="ij")
        dx, dy, dz = dx.ravel(), dy.ravel(), dz.ravel()
        d2 = dx * dx + dy * dy + dz * dz
        keep = d2 <= radius * radius
        dx, dy, dz, d2 = dx[keep], dy[keep], dz[keep], d2[keep]
        order = np.lexsort((dz, dy, dx, d2))
        packed = ((dx[order] + 128) | ((dy[order] + 128) << 8) | ((dz[order] + 128) << 16)).astype(np.int64)
        _offset_tables[key] = torch.from_numpy(packed.astype(np.uint32).view(np.int32)).to(device)
    return _offset_tables[key]


def _fps_nn_xyz(query_xyz, key_xyz, fps_num, radius, max_cluster_samples, dist_thresh, num, q_lin=None, grid=None,
                which=0, home=None, k_lin=None):
    """Index search on float xyz rows.  Returns int32 [num, Q] of key ordinals (-1 = none).
    q_lin/grid: the queries as distinct voxels of one grid -> pruned FPS kernel; with k_lin as well (both sets on the same
    grid) top-K and the ball query run their grid forms (index maps instead of 2048 x N distance sweeps)."""
    dev = query_xyz.device
    Q, Nk = query_xyz.shape[0], key_xyz.shape[0]
    if Q <= fps_num:
        if num != 1:
            # the reference indexes a 1-D tensor twice on this branch (bifuser_n.py:90-93)
            raise IndexError("too many indices for tensor of dimension 1")
        out = torch.full((1, Q), -1, device=dev, dtype=_I32)
        if Q == 0 or Nk == 0:
            return out
        val = torch.empty(Q, 1, device=dev, dtype=_F32)
        nn_ = torch.empty(Q, 1, device=dev, dtype=_I32)
        call("coocc_knn_topk", Q, Nk, 1, ptr(query_xyz), ptr(key_xyz), ptr(val), ptr(nn_))
        call("coocc_knn_threshold", Q, float(dist_thresh), ptr(val), ptr(nn_), ptr(out))
        return out
    if q_lin is not None and grid is not None and (grid[0] + 3) // 4 * ((grid[1] + 3) // 4) * ((grid[2] + 7) // 8) <= FPS_MAX_BUCKETS:
        repr_idx = _fps_voxels(q_lin, grid, fps_num, which, home)
    else:
        repr_idx = torch.empty(1, fps_num, device=dev, dtype=_I32)
        temp = torch.empty(1, Q, device=dev, dtype=_F32)
        call("coocc_furthest_point_sampling", 1, Q, fps_num, ptr(query_xyz), ptr(temp), ptr(repr_idx))
    repr_xyz = query_xyz[repr_idx[0].long()].contiguous()
    val = torch.empty(fps_num, num, device=dev, dtype=_F32)
    nn_ = torch.empty(fps_num, num, device=dev, dtype=_I32)
    group = torch.empty(fps_num, max_cluster_samples, device=dev, dtype=_I32)
    if GRID_SEARCH and q_lin is not None and k_lin is not None and grid is not None:
        X, Y, Z = grid
        maps = torch.empty(2, X * Y * Z, device=dev, dtype=_I32)
        call("coocc_voxel_index_map", ptr(q_lin), Q, X * Y * Z, ptr(maps[0]))
        call("coocc_voxel_index_map", ptr(k_lin), Nk, X * Y * Z, ptr(maps[1]))
        off = offset_table(Z, dev)
        rep = repr_idx[0].contiguous()
        call("coocc_knn_topk_voxels", fps_num, Nk, num, X, Y, Z, ptr(rep), ptr(q_lin), ptr(maps[1]), ptr(off), off.numel(),
             ptr(repr_xyz), ptr(key_xyz), ptr(val), ptr(nn_))
        call("coocc_ball_query_voxels", fps_num, 0.0, float(radius), max_cluster_samples, X, Y, Z, ptr(rep), ptr(q_lin),
             ptr(maps[0]), ptr(group))
    else:
        call("coocc_knn_topk", fps_num, Nk, num, ptr(repr_xyz), ptr(key_xyz), ptr(val), ptr(nn_))
        # the reference recomputes this identical ball query once per k (bifuser_n.py:109)
        call("coocc_ball_query", 1, Q, fps_num, 0.0, float(radius), max_cluster_samples, ptr(repr_xyz), ptr(query_xyz),
             ptr(group))
    winner = torch.empty(num, Q, device=dev, dtype=_I32)
    out = torch.empty(num, Q, device=dev, dtype=_I32)
    call("coocc_knn_assign", fps_num, num, max_cluster_samples, Q, float(dist_thresh), ptr(val), ptr(nn_), ptr(group),
         ptr(winner), ptr(out))
    return out


def _as_rows_view(x):
    """Rows for inputs that already ARE channels-last rows in memory (a Rows object, or the permuted [B,C,X,Y,Z] view of
    one, possibly a column slice of a wider buffer such as the [V,4C] concat rows); None for NCDHW / other layouts."""
    if isinstance(x, Rows):
        return x
    if x.dim() != 5 or not x.is_cuda or x.dtype != _F32:
        return None
    B, C, X, Y, Z = x.shape
    sb, sc, sx, sy, sz = x.stride()
    S = sz
    if sc != 1 or S < C or S % 4 or sy != Z * S or sx != Y * Z * S or (B > 1 and sb != X * Y * Z * S):
        return None
    off = x.storage_offset()
    coff = off % S
    if coff + C > S or (x.data_ptr() - 4 * coff) % 16:
        return None
    base = torch.as_strided(x, (B * X * Y * Z, S), (S, 1), off - coff)
    return Rows(base, B, X, Y, Z, C, coff)


_concat_buffers = {}          # data_ptr -> weak reference of the storage's owner tensor


def _register_concat(t):
    import weakref
    for k in [k for k, r in _concat_buffers.items() if r() is None]:
        del _concat_buffers[k]
    _concat_buffers[t.data_ptr()] = weakref.ref(t)


def _is_concat(t):
    """``t`` shares its memory with a live buffer that ``concat_buffer`` handed out (views keep the base tensor alive)."""
    r = _concat_buffers.get(t.data_ptr())
    base = r() if r is not None else None
    return base is not None and base.data_ptr() == t.data_ptr() and base.numel() == t.numel()


class SearchSlot:
    """Static buffers of one sample in flight (serving loop over captured hipGraphs, ``co_occ_amd.graph``): everything the
    dense stage reads from the search stage lives at fixed addresses with capacity-sized shapes -- the concat rows
    [V,4C] (slot 0 = the pooled camera volume, written by ``lift_splat(out=slot.img_rows())``), the non-empty voxel lists
    [2,V], their lengths [2] ON THE DEVICE, and the two neighbour row tables [K,V]."""

    def __init__(self, C, knum, grid, device):
        X, Y, Z = grid
        V = X * Y * Z
        self.grid, self.C, self.V = (X, Y, Z), C, V
        self.cat4 = torch.empty(V, 4 * C, device=device, dtype=_F32)
        _register_concat(self.cat4)
        self.lin = torch.zeros(2, V, device=device, dtype=_I32)
        self.counts = torch.zeros(2, device=device, dtype=_I32)
        self.rows = torch.zeros(knum, V, device=device, dtype=_I32)
        self.rows_p = torch.zeros(knum, V, device=device, dtype=_I32)

        self.knum = knum
        self.near = torch.zeros(2, knum * V, device=device, dtype=_I32)     # near_img, near_pts (dense [K, count] inside)
        self._native = None

    def img_rows(self):
        X, Y, Z = self.grid
        return Rows(self.cat4, 1, X, Y, Z, self.C, 0, persistent=True)     # rewritten per frame by lift_splat(out=) / the search

    def native_desc(self):
        """coocc_search_desc over this slot's buffers (csrc/search.hip), workspace included; built once."""
        if self._native is None:
            import ctypes
            X, Y, Z = self.grid
            d = _lib.SearchDesc()
            d.cat4 = self.cat4.data_ptr()
            d.C, d.X, d.Y, d.Z, d.K = self.C, X, Y, Z, self.knum
            d.fps_num, d.max_cluster, d.radius, d.dist_thresh = 2048, 200, 6.0, 13.3
            off = offset_table(Z, self.cat4.device)
            d.offsets, d.noff = off.data_ptr(), off.numel()
            d.lin, d.counts = self.lin.data_ptr(), self.counts.data_ptr()
            d.near_img, d.near_pts = self.near[0].data_ptr(), self.near[1].data_ptr()
            d.rows, d.rows_p = self.rows.data_ptr(), self.rows_p.data_ptr()
            need = int(_lib.load().coocc_fuser_search_ws(ctypes.byref(d)))
            ws = torch.empty(need, device=self.cat4.device, dtype=torch.uint8)
            d.ws, d.ws_bytes = ws.data_ptr(), need
            host = (ctypes.c_int32 * 2)()
            d.counts_host = ctypes.addressof(host)
            self._native = (d, ws, host, off)
        return self._native


class SearchResult:
    """Output of ``BiFuser_N.search``: the concat rows (img | pts halves written), the non-empty voxel lists and
    the neighbour row tables of both directions, with the events that mark them ready."""

    def __init__(self, cat4, lin_img, lin_pts):
        self.cat4, self.lin_img, self.lin_pts = cat4, lin_img, lin_pts
        self.rows = self.rows_p = self.near_img = self.near_pts = None
        self.done_main = self.done_side = None
        self.keep = ()
        self.slot = None          # SearchSlot: the dense stage reads capacity-sized buffers + device-side counts
        self.counts = (lin_img.numel(), lin_pts.numel())

    def tensors(self):
        out = [self.cat4.t, self.lin_img, self.lin_pts]
        out += [t for t in (self.rows, self.rows_p, self.near_img, self.near_pts) if t is not None]
        return out + [t for t in self.keep if torch.is_tensor(t)]


@FUSION_LAYERS.register_module()
class BiFuser_N(nn.Module):
    def __init__(self, in_channels, out_channels, knum=1, norm_cfg=None):
        super().__init__()
        self.in_channels, self.out_channels, self.knum = in_channels, out_channels, knum
        # parameter containers with the reference layout (bifuser_n.py:23-36); norm_cfg is
        # accepted and ignored exactly as the reference does
        self.con_enc = nn.Sequential(
            nn.Conv3d(in_channels * 4, out_channels * 2, 3, padding=1, bias=False),
            nn.BatchNorm3d(out_channels * 2), nn.ReLU(True),
            nn.Conv3d(in_channels * 2, out_channels, 3, padding=1, bias=False),
            nn.BatchNorm3d(out_channels), nn.ReLU(True))
        self.knn_enc = nn.Sequential(nn.Linear(in_channels * knum, out_channels), nn.ReLU())
        self._packs = PackCache(self)
        self.last_counts = None
        # callable -> the layers (PackedConv) that read the fused features next (the detector sets it: the encoder's input_proj
        # and the render MLPs' first layers are split-f16 GEMMs outside the Winograd path, so con_enc.3's output transform
        # writes their H2 operand next to the fp32 rows)
        self.output_readers = None

    # ---------------------------------------------------------------- packing
    def set_conenc_tiles(self, tiles):
        """Winograd tile sizes (m of F(m x m, 3x3)) of con_enc.0 / con_enc.3 for THIS model: (4, 4) = the split-f16 engine's
        default (fastest), (2, 2) = the accuracy preset (voxel_feats 2x closer to fp64 than the CPU fp32 reference, r50 rgb maps
        2.9e-5 instead of 7.4e-5 from the oracle, ~5 % slower); None = environment / default again.  ``co_occ_amd.calibrate``
        measures both on a user's checkpoint.  Captured serving graphs re-capture at their next submit."""
        self.conenc_tiles = None if tiles is None else (int(tiles[0]), int(tiles[1]))
        _core_mod.invalidate_packs(self)

    def _packed(self):
        def build():
            d = build_packs()
            # con_enc opens the decoder: its rounding error is amplified by every later layer and goes straight into the
            # render heads, so its Winograd tile size is chosen by parity, not speed.  Measured (profiles/r2_conenc_tiles.txt;
            # rms error of the fine logits vs an fp64 evaluation as a multiple of the CPU fp32 oracle's own error, 4 seeds; and the
            # full-size r50 scene of tests/test_gpu_parity_full.py), tiles of (con_enc.0, con_enc.3):
            #   (2,2) 0.58-0.96, rgb maps 2.9e-5, 104.5 samples/s
            #   (2,4) 0.70-1.20, rgb maps 1.03e-4 (over the 1e-4 bound), 109.7
            #   (4,4) 1.22-1.93 (over the 1.5x bound), 119.1
            # so both layers keep F(2x2); the 0.3 / 1.2 ms the larger tiles would save are left on the table for parity.
            # COOCC_CONENC_TILES overrides (experiments).
            # Round 3: on the split-f16 engine the transform-domain GEMM carries half the error of the fp32-MFMA one (F(4x4):
            # 2.3-3.2e-6 -> 1.0-1.3e-6 rms of the output scale, profiles/r3_h2_check.txt), and (4,4) passes the whole sweep --
            # rms ratios 0.49-0.85, max ratios 0.26-1.43, aggregate 0.72 / 0.69, full r50 rgb maps 5.6e-5
            # (profiles/r3_parity_seed_sweep.txt): it is the default there (+5 % samples/s); the fp32 engine keeps (2,2).
            import os
            from . import core as _core
            default = "4,4" if _core.CONV_ENGINE == "h2" else "2,2"
            forced = getattr(self, "conenc_tiles", None)            # set_conenc_tiles(): per model, wins over the environment
            t0, t3 = forced if forced is not None else [int(v) for v in os.environ.get("COOCC_CONENC_TILES", default).split(",")]
            d["c0"].wino_tile, d["c3"].wino_tile = t0, t3
            d["c0_dense"].wino_tile = t0
            # ... on the grids it is validated on: the OpenOccupancy scene (163 840 voxels, 6x896x1600 maps) renders 1.09e-4 /
            # 1.98e-4 (rgb abs / depth rel) from the CPU oracle with (4,4) and holds 1e-4 with (2,2) (tests/test_gpu_openocc.py),
            # so grids above DENSE_C0_MAX_VOXELS keep (2,2) unless COOCC_CONENC_TILES says otherwise (con_enc0 applies it per call)
            d["conenc_tiles"] = (t0, t3)
            d["conenc_tiles_large"] = (t0, t3) if ("COOCC_CONENC_TILES" in os.environ or forced is not None) else (2, 2)
            return d

        def build_packs():
            C, W0 = self.in_channels, self.con_enc[0].weight
            Co = W0.shape[0]
            dense_idx = list(range(0, C)) + list(range(3 * C, 4 * C))                       # img | fused_pts
            # sparse half (pts | fused_img) as a Linear with one 27*Cout-wide output row per occupied voxel: column (t, n)
            w_cols = W0[:, C:3 * C].reshape(Co, 2 * C, 27).permute(2, 0, 1).reshape(27 * Co, 2 * C).contiguous()
            return dict(
                c0_dense=PackedConv(W0[:, dense_idx].contiguous(), bn=self.con_enc[1], ksize=3, pad=1),
                c0_sparse=PackedConv(w_cols),
                c0=PackedConv(self.con_enc[0].weight, bn=self.con_enc[1], ksize=3, pad=1),
                c3=PackedConv(self.con_enc[3].weight, bn=self.con_enc[4], ksize=3, pad=1),
                knn=PackedConv(self.knn_enc[0].weight, bias=self.knn_enc[0].bias, tap_major=True, taps=self.knum))
        return self._packs.get_modules((self.con_enc, self.knn_enc), build)

    # ---------------------------------------------------------------- K2-K5
    def fps_NN_fast(self, query, key, fps_num, radius, max_cluster_samples, dist_thresh, num):
        """bifuser_n.py:38-125.  query [Q,4], key [Nk,4] integer (b,x,y,z) rows ->
        int64 [Q] (num == 1) or [num,Q]; -1 where no key was assigned."""
        q = query[:, 1:].float().contiguous()
        k = key[:, 1:].float().contiguous()
        out = _fps_nn_xyz(q, k, fps_num, radius, max_cluster_samples, dist_thresh, num).long()
        return out[0] if num == 1 else out

    def concat_buffer(self, B, X, Y, Z, device):
        """A fresh [B*X*Y*Z, 4C] concat buffer and the Rows of its slot 0: a producer that writes the camera volume there
        (``ViewTransformerLiftSplatShootVoxel.lift_splat(out=...)``) saves the prologue its copy.  Only buffers handed out
        here (or owned by a ``SearchSlot``) are completed IN PLACE by ``search``: any other channels-last tensor -- e.g.
        ``feats[:, :C]`` of a wider buffer the caller still uses -- is copied into a fresh concat buffer."""
        C = self.in_channels
        cat4 = torch.empty(B * X * Y * Z, 4 * C, device=device, dtype=_F32)
        _register_concat(cat4)
        return Rows(cat4, B, X, Y, Z, C, 0)

    # ---------------------------------------------------------------- forward
    def search(self, img_voxel_feats, pts_voxel_feats, slot=None):
        """K1..K5 on the current stream: concat rows with the img | pts halves in place, non-empty voxel lists,
        both index searches (the second on a side stream).  Returns a ``SearchResult``; nothing downstream of
        the indices has been launched, so a caller may run this for sample i+1 on its own stream (and host
        thread: there is one device->host read of the two voxel counts) while sample i is in its dense stage
        -- the 2 x 2047 dependent FPS steps occupy one CU each and overlap everything."""
        img_r, pts_r = _as_rows_view(img_voxel_feats), _as_rows_view(pts_voxel_feats)
        one = img_r if img_r is not None else img_voxel_feats
        if isinstance(one, Rows):
            B, C, X, Y, Z = one.B, one.C, one.X, one.Y, one.Z
            dev = one.t.device
        else:
            B, C, X, Y, Z = one.shape
            dev = one.device
        V = X * Y * Z
        if dev.type != "cuda":
            raise _lib.CooccError("BiFuser_N runs on the GPU only (no CPU fallback)")
        flags = torch.empty(2, B * V, device=dev, dtype=torch.uint8)
        if img_r is None and pts_r is None:
            img = img_voxel_feats.float().contiguous()
            pts = pts_voxel_feats.float().contiguous()
            cat4 = torch.empty(B * V, 4 * C, device=dev, dtype=_F32)
            call("coocc_fuser_prepare", ptr(img), ptr(pts), ptr(cat4), ptr(flags[0]), ptr(flags[1]), B, C, V)
        else:
            # a producer handed over channels-last rows (fused lift-splat, sparse LiDAR encoder): no NCDHW round trip; rows
            # that already sit in slot 0 of a [V,4C] concat buffer (lift_splat(out=BiFuser_N.concat_buffer(...))) stay there
            in_place = (img_r is not None and img_r.t.shape[1] == 4 * C and img_r.coff == 0 and img_r.t.is_contiguous()
                        and (_is_concat(img_r.t) or (slot is not None and img_r.t.data_ptr() == slot.cat4.data_ptr())))
            if slot is not None:
                assert in_place and img_r.t.data_ptr() == slot.cat4.data_ptr() and B == 1, \
                    "search(slot=...): the camera rows must already sit in slot 0 of the slot's concat buffer"
            cat4 = img_r.t if in_place else torch.empty(B * V, 4 * C, device=dev, dtype=_F32)
            img = img_r if img_r is not None else img_voxel_feats.float().contiguous()
            pts = pts_r if pts_r is not None else pts_voxel_feats.float().contiguous()
            src = lambda r: (r.data(), 1, r.stride) if isinstance(r, Rows) else (ptr(r), 0, 0)
            call("coocc_fuser_prepare_rows", *src(img), *src(pts), ptr(cat4), ptr(flags[0]), ptr(flags[1]), B, C, V)
        assert slot is None or (img_r is not None or pts_r is not None), "search(slot=...) takes channels-last producers"
        lin = slot.lin if slot is not None else torch.empty(2, B * V, device=dev, dtype=_I32)
        counts = slot.counts if slot is not None else torch.empty(2, device=dev, dtype=_I32)
        ws = torch.empty(2, B * V // 1024 + 2, device=dev, dtype=_I32)
        for i in range(2):
            call("coocc_compact_flags", ptr(flags[i]), B * V, ptr(lin[i]), ptr(counts[i:i + 1]), ptr(ws[i]),
                 ws[i].numel() * 4)
        Ni, Np = (int(v) for v in _lib.host_read(counts))     # the one host sync of the stage (torch.nonzero does two)
        self.last_counts = (Ni, Np)
        lin_img, lin_pts = lin[0, :Ni], lin[1, :Np]
        xyz = torch.empty(Ni + Np, 3, device=dev, dtype=_F32)
        if Ni:
            call("coocc_lin_to_coords", ptr(lin_img), Ni, X, Y, Z, ptr(xyz[:Ni]), None)
        if Np:
            call("coocc_lin_to_coords", ptr(lin_pts), Np, X, Y, Z, ptr(xyz[Ni:]), None)
        xyz_img, xyz_pts = xyz[:Ni], xyz[Ni:]
        K = self.knum
        kw = dict(fps_num=2048, radius=6, max_cluster_samples=200, dist_thresh=13.3, num=K)
        sr = SearchResult(Rows(cat4, B, X, Y, Z, 4 * C), lin_img, lin_pts)
        sr.slot = slot
        cur = torch.cuda.current_stream(dev)
        if Np and Ni:
            vox = (X, Y, Z) if B == 1 else None
            # The two search directions are independent (each is a 2047-step dependent chain on
            # one CU): the img->pts search runs on a side stream next to the pts->img one.
            side = _side_stream(dev, cur)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                # img queries <- nearest pts keys (:150-162); for knum > 1 the reference indexes
                # inds_img with the pts ordinals (:158) -- kept
                sr.near_pts = _fps_nn_xyz(xyz_img, xyz_pts, q_lin=lin_img if vox else None, k_lin=lin_pts if vox else None, grid=vox,
                                          which=1, home=cur, **kw)
                sr.rows_p = slot.rows_p[:, :Ni] if slot is not None else torch.empty(K, Ni, device=dev, dtype=_I32)
                base, nbase = (lin_pts, Np) if K == 1 else (lin_img, Ni)
                for k in range(K):
                    call("coocc_index_rows_i32", ptr(base), nbase, ptr(sr.near_pts[k]), Ni, ptr(sr.rows_p[k]))
                sr.done_side = streams.new_event()
                sr.done_side.record()
            # pts queries <- nearest img keys (bifuser_n.py:137-148)
            sr.near_img = _fps_nn_xyz(xyz_pts, xyz_img, q_lin=lin_pts if vox else None, k_lin=lin_img if vox else None, grid=vox,
                                      which=0, home=cur, **kw)
            sr.rows = slot.rows[:, :Np] if slot is not None else torch.empty(K, Np, device=dev, dtype=_I32)
            for k in range(K):
                call("coocc_index_rows_i32", ptr(lin_img), Ni, ptr(sr.near_img[k]), Np, ptr(sr.rows[k]))
            sr.keep = (xyz, lin, img, pts)       # referenced by kernels still in flight on the side stream
        elif Np or Ni:
            raise IndexError("BiFuser_N: one modality has no non-empty voxel (the reference fails on empty keys)")
        sr.done_main = streams.new_event()
        sr.done_main.record()
        return sr

    def search_native(self, pts_voxel_feats, slot):
        """``search(slot.img_rows(), pts, slot=slot)`` through ONE C-ABI call (``coocc_fuser_search``, csrc/search.hip): the
        launches, the two-stream fork / join and the count read are issued from C++ with the GIL released.  Falls back to
        ``search`` for what the native driver does not cover (a list with <= 2048 voxels, grids too large for the bucketed
        FPS, batch > 1).  The camera rows must already sit in slot 0 of ``slot.cat4``."""
        import ctypes
        X, Y, Z = slot.grid
        dev = slot.cat4.device
        pts_r = _as_rows_view(pts_voxel_feats)
        B = pts_r.B if pts_r is not None else pts_voxel_feats.shape[0]
        if B != 1 or (X + 3) // 4 * ((Y + 3) // 4) * ((Z + 7) // 8) > FPS_MAX_BUCKETS:
            return self.search(slot.img_rows().as_ncdhw(), pts_voxel_feats, slot=slot)
        d, _, host, _ = slot.native_desc()
        if pts_r is not None:
            d.pts, d.pts_rows, d.pts_stride = pts_r.t.data_ptr() + 4 * pts_r.coff, 1, pts_r.stride
            keep = pts_r.t
        else:
            keep = pts_voxel_feats.float().contiguous()
            d.pts, d.pts_rows, d.pts_stride = keep.data_ptr(), 0, 0
        cur = torch.cuda.current_stream(dev)
        side = _side_stream(dev, cur)
        rc = _lib.load().coocc_fuser_search(ctypes.byref(d), ctypes.c_void_p(cur.cuda_stream), ctypes.c_void_p(side.cuda_stream))
        if rc == 1:           # COOCC_SEARCH_SMALL: the reference's other branch
            return self.search(slot.img_rows().as_ncdhw(), pts_voxel_feats, slot=slot)
        _lib.check(rc)
        Ni, Np = int(host[0]), int(host[1])
        self.last_counts = (Ni, Np)
        K, V = self.knum, slot.V
        sr = SearchResult(Rows(slot.cat4, 1, X, Y, Z, 4 * self.in_channels), slot.lin[0, :Ni], slot.lin[1, :Np])
        sr.slot = slot
        sr.near_img = slot.near[0, :K * Np].view(K, Np)
        sr.near_pts = slot.near[1, :K * Ni].view(K, Ni)
        sr.rows, sr.rows_p = slot.rows[:, :Np], slot.rows_p[:, :Ni]
        sr.keep = (keep,)
        sr.done_main = streams.new_event()
        sr.done_main.record()
        sr.done_side = sr.done_main          # the native call joins the side stream before it returns
        return sr

    def finish(self, sr):
        """G1 on the current stream: the two gather -> knn_enc -> gate -> scatter GEMMs that complete the concat
        rows of a ``SearchResult`` (bifuser_n.py:138-169).  Returns (rows [B*V,4C], (lin_img, lin_pts))."""
        cat4 = sr.cat4.t
        dev = cat4.device
        C = self.in_channels
        cur = torch.cuda.current_stream(dev)
        packs = self._packed()
        sr.done_main.wait(cur)
        for t in sr.tensors():
            t.record_stream(cur)             # allocated on the search stream(s), consumed here
        if sr.rows is not None and sr.slot is not None:
            # the search wrote into a SearchSlot: its tables are capacity-strided views, use the device-count form
            sr.done_side.wait(cur)
            self.finish_static(sr.slot)
            self.last_near = (sr.near_img, sr.near_pts)
        elif sr.rows is not None:
            sr.done_side.wait(cur)
            h2 = self._g1_h2(cat4, packs)
            gather_conv_rows(cat4, 0, packs["knn"], sr.rows, sr.lin_pts, cat4, 2 * C, C, C, src_h2=h2)
            gather_conv_rows(cat4, C, packs["knn"], sr.rows_p, sr.lin_img, cat4, 3 * C, 0, C, src_h2=h2)
            self.last_near = (sr.near_img, sr.near_pts)
        return sr.cat4, (sr.lin_img, sr.lin_pts)

    def _g1_h2(self, cat4, packs):
        """H2 rows of the img | pts slots for both gather GEMMs (one conversion launch instead of one per GEMM; COOCC_G1_H2_ONCE=0
        keeps the per-GEMM conversions).  The GEMMs only write the fused slots [2C, 4C), so converting both sources first reads
        the same values."""
        import os
        from . import core
        if os.environ.get("COOCC_G1_H2_ONCE", "1") == "0":
            return None
        sh = core.g1_sources_h2(cat4, self.in_channels, packs["knn"])
        return None if sh is None else (sh, 0, 2 * self.in_channels)

    def finish_static(self, slot):
        """``finish`` over a ``SearchSlot``: the same two gather GEMMs with the row counts read on the device and
        capacity-sized tables -- no argument depends on the sample, so the launches can be captured once (hipGraph) and
        replayed after every search that filled the slot.  The caller orders the streams (events of the SearchResult)."""
        C = self.in_channels
        packs = self._packed()
        cat4 = slot.cat4
        h2 = self._g1_h2(cat4, packs)
        gather_conv_rows(cat4, 0, packs["knn"], slot.rows, slot.lin[1], cat4, 2 * C, C, C, count_dev=slot.counts[1:2], src_h2=h2)
        gather_conv_rows(cat4, C, packs["knn"], slot.rows_p, slot.lin[0], cat4, 3 * C, 0, C, count_dev=slot.counts[0:1], src_h2=h2)
        X, Y, Z = slot.grid
        return Rows(cat4, 1, X, Y, Z, 4 * C)

    def forward_static(self, slot):
        """G1 + con_enc over a ``SearchSlot`` (see ``finish_static``): [V, out] rows."""
        packs = self._packed()
        cat4 = self.finish_static(slot)
        x = self.con_enc0(cat4, slot.lin[1], packs, count_dev=slot.counts[1:2])
        return conv_rows(x, packs["c3"], relu=True, twin_for=self._readers())

    def finish_bookkeeping(self, sr):
        """Make a SearchResult issued on another stream safe to consume on the current one (training path)."""
        cur = torch.cuda.current_stream(sr.cat4.t.device)
        sr.done_main.wait(cur)
        if sr.done_side is not None:
            sr.done_side.wait(cur)
        for t in sr.tensors():
            t.record_stream(cur)
        if sr.rows is not None:
            self.last_near = (sr.near_img, sr.near_pts)

    def fuse(self, img_voxel_feats, pts_voxel_feats, search=None):
        """K1..G1: returns the [B*V, 4C] concat rows (img | pts | fused_img | fused_pts)."""
        return self.finish(search if search is not None else self.search(img_voxel_feats, pts_voxel_feats))

    def forward(self, img_voxel_feats, pts_voxel_feats, search=None):
        """[B,C,X,Y,Z] x2 -> [B,out,X,Y,Z] (bifuser_n.py:127-174).  ``search``: a ``SearchResult`` of the same
        inputs computed ahead of time (cross-sample pipelining)."""
        if not (img_voxel_feats.t if isinstance(img_voxel_feats, Rows) else img_voxel_feats).is_cuda:
            raise _lib.CooccError("BiFuser_N runs on the GPU only (no CPU fallback)")
        if self.training:                    # batch-statistics BN + autograd, as upstream under model.train()
            from . import autograd as ag
            assert img_voxel_feats.shape[0] == 1, "BiFuser_N training path: batch size 1 per GPU (samples_per_gpu=1 upstream)"
            with torch.no_grad():
                sr = search if search is not None else self.search(img_voxel_feats.detach(), pts_voxel_feats.detach())
            (ri, geom), (rp, _) = ag.rows_from_ncdhw(img_voxel_feats), ag.rows_from_ncdhw(pts_voxel_feats)
            self.finish_bookkeeping(sr)
            x, g = ag.con_enc_train(self.con_enc, ag.fuser_fuse_train(self, ri, rp, sr), geom)
            return ag.ncdhw_from_rows(x, g)
        packs = self._packed()
        cat4, (_, lin_pts) = self.fuse(img_voxel_feats, pts_voxel_feats, search)
        x = self.con_enc0(cat4, lin_pts, packs)
        x = conv_rows(x, packs["c3"], relu=True, twin_for=self._readers())
        return x.as_ncdhw()

    def _readers(self):
        return tuple(self.output_readers()) if self.output_readers is not None else ()

    @staticmethod
    def c0_capacity(V):
        """Rows the scatter-form half of con_enc.0 is sized for in the static (hipGraph) form."""
        return int(SPLIT_C0_MAX_DENSITY * V)

    def con_enc0(self, cat4, lin_pts, packs, count_dev=None):
        """con_enc[0:3] (Conv3d 4C -> 2C + BN + ReLU) on the concat rows.  When the LiDAR voxels are a small share of the grid
        the layer is split by input-channel support: img | fused_pts through the Winograd GEMM (half the K), pts | fused_img
        (non-zero on ``lin_pts`` only) in scatter form -- one row-table GEMM over the occupied rows into per-tap
        contributions, summed per output voxel in tap order and handed to the output transform as its residual."""
        from . import core
        C, V, dev = self.in_channels, cat4.V * cat4.B, cat4.t.device
        Np = int(lin_pts.numel())
        pd = packs["c0_dense"]
        if "conenc_tiles" in packs:           # Winograd tiles of con_enc.0 / con_enc.3 by grid size (see _packed)
            t0, t3 = packs["conenc_tiles_large"] if cat4.V > DENSE_C0_MAX_VOXELS else packs["conenc_tiles"]
            packs["c0"].wino_tile = pd.wino_tile = t0
            packs["c3"].wino_tile = t3
        split = SPLIT_C0 or (core.CONV_ENGINE == "h2" and V > DENSE_C0_MAX_VOXELS and "COOCC_SPLIT_C0" not in __import__("os").environ)
        plan = core.wino_plan(cat4, pd, V, 1) if (split and core.CONV_DTYPE == "f32") else None
        if count_dev is not None:
            # static form (hipGraph replay): lin_pts is the capacity-sized list, its length sits on the device; the
            # scatter-form GEMM is sized for the densest sweep the split is used for (the caller checks the host-side count
            # of the sample against ``c0_capacity`` before replaying)
            if plan is None or cat4.B != 1:
                return conv_rows(cat4, packs["c0"], relu=True)      # one dense GEMM: nothing depends on the count
            Np = self.c0_capacity(V)
        elif plan is None or Np == 0 or Np > SPLIT_C0_MAX_DENSITY * V or cat4.B != 1:
            return conv_rows(cat4, packs["c0"], relu=True)
        ps = packs["c0_sparse"]
        Co = pd.Cout
        # P[u][t][n] = W_t[n, :] . cat4[u, C:3C] for the occupied rows u
        P = core.scratch(dev, "c0P", Np * 27 * Co)
        d = _lib.ConvDesc()
        ws = core.workspace(dev)
        src = _lib.DevPtr(cat4.t.data_ptr() + 4 * (cat4.coff + C))
        src._keep = cat4.t
        d.in_, d.w, d.out = src, ptr(ps.w), ptr(P)
        d.scale = d.bias = d.res = d.out_rows = None
        d.gather = ptr(lin_pts)
        d.ws, d.ws_floats = ptr(ws), ws.numel()
        d.M, d.Cin, d.Cout, d.taps = Np, 2 * C, 27 * Co, 1
        d.in_stride, d.out_stride, d.res_stride = cat4.stride, 27 * Co, 0
        d.B = d.Yi = d.Zi = d.Xo = d.Yo = d.Zo = 1
        d.Xi = V                      # number of input rows (lets coocc_conv_fwd pick the pipelined row-table kernel)
        d.ksize, d.stride, d.pad = 1, 1, 0
        d.relu, d.res_mode, d.splitk, d.tile_hint = 0, 0, 1, core.TILE_HINT
        if count_dev is not None:
            d.M_dev, d.gather_stride = ptr(count_dev, _I32), lin_pts.numel()
        kname = core.conv_kernel_name(Np, 27 * Co, True)
        if core.CONV_ENGINE == "h2" and core.H2_DIRECT and (2 * C) % 32 == 0:
            # split-f16 engine: the occupied rows are gathered into a compact H2 operand [Np, 2C] (count on the device in
            # the static form), then a plain GEMM against the [27 Cout, 2C] matrix
            rh = core.scratch(dev, "c0rows", Np * 2 * C)
            call("coocc_rows_to_h2_gather", src, cat4.stride, ptr(lin_pts), Np, ptr(count_dev, _I32) if count_dev is not None else None,
                 2 * C, 1.0, ptr(rh))
            d.in_, d.in_stride, d.w, d.gather, d.gather_stride = ptr(rh), 2 * C, ptr(ps.h2_pack()), None, 0
            d.Xi, d.Xo = Np, Np
            d.mfma_dtype, d.alpha, kname = 3, 1.0, "k_gemm_h2w"
        with core.TIMER.region(kname + " c0-sparse", 2.0 * Np * 2 * C * 27 * Co):
            _lib.conv_fwd(d, dev)
        vmap = torch.empty(V, device=dev, dtype=_I32)
        if count_dev is not None:
            call("coocc_voxel_index_map_dev", ptr(lin_pts), int(lin_pts.numel()), ptr(count_dev, _I32), V, ptr(vmap))
        else:
            call("coocc_voxel_index_map", ptr(lin_pts), Np, V, ptr(vmap))
        S = core.Rows(torch.empty(V, Co, device=dev, dtype=_F32), cat4.B, cat4.X, cat4.Y, cat4.Z, Co)
        call("coocc_sparse_tap_sum", ptr(P), ptr(vmap), cat4.B, cat4.X, cat4.Y, cat4.Z, Co, ptr(pd.scale), ptr(S.t), Co, int(Np))
        out = core.Rows(torch.empty(V, Co, device=dev, dtype=_F32), cat4.B, cat4.X, cat4.Y, cat4.Z, Co)
        return core.conv_rows_wino(cat4, pd, out, True, S, plan, in_ranges=[(0, C), (3 * C, C)])
