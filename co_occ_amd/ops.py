"""``mmdet3d.ops``-shaped entry points of the three native ops on the path,
backed by libcoocc_hip.so.  Signatures, dtypes and return shapes follow the reference
Python wrappers; the ``*_ext`` objects mirror the pybind11 extension modules they call.
"""
import torch

from . import _lib
from ._lib import call, ptr

_I32, _F32 = torch.int32, torch.float32


# ---------------------------------------------------------------- extension-module mirrors
class furthest_point_sample_ext:
    @staticmethod
    def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
        """M/ops/furthest_point_sample/src/furthest_point_sample.cpp:35-46."""
        call("coocc_furthest_point_sampling", b, n, m, ptr(points, _F32), ptr(temp, _F32), ptr(idx, _I32))
        return 1


class ball_query_ext:
    @staticmethod
    def ball_query_wrapper(b, n, m, min_radius, max_radius, nsample, new_xyz, xyz, idx):
        """M/ops/ball_query/src/ball_query.cpp:32-45."""
        call("coocc_ball_query", b, n, m, float(min_radius), float(max_radius), nsample, ptr(new_xyz, _F32),
             ptr(xyz, _F32), ptr(idx, _I32))
        return 1


class bev_pool_ext:
    @staticmethod
    def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
        """M/ops/bev_pool/src/bev_pool.cpp:22-47 -> [b,d,h,w,c]."""
        b, d, h, w = int(b), int(d), int(h), int(w)
        n, c = x.shape
        out = torch.empty(b, d, h, w, c, device=x.device, dtype=_F32)
        call("coocc_bev_pool_forward", ptr(x.contiguous(), _F32), ptr(geom_feats.contiguous(), _I32),
             ptr(interval_lengths.contiguous(), _I32), ptr(interval_starts.contiguous(), _I32), b, d, h, w, n, c,
             interval_lengths.numel(), ptr(out))
        return out

    @staticmethod
    def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, b, d, h, w):
        """M/ops/bev_pool/src/bev_pool.cpp:60-87 -> [n,c]."""
        b, d, h, w = int(b), int(d), int(h), int(w)
        n, c = geom_feats.shape[0], out_grad.shape[4]
        xg = torch.empty(n, c, device=out_grad.device, dtype=_F32)
        call("coocc_bev_pool_backward", ptr(out_grad.contiguous(), _F32), ptr(geom_feats.contiguous(), _I32),
             ptr(interval_lengths.contiguous(), _I32), ptr(interval_starts.contiguous(), _I32), b, d, h, w, n, c,
             interval_lengths.numel(), ptr(xg))
        return xg


# ---------------------------------------------------------------- Python-level ops
def furthest_point_sample(points_xyz, num_points):
    """``furthest_point_sample(points_xyz[B,N,3] f32, num_points) -> [B,num_points] int32``
    (M/ops/furthest_point_sample/furthest_point_sample.py:15-35)."""
    assert points_xyz.is_contiguous()
    B, N = points_xyz.shape[:2]
    out = torch.empty(B, num_points, device=points_xyz.device, dtype=_I32)
    temp = torch.empty(B, N, device=points_xyz.device, dtype=_F32)
    furthest_point_sample_ext.furthest_point_sampling_wrapper(B, N, num_points, points_xyz, temp, out)
    return out


def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    """``ball_query(min_r, max_r, nsample, xyz[B,N,3], center_xyz[B,M,3]) -> [B,M,nsample] int32``
    (M/ops/ball_query/ball_query.py:14-40)."""
    assert center_xyz.is_contiguous() and xyz.is_contiguous()
    assert min_radius < max_radius
    B, N, _ = xyz.shape
    M = center_xyz.shape[1]
    idx = torch.empty(B, M, sample_num, device=xyz.device, dtype=_I32)
    ball_query_ext.ball_query_wrapper(B, N, M, min_radius, max_radius, sample_num, center_xyz, xyz, idx)
    return idx


_pool_ws = {}


def _pool_workspace(device, npts, nvox):
    need = int(_lib.load().coocc_voxel_pool_ws(npts, nvox))
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    if key not in _pool_ws or _pool_ws[key].numel() < need:
        _pool_ws[key] = torch.empty(need, device=device, dtype=torch.uint8)
    return _pool_ws[key]


def pool_ws_clean(ws, npts, nvox):
    """``ws_clean`` argument of the pooling entry points (include/coocc_hip.h): 1 when the last write to this workspace tensor was
    a pooling call of the same (npts, nvox) that returned without error -- it left its histogram zeroed, so the memset launch
    is skipped.  The mark lives on the tensor object and is taken off for the duration of the call (``pool_ws_done`` puts it back
    on success), so an exception leaves the workspace marked dirty."""
    state = getattr(ws, "_coocc_pool_state", None)
    ws._coocc_pool_state = None
    return int(state == (int(npts), int(nvox)))


def pool_ws_done(ws, npts, nvox):
    ws._coocc_pool_state = (int(npts), int(nvox))


def bev_pool_rows(feats, coords, B, D, H, W):
    """Pooling core: returns channels-last rows [B*H*W*D, C] with voxel order (b, x, y, z)
    where x < H, y < W, z < D (the reference's argument naming: D = nz, H = nx, W = ny)."""
    B, D, H, W = int(B), int(D), int(H), int(W)
    n, C = feats.shape
    nvox = B * H * W * D
    out = torch.empty(nvox, C, device=feats.device, dtype=_F32)
    if n == 0:
        return out.zero_()
    ws = _pool_workspace(feats.device, n, nvox)
    call("coocc_bev_pool_coords", ptr(feats.contiguous(), _F32), ptr(coords.long().contiguous(), torch.int64), n, C,
         B, H, W, D, ptr(out), C, ptr(ws), ws.numel(), pool_ws_clean(ws, n, nvox))
    pool_ws_done(ws, n, nvox)
    return out


class _BevPoolFn(torch.autograd.Function):
    """QuickCumsumCuda (M/ops/bev_pool/bev_pool.py:37-80): forward = per-voxel sums, backward = every point receives the
    gradient row of its voxel (bev_pool_cuda.cu:61-84) -- here a row gather by the linear voxel id (``coocc_gather_rows``), no
    sort and no interval bookkeeping to save."""

    @staticmethod
    def forward(ctx, feats, coords, B, D, H, W):
        rows = bev_pool_rows(feats, coords, B, D, H, W)
        c = coords.long()
        # the forward kernel DROPS points outside [0,H) x [0,W) x [0,D) x [0,B) (key = nvox): they must receive a zero gradient,
        # not the row of whatever voxel their out-of-range linear id aliases (coocc_gather_rows skips ids < 0)
        ok = (c[:, 0] >= 0) & (c[:, 0] < H) & (c[:, 1] >= 0) & (c[:, 1] < W) & (c[:, 2] >= 0) & (c[:, 2] < D) & (c[:, 3] >= 0) & (c[:, 3] < B)
        lin = (((c[:, 3] * H + c[:, 0]) * W + c[:, 1]) * D + c[:, 2])
        lin = torch.where(ok, lin, torch.full_like(lin, -1)).to(torch.int32).contiguous()
        ctx.save_for_backward(lin)
        ctx.dims = (B, D, H, W)
        ctx.mark_non_differentiable(lin)
        return rows.view(B, H, W, D, -1).permute(0, 4, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        (lin,) = ctx.saved_tensors
        B, D, H, W = ctx.dims
        C = dout.shape[1]
        drows = dout.permute(0, 3, 4, 2, 1).reshape(B * H * W * D, C).float().contiguous()     # (b, x, y, z) rows
        n = lin.numel()
        dfeats = torch.empty(n, C, device=dout.device, dtype=_F32)
        if n:
            call("coocc_gather_rows", ptr(drows), C, ptr(lin, torch.int32), n, C, ptr(dfeats), C)
        return dfeats, None, None, None, None, None


def bev_pool(feats, coords, B, D, H, W):
    """``bev_pool(feats[n,c], coords[n,4] (x,y,z,b), B, D, H, W) -> [B,c,D,H,W]``
    (M/ops/bev_pool/bev_pool.py:83-97), differentiable in ``feats`` like the reference's autograd Function (:37-80).
    No argsort: sort-free CSR binning on the voxel key (csrc/pool.hip), each voxel sums its rows in ascending row index."""
    assert feats.shape[0] == coords.shape[0]
    B, D, H, W = int(B), int(D), int(H), int(W)
    if feats.requires_grad and torch.is_grad_enabled():
        return _BevPoolFn.apply(feats, coords, B, D, H, W)
    rows = bev_pool_rows(feats, coords, B, D, H, W)            # [(b,x,y,z), c]
    return rows.view(B, H, W, D, -1).permute(0, 4, 3, 1, 2)   # [B,c,D,H,W] view
