"""Volume-rendering regulariser (R1-R3, L1): ``MLP`` (P/utils/nerf_mlp.py:14-105), the inline
render block of ``COOCC_Ray`` (coocc_ray.py:570-627) and the library functions
``volume_sampling`` / ``raw2outputs`` / ``sample_along_camera_ray`` (P/utils/render_ray.py).
"""
import torch
from torch import nn

from ._lib import call, host_f32, ptr
from .core import PackCache, PackedConv, Rows, linear_rows

_F32 = torch.float32
RENDER_BOUNDS = [-50., 50., 1., -50., 50., 1., -5., 3., 1.0]     # hard-coded at coocc_ray.py:577
# camera chunks for overlapping the ALU-bound ray kernel with the store-bound upsample on a side stream.  Measured: the
# extra launches / events cost more host time than the overlap returns at 6 cameras (r101 pair 70 -> 138 us at 2 chunks)
RENDER_CHUNKS = int(__import__("os").environ.get("COOCC_RENDER_CHUNKS", "1"))
_render_sides = {}



_ZVALS = {}


def _zvals(D, dev):
    """linspace(0, D, D) of coocc_ray.py:581 -- a constant per (D, device): kept instead of one launch per sample."""
    key = (int(D), str(dev))
    if key not in _ZVALS:
        _ZVALS[key] = torch.linspace(0, D, D, device=dev)
    return _ZVALS[key]


def _render_side(dev, cur):
    key = (dev.index, cur.cuda_stream)
    if key not in _render_sides:
        _render_sides[key] = torch.cuda.Stream(device=dev, priority=getattr(cur, "priority", 0))
    return _render_sides[key]


class SinusoidalEncoder(nn.Module):
    """Only its ``scales`` buffer matters (state_dict key ``posi_encoder.scales``)."""

    def __init__(self, x_dim, min_deg, max_deg, use_identity=True):
        super().__init__()
        self.register_buffer("scales", torch.tensor([2 ** i for i in range(min_deg, max_deg)]))


class MLP(nn.Module):
    """nerf_mlp.py:14-105 with the hyper-parameters COOCC_Ray uses (skip_layer=None, ReLU,
    xavier init, zero bias).  ``forward`` accepts [..., input_dim] HIP tensors."""

    def __init__(self, input_dim, output_dim=None, net_depth=8, net_width=256, skip_layer=4, **kwargs):
        super().__init__()
        if skip_layer is not None:
            raise NotImplementedError("MLP skip connections are never instantiated on the Co-Occ path")
        self.input_dim, self.output_dim, self.net_depth, self.net_width = input_dim, output_dim, net_depth, net_width
        self.hidden_layers = nn.ModuleList()
        self.posi_encoder = SinusoidalEncoder(3, 0, 10, True)
        in_features = input_dim
        for _ in range(net_depth):
            self.hidden_layers.append(nn.Linear(in_features, net_width))
            in_features = net_width
        self.output_layer = nn.Linear(in_features, output_dim)
        for m in list(self.hidden_layers) + [self.output_layer]:
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
        self._packs = PackCache(self)

    def _packed(self):
        def build():
            return [PackedConv(l.weight, bias=l.bias) for l in list(self.hidden_layers) + [self.output_layer]]
        return self._packs.get_modules((self,), build)

    def forward_rows(self, x2d, out=None, out_coff=0, xh=None):
        """``xh``: an H2 copy of ``x2d`` made by the caller (``core.rows_to_h2``), shared between heads."""
        from . import core
        p = self._packed()
        n = x2d.shape[0]
        if (core.CONV_ENGINE == "h2" and core.CONV_DTYPE == "f32" and core.H2_DIRECT and n >= 8192 and self.input_dim % 32 == 0
                and self.net_width % 32 == 0):
            # split-f16 engine: every hidden layer's epilogue writes the next layer's H2 operand, the output layer (1 or 3
            # columns, padded to one 128-column tile in the pack) writes fp32 into the caller's table
            h = xh if xh is not None else core.rows_to_h2(x2d, self.input_dim, name="mlp_in")
            cin = self.input_dim
            for li, pc in enumerate(p[:-1]):
                nxt = core.scratch(x2d.device, "mlp_h%d" % (li & 1), n * pc.Cout)[:n * pc.Cout].view(n, pc.Cout)
                core.linear_rows_h2(h, n, cin, pc, relu=True, out=nxt, out_h2=True)
                h, cin = nxt, pc.Cout
            if out is None:
                out = torch.empty(n, p[-1].Cout, device=x2d.device, dtype=_F32)
            return core.linear_rows_h2(h, n, cin, p[-1], relu=False, out=out, out_coff=out_coff)
        for pc in p[:-1]:
            x2d = linear_rows(x2d, pc, relu=True)
        return linear_rows(x2d, p[-1], relu=False, out=out, out_coff=out_coff)

    def forward(self, x):
        shp = x.shape
        y = self.forward_rows(x.reshape(-1, shp[-1]).float().contiguous())
        return y.view(*shp[:-1], self.output_dim)


FUSED_HEADS = __import__("os").environ.get("COOCC_FUSED_RENDER_HEADS", "1") != "0"      # 0: layer-by-layer GEMM launches


def _fused_heads_args(sigma_head, rgb_head):
    """Device pointers of the one-launch render heads (csrc/mlp_h2.hip), or None when the heads do not have the shape it covers
    (COOCC_Ray's: input 128, width 256, sigma depth 1, rgb depth 1..4)."""
    import ctypes
    heads = [sigma_head] + ([rgb_head] if rgb_head is not None else [])
    if sigma_head.net_depth != 1 or any(h.input_dim != 128 or h.net_width != 256 for h in heads):
        return None
    if rgb_head is not None and not (1 <= rgb_head.net_depth <= 4 and rgb_head.output_dim == 3):
        return None
    if sigma_head.output_dim != 1:
        return None
    ps = sigma_head._packed()
    keep = [ps[0].h2_pack(), ps[0].bias, sigma_head.output_layer.weight.detach().float().contiguous(),
            sigma_head.output_layer.bias.detach().float().contiguous()]
    args = dict(ws0=keep[0], bs0=keep[1], ws1=keep[2], bs1=keep[3], n=0, wr=None, br=None, wro=None, bro=None)
    if rgb_head is not None:
        pr = rgb_head._packed()
        n = rgb_head.net_depth
        packs = [pr[l].h2_pack() for l in range(n)]
        biases = [pr[l].bias for l in range(n)]
        wro = rgb_head.output_layer.weight.detach().float().contiguous()
        bro = rgb_head.output_layer.bias.detach().float().contiguous()
        keep += packs + biases + [wro, bro]
        args.update(n=n, wr=(ctypes.c_void_p * n)(*[t.data_ptr() for t in packs]), br=(ctypes.c_void_p * n)(*[t.data_ptr() for t in biases]),
                    wro=wro, bro=bro)
    if any(t is None for t in keep) or any(t.data_ptr() % 16 for t in keep):
        return None
    args["keep"] = keep
    return args


def voxel_table(sigma_head, rgb_head, vf, activate=False):
    """R1 per voxel: [V,4] = (sigma_head(f), rgb_head(f)) raw outputs (pointwise heads, F5); ``rgb_head=None`` (depth-only
    branch) leaves the colour columns zero.  ``activate``: the colour columns hold sigmoid(logit) (the form the ray kernel
    reads; the backward path keeps raw logits)."""
    V = vf.t.shape[0]
    dev = vf.t.device
    from . import core
    h2 = core.CONV_ENGINE == "h2" and core.CONV_DTYPE == "f32" and core.H2_DIRECT and V >= 8192 and vf.C % 32 == 0
    whole = vf.coff == 0 and vf.stride == vf.C
    if h2 and FUSED_HEADS and vf.C == 128 and not sigma_head.training:
        # both heads in ONE launch (csrc/mlp_h2.hip): hidden activations stay in LDS, output layers are fp32 dot products
        fa = _cached_fused(sigma_head, rgb_head)
        if fa is not None:
            xh = core.h2_rows(vf) if whole else core.rows_to_h2(vf.t[:, vf.coff:vf.coff + vf.C].contiguous(), vf.C, name="mlp_in")
            table = torch.empty(V, 4, device=dev, dtype=_F32)
            with core.TIMER.region("k_render_heads_h2", 2.0 * V * (128 * 256 * (2 if rgb_head is not None else 1) + 256 +
                                                                 (256 * 256 * (rgb_head.net_depth - 1) + 768 if rgb_head is not None else 0))):
                call("coocc_render_heads_h2", ptr(xh), V, 128, 256, ptr(fa["ws0"]), ptr(fa["bs0"]), ptr(fa["ws1"]), ptr(fa["bs1"]),
                     fa["wr"], fa["br"], fa["n"], ptr(fa["wro"]), ptr(fa["bro"]), ptr(table), int(activate))
            return table
    table = torch.empty(V, 4, device=dev, dtype=_F32) if rgb_head is not None else torch.zeros(V, 4, device=dev, dtype=_F32)
    x = vf.t if whole else vf.t[:, vf.coff:vf.coff + vf.C].contiguous()
    xh = None
    if h2:
        # one H2 copy of the voxel features for both heads: the twin con_enc.3's epilogue wrote (shared with the encoder's
        # input_proj), or one conversion pass
        xh = core.h2_rows(vf) if whole else core.rows_to_h2(x, vf.C, name="mlp_in")
    sigma_head.forward_rows(x, out=table, out_coff=0, xh=xh)
    if rgb_head is not None:
        rgb_head.forward_rows(x, out=table, out_coff=1, xh=xh)
    if activate:
        call("coocc_render_activate_table", ptr(table), V)           # sigmoid(rgb) once per voxel, not per ray sample
    return table


def _cached_fused(sigma_head, rgb_head):
    """``_fused_heads_args`` cached on the sigma head, rebuilt when either head's weights change (their PackCache rebuilds)."""
    ps = sigma_head._packed()
    pr = rgb_head._packed() if rgb_head is not None else None
    c = getattr(sigma_head, "_fused_cache", None)
    if c is None or c[0] is not ps or c[1] is not pr:
        c = (ps, pr, _fused_heads_args(sigma_head, rgb_head))
        sigma_head._fused_cache = c
    return c[2]


def render_block(sigma_head, rgb_head, voxel_feats, gemo, scale=16, depth_only=False, cam_geo=None):
    """coocc_ray.py:570-627: voxel_feats Rows/[1,C,X,Y,Z], gemo [1,N,D,H,W,3] ->
    rgbs [N,16H,16W,3], depths [N,16H,16W] (+ the pre-upsample maps [N,H,W,4]).
    ``depth_only`` (or ``rgb_head=None``): the LiDAR-only branch (:436-484, geometry from ``get_frustum``) -- only the sigma
    head is evaluated and only the depth maps are written; ``rgbs`` is None.
    ``cam_geo`` = (mats [N,39], xs [W], ys [H], ds [D]) (``ViewTransformer._camera_mats``) instead of ``gemo``: the ray kernel
    evaluates get_geometry's chain in place (same bits), the [N,D,H,W,3] tensor is never read."""
    if depth_only or rgb_head is None:
        return _render_depth_only(sigma_head, voxel_feats, gemo, scale)
    from .core import to_rows
    vf = to_rows(voxel_feats)
    if cam_geo is not None:
        mats, xs, ys, ds = cam_geo
        N, D, H, W = mats.shape[0], ds.numel(), ys.numel(), xs.numel()
        g = None
        dev = mats.device
    else:
        B, N, D, H, W, _ = gemo.shape
        assert B == 1
        g = gemo.reshape(N, D, H, W, 3).float().contiguous()
        dev = g.device
    assert vf.B == 1
    table = voxel_table(sigma_head, rgb_head, vf, activate=True)      # sigmoid(rgb) once per voxel, not per ray sample
    zvals = _zvals(D, dev)
    maps = torch.empty(N, H, W, 4, device=dev, dtype=_F32)
    from .core import TIMER
    rgbs = torch.empty(N, H * scale, W * scale, 3, device=dev, dtype=_F32)
    depths = torch.empty(N, H * scale, W * scale, device=dev, dtype=_F32)
    # algorithmic HBM bytes: geom read + table read + small maps written, then small maps read +
    # upsampled maps written.  The ray kernel is ALU-bound (alpha, scan), the x16 upsample store-bound: cameras are
    # processed in chunks and the upsample of chunk c runs on a side stream under the rays of chunk c+1.
    chunks = max(1, min(RENDER_CHUNKS, N))
    per = -(-N // chunks)
    cur = torch.cuda.current_stream(dev)
    side = _render_side(dev, cur) if chunks > 1 else None
    by_rays = (12.0 * N * D * H * W if g is not None else 0.0) + 16.0 * vf.V + 16.0 * N * H * W
    by_up = 16.0 * N * H * W + 16.0 * N * H * W * scale * scale
    with TIMER.region("k_render_nearest+k_upsample_maps", by_rays + by_up):
        for c0 in range(0, N, per):
            nc = min(per, N - c0)
            if g is None:
                call("coocc_render_nearest_cams", ptr(table), vf.X, vf.Y, vf.Z, ptr(mats[c0:c0 + nc]), ptr(xs), ptr(ys), ptr(ds),
                     ptr(zvals), nc, D, H, W, host_f32(RENDER_BOUNDS), 1, ptr(maps[c0:c0 + nc]))
            else:
                call("coocc_render_nearest", ptr(table), vf.X, vf.Y, vf.Z, ptr(g[c0:c0 + nc]), ptr(zvals), nc, D, H, W,
                     host_f32(RENDER_BOUNDS), 1, ptr(maps[c0:c0 + nc]))
            if side is None:
                call("coocc_upsample_maps", ptr(maps[c0:c0 + nc]), nc, H, W, scale, ptr(rgbs[c0:c0 + nc]), ptr(depths[c0:c0 + nc]))
            else:
                ev = torch.cuda.Event()
                ev.record(cur)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    call("coocc_upsample_maps", ptr(maps[c0:c0 + nc]), nc, H, W, scale, ptr(rgbs[c0:c0 + nc]),
                         ptr(depths[c0:c0 + nc]))
        if side is not None:
            cur.wait_stream(side)
            for t in (maps, rgbs, depths):
                t.record_stream(side)
    return rgbs, depths, maps


def _render_depth_only(sigma_head, voxel_feats, gemo, scale):
    from .core import to_rows
    vf = to_rows(voxel_feats)
    B, N, D, H, W, _ = gemo.shape
    assert B == 1 and vf.B == 1
    table = voxel_table(sigma_head, None, vf)
    g = gemo.reshape(N, D, H, W, 3).float().contiguous()
    dev = g.device
    zvals = _zvals(D, dev)
    maps = torch.empty(N, H, W, 4, device=dev, dtype=_F32)
    depths = torch.empty(N, H * scale, W * scale, device=dev, dtype=_F32)
    call("coocc_render_nearest", ptr(table), vf.X, vf.Y, vf.Z, ptr(g), ptr(zvals), N, D, H, W, host_f32(RENDER_BOUNDS), 1, ptr(maps))
    call("coocc_upsample_maps", ptr(maps), N, H, W, scale, None, ptr(depths))
    return None, depths, maps


def render_block_sharded(sigma_head, rgb_head, voxel_feats, gemo, scale=16, rank=None, world=None, events=None):
    """Config 5: ONE scene rendered by `world` ranks.  The flattened (camera, row) space of the
    feature-map rays is cut into `world` contiguous chunks (6 cameras do not divide 8 GPUs); every rank holds the
    fused volume, evaluates the per-voxel table, composites its chunk of rays, the 16-byte-per-ray maps are
    all-gathered (dist.gather_ray_shards, RCCL) and every rank upsamples the full maps (35 us at r101).
    Returns the same (rgbs, depths, maps) as render_block.  ``events``: a list that receives four timing events (start, before
    the gather, after the gather, end) so a caller can separate the sharded compute from the collective (which also carries
    the wait for the slowest rank)."""
    from . import dist as cdist
    from .core import to_rows
    import torch.distributed as tdist
    vf = to_rows(voxel_feats)
    B, N, D, H, W, _ = gemo.shape
    assert B == 1 and vf.B == 1
    if world is None:
        world = tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1
        rank = tdist.get_rank() if world > 1 else 0
    dev = gemo.device

    def mark():
        if events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            events.append(e)

    mark()
    table = voxel_table(sigma_head, rgb_head, vf, activate=True)
    zvals = _zvals(D, dev)
    lo, hi = cdist.shard_range(N * H, rank, world)
    g = gemo.reshape(N, D, H, W, 3).float()
    local = torch.empty(hi - lo, W, 4, device=dev, dtype=_F32)
    r = lo
    while r < hi:                                   # one launch per camera the chunk touches
        cam, h0 = divmod(r, H)
        h1 = min(H, h0 + (hi - r))
        piece = g[cam:cam + 1, :, h0:h1].contiguous()            # [1, D, rows, W, 3]
        call("coocc_render_nearest", ptr(table), vf.X, vf.Y, vf.Z, ptr(piece), ptr(zvals), 1, D, h1 - h0, W,
             host_f32(RENDER_BOUNDS), 1, ptr(local[r - lo:r - lo + (h1 - h0)]))
        r += h1 - h0
    mark()
    maps = cdist.gather_ray_shards(local, N * H).view(N, H, W, 4).contiguous()
    mark()
    rgbs = torch.empty(N, H * scale, W * scale, 3, device=dev, dtype=_F32)
    depths = torch.empty(N, H * scale, W * scale, device=dev, dtype=_F32)
    call("coocc_upsample_maps", ptr(maps), N, H, W, scale, ptr(rgbs), ptr(depths))
    mark()
    return rgbs, depths, maps


def render_losses(rgbs, depths, rgb_gt, depth_gt, D):
    """coocc_ray.py:423-433 -> dict(loss_depth_render, loss_rgb) (forward values)."""
    out = torch.empty(3, device=rgbs.device, dtype=_F32)
    from .core import workspace
    ws = workspace(rgbs.device)
    call("coocc_render_losses", ptr(rgbs.contiguous()), ptr(depths.contiguous()), ptr(rgb_gt.float().contiguous()),
         ptr(depth_gt.float().contiguous()), depths.numel(), int(D), ptr(out), ptr(ws), ws.numel() * 4)
    return dict(loss_depth_render=out[0], loss_rgb=out[1])


def volume_sampling(sample_pts, features, aabb):
    """render_ray.py:28-48: features [1,C,D,W,H], sample_pts [N_rays,N_samples,3] ->
    ([N_rays,N_samples,C], mask [N_rays,N_samples])."""
    from .core import to_rows
    B, C, d0, d1, d2 = features.shape
    assert B == 1
    vol = to_rows(features)
    nr, ns, _ = sample_pts.shape
    pts = sample_pts.reshape(-1, 3).float().contiguous()
    feat = torch.empty(nr * ns, C, device=pts.device, dtype=_F32)
    mask = torch.empty(nr * ns, device=pts.device, dtype=torch.uint8)
    a = [float(v) for v in aabb[0]] + [float(v) for v in aabb[1]]
    call("coocc_volume_sampling", ptr(vol.t), C, d0, d1, d2, ptr(pts), nr * ns, host_f32(a), ptr(feat), ptr(mask))
    return feat.view(nr, ns, C), mask.view(nr, ns).bool()


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False, det=False):
    """render_ray.py:147-191 (host-side torch: [N_rays, N_samples] arithmetic, no kernel needed)."""
    near, far = depth_range
    assert near > 0 and far > 0 and far > near
    near_d = near * torch.ones_like(ray_d[..., 0])
    far_d = far * torch.ones_like(ray_d[..., 0])
    if inv_uniform:
        start = 1. / near_d
        step = (1. / far_d - start) / (N_samples - 1)
        z_vals = 1. / torch.stack([start + i * step for i in range(N_samples)], dim=1)
    else:
        step = (far_d - near_d) / (N_samples - 1)
        z_vals = torch.stack([near_d + i * step for i in range(N_samples)], dim=1)
    if not det:
        mids = .5 * (z_vals[:, 1:] + z_vals[:, :-1])
        upper = torch.cat([mids, z_vals[:, -1:]], dim=-1)
        lower = torch.cat([z_vals[:, 0:1], mids], dim=-1)
        z_vals = lower + (upper - lower) * torch.rand_like(z_vals)
    pts = z_vals.unsqueeze(2) * ray_d.unsqueeze(1) + ray_o.unsqueeze(1)
    return pts, z_vals


def raw2outputs(raw, z_vals, mask=None, white_bkgd=False):
    """render_ray.py:198-249 -> dict(rgb, depth, weights, mask, z_vals)."""
    R, S, _ = raw.shape
    raw = raw.float().contiguous()
    z = z_vals.float().contiguous()
    dev = raw.device
    rgb = torch.empty(R, 3, device=dev, dtype=_F32)
    depth = torch.empty(R, device=dev, dtype=_F32)
    weights = torch.empty(R, S, device=dev, dtype=_F32)
    zmin, zmax = (float(z.min()), float(z.max())) if R else (0.0, 0.0)
    call("coocc_raw2outputs", ptr(raw), ptr(z), R, S, int(white_bkgd), zmin, zmax, ptr(rgb), ptr(depth), ptr(weights))
    if mask is not None:
        mask = mask.float().sum(dim=1) > 8
    return dict(rgb=rgb, depth=depth, weights=weights, mask=mask, z_vals=z_vals)
