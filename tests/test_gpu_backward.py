"""backward kernels vs torch autograd of the same op in plain fp32 on the CPU
(the floating-point reference the brief allows for float kernels; the reference repo has no backward
kernels of its own -- it relies on torch autograd through cudnn)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import co_occ_amd as pkg
from co_occ_amd import autograd as ag
from util import assert_close

pytestmark = pytest.mark.gpu


def _rows(x):       # [B,C,X,Y,Z] -> [B*X*Y*Z, C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def _vol(r, B, X, Y, Z):
    return r.view(B, X, Y, Z, -1).permute(0, 4, 1, 2, 3)


CONV_CASES = [
    # Cin, Cout, grid, k, stride, bn, bias, res, relu
    (8, 12, (6, 5, 4), 3, 1, True, False, True, True),
    (16, 8, (7, 6, 5), 3, 2, True, False, False, True),
    (12, 20, (5, 4, 3), 1, 1, False, True, False, False),
    (64, 17, (4, 4, 2), 1, 1, False, True, False, False),      # class logits: Cout % 4 != 0
    (132, 136, (9, 8, 4), 3, 1, True, True, True, True),       # crosses the 128-wide tiles
    (32, 160, (10, 9, 3), 3, 2, False, False, False, True),
    (16, 24, (7, 6, 1), 3, 1, True, False, True, True),        # Z = 1: only the centre z tap is live (fwd, dgrad, wgrad)
    (16, 24, (7, 6, 2), 3, 2, True, False, False, True),       # Z = 2 -> 1: z taps 1..2 live
    (20, 12, (5, 6, 2), 3, 1, False, True, False, False),      # Z = 2, stride 1: all taps live
]


WINO_CASES = [
    (8, 12, (6, 5, 4), 3, 1, True, False, True, True),          # F(2x2): X, Y < 8
    (132, 136, (9, 8, 4), 3, 1, True, True, True, True),        # F(4x4), ragged tiles, both channel counts cross 128
    (32, 48, (13, 12, 3), 3, 1, False, False, False, False),    # F(4x4), no epilogue at all
]


@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "c%d-%d_%dx%d" % (c[0], c[1], c[2][0], c[2][1]))
def test_conv_backward_winograd_training_path(dev, case, monkeypatch):
    """Forward and dgrad of the training path through the Winograd kernels with device-made weight packs
    (coocc_wino_pack_weights_dev; production switches at core.WINO_MIN_ROWS output rows), wgrad direct."""
    from co_occ_amd import core
    monkeypatch.setattr(core, "WINO_MIN_ROWS", 0)
    monkeypatch.setattr(core, "WINO", 1)
    seen = []
    real = core.conv_rows_wino
    monkeypatch.setattr(core, "conv_rows_wino", lambda *a, **k: (seen.append(a[5][0]), real(*a, **k))[1])
    _run_conv_case(dev, case)
    assert len(seen) == 2 and seen[0] == (4 if min(case[2][:2]) >= 8 else 2), seen      # forward + dgrad


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d-%d_k%d_s%d_z%d" % (c[0], c[1], c[3], c[4], c[2][2]))
def test_conv_backward_vs_torch_autograd(dev, case):
    _run_conv_case(dev, case)


def _run_conv_case(dev, case):
    Cin, Cout, (X, Y, Z), k, stride, use_bn, use_bias, use_res, relu = case
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    B = 2
    x = torch.randn(B, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    bn = torch.nn.BatchNorm3d(Cout).eval()
    bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
    bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(Cout, generator=g) * 0.1)
    pad = k // 2
    Xo, Yo, Zo = ((n + 2 * pad - k) // stride + 1 for n in (X, Y, Z))
    res = torch.randn(B, Cout, Xo, Yo, Zo, generator=g) if use_res else None
    gout = torch.randn(B, Cout, Xo, Yo, Zo, generator=g)

    # fp32 reference on the CPU
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if use_bias else None
    rr = res.clone().requires_grad_() if use_res else None
    y = F.conv3d(xr, wr, br, stride=stride, padding=pad)
    if use_bn:
        y = bn(y)
    if use_res:
        y = y + rr
    if relu:
        y = F.relu(y)
    y.backward(gout)

    xd = _rows(x).to(dev).requires_grad_()
    wd = w.to(dev).requires_grad_()
    bd = b.to(dev).requires_grad_() if use_bias else None
    rd = _rows(res).to(dev).requires_grad_() if use_res else None
    out, geom = ag.conv3d_rows(xd, wd, (B, X, Y, Z), bias=bd, bn=bn if use_bn else None, stride=stride, relu=relu, res2d=rd)
    assert geom == (B, Xo, Yo, Zo)
    assert_close(_vol(out.detach().cpu(), B, Xo, Yo, Zo), y.detach(), what="forward")
    out.backward(_rows(gout).to(dev))
    assert_close(_vol(xd.grad.cpu(), B, X, Y, Z), xr.grad, what="dgrad")
    assert_close(wd.grad.cpu(), wr.grad, what="wgrad")
    if use_bias:
        assert_close(bd.grad.cpu(), br.grad, what="dbias")
    if use_res:
        assert_close(_vol(rd.grad.cpu(), B, Xo, Yo, Zo), rr.grad, what="dres")


def test_linear_backward_and_chain(dev):
    """Two stacked layers (Linear+ReLU -> Linear): gradients flow through both Functions."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(300, 128, generator=g)
    w1, b1 = torch.randn(64, 128, generator=g) / 11, torch.randn(64, generator=g)
    w2, b2 = torch.randn(4, 64, generator=g) / 8, torch.randn(4, generator=g)
    ref = [t.clone().requires_grad_() for t in (x, w1, b1, w2, b2)]
    yr = F.linear(F.relu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    yr.square().sum().backward()
    got = [t.to(dev).requires_grad_() for t in (x, w1, b1, w2, b2)]
    y = ag.linear_rows(ag.linear_rows(got[0], got[1], got[2], relu=True), got[3], got[4])
    assert_close(y.detach().cpu(), yr.detach(), what="forward")
    y.square().sum().backward()
    for a, b, name in zip(got, ref, ("dx", "dw1", "db1", "dw2", "db2")):
        assert_close(a.grad.cpu(), b.grad, what=name)


def test_gather_rows_backward(dev):
    g = torch.Generator().manual_seed(3)
    src = torch.randn(50, 16, generator=g)
    idx = torch.randint(-1, 50, (200,), generator=g).int()
    gout = torch.randn(200, 16, generator=g)
    sr = src.clone().requires_grad_()
    ref = torch.where(idx[:, None] >= 0, sr[idx.clamp(min=0).long()], torch.zeros(()))
    ref.backward(gout)
    sd = src.to(dev).requires_grad_()
    out = ag.gather_rows(sd, idx.to(dev))
    assert torch.equal(out.detach().cpu(), ref.detach())
    out.backward(gout.to(dev))
    assert_close(sd.grad.cpu(), sr.grad, what="gather dsrc")


def _pool_ref(vt_cfg, geom, vol_rows, B):
    """voxel_pooling with torch ops only (differentiable): truncate-then-filter keys + index_add."""
    lo = torch.tensor([vt_cfg["xbound"][0], vt_cfg["ybound"][0], vt_cfg["zbound"][0]])
    dx = torch.tensor([vt_cfg["xbound"][2], vt_cfg["ybound"][2], vt_cfg["zbound"][2]])
    nx = [int(round((vt_cfg[k][1] - vt_cfg[k][0]) / vt_cfg[k][2])) for k in ("xbound", "ybound", "zbound")]
    q = ((geom.reshape(-1, 3) - lo) / dx).long()
    kept = ((q >= 0) & (q < torch.tensor(nx))).all(1)
    b = torch.arange(q.shape[0]) // (q.shape[0] // B)
    key = ((b * nx[0] + q[:, 0]) * nx[1] + q[:, 1]) * nx[2] + q[:, 2]
    out = torch.zeros(B * nx[0] * nx[1] * nx[2], vol_rows.shape[1])
    return out.index_add(0, key[kept], vol_rows[kept]), nx


def test_lift_splat_and_voxel_pool_backward(dev):
    from oracle import cases, ref_cpu
    import co_occ_amd.synth as synth
    c = cases.POOL_CASE
    rig = synth.camera_rig(c["ncam"], c["input_size"], seed=c["seed"])
    vt = pkg.ViewTransformerLiftSplatShootVoxel(grid_config=c["grid_config"], data_config=dict(input_size=c["input_size"]),
                                                downsample=c["downsample"], numC_Trans=c["C"]).to(dev)
    fr = ref_cpu.create_frustum(c["input_size"], c["downsample"], c["grid_config"]["dbound"])
    geom = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    g = torch.Generator().manual_seed(8)
    N, D, (H, W), C = c["ncam"], fr.shape[0], c["fmap"], 12
    depth = torch.softmax(torch.randn(N, D, H, W, generator=g) * 2, dim=1)
    feat = torch.randn(N, C, H, W, generator=g)
    # reference
    dr, fr_ = depth.clone().requires_grad_(), feat.clone().requires_grad_()
    vol = (dr.unsqueeze(1) * fr_.unsqueeze(2)).permute(0, 2, 3, 4, 1).reshape(-1, C)     # rows in (n,d,h,w) order
    ref, nx = _pool_ref(c["grid_config"], geom, vol, 1)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    # fused
    dd, fd = depth.to(dev).requires_grad_(), feat.to(dev).requires_grad_()
    out = ag.lift_splat(vt, dd, fd, geom.to(dev))
    rows = out.permute(0, 2, 3, 4, 1).reshape(-1, C)
    assert_close(rows.detach().cpu(), ref.detach(), what="lift_splat fwd")
    rows.backward(gout.to(dev))
    assert_close(dd.grad.cpu(), dr.grad, what="d depth")
    assert_close(fd.grad.cpu(), fr_.grad, what="d feat")
    # plain voxel pooling of a materialised volume: dx[p] = dout[voxel(p)]
    vol_d = vol.detach().view(1, N, D, H, W, C).to(dev).requires_grad_()
    vr = vol.detach().clone().requires_grad_()
    ref2, _ = _pool_ref(c["grid_config"], geom, vr, 1)
    ref2.backward(gout)
    out2 = ag.voxel_pooling(vt, geom.to(dev), vol_d)
    out2.permute(0, 2, 3, 4, 1).reshape(-1, C).backward(gout.to(dev))
    assert torch.equal(vol_d.grad.reshape(-1, C).cpu(), vr.grad)


def _render_ref(sig_pre, rgb_logit, gemo, grid, scale):
    """coocc_ray.py:574-625 in its per-voxel-table form, plain torch on the CPU (differentiable)."""
    X, Y, Z = grid
    N, D, H, W, _ = gemo.shape
    lo, nx = torch.tensor([-50., -50., -5.]), torch.tensor([100., 100., 8.])
    g = gemo - lo
    inside = ((g >= 0) & (g < nx)).all(-1)
    pts = (g * inside[..., None]).long()
    lin = (pts[..., 0] * Y + pts[..., 1]) * Z + pts[..., 2]
    rgb = torch.sigmoid(rgb_logit[lin] * inside[..., None])
    sigma = F.relu(sig_pre[lin])
    pf = pts.float()
    dists = torch.norm(pf[:, 1:] - pf[:, :-1], dim=-1)
    dists = torch.cat([dists, torch.full_like(dists[:, :1], 1e10)], 1)
    alpha = 1. - torch.exp(-F.relu(sigma * dists))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], 1), 1)[:, :-1]
    w = alpha * T
    rgb_map = (w[..., None] * rgb).sum(1)                               # [N,H,W,3]
    z = torch.linspace(0, D, D).view(1, D, 1, 1)
    depth_map = (w * z).sum(1)
    rgbs = F.interpolate(rgb_map.permute(0, 3, 1, 2), scale_factor=scale, mode='bilinear').permute(0, 2, 3, 1)
    depths = F.interpolate(depth_map.unsqueeze(1), scale_factor=scale, mode='bilinear').squeeze(1)
    return rgbs, depths


def test_render_block_backward(dev):
    """R2 + x16 upsample + render losses: d table through the composite (prefix product / suffix sum) and the
    bilinear adjoint, against torch autograd of the same formulas."""
    g = torch.Generator().manual_seed(21)
    grid, (N, D, H, W), scale = (100, 100, 8), (2, 24, 3, 5), 16
    V = grid[0] * grid[1] * grid[2]
    sig = torch.randn(V, generator=g) * 0.3
    rgbl = torch.randn(V, 3, generator=g)
    # rays marching outwards through the volume, some samples leaving it
    o = torch.tensor([0., 0., -1.]) + torch.randn(N, 1, H, W, 3, generator=g) * 0.3
    d = torch.randn(N, 1, H, W, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True) * torch.tensor([1., 1., 0.15])
    t = torch.linspace(0.5, 70, D).view(1, D, 1, 1, 1)
    gemo = (o + d * t).contiguous()
    rgb_gt = torch.rand(N, H * scale, W * scale, 3, generator=g)
    depth_gt = torch.rand(N, H * scale, W * scale, generator=g) * 60
    depth_gt[depth_gt < 10] = 0     # background pixels
    sr, rr = sig.clone().requires_grad_(), rgbl.clone().requires_grad_()
    rgbs_r, depths_r = _render_ref(sr, rr, gemo, grid, scale)
    dg = ((depth_gt - 1.75) / 0.5).clip(0, D)
    fg = dg > 0
    loss_r = F.mse_loss(depths_r[fg] / D, dg[fg] / D) * 0.7 + F.mse_loss(rgbs_r, rgb_gt) * 1.3
    loss_r.backward()

    table = torch.cat([sig[:, None], rgbl], 1).to(dev).requires_grad_()
    maps = ag.RenderNearestFn.apply(table, gemo.to(dev), grid)
    rgbs, depths = ag.UpsampleMapsFn.apply(maps, scale)
    assert_close(rgbs.detach().cpu(), rgbs_r.detach(), what="rgbs")
    assert_close(depths.detach().cpu(), depths_r.detach(), what="depths")
    L = ag.render_losses(rgbs, depths, rgb_gt.to(dev), depth_gt.to(dev), D)
    loss = L["loss_depth_render"] * 0.7 + L["loss_rgb"] * 1.3
    assert_close(loss.detach().cpu(), loss_r.detach(), what="loss")
    loss.backward()
    want = torch.cat([sr.grad[:, None], rr.grad], 1)
    assert float(want.abs().max()) > 0
    assert_close(table.grad.cpu() / want.abs().max(), want / want.abs().max(), what="d table")


def test_render_block_train_reaches_features_and_heads(dev):
    """render_block_train: gradients flow through the per-voxel MLP heads (ConvRowsFn) to the voxel features."""
    g = torch.Generator().manual_seed(5)
    grid, (N, D, H, W) = (100, 100, 8), (1, 16, 2, 4)
    V, C = 80000, 32
    feats = (torch.randn(V, C, generator=g) * 0.5)
    sig_head = pkg.MLP(input_dim=C, output_dim=1, net_depth=1, net_width=16, skip_layer=None)
    rgb_head = pkg.MLP(input_dim=C, output_dim=3, net_depth=2, net_width=16, skip_layer=None)
    for m in (sig_head, rgb_head):
        for p in m.parameters():
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.2)
    o = torch.tensor([0., 0., -1.]) + torch.randn(N, 1, H, W, 3, generator=g) * 0.3
    d = torch.randn(N, 1, H, W, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True) * torch.tensor([1., 1., 0.15])
    gemo = (o + d * torch.linspace(0.5, 40, D).view(1, D, 1, 1, 1)).contiguous()

    def mlp_ref(m, x):
        for l in m.hidden_layers:
            x = F.relu(F.linear(x, l.weight, l.bias))
        return F.linear(x, m.output_layer.weight, m.output_layer.bias)
    fr = feats.clone().requires_grad_()
    rgbs_r, depths_r = _render_ref(mlp_ref(sig_head, fr)[:, 0], mlp_ref(rgb_head, fr), gemo, grid, 16)
    (rgbs_r.square().mean() + depths_r.mean()).backward()
    ref_grads = [p.grad.clone() for p in list(sig_head.parameters()) + list(rgb_head.parameters())]
    for p in list(sig_head.parameters()) + list(rgb_head.parameters()):
        p.grad = None
    sig_head, rgb_head = sig_head.to(dev), rgb_head.to(dev)
    fd = feats.to(dev).requires_grad_()
    rgbs, depths = ag.render_block_train(sig_head, rgb_head, fd, grid, gemo.to(dev)[None])
    (rgbs.square().mean() + depths.mean()).backward()
    s = float(fr.grad.abs().max())
    assert s > 0
    assert_close(fd.grad.cpu() / s, fr.grad / s, what="d voxel feats")
    for p, r in zip(list(sig_head.parameters()) + list(rgb_head.parameters()), ref_grads):
        sc = max(float(r.abs().max()), 1e-12)
        assert_close(p.grad.cpu() / sc, r / sc, what="d head params")


@pytest.mark.parametrize("conv_path", ["direct", "winograd"])
def test_decoder_trunk_backward_vs_oracle_autograd(dev, monkeypatch, conv_path):
    """con_enc -> CustomResNet3D(18) -> FPN3D: every conv weight gradient and the input gradient of the
    differentiable trunk vs torch autograd through the oracle's restatement (frozen BN statistics).
    conv_path=winograd forces forward, dgrad and wgrad of every 3x3x3 stride-1 layer through the Winograd kernels
    (production does so from core.WINO_MIN_ROWS output rows up)."""
    from oracle import ref_cpu
    from co_occ_amd import core
    monkeypatch.setattr(core, "WINO_MIN_ROWS", 0 if conv_path == "winograd" else 1 << 30)
    import co_occ_amd.synth as synth
    C, planes, fpn_out, grid = 8, [16, 32, 64, 128], 32, (12, 10, 4)
    bn = dict(type="BN3d")           # eps 1e-5, the value the oracle restates
    fuser = pkg.BiFuser_N(C, C, knum=2).eval()
    enc = pkg.CustomResNet3D(depth=18, block_inplanes=planes, n_input_channels=C, norm_cfg=bn).eval()
    neck = pkg.FPN3D(in_channels=planes, out_channels=fpn_out, norm_cfg=bn).eval()
    for i, m in enumerate((fuser, enc, neck)):
        m.load_state_dict(synth.random_state_dict(m.state_dict(), seed=70 + i))
    g = torch.Generator().manual_seed(77)
    X, Y, Z = grid
    x = torch.randn(1, X, Y, Z, 4 * C, generator=g)

    # oracle with autograd: leaf copies of every tensor
    def leaf(sd):
        return {k: (v.clone().float().requires_grad_() if (v.is_floating_point() and "running_" not in k) else v.clone())
                for k, v in sd.items()}
    sd_f, sd_e, sd_n = leaf(fuser.state_dict()), leaf(enc.state_dict()), leaf(neck.state_dict())
    xr = x.clone().requires_grad_()
    outs_r = ref_cpu.fpn3d_forward(sd_n, ref_cpu.resnet3d_forward(sd_e, ref_cpu.con_enc(sd_f, xr)))
    gouts = [torch.randn(o.shape, generator=g) for o in outs_r]
    sum((o * go).sum() for o, go in zip(outs_r, gouts)).backward()

    fuser, enc, neck = fuser.to(dev), enc.to(dev), neck.to(dev)
    xd = x.reshape(-1, 4 * C).to(dev).requires_grad_()
    outs = ag.trunk_forward_train(fuser.con_enc, enc, neck, xd, (1, X, Y, Z))
    loss = 0
    for (o, geo), o_r, go in zip(outs, outs_r, gouts):
        assert tuple(geo[1:]) == tuple(o_r.shape[2:])
        assert_close(_vol(o.detach().cpu(), *geo), o_r.detach(), what="trunk forward")
        loss = loss + (o * _rows(go).to(dev)).sum()
    loss.backward()
    assert_close(xd.grad.cpu().view(1, X, Y, Z, 4 * C), xr.grad, what="d input")
    n = 0
    for mod, sd in ((fuser, sd_f), (enc, sd_e), (neck, sd_n)):
        for name, p in mod.named_parameters():
            if p.dim() == 5:                  # conv weights (BN affine parameters are frozen)
                assert p.grad is not None, name
                assert_close(p.grad.cpu(), sd[name].grad, what=name)
                n += 1
    assert n == 2 + 1 + 16 + 3 + 8           # con_enc, input_proj, 8 blocks x 2, 3 downsamples, FPN 4 + 4


def test_decoder_with_coarse_head_backward_vs_oracle_autograd(dev):
    """con_enc -> ResNet3D -> FPN3D -> OccHead coarse branch (3x3x3 per level, softmax-weighted multi-level trilinear mix,
    1x1 prediction stack): logits and every conv weight gradient vs torch autograd through the oracle."""
    from oracle import ref_cpu
    import co_occ_amd.synth as synth
    C, planes, fpn_out, grid = 8, [16, 32, 64, 128], 32, (16, 12, 8)
    cfg = synth.model_cfg(C=C, block_inplanes=planes, out_channels=fpn_out)
    bn = dict(type="BN3d")
    fuser = pkg.BiFuser_N(C, C, knum=2).eval()
    enc = pkg.CustomResNet3D(depth=18, block_inplanes=planes, n_input_channels=C, norm_cfg=bn).eval()
    neck = pkg.FPN3D(in_channels=planes, out_channels=fpn_out, norm_cfg=bn).eval()
    hcfg = dict(cfg["pts_bbox_head"], norm_cfg=bn, sample_from_img=False, sample_from_voxel=False, cascade_ratio=1)
    head = pkg.build_head(hcfg).eval()
    for i, m in enumerate((fuser, enc, neck, head)):
        m.load_state_dict(synth.random_state_dict(m.state_dict(), seed=80 + i))
    g = torch.Generator().manual_seed(81)
    X, Y, Z = grid
    x = torch.randn(1, X, Y, Z, 4 * C, generator=g)

    def leaf(sd):
        return {k: (v.clone().float().requires_grad_() if (v.is_floating_point() and "running_" not in k) else v.clone())
                for k, v in sd.items()}
    sds = [leaf(m.state_dict()) for m in (fuser, enc, neck, head)]
    xr = x.clone().requires_grad_()
    sem_r = ref_cpu.fpn3d_forward(sds[2], ref_cpu.resnet3d_forward(sds[1], ref_cpu.con_enc(sds[0], xr)))
    hr = ref_cpu.occhead_coarse(sds[3], sem_r, soft_weights=True)
    gocc = torch.randn(hr["occ"].shape, generator=g)
    gfeat = torch.randn(hr["out_voxel_feats"].shape, generator=g) * 0.1
    ((hr["occ"] * gocc).sum() + (hr["out_voxel_feats"] * gfeat).sum()).backward()

    mods = [m.to(dev) for m in (fuser, enc, neck, head)]
    xd = x.reshape(-1, 4 * C).to(dev).requires_grad_()
    feats = ag.trunk_forward_train(mods[0].con_enc, mods[1], mods[2], xd, (1, X, Y, Z))
    ovf, occ = ag.occhead_coarse_train(mods[3], feats)
    assert_close(_vol(occ.detach().cpu(), 1, X, Y, Z), hr["occ"].detach(), what="coarse logits")
    assert_close(_vol(ovf.detach().cpu(), 1, X, Y, Z), hr["out_voxel_feats"].detach(), what="out_voxel_feats")
    ((occ * _rows(gocc).to(dev)).sum() + (ovf * _rows(gfeat).to(dev)).sum()).backward()
    assert_close(xd.grad.cpu().view(1, X, Y, Z, 4 * C), xr.grad, what="d input")
    n = 0
    for mod, sd in zip(mods, sds):
        for name, p in mod.named_parameters():
            if p.dim() == 5 and sd[name].grad is not None:
                assert p.grad is not None, name
                sc = max(1.0, float(sd[name].grad.abs().max()))
                assert_close(p.grad.cpu() / sc, sd[name].grad / sc, what=name)
                n += 1
    assert n == 30 + 4 + 2 + 2           # trunk, 4 occ_convs, 2 soft-weight convs, 2 prediction convs


def test_groupnorm_rows_backward(dev):
    g = torch.Generator().manual_seed(17)
    n, C, groups = 333, 64, 16
    x = torch.randn(n, C, generator=g)
    gn = torch.nn.GroupNorm(groups, C)
    gn.weight.data.copy_(torch.rand(C, generator=g) + 0.5)
    gn.bias.data.copy_(torch.randn(C, generator=g) * 0.2)
    gout = torch.randn(n, C, generator=g)
    xr = x.clone().requires_grad_()
    F.relu(gn(xr)).backward(gout)
    xd = x.to(dev).requires_grad_()
    wd, bd = gn.weight.detach().clone().to(dev).requires_grad_(), gn.bias.detach().clone().to(dev).requires_grad_()
    y = ag.GroupNormRowsFn.apply(xd, wd, bd, groups, gn.eps, True)
    assert_close(y.detach().cpu(), F.relu(gn(x)).detach(), what="gn forward")
    y.backward(gout.to(dev))
    assert_close(xd.grad.cpu(), xr.grad, what="gn dx")
    assert_close(wd.grad.cpu(), gn.weight.grad, what="gn dgamma")
    assert_close(bd.grad.cpu(), gn.bias.grad, what="gn dbeta")


def test_fine_branch_backward_vs_torch(dev):
    """Voxel-only fine branch: trilinear grid_sample adjoint (atomics) + Linear + GroupNorm + ReLU + Linear against torch
    autograd of F.grid_sample (align_corners=False, zeros) with the reference's coordinate normalisation."""
    import co_occ_amd.synth as synth
    g = torch.Generator().manual_seed(23)
    C, (X, Y, Z), r = 128, (6, 5, 4), 2
    cfg = synth.model_cfg(C=8, block_inplanes=(16, 32, 64, 128), out_channels=256, final_occ_size=(X * r, Y * r, Z * r))
    head = pkg.build_head(dict(cfg["pts_bbox_head"], norm_cfg=dict(type="BN3d"), sample_from_img=False)).eval()
    head.load_state_dict(synth.random_state_dict(head.state_dict(), seed=3))
    vol = torch.randn(1, C, X, Y, Z, generator=g)
    sel = torch.randperm(X * Y * Z, generator=g)[:37].sort().values.int()
    gout = torch.randn(37 * r ** 3, 17, generator=g)
    # torch reference (occ_head.py:199-214): fine coords, normalise by (final - 1), grid_sample on the permuted volume
    vr = vol.clone().requires_grad_()
    cz, cy, cx = sel % Z, (sel // Z) % Y, sel // (Y * Z)
    offs = torch.stack(torch.meshgrid(torch.arange(r), torch.arange(r), torch.arange(r), indexing="ij"), -1).view(-1, 3)
    fine = torch.cat([torch.stack([cx * r + o[0], cy * r + o[1], cz * r + o[2]], 0) for o in offs], 1)     # [3, r^3 n], offset-major
    norm = (fine.float() / torch.tensor([X * r - 1, Y * r - 1, Z * r - 1.0]).view(3, 1) - 0.5) * 2
    grid = norm.t().view(1, 1, 1, -1, 3)
    samp = F.grid_sample(vr.permute(0, 1, 4, 3, 2), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    feat_r = samp[0, :, 0, 0].t()
    fm = head.fine_mlp
    out_r = fm[3](F.relu(fm[1](fm[0](feat_r))))
    out_r.backward(gout)
    ref_grads = [p.grad.clone() for p in fm.parameters()]
    for p in fm.parameters():
        p.grad = None
    head = head.to(dev)
    vd = _rows(vol).to(dev).requires_grad_()
    out, fine_xyz = ag.fine_branch_train(head, vd, (1, X, Y, Z), sel.to(dev))
    assert torch.equal(fine_xyz.cpu(), fine.long())
    assert_close(out.detach().cpu(), out_r.detach(), what="fine logits")
    out.backward(gout.to(dev))
    assert_close(_vol(vd.grad.cpu(), 1, X, Y, Z), vr.grad, what="d out_voxel_feats")
    for p, rg in zip(head.fine_mlp.parameters(), ref_grads):
        sc = max(1.0, float(rg.abs().max()))
        assert_close(p.grad.cpu() / sc, rg / sc, what="fine_mlp grads")


def test_full_head_backward_vs_oracle_autograd(dev):
    """OccHead coarse + fine (voxel and camera sampling) from FPN levels to fine logits: gradients w.r.t. the level inputs,
    the image features and every head parameter vs torch autograd through the oracle (occ_head.py:149-237)."""
    from oracle import cases, ref_cpu
    import co_occ_amd.synth as synth
    from co_occ_amd._lib import call as ccall, ptr as cptr
    c = cases.DECODER_CASE
    grid = (8, 6, 4)
    final = tuple(2 * v for v in grid)
    # the head hard-codes 128 voxel channels into fine_mlp (occ_head.py:96), i.e. FPN levels of 256 channels
    cfg = synth.model_cfg(C=c["C"], block_inplanes=c["block_inplanes"], out_channels=256, cascade_ratio=2, final_occ_size=final,
                          point_cloud_range=c["point_cloud_range"])
    head = pkg.build_head(dict(cfg["pts_bbox_head"], norm_cfg=dict(type="BN3d"))).eval()
    head.load_state_dict(synth.random_state_dict(head.state_dict(), seed=41))
    g = torch.Generator().manual_seed(42)
    sem = [torch.randn(1, 256, *[max(1, -(-v // 2 ** l)) for v in grid], generator=g) for l in range(4)]
    rig = synth.camera_rig(c["ncam"], c["input_size"], seed=42)
    img = synth.image_feats(c["ncam"], c["fmap"], 512, seed=42)
    tr = synth.rig_transform(rig)
    sd = {k: (v.clone().float().requires_grad_() if (v.is_floating_point() and "running_" not in k) else v.clone())
          for k, v in head.state_dict().items()}
    sem_r = [t.clone().requires_grad_() for t in sem]
    img_r = img.clone().requires_grad_()
    want = ref_cpu.occhead_forward(sd, sem_r, [img_r], tr, 2, final, c["point_cloud_range"])
    gfine = torch.randn(want["fine_output"].shape, generator=g)
    gocc = torch.randn(want["output_voxels"].shape, generator=g)
    ((want["fine_output"] * gfine).sum() + (want["output_voxels"] * gocc).sum()).backward()

    head = head.to(dev)
    X, Y, Z = grid
    feats = []
    sem_d = []
    for t in sem:
        rows = _rows(t).to(dev).requires_grad_()
        sem_d.append(rows)
        feats.append((rows, (1,) + tuple(t.shape[2:])))
    img_d = img.to(dev).requires_grad_()
    ovf, occ = ag.occhead_coarse_train(head, feats)
    # selection as in the inference head: argmax != empty -> ascending coarse voxel list (index work, no gradient)
    flags = torch.empty(X * Y * Z, device=dev, dtype=torch.uint8)
    occ_c = occ.detach().contiguous()
    ccall("coocc_argmax_flags", cptr(occ_c), X * Y * Z, occ_c.shape[1], occ_c.shape[1], 0, cptr(flags))
    lin = torch.nonzero(flags)[:, 0].int()
    fine, fine_xyz = ag.fine_branch_train(head, ovf, (1, X, Y, Z), lin, [img_d[None] if img_d.dim() == 4 else img_d],
                                          tuple(t.to(dev) if torch.is_tensor(t) else t for t in tr))
    assert np.array_equal(fine_xyz.cpu().numpy(), want["fine_coord"].numpy())
    assert_close(fine.detach().cpu(), want["fine_output"].detach(), what="fine logits")
    ((fine * gfine.to(dev)).sum() + (occ * _rows(gocc).to(dev)).sum()).backward()
    for a, b, l in zip(sem_d, sem_r, range(4)):
        sc = max(1.0, float(b.grad.abs().max()))
        assert_close(_vol(a.grad.cpu(), 1, *b.shape[2:]) / sc, b.grad / sc, what="d sem level %d" % l)
    sc = max(1e-12, float(img_r.grad.abs().max()))
    assert_close(img_d.grad.cpu() / sc, img_r.grad / sc, what="d img_feats")
    n = 0
    for name, p in head.named_parameters():
        rg = sd[name].grad
        if rg is None or "bn" in name or (".1." in name and p.dim() == 1 and "mlp" not in name):
            continue                      # frozen BatchNorm affine parameters
        assert p.grad is not None, name
        sc = max(1e-12, float(rg.abs().max()))
        assert_close(p.grad.cpu().view_as(rg) / sc, rg / sc, what=name)
        n += 1
    assert n >= 8 + 6 + 6        # conv weights/biases, three Linear layers, three GroupNorms


def test_conv_batchnorm_training_mode_vs_torch(dev):
    """Conv3d -> BatchNorm3d with BATCH statistics -> (+res) -> ReLU, forward, running-stat update and all gradients."""
    g = torch.Generator().manual_seed(31)
    B, Cin, Cout, (X, Y, Z) = 2, 16, 24, (7, 6, 5)
    x = torch.randn(B, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) / (Cin * 27) ** 0.5
    res = torch.randn(B, Cout, X, Y, Z, generator=g)
    gout = torch.randn(B, Cout, X, Y, Z, generator=g)
    bn_r = torch.nn.BatchNorm3d(Cout, momentum=0.1).train()
    bn_r.weight.data.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn_r.bias.data.copy_(torch.randn(Cout, generator=g) * 0.1)
    bn_d = torch.nn.BatchNorm3d(Cout, momentum=0.1).train()
    bn_d.load_state_dict(bn_r.state_dict())
    xr, wr, rr = x.clone().requires_grad_(), w.clone().requires_grad_(), res.clone().requires_grad_()
    yr = F.relu(bn_r(F.conv3d(xr, wr, padding=1)) + rr)
    yr.backward(gout)
    bn_d = bn_d.to(dev)
    xd, wd, rd = _rows(x).to(dev).requires_grad_(), w.to(dev).requires_grad_(), _rows(res).to(dev).requires_grad_()
    y, geo = ag.conv3d_bn_train_rows(xd, wd, (B, X, Y, Z), bn_d, relu=True, res2d=rd)
    assert_close(_vol(y.detach().cpu(), B, X, Y, Z), yr.detach(), what="bn-train forward")
    assert_close(bn_d.running_mean.cpu(), bn_r.running_mean, tol=1e-6, what="running_mean")
    assert_close(bn_d.running_var.cpu(), bn_r.running_var, tol=1e-6, what="running_var")
    assert int(bn_d.num_batches_tracked) == 1
    y.backward(_rows(gout).to(dev))
    assert_close(_vol(xd.grad.cpu(), B, X, Y, Z), xr.grad, what="dx")
    assert_close(wd.grad.cpu(), wr.grad, what="dw")
    assert_close(_vol(rd.grad.cpu(), B, X, Y, Z), rr.grad, what="dres")
    assert_close(bn_d.weight.grad.cpu(), bn_r.weight.grad, what="dgamma")
    assert_close(bn_d.bias.grad.cpu(), bn_r.bias.grad, what="dbeta")


def test_syncbn_two_ranks_one_gpu():
    """SyncBN (norm_cfg type='SyncBN' upstream): batch statistics and the two backward sums all-reduced across ranks.
    Two processes share the one GPU of the test box; gloo carries the collectives."""
    import socket
    import torch.multiprocessing as mp
    import syncbn_worker
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(syncbn_worker.run, args=(2, port), nprocs=2, join=True)


@pytest.mark.parametrize("case", ["fuser_k2", "fuser_small_k1", "fuser_k2_far"])
def test_fuser_g1_backward_vs_oracle_autograd(dev, case):
    """G1 of BiFuser_N as a differentiable composition (gather -> knn_enc -> gate -> scatter, autograd.fuser_fuse_train)
    over the neighbour tables of the inference search: forward equals the inference concat rows; the gradients of both
    feature volumes and of knn_enc equal torch autograd through the oracle's restatement of bifuser_n.py:127-171
    (incl. the -1 -> last-row quirk and the small-Q branch)."""
    from oracle import cases, ref_cpu
    import co_occ_amd.synth as synth
    c = cases.FUSER_CASES[case]
    img, pts = cases.fuser_inputs(c)
    C = c["C"]
    f = pkg.BiFuser_N(C, C, c["knum"])
    sd = synth.random_state_dict(f.state_dict(), seed=c["seed"])
    f.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    # oracle with autograd
    ir, pr = img.clone().requires_grad_(), pts.clone().requires_grad_()
    sdr = {k: (v.clone().float().requires_grad_() if k.startswith("knn_enc") else v.clone()) for k, v in sd.items()}
    allf = ref_cpu.bifuser_fuse(sdr, ir, pr, c["knum"])["all_feats"]                     # [1,X,Y,Z,4C]
    G = torch.randn(allf.shape, generator=g)
    (allf * G).sum().backward()
    # ours
    f = f.to(dev).eval()
    with torch.no_grad():
        sr = f.search(img.to(dev), pts.to(dev))
        want_rows = f.finish(sr)[0].t.clone()
    torch.cuda.synchronize()
    V = img.shape[2] * img.shape[3] * img.shape[4]
    rows = lambda t: t.permute(0, 2, 3, 4, 1).reshape(V, -1).contiguous()
    xi, xp = rows(img).to(dev).requires_grad_(), rows(pts).to(dev).requires_grad_()
    out = ag.fuser_fuse_train(f, xi, xp, sr)
    assert_close(out.detach().cpu(), want_rows.cpu(), tol=1e-6, what="fuse forward vs inference rows")
    assert_close(out.detach().cpu().view(allf.shape), allf.detach(), what="fuse forward vs oracle")
    (out * G.view(V, -1).to(dev)).sum().backward()
    assert_close(xi.grad.cpu(), rows(ir.grad), what="d img feats")
    assert_close(xp.grad.cpu(), rows(pr.grad), what="d pts feats")
    assert_close(f.knn_enc[0].weight.grad.cpu(), sdr["knn_enc.0.weight"].grad, what="d knn_enc.weight")
    assert_close(f.knn_enc[0].bias.grad.cpu(), sdr["knn_enc.0.bias"].grad, what="d knn_enc.bias")


def test_detector_forward_train_hot_path_matches_inference_and_backpropagates(dev):
    """COOCC_Ray.forward_train_hot_path: same numbers as the inference forward (fused volume, coarse logits, fine logits
    of the same coarse voxels, rendered maps) and a backward pass that reaches the two input volumes, the image features
    and every parameter of fuser / encoder / neck / head / render heads."""
    import co_occ_amd.synth as synth
    grid, C = (50, 50, 8), 32
    cfg = synth.model_cfg(C=C, knum=2, final_occ_size=(100, 100, 16), point_cloud_range=(-25, -25, -5.0, 25, 25, 3.0),
                          input_size=(64, 176))
    model = pkg.build_detector(cfg)
    model.load_state_dict(synth.random_state_dict(model.state_dict(), seed=21))
    model = model.to(dev).eval()
    img, pts = synth.voxel_inputs(grid, C=C, seed=31)
    rig = synth.camera_rig(6, (64, 176), seed=31)
    r = {k: v.to(dev) for k, v in rig.items() if torch.is_tensor(v)}
    gemo = model.img_view_transformer.get_geometry(r["rots"], r["trans"], r["intrins"], r["post_rots"], r["post_trans"], r["bda"])
    img_feats = [synth.image_feats(6, (4, 11), 512, seed=31).to(dev).requires_grad_()]
    tr = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synth.rig_transform(rig))
    # the render block's bounds are those of a 100x100x8 volume (coocc_ray.py:577): render only checked for finiteness here
    with torch.no_grad():
        want = model.forward_hot_path(img.to(dev), pts.to(dev), gemo, [img_feats[0].detach()], tr, render=False)
    xi, xp = img.to(dev).requires_grad_(), pts.to(dev).requires_grad_()
    lin_all = torch.nonzero(want["pred_c"][0].argmax(0).flatten() != model.pts_bbox_head.empty_idx).flatten().int()
    out = model.forward_train_hot_path(xi, xp, gemo, img_feats, tr, coarse_lin=lin_all, render=False)
    V = grid[0] * grid[1] * grid[2]
    assert_close(out["voxel_rows"].detach().cpu(), want["voxel_feats"].permute(0, 2, 3, 4, 1).reshape(V, -1).cpu(), what="voxel_feats")
    assert_close(out["logit_rows"].detach().cpu(), want["pred_c"].permute(0, 2, 3, 4, 1).reshape(V, -1).cpu(), what="coarse logits")
    assert torch.equal(out["fine_xyz"].cpu(), want["output_coords_fine"][0].cpu())
    # the inference path runs the split-f16 engine, the training path the fp32-MFMA kernels: two roundings of the same sums,
    # amplified by the fine branch's per-row GroupNorms (tests/test_gpu_parity_full.py judges each against fp64)
    assert_close(out["fine_logits"].detach().cpu(), want["output_voxels_fine"][0].cpu(), tol=5e-4, what="fine logits")
    loss = out["logit_rows"].square().mean() + out["fine_logits"].square().mean()
    loss.backward()
    for name, t in (("img volume", xi), ("pts volume", xp), ("img feats", img_feats[0])):
        assert t.grad is not None and torch.isfinite(t.grad).all() and float(t.grad.abs().max()) > 0, name
    skip = ("sigma_head", "rgb_head", "img_view_transformer")
    missing = [n for n, p in model.named_parameters()
               if p.requires_grad and not n.startswith(skip) and p.dim() > 1 and (p.grad is None or not torch.isfinite(p.grad).all())]
    assert not missing, missing[:5]
