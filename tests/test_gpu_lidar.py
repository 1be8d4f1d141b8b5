"""LiDAR-side producer (hard voxelisation, HardSimpleVFE, sparse encoder) vs the golden output of the
reference's numpy voxel generator and vs the oracle's dense restatement of the spconv layers."""
import numpy as np
import pytest
import torch

import co_occ_amd.synth as synth
from co_occ_amd import lidar as L
from oracle import cases, ref_lidar
from util import assert_close

pytestmark = pytest.mark.gpu


def test_voxelization_vs_golden_bit_exact(dev, golden):
    c, g = cases.LIDAR_CASE, golden("voxelize")
    pts = cases.lidar_points(c)
    vox_layer = L.Voxelization(c["voxel_size"], c["point_cloud_range"], c["max_points"], (c["max_voxels"], c["max_voxels"])).eval()
    vox, coors, num = vox_layer(torch.from_numpy(pts).to(dev))
    assert np.array_equal(coors.cpu().numpy(), g["coors"])            # order of first appearance, (z,y,x)
    assert np.array_equal(num.cpu().numpy(), g["num"])
    assert np.array_equal(vox.cpu().numpy(), g["voxels"])             # first max_points points, in point order, zero padded
    m = L.HardSimpleVFE(num_features=5)(vox, num, coors)
    want = ref_lidar.vfe_mean(g["voxels"], g["num"], 5)
    assert_close(m.cpu(), want, tol=1e-6, what="vfe mean")


@pytest.mark.parametrize("n,max_voxels", [(0, 50), (1, 50), (777, 10 ** 6), (20000, 300)])
def test_voxelization_edge_sizes_vs_oracle(dev, n, max_voxels):
    """Empty sweep, single point, no truncation, heavy max_voxels truncation."""
    c = cases.LIDAR_CASE
    pts = cases.lidar_points(dict(c, seed=c["seed"] + n), n=max(n, 1))[:n]
    layer = L.Voxelization(c["voxel_size"], c["point_cloud_range"], 3, (max_voxels, max_voxels)).eval()
    vox, coors, num = layer(torch.from_numpy(pts).to(dev).view(-1, c["F"]))
    o_vox, o_coors, o_num = ref_lidar.hard_voxelize(pts.reshape(-1, c["F"]), c["voxel_size"], c["point_cloud_range"], 3, max_voxels)
    assert np.array_equal(coors.cpu().numpy(), o_coors) and np.array_equal(num.cpu().numpy(), o_num)
    assert np.array_equal(vox.cpu().numpy(), o_vox)


def _seeded_encoder(cls, shape_xyz, seed, dev):
    enc = cls(4, dict(type="BN1d"), 16, 128, list(shape_xyz)).eval()
    enc.load_state_dict(synth.random_state_dict(enc.state_dict(), seed=seed))
    return enc.to(dev), {k: v.clone() for k, v in enc.state_dict().items()}


@pytest.mark.parametrize("variant", ["8x", "4x"])
def test_sparse_encoder_vs_dense_oracle(dev, variant):
    """SparseLiDAREnc{8x,4x}: rule-book gather GEMMs vs masked dense convolutions (spconv semantics restated; unpinned)."""
    g = np.random.default_rng(3)
    shape_xyz = (32, 24, 16)
    W, H, D = shape_xyz
    M = 900
    lin = g.choice(W * H * D, M, replace=False)
    coors = np.stack([lin // (W * H), (lin // W) % H, lin % W], 1).astype(np.int32)        # (z, y, x), arbitrary order
    feats = g.standard_normal((M, 4)).astype(np.float32)
    cls = L.SparseLiDAREnc8x if variant == "8x" else L.SparseLiDAREnc4x
    enc, sd = _seeded_encoder(cls, shape_xyz, 13, dev)
    with torch.no_grad():
        out = enc(torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev), 1)
    want, mask = ref_lidar.sparse_encoder_forward({k: v.cpu() for k, v in sd.items()}, feats, coors, (D, H, W), variant)
    f = 8 if variant == "8x" else 4
    assert tuple(out["x"].shape) == tuple(want.shape) == (1, 128, W // f, H // f, D // f)
    assert_close(out["x"].cpu(), want, what="dense lidar volume")
    sp = out["pts_feats"][0]
    assert sp.coors.shape[0] == int(mask.sum())                         # same active set
    # batch-indexed coors (b, z, y, x) are accepted like upstream
    coors4 = np.concatenate([np.zeros((M, 1), np.int32), coors], 1)
    with torch.no_grad():
        out4 = enc(torch.from_numpy(feats).to(dev), torch.from_numpy(coors4).to(dev), 1)
    assert torch.equal(out4["x"], out["x"])


def test_lidar_producer_end_to_end(dev):
    """points -> Voxelization -> HardSimpleVFE -> SparseLiDAREnc8x -> [1,128,X/8,Y/8,Z/8] volume, vs the oracle chain."""
    c = dict(cases.LIDAR_CASE, voxel_size=(0.25, 0.25, 0.25), max_voxels=20000, max_points=10)
    pts = cases.lidar_points(c, n=4000)[:, :4]
    shape_xyz = tuple(int(round((c["point_cloud_range"][3 + i] - c["point_cloud_range"][i]) / c["voxel_size"][i])) for i in range(3))
    vox_layer = L.Voxelization(c["voxel_size"], c["point_cloud_range"], c["max_points"], (c["max_voxels"],) * 2).eval()
    enc, sd = _seeded_encoder(L.SparseLiDAREnc8x, shape_xyz, 17, dev)
    with torch.no_grad():
        vox, coors, num = vox_layer(torch.from_numpy(pts).to(dev))
        feats = L.HardSimpleVFE(num_features=5)(vox, num, coors)
        out = enc(feats, coors, 1)
    o_vox, o_coors, o_num = ref_lidar.hard_voxelize(pts, c["voxel_size"], c["point_cloud_range"], c["max_points"], c["max_voxels"])
    o_feats = ref_lidar.vfe_mean(o_vox, o_num, 5)
    want, _ = ref_lidar.sparse_encoder_forward({k: v.cpu() for k, v in sd.items()}, o_feats, o_coors, shape_xyz[::-1], "8x")
    assert_close(out["x"].cpu(), want, what="pts_voxel_feats")
    assert float(out["x"].abs().max()) > 0


def test_detector_extract_pts_feat_from_reference_config_slice(dev):
    """COOCC_Ray built with the pts_* entries of coocc_multi_r50_256x704.py:121-135 (smaller sparse shape)."""
    import co_occ_amd as pkg
    cfg = synth.model_cfg(C=128)
    cfg.update(pts_voxel_layer=dict(max_num_points=10, point_cloud_range=[-8., -8., -2., 8., 8., 2.], voxel_size=[0.125] * 3,
                                    max_voxels=(90000, 120000)),
               pts_voxel_encoder=dict(type='HardSimpleVFE', num_features=5),
               pts_middle_encoder=dict(type='SparseLiDAREnc8x', input_channel=4, base_channel=16, out_channel=128,
                                       norm_cfg=dict(type='SyncBN', requires_grad=True), sparse_shape_xyz=[128, 128, 32]))
    model = pkg.build_detector(cfg).to(dev).eval()
    pts = torch.from_numpy(cases.lidar_points(cases.LIDAR_CASE, n=5000)[:, :4]).to(dev)
    with torch.no_grad():
        vol, sp = model.extract_pts_feat([pts])
    assert tuple(vol.shape) == (1, 128, 16, 16, 4) and sp[0].coors.shape[0] > 0
    keys = set(model.state_dict().keys())
    assert {"pts_middle_encoder.conv_input.0.weight", "pts_middle_encoder.conv1.0.0.weight", "pts_middle_encoder.conv3.2.net.4.running_var",
            "pts_middle_encoder.conv_out.1.bias"} <= keys


def test_sparse_layers_match_hand_computed_fixture(dev):
    """The HIP rule-table layers (index map -> SubM / SparseConv3d(k3,s2,p1) tables -> row-table GEMM) against the hand-computed
    5x5x5 known-answer vector (tests/golden/sparse_rules_5x5x5.json): active sets, output coordinates and values exactly."""
    from test_oracle import _hand_fixture
    fx, W = _hand_fixture()
    coors = torch.tensor(fx["coors_zyx"], dtype=torch.int32, device=dev)
    feats = torch.tensor(fx["feats"], dtype=torch.float32, device=dev)
    x = torch.cat([feats, feats.new_zeros(feats.shape[0], 2)], 1).contiguous()            # rows padded to 4 channels
    conv = L._SpConv(2, 2, 3).to(dev)
    with torch.no_grad():
        conv.weight.copy_(W.to(dev))
    pc = conv.packed()
    cur = L.SparseRows(None, coors, fx["grid_zyx"])
    got = L.sparse_conv(x, 4, pc, cur.subm_table("t"), relu=False)
    assert np.array_equal(got.cpu().numpy(), np.asarray(fx["subm_k3"]["out"], np.float32))
    coors_o, shape_o, table = cur.downsample(3, 2, 1)
    assert list(shape_o) == fx["sparse_k3_s2_p1"]["out_shape_zyx"]
    assert np.array_equal(coors_o.cpu().numpy(), np.asarray(fx["sparse_k3_s2_p1"]["coors_zyx"]))
    got = L.sparse_conv(x, 4, pc, table, relu=False)
    assert np.array_equal(got.cpu().numpy(), np.asarray(fx["sparse_k3_s2_p1"]["out"], np.float32))


@pytest.mark.parametrize("cin,cout,res,M_out", [(32, 32, True, 66003), (64, 64, True, 66003), (32, 64, False, 66003), (64, 32, False, 66003),
                                                 (128, 128, True, 40037), (64, 128, False, 40037)])
def test_narrow_rule_book_gemm_equals_the_wide_tile_kernel(dev, monkeypatch, cin, cout, res, M_out):
    """``k_gemm_h2n`` (rule-book GEMMs with <= 64 output channels on 256-row tiles, every wave on real columns; taken at >= 512
    row tiles) against ``k_gemm_h2w<table>`` (``COOCC_H2_NARROW=0``): the same bits -- rows, H2 twin and residual path -- on a
    random table with a third of its entries empty and a ragged last tile; and against an fp64 gather-matmul on a sample of rows."""
    if not (L.LIDAR_H2 and L.core.CONV_ENGINE == "h2"):
        pytest.skip("split-f16 engine off")
    g = torch.Generator().manual_seed(41)
    M_in, taps = 60000, 27            # 128 output channels: the 64-row-tile form (k_gemm_h2n<4>, 128 .. 1023 row tiles)
    feats = torch.randn(M_in, cin, generator=g).to(dev)
    table = torch.randint(0, M_in, (taps, M_out), generator=g, dtype=torch.int32)
    table[torch.rand(taps, M_out, generator=g) < 0.33] = -1
    table = table.to(dev)
    conv = L._SpConv(cin, cout, 3).to(dev)
    bn = torch.nn.BatchNorm1d(cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5); bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(cout, generator=g) * 0.1); bn.running_var.copy_(torch.rand(cout, generator=g) + 0.5)
    pc = conv.packed(bn=bn)
    r = torch.randn(M_out, cout, generator=g).to(dev) if res else None
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("COOCC_H2_NARROW", flag)
        o, oh = L.sparse_conv(feats, cin, pc, table, relu=True, res=r, twin=True)
        torch.cuda.synchronize()
        outs[flag] = (o.clone(), None if oh is None else oh.clone())
    assert torch.equal(outs["1"][0], outs["0"][0])
    if outs["0"][1] is not None:
        assert torch.equal(outs["1"][1], outs["0"][1])
    # fp64 reference on 512 sampled rows (incl. the last, ragged tile)
    rows = torch.cat([torch.randint(0, M_out, (500,), generator=g), torch.arange(M_out - 12, M_out)]).to(dev)
    w = conv.weight.detach().double().reshape(cout, taps, cin)
    t = table[:, rows.long()].long()                                              # [taps, n]
    x = torch.where((t >= 0)[..., None], feats.double()[t.clamp(min=0)], torch.zeros((), device=dev, dtype=torch.float64))
    y = torch.einsum("tnc,otc->no", x, w)
    s_ = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).double()
    y = y * s_ + (bn.bias.double() - bn.running_mean.double() * s_)
    if res:
        y = y + r[rows.long()].double()
    y = torch.relu(y)
    assert_close(outs["1"][0][rows.long()].cpu(), y.float().cpu(), tol=1e-5, what="narrow rule-book GEMM vs fp64")


def _grad_close(got, want, what, tol=2e-4):
    """Gradient tensors: error relative to the tensor's own largest entry (gradients of this encoder span 1e-6 .. 1e2)."""
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    sc = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= tol * max(sc, 1e-12) + 1e-9, "%s: max |d| %.3e of scale %.3e" % (what, err, sc)


@pytest.mark.parametrize("kind", ["subm", "down", "subm_4_to_16"])
def test_sparse_conv_fn_gradients_vs_dense_autograd(dev, kind):
    """``SparseConvFn`` (round 5: the LiDAR encoder trains): forward, dgrad through the transposed rule book and wgrad through
    the forward rule book of one SubMConv3d / SparseConv3d(k3, s2, p1), against torch autograd through the oracle's masked-dense
    restatement (oracle/ref_lidar.py)."""
    g = np.random.default_rng(5)
    W, H, D = 14, 11, 9
    M, Cin, Cout = 400, 8, 20
    if kind == "subm_4_to_16":            # conv_input's shape: the dgrad GEMM has 4 output columns
        kind, Cin, Cout = "subm", 4, 16
    lin = g.choice(W * H * D, M, replace=False)
    coors = np.stack([lin // (W * H), (lin // W) % H, lin % W], 1).astype(np.int32)
    feats = torch.from_numpy(g.standard_normal((M, Cin)).astype(np.float32))
    w = torch.from_numpy((g.standard_normal((Cout, 3, 3, 3, Cin)) * 0.1).astype(np.float32))
    b = torch.from_numpy(g.standard_normal(Cout).astype(np.float32))
    # oracle
    fo, wo, bo = feats.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    if kind == "subm":
        want = ref_lidar.subm_conv3d(fo, coors, (D, H, W), wo, bo)
        want_coors = torch.from_numpy(coors).long()
    else:
        want, want_coors, _ = ref_lidar.sparse_conv3d(fo, coors, (D, H, W), wo)
        want = want + bo
    # HIP
    fx, wx, bx = feats.to(dev).requires_grad_(), w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    conv = L._SpConv(Cin, Cout, 3, bias=True).to(dev)
    cur = L.SparseRows(None, torch.from_numpy(coors).to(dev), (D, H, W))
    if kind == "subm":
        tb = cur.subm_table("k")
        got = L.SparseConvFn.apply(fx, wx, bx, tb, tb.flip(0).contiguous())
        got_coors = torch.from_numpy(coors).long()
    else:
        co, shape_o, tb = cur.downsample(3, 2, 1)
        got = L.SparseConvFn.apply(fx, wx, bx, tb, L._inverse_table(tb, M))
        got_coors = co.cpu().long()
    # rows may come in another order: compare through the coordinates
    def keyed(rows, cc):
        k = (cc[:, 0] * 1000 + cc[:, 1]) * 1000 + cc[:, 2]
        return rows[torch.argsort(k)]
    assert got.shape == want.shape
    assert_close(keyed(got.detach().cpu(), got_coors), keyed(want.detach(), want_coors), what="sparse conv forward (%s)" % kind)
    up = torch.from_numpy(g.standard_normal(tuple(want.shape)).astype(np.float32))
    # the same upstream gradient per output voxel (match rows by coordinate)
    kw = (want_coors[:, 0] * 1000 + want_coors[:, 1]) * 1000 + want_coors[:, 2]
    kg = (got_coors[:, 0] * 1000 + got_coors[:, 1]) * 1000 + got_coors[:, 2]
    pos = {int(k): i for i, k in enumerate(kw.tolist())}
    perm = torch.tensor([pos[int(k)] for k in kg.tolist()])
    (want * up).sum().backward()
    (got * up[perm].to(dev)).sum().backward()
    _grad_close(fx.grad, fo.grad, "d input")
    _grad_close(wx.grad, wo.grad, "d weight")
    _grad_close(bx.grad, bo.grad, "d bias")


@pytest.mark.parametrize("variant", ["8x", "4x"])
def test_sparse_encoder_trains_forward_and_backward_vs_oracle(dev, variant):
    """``SparseLiDAREnc{8x,4x}`` under ``train()`` (sparse_lidar_enc.py:125-176 as ``tools/train.py`` runs it): BatchNorm1d with
    the batch statistics of the active rows, running statistics updated, gradients of every parameter and of the input features
    against torch autograd through the oracle's dense restatement with training-mode BN."""
    g = np.random.default_rng(7)
    shape_xyz = (32, 24, 16)
    W, H, D = shape_xyz
    M = 900
    lin = g.choice(W * H * D, M, replace=False)
    coors = np.stack([lin // (W * H), (lin // W) % H, lin % W], 1).astype(np.int32)
    feats = g.standard_normal((M, 4)).astype(np.float32)
    cls = L.SparseLiDAREnc8x if variant == "8x" else L.SparseLiDAREnc4x
    enc, sd = _seeded_encoder(cls, shape_xyz, 17, dev)
    enc.train()
    rm0 = enc.conv2[0][1].running_mean.clone()
    fx = torch.from_numpy(feats).to(dev).requires_grad_()
    out = enc(fx, torch.from_numpy(coors).to(dev), 1)["x"]
    osd = {k: (v.cpu().clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.cpu().clone()) for k, v in sd.items()}
    fo = torch.from_numpy(feats).requires_grad_()
    want, mask = ref_lidar.sparse_encoder_forward(osd, fo, coors, (D, H, W), variant, train_bn=True)
    assert tuple(out.shape) == tuple(want.shape)
    # batch-statistics BN over the few hundred active rows of the deep stages amplifies fp32 rounding (the fp32 oracle itself is
    # 1e-4 .. 1e-3 of the scale away from its fp64 evaluation here): judged against the fp64 anchor, as the fine branch is
    sd64 = {k: (v.cpu().double() if v.dtype.is_floating_point else v.cpu()) for k, v in sd.items()}
    with torch.no_grad():
        w64, _ = ref_lidar.sparse_encoder_forward(sd64, torch.from_numpy(feats).double(), coors, (D, H, W), variant, train_bn=True)
    sc = max(1.0, float(w64.abs().max()))
    e_hip, e_o32 = float((out.detach().cpu().double() - w64).abs().max()) / sc, float((want.detach().double() - w64).abs().max()) / sc
    print("lidar encoder (train, %s): HIP vs fp64 %.2e, fp32 oracle vs fp64 %.2e" % (variant, e_hip, e_o32))
    assert e_hip <= max(3.0 * e_o32, 1e-4)
    assert not torch.equal(rm0, enc.conv2[0][1].running_mean)            # running statistics move as nn.BatchNorm1d moves them
    up = torch.from_numpy(g.standard_normal(tuple(want.shape)).astype(np.float32))
    (want * up).sum().backward()
    (out * up.to(dev)).sum().backward()
    # the same anchor for the gradients: autograd through the oracle in fp64
    osd64 = {k: (v.clone().requires_grad_() if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd64.items()}
    fo64 = torch.from_numpy(feats).double().requires_grad_()
    w64g, _ = ref_lidar.sparse_encoder_forward(osd64, fo64, coors, (D, H, W), variant, train_bn=True)
    (w64g * up.double()).sum().backward()
    bad, n, worst = [], 0, (0.0, 0.0)
    for name, got_g, g32, g64 in [("d voxel features", fx.grad, fo.grad, fo64.grad)] + [(k, q.grad, osd[k].grad, osd64[k].grad)
                                                                                          for k, q in enc.named_parameters()]:
        assert got_g is not None, name
        sc = max(float(g64.abs().max()), 1e-12)
        e_hip = float((got_g.detach().cpu().double() - g64).abs().max()) / sc
        e_o32 = float((g32.double() - g64).abs().max()) / sc
        worst = max(worst, (e_hip, e_o32))
        if not e_hip <= max(3.0 * e_o32, 1e-3):
            bad.append("%s: HIP vs fp64 %.2e, fp32 oracle vs fp64 %.2e" % (name, e_hip, e_o32))
        n += 1
    print("lidar encoder gradients (%s): worst HIP vs fp64 %.2e (fp32 oracle there: %.2e) over %d tensors" % (variant, worst[0], worst[1], n))
    assert not bad, "%d of %d gradients off:\n%s" % (len(bad), n, "\n".join(bad[:12]))
    assert n >= 40
    # eval() afterwards is the inference path again (folded running statistics, no autograd)
    enc.eval()
    with torch.no_grad():
        ev = enc(torch.from_numpy(feats).to(dev), torch.from_numpy(coors).to(dev), 1)["x"]
    assert not ev.requires_grad and tuple(ev.shape) == tuple(want.shape)
