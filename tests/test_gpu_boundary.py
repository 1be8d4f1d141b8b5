"""Drop-in boundary at detector level, GPU half: ``get_frustum`` and the depth-only render branch against golden vectors of
the unmodified reference, the render regulariser scalars (L1) of its ``forward_train``, and the reference call signatures
``simple_test(img_metas, img, points=..., gt_occ=...)`` / ``forward(return_loss=True, ...)`` end to end with stand-in
upstream encoders (the 2D backbone and DepthNet are outside the hot path; any nn.Module can be attached)."""
import numpy as np
import pytest
import torch
from torch import nn

import co_occ_amd as pkg
import co_occ_amd.synth as synth
from co_occ_amd import autograd as ag, render as R
from oracle import cases, ref_cpu
from util import assert_close, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["nus", "kitti"])
def test_get_frustum_vs_golden(dev, golden, tag):
    """coocc_ray.py:732-776 (nuScenes 3x3 matrices; KITTI-style 4x4 intrinsics with a shift column + 4x4 bda)."""
    c = cases.FRUSTUM_CASE
    a = cases.frustum_inputs(c, tag == "kitti")
    got = pkg.get_frustum(*[t.to(dev) for t in a[:6]], a[6], c["scale"])
    want = golden("frustum")[tag]
    assert tuple(got.shape) == want.shape
    assert_close(got.cpu(), want, tol=1e-5, what="get_frustum " + tag)
    # ... and the view transformer's own get_frustum / get_geometry accept the same matrices
    vt = pkg.build_neck(dict(synth.model_cfg(input_size=c["input_size"])["img_view_transformer"])).to(dev)
    assert_close(vt.get_frustum(*[t.to(dev) for t in a[:6]], c["scale"]).cpu(), want, tol=1e-5)


def _heads(c, dev):
    sig, rgb = R.MLP(128, 1, net_depth=1, skip_layer=None), R.MLP(128, 3, net_depth=3, skip_layer=None)
    sig.load_state_dict(synth.random_state_dict(sig.state_dict(), c["seed"]))
    rgb.load_state_dict(synth.random_state_dict(rgb.state_dict(), c["seed"] + 1))
    return sig.to(dev), rgb.to(dev)


def test_train_render_losses_vs_reference_forward_train_golden(dev, golden):
    """L1 (coocc_ray.py:423-433 camera branch, :486-496 LiDAR-only branch): the scalars the unmodified forward_train
    produced, through the differentiable HIP render block + loss kernels, and through the inference render block."""
    g = golden("train_render")
    c = cases.TRAIN_RENDER_CASE
    vf, rig, imgs, depth = cases.train_render_inputs(c)
    sig, rgb = _heads(c, dev)
    mats = [rig[k].to(dev) for k in ("rots", "trans", "intrins", "post_rots", "post_trans", "bda")]
    H, W = c["input_size"]
    gemo = pkg.get_frustum(*mats, (H, W), 16)                       # == get_geometry at downsample 16
    X, Y, Z = c["grid"]
    rows = vf.to(dev).permute(0, 2, 3, 4, 1).reshape(X * Y * Z, c["C"]).contiguous().requires_grad_(True)
    rgbs, depths = ag.render_block_train(sig, rgb, rows, (X, Y, Z), gemo)
    D = gemo.shape[2]
    l = ag.render_losses(rgbs, depths, imgs[0].permute(0, 2, 3, 1).contiguous().to(dev), depth[0].to(dev), D)
    assert abs(float(l["loss_depth_render"].detach()) - float(g["loss_depth_render"])) < 2e-6
    assert abs(float(l["loss_rgb"].detach()) - float(g["loss_rgb"])) < 2e-6
    (l["loss_depth_render"] + l["loss_rgb"]).backward()
    assert torch.isfinite(rows.grad).all() and float(rows.grad.abs().sum()) > 0
    # LiDAR-only branch: no colour head, geometry from get_frustum
    _, d2 = ag.render_block_train(sig, None, rows.detach(), (X, Y, Z), gemo)
    l2 = pkg.COOCC_Ray.render_losses_train(None, d2, None, depth[0].to(dev), D)
    assert abs(float(l2["loss_depth_render"]) - float(g["lidar_loss_depth_render"])) < 2e-6
    # inference render block: depth_only gives the same depth maps as the full render and no colour maps
    with torch.no_grad():
        r_full, d_full, _ = R.render_block(sig, rgb, vf.to(dev), gemo, 16)
        r_none, d_only, _ = R.render_block(sig, rgb, vf.to(dev), gemo, 16, depth_only=True)
    assert r_none is None and torch.equal(d_only, d_full)
    assert_close(d_full.cpu(), depths.detach().cpu(), tol=1e-6)
    want = R.render_losses(r_full, d_full, imgs[0].permute(0, 2, 3, 1).contiguous().to(dev), depth[0].to(dev), D)
    assert abs(float(want["loss_rgb"]) - float(g["loss_rgb"])) < 2e-6


class _StubBackbone(nn.Module):
    """Stand-in for the mmdet ResNet + SECONDFPN image encoder (upstream of the hot path): stride-16 patch embedding."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 512, 16, stride=16)

    def forward(self, x):
        return [torch.relu(self.conv(x))]


class _StubDepthNet(nn.Module):
    """Stand-in for DepthNet (DCN + ASPP, upstream): (x [BN,512,h,w], mlp_input [B,N,27]) -> [BN, D + C, h, w]."""

    def __init__(self, D=112, C=128):
        super().__init__()
        self.conv = nn.Conv2d(512, D + C, 1)
        self.se = nn.Linear(27, D + C)

    def forward(self, x, mlp_input):
        return self.conv(x) * torch.sigmoid(self.se(mlp_input.reshape(-1, mlp_input.shape[-1])))[..., None, None]


def _full_model(dev, seed=2):
    torch.manual_seed(seed)
    cfg = synth.model_cfg()
    cfg["img_view_transformer"]["depth_net"] = _StubDepthNet()
    cfg.update(img_backbone=_StubBackbone(),
               pts_voxel_layer=dict(max_num_points=10, point_cloud_range=[-50, -50, -5, 50, 50, 3], voxel_size=[0.125] * 3,
                                    max_voxels=(90000, 120000)),
               pts_voxel_encoder=dict(type='HardSimpleVFE', num_features=5),
               pts_middle_encoder=dict(type='SparseLiDAREnc8x', input_channel=4, base_channel=16, out_channel=128,
                                       norm_cfg=dict(type='SyncBN', requires_grad=True), sparse_shape_xyz=[800, 800, 64]))
    model = pkg.build_detector(cfg)
    sd = synth.random_state_dict({k: v for k, v in model.state_dict().items() if not k.startswith(("img_backbone", "img_view_transformer"))}, seed=seed)
    model.load_state_dict(sd, strict=False)
    return model.to(dev)


def _sample(dev, seed=5):
    g = np.random.default_rng(seed)
    rig = synth.camera_rig(6, (256, 704), seed=seed)
    imgs = torch.from_numpy(g.random((1, 6, 3, 256, 704), dtype=np.float32))
    depth = torch.from_numpy((g.uniform(0.5, 70.0, (1, 6, 256, 704)) * (g.random((1, 6, 256, 704)) < 0.05)).astype(np.float32))
    # a sweep that stays SPARSER than the camera volume: with knum > 1 the reference indexes inds_img with pts ordinals
    # (bifuser_n.py:158) and raises IndexError as soon as the LiDAR list is the longer one
    n = 6000
    pts = np.concatenate([g.uniform(-22, 22, (n, 2)), g.normal(-1.5, 1.0, (n, 1)).clip(-4.9, 2.9), g.random((n, 2))], 1).astype(np.float32)
    gt = g.integers(1, 17, (1, 200, 200, 16)).astype(np.uint8)
    gt[g.random(gt.shape) < 0.8] = 0
    gt[g.random(gt.shape) < 0.02] = 255
    # img_inputs as the dataset pipeline collates it: ..., [7] = sparse LiDAR depth maps, [-1] = (H, W) of the input images
    img_inputs = [imgs] + [rig[k] for k in ("rots", "trans", "intrins", "post_rots", "post_trans", "bda")] + [depth, rig["input_size"]]
    mv = lambda t: t.to(dev) if torch.is_tensor(t) else t
    return [mv(t) for t in img_inputs], [torch.from_numpy(pts).to(dev)], torch.from_numpy(gt).to(dev)


def test_simple_test_with_the_reference_call_signature(dev):
    """tools/test.py path: ``model(return_loss=False, img_metas=, img_inputs=, points=, gt_occ=)`` -> the reference's result
    dict (coocc_ray.py:639-656) with numpy confusion matrices; logits equal the oracle run on the same encoder outputs."""
    model = _full_model(dev).eval()
    img_inputs, points, gt = _sample(dev)
    with torch.no_grad():
        out = model(return_loss=False, img_metas=None, img_inputs=img_inputs, points=points, gt_occ=gt)
        ivf, depth, img_feats, geom = model.extract_img_feat(img_inputs, None)
        pvf, _ = model.extract_pts_feat(points)
    assert {"SC_metric", "SSC_metric", "pred_c", "pred_f", "output_voxels", "target_voxels", "SSC_metric_fine"} <= set(out)
    assert isinstance(out["SC_metric"], np.ndarray) and out["SC_metric"].shape == (2, 2) and out["SSC_metric"].shape == (17, 17)
    valid = int((gt != 255).sum())
    assert int(out["SC_metric"].sum()) == valid and int(out["SSC_metric"].sum()) == valid and int(out["SSC_metric_fine"].sum()) == valid
    assert tuple(out["pred_c"].shape) == (1, 17, 100, 100, 8) and tuple(out["pred_f"].shape) == (1, 17, 200, 200, 16)
    assert out["target_voxels"] is gt and out["output_voxels"] is out["pred_c"]
    assert tuple(ivf.shape) == (1, 128, 100, 100, 8) and tuple(pvf.shape) == (1, 128, 100, 100, 8) and tuple(depth.shape) == (6, 112, 16, 44)
    # oracle on the very same encoder outputs (the encoders themselves are upstream of the path)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tr = tuple(t.cpu() if torch.is_tensor(t) else t for t in img_inputs[1:])
    o = ref_cpu.hot_path_forward(sd, ivf.cpu().contiguous(), pvf.cpu().contiguous(), geom.cpu(), [img_feats[0].cpu()], tr, knum=2,
                                 literal_render=False)
    assert_close(out["pred_c"].cpu(), o["output_voxels"], what="pred_c via simple_test")
    assert_close(out["rgbs"].cpu(), o["rgbs"], tol=1e-4, what="rgbs via simple_test")
    sc = ref_cpu.evaluation_semantic(o["output_voxels"], gt.cpu(), 'SC')[0]
    assert np.abs(out_sc_coarse(model, out, gt) - sc).sum() <= 0.002 * valid


def out_sc_coarse(model, out, gt):
    from co_occ_amd.evaluation import evaluation_semantic
    return evaluation_semantic(out["pred_c"], gt, 'SC')[0].cpu().numpy()


def test_forward_train_with_the_reference_call_signature(dev):
    """tools/train.py path: ``model(return_loss=True, points=, img_metas=, img_inputs=, gt_occ=)`` -> the reference's loss
    dict (coocc_ray.py:339-434); training-mode BN (batch statistics, running stats updated); gradients reach every
    hot-path parameter -- the sparse LiDAR encoder included (round 5: it trains like upstream, batch-statistics BN1d) -- and
    flow back into the upstream encoders; ``freeze_lidar_encoder()`` keeps the LiDAR branch fixed."""
    model = _full_model(dev)
    model.freeze_lidar_encoder()
    model.train()
    assert not model.pts_middle_encoder.training and all(not prm.requires_grad for prm in model.pts_middle_encoder.parameters())
    model.freeze_lidar_encoder(False)
    model.train()                        # what an unchanged tools/train.py does
    assert model.pts_middle_encoder.training and model.semantic_encoder.training
    assert all(prm.requires_grad for prm in model.pts_middle_encoder.parameters())
    img_inputs, points, gt = _sample(dev)
    rm0 = model.semantic_encoder.layers[0][0].bn1.running_mean.clone()
    lrm0 = model.pts_middle_encoder.conv1[0][1].running_mean.clone()
    losses = model(return_loss=True, points=points, img_metas=None, img_inputs=img_inputs, gt_occ=gt,
                   generator=torch.Generator(device=dev).manual_seed(0))
    want = {"loss_depth", "loss_depth_render", "loss_rgb"} | {"loss_voxel_%s_%s" % (a, b) for a in ("ce", "sem_scal", "geo_scal", "lovasz")
                                                               for b in ("c_0", "fine")}
    assert set(losses) == want, sorted(set(losses) ^ want)
    assert all(torch.isfinite(v).all() for v in losses.values())
    sum(losses.values()).backward()
    missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None
               and "frustum" not in n and n.split(".")[-1] not in ("dx", "bx", "nx")]
    assert not missing, missing[:8]
    assert float(model.pts_middle_encoder.conv_input[0].weight.grad.abs().sum()) > 0     # ... and the first sparse convolution
    assert not torch.equal(lrm0, model.pts_middle_encoder.conv1[0][1].running_mean)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert float(model.img_backbone.conv.weight.grad.abs().sum()) > 0          # lift (x) splat backward reaches the image branch
    assert float(model.img_view_transformer.depth_net.conv.weight.grad.abs().sum()) > 0
    assert not torch.equal(rm0, model.semantic_encoder.layers[0][0].bn1.running_mean)   # batch statistics were used


def test_modules_in_training_mode_use_batch_statistics_and_autograd(dev):
    """under model.train() the drop-in modules must not silently return eval-mode numbers.  forward()
    normalises with batch statistics like nn.BatchNorm3d.train() and carries grad_fn; the folded-BN row paths refuse."""
    torch.manual_seed(0)
    enc = pkg.build_backbone(dict(type='CustomResNet3D', depth=10, n_input_channels=8, block_inplanes=[8, 16, 16, 32],
                                  out_indices=(0, 1, 2, 3), norm_cfg=dict(type='BN3d', requires_grad=True))).to(dev).train()
    x = torch.randn(1, 8, 12, 10, 4, device=dev, requires_grad=True)
    outs = enc(x)
    assert all(o.grad_fn is not None for o in outs)
    # torch reference: same parameters, training-mode BatchNorm3d
    import torch.nn.functional as F
    bn = enc.input_proj[1]
    y = F.conv3d(x, enc.input_proj[0].weight)
    mean, var = y.mean((0, 2, 3, 4)), y.var((0, 2, 3, 4), unbiased=False)
    y = torch.relu((y - mean.view(1, -1, 1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1, 1) + bn.eps) * bn.weight.view(1, -1, 1, 1, 1)
                   + bn.bias.view(1, -1, 1, 1, 1))
    sum(o.sum() for o in outs).backward()
    assert torch.isfinite(x.grad).all()
    with pytest.raises(RuntimeError):
        enc.forward_rows(x)
    enc.eval()
    with torch.no_grad():
        e = enc(x)
    assert all(o.grad_fn is None for o in e) and rel_err(e[0].cpu(), outs[0].detach().cpu()) > 1e-3     # different statistics
