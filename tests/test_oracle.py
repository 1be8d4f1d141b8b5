"""The oracle itself: upstream known-answer tests for the two CUDA ops that have them, and the
CPU restatement against the golden vectors produced from the real reference (no GPU needed)."""
import numpy as np
import pytest
import torch

from oracle import cases, native, ref_cpu
import co_occ_amd.synth as synth
from util import assert_close

# mmdetection3d/tests/test_models/test_common_modules/test_pointnet_ops.py:10-24
FPS_XYZ = np.array([[[-0.2748, 1.0020, -1.1674], [0.1015, 1.3952, -1.2681], [-0.8070, 2.4137, -0.5845],
                     [-1.0001, 2.1982, -0.5859], [0.3841, 1.8983, -0.7431]],
                    [[-1.0696, 3.0758, -0.1899], [-0.2559, 3.5521, -0.1402], [0.8164, 4.0081, -0.1839],
                     [-1.1000, 3.0213, -0.8205], [-0.0518, 3.7251, -0.3950]]], np.float32)
FPS_EXPECTED = np.array([[0, 2, 4], [0, 2, 1]])
# same file :27-74
BQ_NEW = np.array([[[-0.0740, 1.3147, -1.3625], [-2.2769, 2.7817, -0.2334], [-0.4003, 2.4666, -0.5116],
                    [-0.0740, 1.3147, -1.3625], [-0.0740, 1.3147, -1.3625]],
                   [[-2.0289, 2.4952, -0.1708], [-2.0668, 6.0278, -0.4875], [0.4066, 1.4211, -0.2947],
                    [-2.0289, 2.4952, -0.1708], [-2.0289, 2.4952, -0.1708]]], np.float32)
BQ_XYZ = np.array([[[-0.0740, 1.3147, -1.3625], [0.5555, 1.0399, -1.3634], [-0.4003, 2.4666, -0.5116],
                    [-0.5251, 2.4379, -0.8466], [-0.9691, 1.1418, -1.3733], [-0.2232, 0.9561, -1.3626],
                    [-2.2769, 2.7817, -0.2334], [-0.2822, 1.3192, -1.3645], [0.1533, 1.5024, -1.0432],
                    [0.4917, 1.1529, -1.3496]],
                   [[-2.0289, 2.4952, -0.1708], [-0.7188, 0.9956, -0.5096], [-2.0668, 6.0278, -0.4875],
                    [-1.9304, 3.3092, 0.6610], [0.0949, 1.4332, 0.3140], [-1.2879, 2.0008, -0.7791],
                    [-0.7252, 0.9611, -0.6371], [0.4066, 1.4211, -0.2947], [0.3220, 1.4447, 0.3548],
                    [-0.9744, 2.3856, -1.2000]]], np.float32)
BQ_EXPECTED_02 = np.array([[[0, 0, 0, 0, 0], [6, 6, 6, 6, 6], [2, 2, 2, 2, 2], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]],
                           [[0, 0, 0, 0, 0], [2, 2, 2, 2, 2], [7, 7, 7, 7, 7], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]])
BQ_EXPECTED_DILATED = np.array([[[0, 5, 7, 0, 0], [6, 6, 6, 6, 6], [2, 3, 2, 2, 2], [0, 5, 7, 0, 0], [0, 5, 7, 0, 0]],
                                [[0, 0, 0, 0, 0], [2, 2, 2, 2, 2], [7, 7, 7, 7, 7], [0, 0, 0, 0, 0], [0, 0, 0, 0, 0]]])


def test_fps_upstream_kat():
    assert np.array_equal(native.fps(FPS_XYZ, 3), FPS_EXPECTED)


def test_ball_query_upstream_kat():
    assert np.array_equal(native.ball_query(0, 0.2, 5, BQ_XYZ, BQ_NEW), BQ_EXPECTED_02)
    assert np.array_equal(native.ball_query(0.2, 0.4, 5, BQ_XYZ, BQ_NEW), BQ_EXPECTED_DILATED)


def test_fps_tie_rule_closed_form():
    """F6: among points of maximal temp the kernel picks min (bitrev10(k mod 1024), k)."""
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 6, (1, 3000, 3)).astype(np.float32)        # heavy ties
    idx = native.fps(pts, 64)[0]
    temp = np.full(3000, 1e10, np.float32)
    br = np.array([int(format(k % 1024, "010b")[::-1], 2) for k in range(3000)])
    old = 0
    for j in range(1, 64):
        d = ((pts[0] - pts[0, old]) ** 2).sum(1).astype(np.float32)
        temp = np.minimum(temp, d)
        cand = np.flatnonzero(temp == temp.max())
        old = cand[np.lexsort((cand, br[cand]))[0]]
        assert idx[j] == old


def test_knn_topk_matches_stable_sort():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 12, (50, 3)).astype(np.float32)
    k = rng.integers(0, 12, (400, 3)).astype(np.float32)
    val, idx = native.knn_topk(q, k, 4)
    d = torch.norm(torch.from_numpy(q)[:, None] - torch.from_numpy(k)[None], dim=-1)
    v, i = torch.sort(d, dim=1, stable=True)
    assert np.array_equal(idx, i[:, :4].numpy()) and np.allclose(val, v[:, :4].numpy(), atol=0)


def test_bev_pool_matches_cumsum_trick():
    """Cross-check against the pure-torch QuickCumsum formulation the reference also ships
    (ViewTransformerLSSBEVDepth.py:27-62,222-236)."""
    rng = np.random.default_rng(2)
    n, C, B, X, Y, Z = 5000, 6, 1, 9, 7, 3
    feats = torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32))
    coords = torch.from_numpy(np.stack([rng.integers(0, X, n), rng.integers(0, Y, n), rng.integers(0, Z, n),
                                        np.zeros(n, np.int64)], 1))
    got = ref_cpu.bev_pool(feats, coords, B, Z, X, Y)                 # [B,C,Z,X,Y]
    dense = torch.zeros(B, Z, X, Y, C, dtype=torch.float64)
    dense.index_put_((coords[:, 3], coords[:, 2], coords[:, 0], coords[:, 1]), feats.double(), accumulate=True)
    assert torch.allclose(got.double(), dense.permute(0, 4, 1, 2, 3), atol=1e-5)


@pytest.mark.parametrize("name", sorted(cases.FUSER_CASES))
def test_fuser_oracle_vs_golden(name, golden):
    c, g = cases.FUSER_CASES[name], golden(name)
    img, pts = cases.fuser_inputs(c)
    from co_occ_amd.fuser import BiFuser_N
    sd = synth.random_state_dict(BiFuser_N(c["C"], c["C"], c["knum"]).state_dict(), seed=c["seed"])
    assert sorted(sd) == list(g["keys"])
    o = ref_cpu.bifuser_fuse(sd, img, pts, c["knum"])
    assert o["inds_img"].shape[0] == int(g["n_img"]) and o["inds_pts"].shape[0] == int(g["n_pts"])
    assert np.array_equal(o["near_img"].numpy(), g["near_img"]) and np.array_equal(o["near_pts"].numpy(), g["near_pts"])
    out = ref_cpu.con_enc(sd, o["all_feats"])
    assert np.allclose(out.numpy(), g["out"], atol=1e-5)


def test_decoder_oracle_vs_golden(golden):
    c, g = cases.DECODER_CASE, golden("decoder")
    x, rig, img_feats = cases.decoder_inputs(c)
    import co_occ_amd as pkg
    cfg = synth.model_cfg(C=c["C"], block_inplanes=c["block_inplanes"], out_channels=c["fpn_out"],
                          cascade_ratio=c["cascade_ratio"], final_occ_size=c["final_occ_size"],
                          point_cloud_range=c["point_cloud_range"])
    enc, neck, head = pkg.build_backbone(cfg["semantic_encoder"]), pkg.build_neck(cfg["semantic_neck"]), pkg.build_head(cfg["pts_bbox_head"])
    sds = [synth.random_state_dict(m.state_dict(), seed=c["seed"]) for m in (enc, neck, head)]
    for sd, key in zip(sds, ("enc_keys", "neck_keys", "head_keys")):
        assert sorted(sd) == list(g[key])
    mid = ref_cpu.resnet3d_forward(sds[0], x)
    sem = ref_cpu.fpn3d_forward(sds[1], mid)
    h = ref_cpu.occhead_forward(sds[2], sem, img_feats, synth.rig_transform(rig), c["cascade_ratio"], c["final_occ_size"],
                                c["point_cloud_range"])
    for i in range(4):
        assert_close(mid[i], g["mid%d" % i], what="mid%d" % i)
        assert_close(sem[i], g["sem%d" % i], what="sem%d" % i)
    assert_close(h["output_voxels"], g["occ"], what="occ")
    assert np.array_equal(h["fine_coord"].numpy(), g["fine_coord"])
    assert_close(h["fine_output"], g["fine_output"], what="fine")


def test_pool_geometry_oracle_vs_golden(golden):
    c, g = cases.POOL_CASE, golden("pool_geometry")
    rig = synth.camera_rig(c["ncam"], c["input_size"], seed=c["seed"])
    fr = ref_cpu.create_frustum(c["input_size"], c["downsample"], c["grid_config"]["dbound"])
    geom = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    assert np.allclose(geom.numpy(), g["geom"], atol=1e-5)
    vol = synth.lifted_volume(c["ncam"], fr.shape[0], c["fmap"], c["C"], seed=c["seed"])
    dx, bx, nx = ref_cpu.gen_dx_bx(c["grid_config"]["xbound"], c["grid_config"]["ybound"], c["grid_config"]["zbound"])
    pooled = ref_cpu.voxel_pooling(torch.from_numpy(g["geom"]), vol, dx, bx, nx)[0]
    nz = torch.nonzero(pooled.abs().sum(0))
    assert np.array_equal(nz.numpy(), g["pooled_nz_idx"])
    assert np.allclose(pooled[:, nz[:, 0], nz[:, 1], nz[:, 2]].t().numpy(), g["pooled_nz_val"], atol=1e-5)


def test_render_oracle_vs_golden(golden):
    c, g = cases.RENDER_CASE, golden("render")
    vf, rig = cases.render_inputs(c)
    from co_occ_amd.render import MLP
    ssd = synth.random_state_dict(MLP(128, 1, net_depth=1, skip_layer=None).state_dict(), c["seed"])
    rsd = synth.random_state_dict(MLP(128, 3, net_depth=3, skip_layer=None).state_dict(), c["seed"] + 1)
    fr = ref_cpu.create_frustum(c["input_size"], c["downsample"], [2.0, 58.0, 0.5])
    gemo = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    for literal in (True, False):
        rgbs, depths = ref_cpu.render_block(ssd, rsd, vf, gemo, literal=literal)
        assert np.allclose(rgbs.numpy(), g["rgbs"], atol=1e-5) and np.allclose(depths.numpy(), g["depths"], atol=1e-4)
    r, d = ref_cpu.render_camera(ssd, rsd, vf[0], gemo[0, 0], literal=False)
    assert np.allclose(r.numpy(), g["rgb_maps"][0], atol=1e-5) and np.allclose(d.numpy(), g["depth_maps"][0], atol=1e-4)


def test_rays_oracle_vs_golden(golden):
    c, g = cases.RAY_CASE, golden("rays")
    vol, o, d, raw = cases.ray_inputs(c)
    pts, z = ref_cpu.sample_along_camera_ray(o, d, c["near_far"], c["n_samples"])
    assert np.allclose(pts.numpy(), g["pts"], atol=1e-6)
    pi, zi = ref_cpu.sample_along_camera_ray(o, d, c["near_far"], c["n_samples"], inv_uniform=True)
    assert np.allclose(pi.numpy(), g["pts_inv"], atol=1e-5)
    feat, mask = ref_cpu.volume_sampling(pts, vol, c["aabb"])
    assert np.allclose(feat.numpy(), g["feat"], atol=1e-6) and np.array_equal(mask.numpy(), g["mask"])
    r = ref_cpu.raw2outputs(raw, z, mask, False)
    assert np.allclose(r["rgb"].numpy(), g["rgb"], atol=1e-6) and np.allclose(r["depth"].numpy(), g["depth"], atol=1e-6)
    assert np.allclose(ref_cpu.raw2outputs(raw, z, mask, True)["rgb"].numpy(), g["rgb_white"], atol=1e-6)
    assert np.allclose(ref_cpu.get_weights(raw[..., 3:4], z).numpy(), g["get_weights"], atol=1e-6)


def test_eval_oracle_vs_golden(golden):
    """coocc_ray.py:659-684,726-730: SC / SSC / visible-SSC confusion matrices of the unmodified reference."""
    c, g = cases.EVAL_CASE, golden("eval")
    for tag, same in (("coarse", False), ("full", True)):
        pred, gt, vis = cases.eval_inputs(c, same)
        sc = ref_cpu.evaluation_semantic(pred, gt, 'SC', vis)[0]
        ssc, occ, _ = ref_cpu.evaluation_semantic(pred, gt, 'SSC', vis)
        assert np.array_equal(sc, g[tag + "_sc"]) and np.array_equal(ssc, g[tag + "_ssc"])
        assert np.array_equal(occ, g[tag + "_occ"])
        assert ssc.sum() == int((gt != 255).sum()) and sc.sum() == ssc.sum()


def test_voxelize_oracle_vs_golden(golden):
    """voxelization_cpu.cpp:44-104 restated == the reference's numpy points_to_voxel (golden), incl. the max_voxels and
    max_points truncations; HardSimpleVFE mean."""
    from oracle import ref_lidar
    c, g = cases.LIDAR_CASE, golden("voxelize")
    pts = cases.lidar_points(c)
    vox, coors, num = ref_lidar.hard_voxelize(pts, c["voxel_size"], c["point_cloud_range"], c["max_points"], c["max_voxels"])
    assert np.array_equal(vox, g["voxels"]) and np.array_equal(coors, g["coors"]) and np.array_equal(num, g["num"])
    assert vox.shape[0] == c["max_voxels"] and int(num.max()) == c["max_points"]
    m = ref_lidar.vfe_mean(vox, num, 4)
    assert m.shape == (vox.shape[0], 4) and torch.allclose(m[0], torch.from_numpy(vox[0, :num[0], :4]).mean(0))


@pytest.mark.parametrize("shape,n", [((6, 9, 8), 40), ((5, 12, 7), 90), ((4, 4, 4), 3)])
def test_sparse_conv_rules_match_vendored_spconv_v1_rulebook(shape, n):
    """The LiDAR encoder's sparse convolutions are restated as MASKED DENSE convolutions (oracle/ref_lidar.py) because spconv
    2.3.6 is absent.  mmdetection3d vendors spconv v1 (M/ops/spconv): its rule-book generation (geometry.h:25-86, :144-192,
    :247-297; output size ops.py:20-31), restated in ref_lidar.spconv_v1_*, must give the same active sets and values as the
    masked-dense form -- SubMConv3d(k3) and SparseConv3d(k3, s2, p1), the two layer types of sparse_lidar_enc.py."""
    import torch.nn.functional as F
    from oracle import ref_lidar as RL
    g = np.random.default_rng(sum(shape) + n)
    D, H, W = shape
    lin = g.choice(D * H * W, size=min(n, D * H * W), replace=False)
    coors = np.stack([lin // (H * W), (lin // W) % H, lin % W], 1)
    Cin, Cout = 4, 6
    feats = torch.from_numpy(g.standard_normal((len(lin), Cin)).astype(np.float32))
    w2 = torch.from_numpy(g.standard_normal((Cout, 3, 3, 3, Cin)).astype(np.float32))      # spconv 2.x layout [Cout,kd,kh,kw,Cin]
    w1 = w2.permute(1, 2, 3, 4, 0).contiguous()                                             # v1 layout [kd,kh,kw,Cin,Cout]
    vol, mask = RL._to_dense(feats, torch.from_numpy(coors), shape)
    wt = w2.permute(0, 4, 1, 2, 3).contiguous()
    # SubMConv3d: outputs exactly at the active inputs
    out, oc, osz = RL.spconv_v1_conv(feats, coors, shape, w1, subm=True)
    dense = F.conv3d(vol, wt, padding=1) * mask
    assert list(osz) == list(shape) and np.array_equal(oc, coors)
    got = dense[0][:, coors[:, 0], coors[:, 1], coors[:, 2]].t()
    assert torch.allclose(got, out, atol=1e-5)
    # SparseConv3d(k3, s2, p1): an output is active iff its receptive field holds an active input
    out, oc, osz = RL.spconv_v1_conv(feats, coors, shape, w1, k=3, s=2, p=1)
    newmask = F.max_pool3d(mask.float(), 3, 2, 1) > 0
    dense = F.conv3d(vol, wt, stride=2, padding=1)
    assert list(osz) == list(dense.shape[2:])
    act = newmask[0, 0].nonzero().numpy()
    assert {tuple(c) for c in oc.tolist()} == {tuple(c) for c in act.tolist()}
    got = dense[0][:, oc[:, 0], oc[:, 1], oc[:, 2]].t()
    assert torch.allclose(got, out, atol=1e-5)
    assert float((dense * ~newmask).abs().max()) == 0.0          # nothing outside the active set


def _hand_fixture():
    import json
    import os
    from conftest import GOLDEN
    fx = json.load(open(os.path.join(GOLDEN, "sparse_rules_5x5x5.json")))
    W = torch.zeros(2, 3, 3, 3, 2)                       # spconv 2.x layout [Cout, kd, kh, kw, Cin]
    for co in range(2):
        for ci in range(2):
            for kz in range(3):
                for ky in range(3):
                    for kx in range(3):
                        W[co, kz, ky, kx, ci] = (1, -1)[co] * (1, 10)[ci] * (1 + kx + 3 * ky + 9 * kz)
    return fx, W


def test_sparse_conv_rules_match_hand_computed_fixture():
    """SubMConv3d / SparseConv3d(k3, s2, p1) on a 5x5x5 grid against outputs derived BY HAND (tests/golden/sparse_rules_5x5x5.json,
    every number justified in its 'derivation'): the masked-dense form the encoder oracle uses AND the spconv-v1 rule book
    (vendored geometry.h restated) give exactly those active sets and values -- weight layout [Cout, kd, kh, kw, Cin] included."""
    from oracle import ref_lidar as RL
    fx, W = _hand_fixture()
    feats, coors, shape = np.asarray(fx["feats"], np.float32), np.asarray(fx["coors_zyx"]), fx["grid_zyx"]
    got = RL.subm_conv3d(feats, coors, shape, W)
    assert np.array_equal(got.numpy(), np.asarray(fx["subm_k3"]["out"], np.float32))
    rows, oc, osz = RL.sparse_conv3d(feats, coors, shape, W)
    assert osz == fx["sparse_k3_s2_p1"]["out_shape_zyx"]
    assert np.array_equal(oc.numpy(), np.asarray(fx["sparse_k3_s2_p1"]["coors_zyx"]))       # ascending (z,y,x): the fixture's order
    assert np.array_equal(rows.numpy(), np.asarray(fx["sparse_k3_s2_p1"]["out"], np.float32))
    # the vendored v1 rule book (weights [kd, kh, kw, Cin, Cout])
    w1 = W.permute(1, 2, 3, 4, 0).contiguous().numpy()
    out, oc1, _ = RL.spconv_v1_conv(feats, coors, shape, w1, subm=True)
    assert np.array_equal(oc1, coors) and np.array_equal(out.numpy(), np.asarray(fx["subm_k3"]["out"], np.float32))
    out, oc1, osz1 = RL.spconv_v1_conv(feats, coors, shape, w1, k=3, s=2, p=1)
    assert osz1 == fx["sparse_k3_s2_p1"]["out_shape_zyx"]
    order = np.lexsort((oc1[:, 2], oc1[:, 1], oc1[:, 0]))                                     # v1 creates sites in first-touch order
    assert np.array_equal(oc1[order], np.asarray(fx["sparse_k3_s2_p1"]["coors_zyx"]))
    assert np.array_equal(out.numpy()[order], np.asarray(fx["sparse_k3_s2_p1"]["out"], np.float32))
