"""CPU restatement of Co-Occ's fused-voxel hot path (the parity oracle).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  Float work runs on
torch CPU fp32 (the same ATen semantics the reference relies on), integer/index
work on numpy + oracle/c/coocc_oracle.c.  Every function cites the reference
file:line it restates (P/ = projects/mmdet3d_plugin/, M/ = mmdetection3d/mmdet3d/).

Weights are passed as plain ``{key: tensor}`` dicts whose keys are the reference
modules' ``state_dict`` keys, so a reference checkpoint slice
can be fed in unchanged.

Parity pinning: validated in the build container against the unmodified reference
modules (oracle/gen_golden.py -> tests/golden/*.npz); the three CUDA ops are pinned
by the upstream known-answer tests (tests/test_oracle.py).

Determinism rules the reference leaves open, fixed here:
  * top-K ties ordered by (d^2, key index);
  * duplicate-index assignment = last (highest centre ordinal) writer wins;
  * FPS ties resolved exactly as the CUDA block reduction does (C emulation);
  * bev_pool sums rows of one voxel in ascending point index (stable sort).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import native

BN_EPS = 1e-5
# None: exact arithmetic in the dtype of the inputs.  torch.bfloat16: every Conv3d of C0-C3 rounds BOTH operands to bf16
# first (fp32 accumulate and epilogue): the emulation of the product's reduced-precision MFMA path (configs[4]).
CONV_OPERAND_DTYPE = None


def _conv3d(x, w, bias=None, **kw):
    if CONV_OPERAND_DTYPE is not None:
        x, w = x.to(CONV_OPERAND_DTYPE).to(x.dtype), w.to(CONV_OPERAND_DTYPE).to(w.dtype)
    return F.conv3d(x, w, bias, **kw)


# --------------------------------------------------------------------------- helpers
def _bn(x, sd, prefix, eps=BN_EPS):
    """Eval-mode BatchNorm (running stats), any dimensionality."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd.get(prefix + ".weight"), sd.get(prefix + ".bias"), False, 0.0, eps)


def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def to_dtype(obj, dtype):
    """Cast every floating tensor in a (nested) dict / list / tuple to ``dtype`` -- the fp64 ANCHOR mode of the oracle:
    every function below computes in the dtype of its inputs, so ``to_dtype(sd, torch.float64)`` + fp64 inputs evaluate
    the same restatement in double precision (used by the parity tests to judge fp32 errors of ill-conditioned outputs:
    ``err(HIP, fp64) <= c * err(oracle fp32, fp64)``)."""
    if torch.is_tensor(obj):
        return obj.to(dtype) if obj.is_floating_point() else obj
    if isinstance(obj, dict):
        return {k: to_dtype(v, dtype) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_dtype(v, dtype) for v in obj)
    return obj


# --------------------------------------------------------------------------- K1..K5
def voxel_nonzero(feats):
    """K1: ``torch.nonzero(feats.sum(1))`` (P/coocc/fuser/bifuser_n.py:130-131).
    feats [B,C,X,Y,Z] -> [N,4] int64 rows (b,x,y,z) in lexicographic order."""
    return torch.nonzero(feats.sum(1))


def fps_nn_fast(query, key, fps_num=2048, radius=6, max_cluster_samples=200, dist_thresh=13.3, num=1):
    """K2-K5: ``BiFuser_N.fps_NN_fast`` (bifuser_n.py:38-125) under the canonical rules.

    query [Q,4], key [Nk,4] int64 (b,x,y,z).  Returns [Q] (num==1) or [num,Q] int64;
    -1 = unassigned.  The ``num>1 and Q<=fps_num`` branch raises IndexError in the
    reference (bifuser_n.py:90-93 indexes a 1-D tensor twice); restated as such."""
    q = query[:, 1:].float().numpy()
    k = key[:, 1:].float().numpy()
    Q = q.shape[0]
    if Q <= fps_num:
        if num != 1:
            raise IndexError("too many indices for tensor of dimension 1")
        val, nn = native.knn_topk(q, k, 1)  # dist.min(-1) returns the first minimum (F4)
        out = np.where(val[:, 0] < np.float32(dist_thresh), nn[:, 0], -1)
        return torch.from_numpy(out.astype(np.int64))
    repr_idx = native.fps(q[None], fps_num)[0].astype(np.int64)          # :63 / :97
    repr_q = q[repr_idx]
    val, nn = native.knn_topk(repr_q, k, num)                              # :101-103
    group = native.ball_query(0, radius, max_cluster_samples, q[None], repr_q[None])[0]  # :109
    out = native.knn_assign(val, nn, group, Q, dist_thresh)                # :104-125
    out = torch.from_numpy(out)
    return out[0] if num == 1 else out


def fps_nn_fast_parts(query, key, fps_num=2048, radius=6, max_cluster_samples=200, num=1):
    """Intermediate tensors of the large path, for per-kernel parity tests."""
    q = query[:, 1:].float().numpy()
    k = key[:, 1:].float().numpy()
    repr_idx = native.fps(q[None], fps_num)[0]
    repr_q = q[repr_idx.astype(np.int64)]
    val, nn = native.knn_topk(repr_q, k, num)
    group = native.ball_query(0, radius, max_cluster_samples, q[None], repr_q[None])[0]
    return dict(repr_idx=repr_idx, val=val, nn=nn, group=group)


# --------------------------------------------------------------------------- G1 + C0
def bifuser_fuse(sd, img_voxel_feats, pts_voxel_feats, knum):
    """K1..G1 of ``BiFuser_N.forward`` (bifuser_n.py:127-171): returns the 4C concat
    volume [B,X,Y,Z,4C] plus the index lists (the reference quirks are kept: -1 wraps to
    the last row, and for knum>1 the pts branch indexes ``inds_img`` (:158))."""
    B, C, X, Y, Z = img_voxel_feats.shape
    inds_img = voxel_nonzero(img_voxel_feats)
    inds_pts = voxel_nonzero(pts_voxel_feats)
    img_cl = img_voxel_feats.permute(0, 2, 3, 4, 1)
    pts_cl = pts_voxel_feats.permute(0, 2, 3, 4, 1)

    def at(vol, ind):
        return vol[ind[:, 0], ind[:, 1], ind[:, 2], ind[:, 3]]

    def knn_enc(x):
        return F.relu(F.linear(x, sd["knn_enc.0.weight"], sd["knn_enc.0.bias"]))

    sel_pts = at(pts_cl, inds_pts)
    near_img = fps_nn_fast(inds_pts, inds_img, num=knum)
    if knum == 1:
        g = at(img_cl, inds_img[near_img])
    else:
        g = torch.cat([at(img_cl, inds_img[near_img[i]]) for i in range(knum)], 1)
    fused_img_rows = knn_enc(g) * sel_pts

    sel_img = at(img_cl, inds_img)
    near_pts = fps_nn_fast(inds_img, inds_pts, num=knum)
    if knum == 1:
        g = at(pts_cl, inds_pts[near_pts])
    else:
        g = torch.cat([at(pts_cl, inds_img[near_pts[i]]) for i in range(knum)], 1)   # sic (:158)
    fused_pts_rows = knn_enc(g) * sel_img

    fused_img = torch.zeros(B, X, Y, Z, C, dtype=img_voxel_feats.dtype)
    fused_img[inds_pts[:, 0], inds_pts[:, 1], inds_pts[:, 2], inds_pts[:, 3]] = fused_img_rows
    fused_pts = torch.zeros(B, X, Y, Z, C, dtype=img_voxel_feats.dtype)
    fused_pts[inds_img[:, 0], inds_img[:, 1], inds_img[:, 2], inds_img[:, 3]] = fused_pts_rows
    allf = torch.cat([img_cl, pts_cl, fused_img, fused_pts], -1)
    return dict(all_feats=allf, inds_img=inds_img, inds_pts=inds_pts, near_img=near_img, near_pts=near_pts)


def con_enc(sd, all_feats_cl):
    """C0: ``con_enc`` (bifuser_n.py:23-30,172-173). in [B,X,Y,Z,4C] -> [B,C,X,Y,Z]."""
    x = all_feats_cl.permute(0, 4, 1, 2, 3)
    x = F.relu(_bn(_conv3d(x, sd["con_enc.0.weight"], padding=1), sd, "con_enc.1"))
    x = F.relu(_bn(_conv3d(x, sd["con_enc.3.weight"], padding=1), sd, "con_enc.4"))
    return x


def bifuser_forward(sd, img_voxel_feats, pts_voxel_feats, knum):
    return con_enc(sd, bifuser_fuse(sd, img_voxel_feats, pts_voxel_feats, knum)["all_feats"])


# --------------------------------------------------------------------------- C1..C3
RESNET_LAYERS = {10: [1, 1, 1, 1], 18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}


def resnet3d_forward(sd, x, depth=18, block_strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3)):
    """C1: ``CustomResNet3D.forward`` (P/coocc/backbones/resnet3d.py:196-205) with
    ``BasicBlock`` (:34-64) and shortcut type B (:181-184)."""
    x = F.relu(_bn(_conv3d(x, sd["input_proj.0.weight"]), sd, "input_proj.1"))
    res = []
    for s, nblocks in enumerate(RESNET_LAYERS[depth]):
        for b in range(nblocks):
            p = "layers.%d.%d." % (s, b)
            stride = block_strides[s] if b == 0 else 1
            out = F.relu(_bn(_conv3d(x, sd[p + "conv1.weight"], stride=stride, padding=1), sd, p + "bn1"))
            out = _bn(_conv3d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2")
            if (p + "downsample.0.weight") in sd:
                x = _bn(_conv3d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1")
            x = F.relu(out + x)
        if s in out_indices:
            res.append(x)
    return res


def fpn3d_forward(sd, inputs):
    """C2: ``FPN3D.forward`` (P/coocc/necks/fpn3d.py:70-108), BN-family norm."""
    lat = [F.relu(_bn(_conv3d(x, sd["lateral_convs.%d.0.conv.weight" % i]), sd, "lateral_convs.%d.0.bn" % i))
           for i, x in enumerate(inputs)]
    for i in range(len(lat) - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="trilinear",
                                                align_corners=False)
    return [F.relu(_bn(_conv3d(x, sd["fpn_convs.%d.0.conv.weight" % i], padding=1), sd, "fpn_convs.%d.0.bn" % i))
            for i, x in enumerate(lat)]


def _seq_1x1(sd, x, name):
    """conv1x1 -> BN -> ReLU -> conv1x1 (occ_pred_conv / voxel_soft_weights, occ_head.py:113-132)."""
    x = F.relu(_bn(_conv3d(x, sd[name + ".0.weight"], sd.get(name + ".0.bias")), sd, name + ".1"))
    return _conv3d(x, sd[name + ".3.weight"], sd.get(name + ".3.bias"))


def occhead_coarse(sd, voxel_feats, soft_weights=True):
    """C3: ``OccHead.forward_coarse_voxel`` (P/coocc/dense_heads/occ_head.py:149-171)."""
    occs = [F.relu(_bn(_conv3d(x, sd["occ_convs.%d.0.weight" % i], sd.get("occ_convs.%d.0.bias" % i), padding=1),
                       sd, "occ_convs.%d.1" % i)) for i, x in enumerate(voxel_feats)]
    n = len(occs)
    if soft_weights:
        w = torch.softmax(_seq_1x1(sd, occs[0], "voxel_soft_weights"), dim=1)
    else:
        w = torch.ones(occs[0].shape[0], n, 1, 1, 1, dtype=occs[0].dtype) / n
    size = occs[0].shape[2:]
    out = 0
    for f, wi in zip(occs, torch.unbind(w, dim=1)):
        f = F.interpolate(f, size=list(size), mode="trilinear", align_corners=False)
        out = out + f * wi.unsqueeze(1)
    return dict(out_voxel_feats=out, occ=_seq_1x1(sd, out, "occ_pred_conv"))


def coarse_to_fine_coordinates(coarse, ratio):
    """P/utils/coordinate_transform.py:3-21 (test-time branch: no random top-k).
    coarse [3,N] int64 -> [3, ratio^3*N], offset-major ordering."""
    r = torch.arange(ratio)
    off = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), dim=3).reshape(-1, 3)   # [r^3,3]
    fine = coarse[None] * ratio + off[:, :, None]                                     # [r^3,3,N]
    return fine.permute(1, 0, 2).reshape(3, -1)


def project_points_on_img(points, rots, trans, intrins, post_rots, post_trans, bda_mat, pts_range,
                          W_img, H_img, W_occ, H_occ, D_occ):
    """P/utils/coordinate_transform.py:25-65, nuScenes branch. points [1,N,3] ->
    uv [n_cam,N,1,2] in [-1,1], mask [1,N,n_cam]... returned as in the reference."""
    voxel_size = (pts_range[3:] - pts_range[:3]) / torch.tensor([W_occ - 1, H_occ - 1, D_occ - 1]).to(pts_range.dtype)
    points = points * voxel_size[None, None] + pts_range[:3][None, None]
    points = (bda_mat.inverse() @ points.unsqueeze(-1)).squeeze(-1)
    points = points.view(-1, 1, 3) - trans.view(1, -1, 3)
    points = rots.inverse().unsqueeze(0) @ points.unsqueeze(-1)
    points = (intrins.unsqueeze(0) @ points).squeeze(-1)
    d = points[..., 2:3]
    uv = points[..., :2] / (d + 1e-5)
    uv = (post_rots[..., :2, :2].unsqueeze(0) @ uv.unsqueeze(-1)).squeeze(-1) + post_trans[..., :2].unsqueeze(0)
    uv = torch.stack([(uv[..., 0] / (W_img - 1) - 0.5) * 2, (uv[..., 1] / (H_img - 1) - 0.5) * 2], -1)
    mask = (d[..., 0] > 1e-5) & (uv[..., 0] > -1) & (uv[..., 0] < 1) & (uv[..., 1] > -1) & (uv[..., 1] < 1)
    return uv.permute(2, 1, 0, 3), mask


def occhead_forward(sd, voxel_feats, img_feats, transform, cascade_ratio=2, final_occ_size=(200, 200, 16),
                    point_cloud_range=(-50, -50, -5.0, 50, 50, 3.0), empty_idx=0, soft_weights=True, fine_subset=None):
    """C3+C4: ``OccHead.forward`` eval branch (occ_head.py:173-265) with
    sample_from_voxel and sample_from_img, nuScenes data type, B == 1.

    transform = (rots, trans, intrins, post_rots, post_trans, bda, ..., (H_img, W_img))
    exactly as ``img_inputs[1:]`` (coocc_ray.py:527); only entries 0-5 and -1 are read."""
    o = occhead_coarse(sd, voxel_feats, soft_weights)
    ovf, coarse = o["out_voxel_feats"], o["occ"]
    res = dict(output_voxels=coarse, out_voxel_feats=ovf, fine_output=None, fine_coord=None)
    if cascade_ratio == 1:
        return res
    mask = coarse.argmax(1) != empty_idx                       # [B,W,H,D]
    assert mask.sum() > 0, "no foreground in coarse voxel"
    _, W, H, D = mask.shape
    coarse_coord = torch.nonzero(mask[0]).t()                  # [3,N] ascending (x,y,z) == masked meshgrid
    fine = coarse_to_fine_coordinates(coarse_coord, cascade_ratio)
    fine_all = fine
    if fine_subset is not None:
        # test hook for grids whose fine branch has ~1e7 points (OpenOccupancy, cascade 4): every fine point is an independent
        # row of the branch, so evaluating a subset of the columns gives exactly the rows of the full evaluation.
        # fine_subset: callable(n_fine) -> LongTensor of column indices
        sel = fine_subset(fine.shape[1]) if callable(fine_subset) else fine_subset
        fine = fine[:, sel]
    dt = ovf.dtype
    new_coord = fine[None].permute(0, 2, 1).to(dt).contiguous()
    g = fine.to(dt)
    g = torch.stack([(g[i] / (final_occ_size[i] - 1) - 0.5) * 2 for i in range(3)], 0)
    grid = g[None, None, None].permute(0, 4, 1, 2, 3)          # [1,N,1,1,3]
    vox = F.grid_sample(ovf[0:1].permute(0, 1, 4, 3, 2), grid, mode="bilinear", padding_mode="zeros",
                        align_corners=False)[0, :, :, 0, 0].permute(1, 0)          # [N,128]
    # image branch (occ_head.py:193-197, 222-236)
    f = img_feats[0]
    B_i, N_i, C_i, W_i, H_i = f.shape
    f = F.conv2d(f.reshape(-1, C_i, W_i, H_i), sd["img_mlp_0.0.weight"], sd["img_mlp_0.0.bias"])
    f = F.relu(F.group_norm(f, 16, sd["img_mlp_0.1.weight"], sd["img_mlp_0.1.bias"]))
    f = f.reshape(B_i, N_i, -1, W_i, H_i)
    pr = torch.tensor(np.array(point_cloud_range)).float().to(dt)
    uv, m = project_points_on_img(new_coord, transform[0][0:1], transform[1][0:1], transform[2][0:1],
                                  transform[3][0:1], transform[4][0:1], transform[5][0:1], pr,
                                  W_img=transform[-1][1][0:1], H_img=transform[-1][0][0:1],
                                  W_occ=W * cascade_ratio, H_occ=H * cascade_ratio, D_occ=D * cascade_ratio)
    s = F.grid_sample(f[0].contiguous(), uv.contiguous(), align_corners=True, mode="bilinear", padding_mode="zeros")
    s = (s * m.permute(2, 1, 0)[:, None]).sum(0)[:, :, 0].permute(1, 0)            # [N,128]
    s = F.linear(s, sd["img_mlp.0.weight"], sd["img_mlp.0.bias"])
    s = F.relu(F.group_norm(s, 16, sd["img_mlp.1.weight"], sd["img_mlp.1.bias"]))
    x = torch.cat([vox, s], 1)
    x = F.linear(x, sd["fine_mlp.0.weight"], sd["fine_mlp.0.bias"])
    x = F.relu(F.group_norm(x, 16, sd["fine_mlp.1.weight"], sd["fine_mlp.1.bias"]))
    x = F.linear(x, sd["fine_mlp.3.weight"], sd["fine_mlp.3.bias"])
    res.update(fine_output=x, fine_coord=fine, fine_coord_all=fine_all)
    return res


def scatter_fine(fine_pred, fine_coord, out_size, empty_idx=0):
    """``simple_test`` fine scatter (P/coocc/detectors/coocc_ray.py:546-550)."""
    pred = torch.full((1, fine_pred.shape[1]) + tuple(out_size), float(empty_idx), dtype=fine_pred.dtype)
    pred[:, :, fine_coord[0], fine_coord[1], fine_coord[2]] = fine_pred.permute(1, 0)[None]
    return pred


# --------------------------------------------------------------------------- P1, P2
def create_frustum(input_size, downsample, dbound):
    """P/coocc/image2bev/ViewTransformerLSSBEVDepth.py:104-115."""
    ogfH, ogfW = input_size
    fH, fW = ogfH // downsample, ogfW // downsample
    ds = torch.arange(*dbound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
    D = ds.shape[0]
    xs = torch.linspace(0, ogfW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
    ys = torch.linspace(0, ogfH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
    return torch.stack((xs, ys, ds), -1)


def get_geometry(frustum, rots, trans, intrins, post_rots, post_trans, bda):
    """P1: ``get_geometry`` (ViewTransformerLSSBEVDepth.py:117-150), 3x3 intrinsics."""
    B, N, _ = trans.shape
    p = frustum - post_trans.view(B, N, 1, 1, 1, 3)
    p = torch.inverse(post_rots).view(B, N, 1, 1, 1, 3, 3).matmul(p.unsqueeze(-1))
    p = torch.cat((p[..., :2, :] * p[..., 2:3, :], p[..., 2:3, :]), 5)
    if intrins.shape[3] == 4:                                # KITTI 3x4 / 4x4 intrinsics (:136-139)
        p = p - intrins[:, :, :3, 3].view(B, N, 1, 1, 1, 3, 1)
        intrins = intrins[:, :, :3, :3]
    combine = rots.matmul(torch.inverse(intrins))
    p = combine.view(B, N, 1, 1, 1, 3, 3).matmul(p).squeeze(-1)
    p = p + trans.view(B, N, 1, 1, 1, 3)
    if bda.shape[-1] == 4:
        p = torch.cat((p, torch.ones(*p.shape[:-1], 1, dtype=p.dtype)), dim=-1)
        p = bda.view(B, 1, 1, 1, 1, 4, 4).matmul(p.unsqueeze(-1)).squeeze(-1)[..., :3]
    else:
        p = bda.view(B, 1, 1, 1, 1, 3, 3).matmul(p.unsqueeze(-1)).squeeze(-1)
    return p


def get_frustum(rots, trans, intrins, post_rots, post_trans, bda, input_size, scale):
    """Module-level ``get_frustum`` of the detector (P/coocc/detectors/coocc_ray.py:732-776): the get_geometry chain on a
    frustum of input_size // scale pixels with the depth bins 2.0 .. 58.0 step 0.5 hard-coded."""
    H, W = int(input_size[0]), int(input_size[1])
    return get_geometry(create_frustum((H, W), scale, [2.0, 58.0, 0.5]), rots, trans, intrins, post_rots, post_trans, bda)


def gen_dx_bx(xbound, ybound, zbound):
    """ViewTransformerLSSBEVDepth.py:21-25."""
    dx = torch.Tensor([r[2] for r in (xbound, ybound, zbound)])
    bx = torch.Tensor([r[0] + r[2] / 2.0 for r in (xbound, ybound, zbound)])
    nx = torch.Tensor([(r[1] - r[0]) / r[2] for r in (xbound, ybound, zbound)])
    return dx, bx, nx


def bev_pool(feats, coords, B, D, H, W):
    """P2: ``bev_pool`` (M/ops/bev_pool/bev_pool.py:83-97) with a STABLE rank sort so each
    voxel sums its rows in ascending point index.  -> [B,C,D,H,W]."""
    B, D, H, W = int(B), int(D), int(H), int(W)
    ranks = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    order = torch.argsort(ranks, stable=True)
    feats, coords, ranks = feats[order], coords[order], ranks[order]
    kept = torch.ones(feats.shape[0], dtype=torch.bool)
    kept[1:] = ranks[1:] != ranks[:-1]
    starts = torch.where(kept)[0].int()
    lengths = torch.zeros_like(starts)
    lengths[:-1] = starts[1:] - starts[:-1]
    lengths[-1] = feats.shape[0] - starts[-1]
    out = native.bev_pool_forward(feats.numpy(), coords.int().numpy(), lengths.numpy(), starts.numpy(), B, D, H, W)
    return torch.from_numpy(out).permute(0, 4, 1, 2, 3).contiguous()


def voxel_pooling(geom_feats, x, dx, bx, nx):
    """P2: ``voxel_pooling`` (P/coocc/image2bev/ViewTransformerLSSVoxel.py:100-123):
    truncate-then-filter quantisation, bev_pool, -> [B,C,X,Y,Z]."""
    B, N, D, H, W, C = x.shape
    Np = B * N * D * H * W
    x = x.reshape(Np, C)
    g = ((geom_feats - (bx - dx / 2.)) / dx).long().view(Np, 3)
    bix = torch.cat([torch.full([Np // B, 1], ix, dtype=torch.long) for ix in range(B)])
    g = torch.cat((g, bix), 1)
    kept = (g[:, 0] >= 0) & (g[:, 0] < nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < nx[1]) & (g[:, 2] >= 0) & (g[:, 2] < nx[2])
    final = bev_pool(x[kept], g[kept], B, nx[2], nx[0], nx[1])
    return final.permute(0, 1, 3, 4, 2)


# --------------------------------------------------------------------------- R1..R3, L1
def mlp_forward(sd, x, net_depth):
    """R1: ``MLP.forward`` (P/utils/nerf_mlp.py:92-105), skip_layer=None as instantiated
    at coocc_ray.py:112-113."""
    for i in range(net_depth):
        x = F.relu(F.linear(x, sd["hidden_layers.%d.weight" % i], sd["hidden_layers.%d.bias" % i]))
    return F.linear(x, sd["output_layer.weight"], sd["output_layer.bias"])


RENDER_BOUNDS = ([-50., 50., 1.], [-50., 50., 1.], [-5., 3., 1.0])   # hard-coded at coocc_ray.py:577


def render_camera(sigma_sd, rgb_sd, voxel_feats, geom, literal=True):
    """R2, one camera: coocc_ray.py:575-616 (test) == :368-411 (train); ``rgb_sd=None`` is the depth-only branch
    (:441-478, no colour head: rgb_map is returned as zeros).
    voxel_feats [C,X,Y,Z], geom [D,H,W,3] ego metres -> rgb_map [H,W,3], depth_map [H,W]
    (pre-upsample).  ``literal=False`` evaluates the heads once per voxel (F5)."""
    dx, bx, nx = (t.to(geom.dtype) for t in gen_dx_bx(*RENDER_BOUNDS))
    g = (geom - (bx - dx / 2.)) / dx
    inside = (g[..., 0] >= 0) & (g[..., 0] < nx[0]) & (g[..., 1] >= 0) & (g[..., 1] < nx[1]) \
        & (g[..., 2] >= 0) & (g[..., 2] < nx[2])
    g = g * inside.unsqueeze(-1)                    # geom[~inside] *= 0
    D, H, W, _ = g.shape
    pts = g.long().permute(1, 2, 0, 3)              # [H,W,D,3]
    mask = inside.permute(1, 2, 0)
    if literal:
        feat = voxel_feats[:, pts[..., 0], pts[..., 1], pts[..., 2]].permute(1, 2, 3, 0)
        rgb = mlp_forward(rgb_sd, feat, 3) if rgb_sd is not None else torch.zeros(*feat.shape[:-1], 3, dtype=feat.dtype)
        sigma = mlp_forward(sigma_sd, feat, 1).squeeze(-1)
    else:
        C = voxel_feats.shape[0]
        tab = voxel_feats.reshape(C, -1).t()
        sig_t = mlp_forward(sigma_sd, tab, 1)
        rgb_t = mlp_forward(rgb_sd, tab, 3) if rgb_sd is not None else torch.zeros(tab.shape[0], 3, dtype=tab.dtype)
        lin = (pts[..., 0] * voxel_feats.shape[2] + pts[..., 1]) * voxel_feats.shape[3] + pts[..., 2]
        rgb, sigma = rgb_t[lin], sig_t[lin].squeeze(-1)
    rgb = torch.sigmoid(rgb * mask.unsqueeze(-1))   # rgb[~mask] = 0 ; sigmoid
    sigma = F.relu(sigma)
    p = pts.to(geom.dtype)
    dists = torch.norm(p[:, :, 1:, :] - p[:, :, :-1, :], dim=-1)
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    alpha = 1. - torch.exp(-F.relu(sigma * dists))
    trans = torch.cumprod(torch.cat([torch.ones(H, W, 1, dtype=alpha.dtype), 1. - alpha + 1e-10], -1), -1)[:, :, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights.unsqueeze(-1) * rgb, dim=-2)
    z_vals = torch.linspace(0, D, D).reshape(1, 1, D).to(alpha.dtype)
    depth_map = torch.sum(weights * z_vals, dim=-1)
    return rgb_map, depth_map


def upsample16(rgb_map, depth_map, scale=16):
    """coocc_ray.py:617-622: bilinear x16, align_corners=False."""
    d = F.interpolate(depth_map[None, None], scale_factor=scale, mode="bilinear")[0, 0]
    c = F.interpolate(rgb_map.permute(2, 0, 1)[None], scale_factor=scale, mode="bilinear").permute(0, 2, 3, 1)[0]
    return c, d


def render_block(sigma_sd, rgb_sd, voxel_feats, gemo, literal=True):
    """R2, all cameras. voxel_feats [1,C,X,Y,Z], gemo [1,N,D,H,W,3] ->
    rgbs [N,16H,16W,3], depths [N,16H,16W] (coocc_ray.py:570-627)."""
    B, N, D, H, W, _ = gemo.shape
    assert B == 1
    rgbs, depths = [], []
    for i in range(N):
        r, d = render_camera(sigma_sd, rgb_sd, voxel_feats[0], gemo[0, i], literal)
        r, d = upsample16(r, d)
        rgbs.append(r)
        depths.append(d)
    return torch.stack(rgbs), torch.stack(depths)


def render_losses(rgbs, depths, rgb_gt, depth_gt, D):
    """L1: coocc_ray.py:423-433."""
    d_bound = [2., 58., 0.5]
    dg = ((depth_gt - (d_bound[0] - d_bound[2] / 2.)) / d_bound[2]).clip(0, D)
    fg = dg > 0
    return dict(loss_depth_render=F.mse_loss(depths[fg] / D, dg[fg] / D), loss_rgb=F.mse_loss(rgbs, rgb_gt))


def train_render_losses(sigma_sd, rgb_sd, voxel_feats, gemo, imgs, depth_gt):
    """Render regulariser of ``forward_train``: camera branch (coocc_ray.py:358-434; imgs [N,3,H,W], depth_gt [N,H,W]) or, with
    ``rgb_sd=None`` / ``imgs=None``, the LiDAR-only depth branch (:436-496) whose ``gemo`` comes from ``get_frustum``."""
    D = gemo.shape[2]
    rgbs, depths = render_block(sigma_sd, rgb_sd, voxel_feats, gemo, literal=True)
    if rgb_sd is None or imgs is None:
        dg = ((depth_gt - 1.75) / 0.5).clip(0, D)
        fg = dg > 0
        return dict(loss_depth_render=F.mse_loss(depths[fg] / D, dg[fg] / D))
    return render_losses(rgbs, depths, imgs.permute(0, 2, 3, 1), depth_gt, D)


def volume_sampling(sample_pts, features, aabb):
    """R3: P/utils/render_ray.py:28-48.  features [1,C,D,W,H]; pts [N_rays,N_samples,3]."""
    B, C = features.shape[:2]
    aabb = torch.Tensor(aabb)
    nr, ns, _ = sample_pts.shape
    p = sample_pts.view(1, nr * ns, 1, 1, 3)
    norm = (p - aabb[0]) * (1.0 / (aabb[1] - aabb[0]) * 2) - 1
    feat = F.grid_sample(features, norm, align_corners=True, padding_mode="border")
    masks = ((norm < 1) & (norm > -1)).float().sum(dim=-1).view(nr, ns) == 3
    return feat.view(C, nr, ns).permute(1, 2, 0).contiguous(), masks


def sample_along_camera_ray(ray_o, ray_d, depth_range, N_samples, inv_uniform=False):
    """R3: P/utils/render_ray.py:147-191 with det=True (the random jitter of the
    non-deterministic branch has no parity definition)."""
    near = depth_range[0] * torch.ones_like(ray_d[..., 0])
    far = depth_range[1] * torch.ones_like(ray_d[..., 0])
    if inv_uniform:
        start = 1. / near
        step = (1. / far - start) / (N_samples - 1)
        z = 1. / torch.stack([start + i * step for i in range(N_samples)], dim=1)
    else:
        step = (far - near) / (N_samples - 1)
        z = torch.stack([near + i * step for i in range(N_samples)], dim=1)
    pts = z.unsqueeze(2) * ray_d.unsqueeze(1) + ray_o.unsqueeze(1)
    return pts, z


def raw2outputs(raw, z_vals, mask=None, white_bkgd=False):
    """R3: P/utils/render_ray.py:198-249."""
    rgb, sigma = raw[:, :, :3], raw[:, :, 3]
    alpha = 1. - torch.exp(-sigma)
    T = torch.cumprod(1. - alpha + 1e-10, dim=-1)[:, :-1]
    T = torch.cat((torch.ones_like(T[:, 0:1]), T), dim=-1)
    w = alpha * T
    rgb_map = torch.sum(w.unsqueeze(2) * rgb, dim=1)
    if white_bkgd:
        rgb_map = rgb_map + (1. - torch.sum(w, dim=-1, keepdim=True))
    if mask is not None:
        mask = mask.float().sum(dim=1) > 8
    depth = torch.sum(w * z_vals, dim=-1) / (torch.sum(w, dim=-1) + 1e-8)
    depth = torch.clamp(depth, z_vals.min(), z_vals.max())
    return dict(rgb=rgb_map, depth=depth, weights=w, mask=mask, alpha=alpha, z_vals=z_vals, transparency=T)


def get_weights(sigma, z_vals):
    """``COOCC_Ray.get_weights`` (coocc_ray.py:199-213)."""
    alpha = 1. - torch.exp(-sigma.squeeze(-1))
    T = torch.cumprod(1. - alpha + 1e-10, dim=-1)[:, :-1]
    T = torch.cat((torch.ones_like(T[:, 0:1]), T), dim=-1)
    return alpha * T


# --------------------------------------------------------------------------- whole path
def hot_path_forward(sd, img_voxel_feats, pts_voxel_feats, gemo, img_feats, transform, knum=2,
                     cascade_ratio=2, final_occ_size=(200, 200, 16), literal_render=True, dtype=None,
                     point_cloud_range=(-50, -50, -5.0, 50, 50, 3.0), render=True, fine_subset=None):
    """``COOCC_Ray.simple_test`` between the encoders and the metrics (coocc_ray.py:523-627)
    with ``test_rendering=True``: fuser -> encoder -> neck -> head -> render.
    ``dtype=torch.float64`` evaluates the same restatement in double precision (the anchor of the fp32 parity tests)."""
    if dtype is not None:
        sd, img_voxel_feats, pts_voxel_feats, gemo, img_feats, transform = to_dtype(
            (dict(sd), img_voxel_feats, pts_voxel_feats, gemo, img_feats, tuple(transform)), dtype)
    vf = bifuser_forward(_sub(sd, "occ_fuser."), img_voxel_feats, pts_voxel_feats, knum)
    mid = resnet3d_forward(_sub(sd, "semantic_encoder."), vf)
    sem = fpn3d_forward(_sub(sd, "semantic_neck."), mid)
    head = occhead_forward(_sub(sd, "pts_bbox_head."), sem, img_feats, transform, cascade_ratio, final_occ_size,
                           point_cloud_range, fine_subset=fine_subset)
    rgbs = depths = None
    if render:
        rgbs, depths = render_block(_sub(sd, "sigma_head."), _sub(sd, "rgb_head."), vf, gemo, literal_render)
    return dict(voxel_feats=vf, output_voxels=head["output_voxels"], fine_output=head["fine_output"],
                fine_coord=head["fine_coord"], fine_coord_all=head.get("fine_coord_all"), rgbs=rgbs, depths=depths)


def fast_hist(pred, label, max_label=18):
    """coocc_ray.py:726-730."""
    bc = np.bincount(max_label * label.flatten().astype(int) + pred.flatten(), minlength=max_label ** 2)
    return bc[:max_label ** 2].reshape(max_label, max_label)


def evaluation_semantic(pred, gt, eval_type, visible_mask=None, empty_idx=0):
    """coocc_ray.py:659-684: trilinear resample to the gt size, argmax, confusion matrices over gt != 255.
    Also returns the resampled logits so tests can bound argmax flips by the top-2 margin."""
    _, H, W, D = gt.shape
    up = F.interpolate(pred, size=[H, W, D], mode='trilinear', align_corners=False).contiguous()
    p = torch.argmax(up[0], dim=0).numpy()
    g = gt[0].numpy().astype(int)
    noise_mask = g != 255
    if eval_type == 'SC':
        g = g.copy()
        g[g != empty_idx] = 1
        p[p != empty_idx] = 1
        return fast_hist(p[noise_mask], g[noise_mask], max_label=2), None, up
    hist_occ = None
    if visible_mask is not None:
        mask = noise_mask & (visible_mask[0].numpy() != 0)
        hist_occ = fast_hist(p[mask], g[mask], max_label=17)
    return fast_hist(p[noise_mask], g[noise_mask], max_label=17), hist_occ, up


def predict_labels(pred, size):
    """P/coocc/apis/test.py:67-68 (and :198-201 for pred_f / pred_c): F.interpolate(trilinear, align_corners=False) to the
    ground-truth size, argmax over classes; save_output_nuscenes narrows to uint8 (P/coocc/apis/utils.py:65).
    Returns (labels u8 [1,H,W,D], resampled logits) -- the logits let tests bound argmax flips by the top-2 margin."""
    up = F.interpolate(pred, size=list(size), mode='trilinear', align_corners=False).contiguous()
    return torch.argmax(up, dim=1).numpy().astype(np.uint8), up
