"""CPU restatement of the LiDAR-side producer.  TEST INFRASTRUCTURE ONLY (tests/, smoke,
bench cpu_baseline) -- the product never imports this.

* ``hard_voxelize``: mmdetection3d/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:44-104 (== the deterministic path of
  voxelization_cuda.cu and the numpy ``points_to_voxel`` of mmdet3d/core/voxel/voxel_generator.py:76-209, which the
  reference's own test compares it with).  PINNED: tests/golden/voxelize.npz holds the output of the reference's
  ``points_to_voxel`` run here (oracle/gen_golden.py, numba stubbed out).
* ``vfe_mean``: mmdet3d/models/voxel_encoders/voxel_encoder.py:43-45.
* ``sparse_encoder_forward``: P/coocc/voxel_encoder/sparse_lidar_enc.py:66-190 with spconv 2.3.6's SubMConv3d /
  SparseConv3d written as masked dense convolutions.  **Parity unpinned**: spconv is an un-vendored dependency
  (docs/requirements_ref.txt:166) and cannot be imported or built here, so this restates its published semantics
  (SubM: outputs only at active inputs; SparseConv3d: an output is active iff its receptive field holds an active input;
  weights [Cout, kd, kh, kw, Cin]; ``dense()`` zero-fills).
"""
import numpy as np
import torch
import torch.nn.functional as F


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    points = np.asarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    rg = np.asarray(coors_range, dtype=np.float32)
    grid = np.round((rg[3:] - rg[:3]) / vs).astype(np.int32)
    F_ = points.shape[1]
    voxels = np.zeros((max_voxels, max_points, F_), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    lut = {}
    n = 0
    for i in range(points.shape[0]):
        c = np.floor((points[i, :3] - rg[:3]) / vs).astype(np.int64)
        if (c < 0).any() or (c >= grid).any():
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        idx = lut.get(key, -1)
        if idx == -1:
            if n >= max_voxels:
                continue
            idx = n
            n += 1
            lut[key] = idx
            coors[idx] = key
        if num[idx] < max_points:
            voxels[idx, num[idx]] = points[i]
            num[idx] += 1
    return voxels[:n], coors[:n], num[:n]


def vfe_mean(voxels, num_points, num_features):
    v = torch.as_tensor(voxels)
    return v[:, :, :num_features].sum(1) / torch.as_tensor(num_points).type_as(v).view(-1, 1)


def _to_dense(feats, coors, shape):
    D, H, W = shape
    vol = torch.zeros(1, feats.shape[1], D, H, W)
    mask = torch.zeros(1, 1, D, H, W, dtype=torch.bool)
    z, y, x = coors[:, 0].long(), coors[:, 1].long(), coors[:, 2].long()
    vol[0, :, z, y, x] = feats.t()
    mask[0, 0, z, y, x] = True
    return vol, mask


def _w(sd, key):
    """spconv 2.x weight [Cout, kd, kh, kw, Cin] -> torch conv3d [Cout, Cin, kd, kh, kw]."""
    return sd[key].permute(0, 4, 1, 2, 3).contiguous()


def _bn1d(x, mask, sd, prefix, eps=1e-5):
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    y = (x - m.view(1, -1, 1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1, 1) + eps) * sd[prefix + ".weight"].view(1, -1, 1, 1, 1) + \
        sd[prefix + ".bias"].view(1, -1, 1, 1, 1)
    return y * mask


def _gn_active(x, mask, sd, prefix, groups=16):
    """nn.GroupNorm applied to the [N, C] feature rows of the active voxels only."""
    idx = mask[0, 0].nonzero()
    rows = x[0][:, idx[:, 0], idx[:, 1], idx[:, 2]].t()
    rows = F.group_norm(rows, groups, sd[prefix + ".weight"], sd[prefix + ".bias"])
    out = torch.zeros_like(x)
    out[0][:, idx[:, 0], idx[:, 1], idx[:, 2]] = rows.t()
    return out


def sparse_encoder_forward(sd, feats, coors, shape_zyx, variant="8x", bn_eps=1e-5):
    """-> dense [1, C, W, H, D] (= x.dense().permute(0,1,4,3,2)) and the final active mask."""
    x, mask = _to_dense(torch.as_tensor(feats).float(), torch.as_tensor(coors), shape_zyx)
    subm = lambda x_, key, bias=None: F.conv3d(x_, _w(sd, key), sd.get(bias) if bias else None, padding=1) * mask
    x = F.relu(_gn_active(subm(x, "conv_input.0.weight", "conv_input.0.bias"), mask, sd, "conv_input.1")) * mask
    for stage in ("conv1", "conv2", "conv3"):
        i = 0
        if (stage + ".0.0.weight") in sd:                       # post_act_block: SparseConv3d(k3, s2, p1) + BN + ReLU
            newmask = F.max_pool3d(mask.float(), 3, 2, 1) > 0
            x = F.conv3d(x, _w(sd, stage + ".0.0.weight"), stride=2, padding=1)
            mask = newmask
            x = F.relu(_bn1d(x, mask, sd, stage + ".0.1", bn_eps)) * mask
            i = 1
        for b in (i, i + 1):                                   # two SparseBasicBlocks
            p = "%s.%d.net." % (stage, b)
            h = F.relu(_bn1d(F.conv3d(x, _w(sd, p + "0.weight"), padding=1) * mask, mask, sd, p + "1", bn_eps)) * mask
            h = _bn1d(F.conv3d(h, _w(sd, p + "3.weight"), padding=1) * mask, mask, sd, p + "4", bn_eps)
            x = F.relu(h + x) * mask
    x = F.conv3d(x, _w(sd, "conv_out.0.weight"), sd.get("conv_out.0.bias"), padding=1) * mask
    x = F.relu(_gn_active(x, mask, sd, "conv_out.1")) * mask
    return x.permute(0, 1, 4, 3, 2).contiguous(), mask
