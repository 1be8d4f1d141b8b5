"""CPU restatement of the LiDAR-side producer.  TEST INFRASTRUCTURE ONLY (tests/, smoke,
bench cpu_baseline) -- the product never imports this.

* ``hard_voxelize``: mmdetection3d/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:44-104 (== the deterministic path of
  voxelization_cuda.cu and the numpy ``points_to_voxel`` of mmdet3d/core/voxel/voxel_generator.py:76-209, which the
  reference's own test compares it with).  PINNED: tests/golden/voxelize.npz holds the output of the reference's
  ``points_to_voxel`` run here (oracle/gen_golden.py, numba stubbed out).
* ``vfe_mean``: mmdet3d/models/voxel_encoders/voxel_encoder.py:43-45.
* ``sparse_encoder_forward``: P/coocc/voxel_encoder/sparse_lidar_enc.py:66-190 with spconv 2.3.6's SubMConv3d /
  SparseConv3d written as masked dense convolutions.  spconv 2.3.6 is an un-vendored dependency
  (docs/requirements_ref.txt:166) and cannot be imported or built here (the vendored v1 sources need <cuda_runtime_api.h>
  even for their CPU functors: M/ops/spconv/include/tensorview/tensorview.h:16), so no golden from the real library
  exists ("parity unpinned" against the LIBRARY).  The layer semantics are instead pinned by three independent sources
  that must all agree (tests/test_oracle.py, tests/test_gpu_lidar.py):
    1. the published definitions -- Graham, Engelcke, van der Maaten, "3D Semantic Segmentation with Submanifold Sparse
       Convolutional Networks" (CVPR 2018), sec. 3: a submanifold convolution SC(m, n, f, s = 1) computes an output ONLY at
       sites whose CENTRE input is active and sums over the active inputs of its receptive field (the active set never
       dilates); Yan, Mao, Li, "SECOND: Sparsely Embedded Convolutional Detection" (Sensors 2018), sec. 3.1: a regular
       sparse convolution makes an output site active iff ANY input of its receptive field is active, via a rule book
       R[k] = {(input row, output row)} per kernel offset; spconv's own docs (docs/USAGE.md: "SubMConv3d ... indices
       unchanged"; SparseConv3d output shape (in + 2p - d(k-1) - 1)//s + 1; ``.dense()`` zero-fills the inactive sites;
       docs/SPCONV_2_BREAKING_CHANGEs.md: 2.x weights are KRSC = [out_channels, *kernel_size, in_channels], v1 RSCK =
       [*kernel_size, in_channels, out_channels]).  Both are cross-correlations (tap k reads input o*s - p + k);
    2. the spconv v1 rule-book code mmdetection3d vendors (M/ops/spconv/include/spconv/geometry.h, ops.py), restated in
       ``spconv_v1_*`` below (tests/test_oracle.py::test_sparse_conv_rules_match_vendored_spconv_v1_rulebook);
    3. a HAND-COMPUTED known-answer vector on a 5x5x5 grid (tests/golden/sparse_rules_5x5x5.json: five active voxels,
       two channels, weights 1 + kx + 3 ky + 9 kz in the 2.x layout; every output derived on paper in the file) that the
       masked-dense form, the v1 rule book AND the HIP rule-table kernels all reproduce exactly
       (test_sparse_conv_rules_match_hand_computed_fixture, test_sparse_layers_match_hand_computed_fixture).
  Row ORDER of a SparseConv3d output is library-internal (hash-table order) and not observable: every consumer is
  row-wise (BN1d / GroupNorm / ReLU) and the encoder ends in ``dense()``.
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
    vol = torch.zeros(1, feats.shape[1], D, H, W, dtype=feats.dtype)
    mask = torch.zeros(1, 1, D, H, W, dtype=torch.bool)
    z, y, x = coors[:, 0].long(), coors[:, 1].long(), coors[:, 2].long()
    vol[0, :, z, y, x] = feats.t()
    mask[0, 0, z, y, x] = True
    return vol, mask


def _w(sd, key):
    """spconv 2.x weight [Cout, kd, kh, kw, Cin] -> torch conv3d [Cout, Cin, kd, kh, kw]."""
    return sd[key].permute(0, 4, 1, 2, 3).contiguous()


def _bn1d(x, mask, sd, prefix, eps=1e-5, train=False):
    """BatchNorm1d over the [N, C] rows of the active voxels (sparse_lidar_enc.py:17-62 applies it to ``x.features``).
    ``train``: batch statistics over the active rows (biased variance), as ``model.train()`` does upstream."""
    m, v = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        n = mask.sum()
        m = (x * mask).sum(dim=(0, 2, 3, 4)) / n
        v = (((x - m.view(1, -1, 1, 1, 1)) * mask) ** 2).sum(dim=(0, 2, 3, 4)) / n
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


def subm_conv3d(feats, coors, shape_zyx, weight_2x, bias=None):
    """One SubMConv3d(k3) in the masked-dense form used below: weight [Cout, kd, kh, kw, Cin] -> rows at ``coors`` (same order)."""
    x, mask = _to_dense(torch.as_tensor(feats).float(), torch.as_tensor(coors), shape_zyx)
    y = F.conv3d(x, torch.as_tensor(weight_2x).float().permute(0, 4, 1, 2, 3).contiguous(), bias, padding=1) * mask
    c = torch.as_tensor(coors).long()
    return y[0][:, c[:, 0], c[:, 1], c[:, 2]].t()


def sparse_conv3d(feats, coors, shape_zyx, weight_2x, k=3, s=2, p=1):
    """One SparseConv3d(k, s, p) in the masked-dense form used below -> (rows, out coords in ascending (z,y,x) order, out shape)."""
    x, mask = _to_dense(torch.as_tensor(feats).float(), torch.as_tensor(coors), shape_zyx)
    newmask = F.max_pool3d(mask.float(), k, s, p) > 0
    y = F.conv3d(x, torch.as_tensor(weight_2x).float().permute(0, 4, 1, 2, 3).contiguous(), stride=s, padding=p) * newmask
    idx = newmask[0, 0].nonzero()
    return y[0][:, idx[:, 0], idx[:, 1], idx[:, 2]].t(), idx, list(y.shape[2:])


def sparse_encoder_forward(sd, feats, coors, shape_zyx, variant="8x", bn_eps=1e-5, train_bn=False):
    """-> dense [1, C, W, H, D] (= x.dense().permute(0,1,4,3,2)) and the final active mask.  ``train_bn``: BatchNorm1d with batch
    statistics (training mode); every operation is a torch op, so the result is differentiable in ``sd`` and ``feats``.
    Evaluated in the dtype of ``sd`` (float64 state dict = the fp64 anchor of the training-mode test)."""
    x, mask = _to_dense(torch.as_tensor(feats).to(sd["conv_input.0.weight"].dtype), torch.as_tensor(coors), shape_zyx)
    subm = lambda x_, key, bias=None: F.conv3d(x_, _w(sd, key), sd.get(bias) if bias else None, padding=1) * mask
    x = F.relu(_gn_active(subm(x, "conv_input.0.weight", "conv_input.0.bias"), mask, sd, "conv_input.1")) * mask
    for stage in ("conv1", "conv2", "conv3"):
        i = 0
        if (stage + ".0.0.weight") in sd:                       # post_act_block: SparseConv3d(k3, s2, p1) + BN + ReLU
            newmask = F.max_pool3d(mask.float(), 3, 2, 1) > 0
            x = F.conv3d(x, _w(sd, stage + ".0.0.weight"), stride=2, padding=1)
            mask = newmask
            x = F.relu(_bn1d(x, mask, sd, stage + ".0.1", bn_eps, train_bn)) * mask
            i = 1
        for b in (i, i + 1):                                   # two SparseBasicBlocks
            p = "%s.%d.net." % (stage, b)
            h = F.relu(_bn1d(F.conv3d(x, _w(sd, p + "0.weight"), padding=1) * mask, mask, sd, p + "1", bn_eps, train_bn)) * mask
            h = _bn1d(F.conv3d(h, _w(sd, p + "3.weight"), padding=1) * mask, mask, sd, p + "4", bn_eps, train_bn)
            x = F.relu(h + x) * mask
    x = F.conv3d(x, _w(sd, "conv_out.0.weight"), sd.get("conv_out.0.bias"), padding=1) * mask
    x = F.relu(_gn_active(x, mask, sd, "conv_out.1")) * mask
    return x.permute(0, 1, 4, 3, 2).contiguous(), mask


# ----------------------------------------------------------------------------- spconv v1 rule books (cross-check)
# spconv 2.3.6 (the reference's dependency) is absent, but mmdetection3d VENDORS spconv v1 (M/ops/spconv/), whose rule-book
# generation defines the same SubMConv3d / SparseConv3d semantics.  Restated here from the vendored sources so that the
# masked-dense formulation above can be checked against them (tests/test_oracle.py::test_sparse_conv_rules_*):
#   * output size:   M/ops/spconv/ops.py:20-31        (in + 2p - d(k-1) - 1) // s + 1
#   * valid outputs of one input position + kernel offset:  include/spconv/geometry.h:25-86 (getValidOutPos)
#   * SparseConv3d:  geometry.h:144-192 (getIndicePairsConv) -- an output site is created by the FIRST input that reaches it
#   * SubMConv3d:    geometry.h:247-297 (getIndicePairsSubM) -- outputs = inputs, pairs only where the output site is active
def spconv_v1_out_shape(shape, k, s, p, d=1):
    return [(n + 2 * p - d * (k - 1) - 1) // s + 1 for n in shape]


def spconv_v1_valid_out_pos(pos, k, s, p, out_shape, d=1):
    """geometry.h:25-86 for one input position (z, y, x): [(out_pos, kernel_offset)], offset = (kz*k + ky)*k + kx."""
    res = []
    rng = []
    for i in range(3):
        lo = (pos[i] - (k - 1) * d - 1 + s + p) // s
        hi = (pos[i] + p) // s
        rng.append(range(hi, lo - 1, -d) if hi >= lo else range(0))
    for oz in rng[0]:
        for oy in rng[1]:
            for ox in rng[2]:
                o = (oz, oy, ox)
                if any(v < 0 or v > out_shape[i] - 1 for i, v in enumerate(o)):
                    continue
                off, m = 0, 1
                for j in (2, 1, 0):
                    off += m * ((pos[j] - o[j] * s + p) // d)
                    m *= k
                res.append((o, off))
    return res


def spconv_v1_conv(feats, coors, shape, weight_kkkio, k=3, s=1, p=1, subm=False):
    """Rule-book sparse convolution with spconv-v1 semantics.  feats [N,Cin], coors [N,3] (z,y,x), weight [k,k,k,Cin,Cout]
    indexed by the kernel offset -> (out_feats [M,Cout], out_coors [M,3], out_shape)."""
    feats = torch.as_tensor(feats).float()
    coors = [tuple(int(v) for v in c) for c in np.asarray(coors)]
    out_shape = list(shape) if subm else spconv_v1_out_shape(shape, k, s, p)
    W = torch.as_tensor(weight_kkkio).float().reshape(k ** 3, weight_kkkio.shape[3], weight_kkkio.shape[4])
    if subm:
        grid = {c: j for j, c in enumerate(coors)}
        out_coors = list(coors)
    else:
        grid, out_coors = {}, []
    pairs = []
    for j, c in enumerate(coors):
        for o, off in spconv_v1_valid_out_pos(c, k, 1 if subm else s, p, out_shape):
            if subm:
                if o in grid:
                    pairs.append((off, j, grid[o]))
            else:
                if o not in grid:
                    grid[o] = len(out_coors)
                    out_coors.append(o)
                pairs.append((off, j, grid[o]))
    out = torch.zeros(len(out_coors), W.shape[2])
    for off, j, oi in pairs:
        out[oi] += feats[j] @ W[off]
    return out, np.asarray(out_coors, dtype=np.int64).reshape(-1, 3), out_shape
