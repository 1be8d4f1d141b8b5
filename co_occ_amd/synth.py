"""Seeded synthetic inputs and weights.  numpy ``default_rng`` only, so the
same arrays are reproduced on every box / torch version; used by tests, the golden-vector
generator, ``bench.py`` and ``__graft_entry__.smoke()``.  There is no dataset or checkpoint
access in this environment: data = "synthetic", weights = random.
"""
import math
import zlib

import numpy as np
import torch

CONFIGS = {
    # name: fused grid, cams, feature map (fH,fW), knum, channels
    # configs[0]: dense enough that both voxel lists exceed 2048 (knum > 1 needs the large path)
    "config1": dict(grid=(50, 50, 4), ncam=1, fmap=(16, 44), knum=4, C=128, p_img=0.8, p_pts=0.5),
    "r50": dict(grid=(100, 100, 8), ncam=6, fmap=(16, 44), knum=2, C=128),          # coocc_multi_r50_256x704
    "r101": dict(grid=(100, 100, 8), ncam=6, fmap=(56, 100), knum=2, C=128),        # coocc_multi_r101_896x1600
    "stress200": dict(grid=(200, 200, 16), ncam=6, fmap=(16, 44), knum=2, C=128),   # north_star stress grid
    # north_star's literal stress workload: the 200x200x16 FUSED grid together with the 6 x 896 x 1600 frames
    "stress200_r101": dict(grid=(200, 200, 16), ncam=6, fmap=(56, 100), knum=2, C=128),
    # configs[4]: coocc_multi_r101_openoccupancy.py -- 512x512x40 occupancy grid, fused grid 128x128x10 (0.8 m), cascade 4
    "openocc": dict(grid=(128, 128, 10), ncam=6, fmap=(56, 100), knum=2, C=128, cascade_ratio=4, final_occ_size=(512, 512, 40),
                    point_cloud_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), input_size=(896, 1600)),
}


def pool_bounds(cfgname):
    """xbound / ybound / zbound of the view transformer for a CONFIGS entry (1 m voxels centred on the ego vehicle unless
    the entry names its own point_cloud_range, as the OpenOccupancy config does)."""
    c = CONFIGS[cfgname]
    X, Y, Z = c["grid"]
    if "point_cloud_range" in c:
        lo, hi = c["point_cloud_range"][:3], c["point_cloud_range"][3:]
        return dict(xbound=[lo[0], hi[0], (hi[0] - lo[0]) / X], ybound=[lo[1], hi[1], (hi[1] - lo[1]) / Y],
                    zbound=[lo[2], hi[2], (hi[2] - lo[2]) / Z])
    return dict(xbound=[-X / 2.0, X / 2.0, 1.0], ybound=[-Y / 2.0, Y / 2.0, 1.0], zbound=[-5.0, -5.0 + Z, 1.0])


def _rng(seed, key=""):
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(key.encode())])


def random_state_dict(reference_sd, seed=0, gain=1.0):
    """Fill a state_dict-shaped mapping with seeded values: kaiming-scaled weights, non-trivial
    eval-mode norm statistics (mean~N(0,.1), var~U(.5,1.5), gamma~U(.5,1.5), beta~N(0,.1)).
    ``gain`` multiplies every conv / linear weight (gain < 1 keeps the decoder's logits O(1-10), where the
    north_star's ABSOLUTE 1e-4 bound is meaningful; the kaiming default lets them grow to ~1e2)."""
    out = {}
    for k, v in reference_sd.items():
        shape = tuple(v.shape)
        g = _rng(seed, k)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shape, dtype=v.dtype)
            continue
        if k.endswith("posi_encoder.scales") or k.rsplit(".", 1)[-1] in ("dx", "bx", "nx", "frustum"):
            out[k] = v.clone()              # constants, not weights (the view transformer keeps its grid / frustum as Parameters)
            continue
        if k.endswith("running_mean"):
            a = g.normal(0, 0.1, shape)
        elif k.endswith("running_var"):
            a = g.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and k.endswith("weight"):
            a = g.uniform(0.5, 1.5, shape)
        elif len(shape) == 1:
            a = g.normal(0, 0.1, shape)
        elif len(shape) == 0:
            a = g.normal(0, 1.0, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = g.normal(0, gain * math.sqrt(2.0 / fan_in), shape)
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32)).to(v.dtype)
    return out


def voxel_inputs(grid, C=128, seed=1234, p_img=0.65, p_pts=0.12):
    """Per-modality voxel volumes [1,C,X,Y,Z]: camera volume dense-ish, LiDAR volume sparse,
    non-negative and clustered in the lower z range."""
    X, Y, Z = grid
    g = _rng(seed, "voxels")
    img = g.standard_normal((1, C, X, Y, Z), dtype=np.float32)
    img *= (g.random((1, 1, X, Y, Z)) < p_img)
    zprof = np.array([1.6 if 1 <= z <= 5 else 0.4 for z in range(Z)], dtype=np.float64)
    zprof = zprof / zprof.mean() * p_pts
    pts = np.maximum(g.standard_normal((1, C, X, Y, Z), dtype=np.float32), 0)
    pts *= (g.random((1, 1, X, Y, Z)) < zprof.reshape(1, 1, 1, 1, Z))
    return torch.from_numpy(img.astype(np.float32)), torch.from_numpy(pts.astype(np.float32))


def camera_rig(ncam=6, input_size=(256, 704), seed=1234):
    """nuScenes-like rig: rots [1,N,3,3] (camera->ego), trans, intrins, post_rots = I, post_trans = 0,
    bda = I.  Camera frame: x right, y down, z forward; ego: x forward, y left, z up."""
    H, W = input_size
    yaws = [-55, 0, 55, 110, 180, -110][:ncam] if ncam <= 6 else list(np.linspace(-180, 180, ncam, endpoint=False))
    base = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=np.float64)      # cam axes in ego
    rots, trans = [], []
    g = _rng(seed, "rig")
    for a in yaws:
        c, s = math.cos(math.radians(a)), math.sin(math.radians(a))
        Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)
        rots.append(Rz @ base)
        trans.append(np.array([1.5 * c, 0.5 * s + 0.1 * g.standard_normal(), 1.5]))
    fx = 1266.0 * (W / 1600.0)
    K = np.array([[fx, 0, W / 2.0], [0, fx, H / 2.0], [0, 0, 1]], dtype=np.float64)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    n = len(yaws)
    return dict(rots=f32(np.stack(rots))[None], trans=f32(np.stack(trans))[None],
                intrins=f32(np.stack([K] * n))[None], post_rots=torch.eye(3).repeat(1, n, 1, 1),
                post_trans=torch.zeros(1, n, 3), bda=torch.eye(3)[None],
                input_size=(torch.tensor([H]), torch.tensor([W])))


def rig_transform(rig):
    """``img_inputs[1:]``-shaped tuple consumed by OccHead (occ_head.py:217-221)."""
    return (rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"],
            rig["input_size"])


def image_feats(ncam, fmap, channels=512, seed=1234):
    g = _rng(seed, "imgfeat")
    return torch.from_numpy(g.standard_normal((1, ncam, channels) + tuple(fmap), dtype=np.float32))


def lift_inputs(ncam, D, fmap, C, seed=1234):
    """What DepthNet hands to Lift-Splat (LSSVoxel.py:131-133): softmax depth distribution [N,D,fH,fW] and context features
    [N,C,fH,fW] (the same arrays ``lifted_volume`` multiplies out)."""
    g = _rng(seed, "lift")
    fH, fW = fmap
    logit = g.normal(0, 2.0, (ncam, D, fH, fW)).astype(np.float32)
    e = np.exp(logit - logit.max(1, keepdims=True))
    depth = e / e.sum(1, keepdims=True)
    ctx = g.standard_normal((ncam, C, fH, fW), dtype=np.float32)
    return torch.from_numpy(depth.astype(np.float32)), torch.from_numpy(ctx)


def lifted_volume(ncam, D, fmap, C, seed=1234):
    """Lift output [1,N,D,fH,fW,C] = softmax-depth (x) context (LSSVoxel.py:135-143)."""
    g = _rng(seed, "lift")
    fH, fW = fmap
    logit = g.normal(0, 2.0, (ncam, D, fH, fW)).astype(np.float32)
    e = np.exp(logit - logit.max(1, keepdims=True))
    depth = e / e.sum(1, keepdims=True)
    ctx = g.standard_normal((ncam, C, fH, fW), dtype=np.float32)
    vol = depth[:, None] * ctx[:, :, None]                       # [N,C,D,H,W]
    return torch.from_numpy(np.ascontiguousarray(vol.transpose(0, 2, 3, 4, 1))[None].astype(np.float32))


def model_cfg_openocc(rendering=True):
    """The hot-path slice of projects/configs/coocc_nusc/coocc_multi_r101_openoccupancy.py (configs[4])."""
    c = CONFIGS["openocc"]
    cfg = model_cfg(C=c["C"], knum=c["knum"], cascade_ratio=c["cascade_ratio"], final_occ_size=c["final_occ_size"],
                    point_cloud_range=c["point_cloud_range"], input_size=c["input_size"], rendering=rendering)
    cfg["scale"] = 4
    cfg["img_view_transformer"].update(scale=4, grid_config=dict(xbound=[-51.2, 51.2, 0.8], ybound=[-51.2, 51.2, 0.8],
                                                                 zbound=[-5.0, 3.0, 0.8], dbound=[2.0, 58.0, 0.5]))
    return cfg


def model_cfg(C=128, knum=2, block_inplanes=(128, 256, 512, 1024), out_channels=256, num_cls=17,
              cascade_ratio=2, final_occ_size=(200, 200, 16), point_cloud_range=(-50, -50, -5.0, 50, 50, 3.0),
              input_size=(256, 704), rendering=True):
    """The hot-path slice of projects/configs/coocc_nusc/coocc_multi_r50_256x704.py:76-180."""
    bn = dict(type='SyncBN', requires_grad=True)
    return dict(
        type='COOCC_Ray', use_rendering=rendering, test_rendering=rendering, empty_idx=0, scale=16,
        occ_fuser=dict(type='BiFuser_N', knum=knum, in_channels=C, out_channels=C),
        semantic_encoder=dict(type='CustomResNet3D', depth=18, n_input_channels=C, block_inplanes=list(block_inplanes),
                              out_indices=(0, 1, 2, 3), norm_cfg=bn),
        semantic_neck=dict(type='FPN3D', with_cp=True, in_channels=list(block_inplanes), out_channels=out_channels,
                           norm_cfg=bn),
        pts_bbox_head=dict(type='OccHead', norm_cfg=bn, soft_weights=True, cascade_ratio=cascade_ratio,
                           sample_from_voxel=True, sample_from_img=True, final_occ_size=list(final_occ_size),
                           fine_topk=15000, empty_idx=0, num_level=4, in_channels=[out_channels] * 4,
                           out_channel=num_cls, point_cloud_range=list(point_cloud_range)),
        img_view_transformer=dict(type='ViewTransformerLiftSplatShootVoxel', scale=16, loss_depth_weight=1.0,
                                  loss_depth_type='bce', numC_Trans=C, vp_megvii=False,
                                  grid_config=dict(xbound=[-50, 50, 1.0], ybound=[-50, 50, 1.0], zbound=[-5.0, 3.0, 1.0],
                                                   dbound=[2.0, 58.0, 0.5]),
                                  data_config=dict(input_size=tuple(input_size))),
    )


def lidar_cfg(point_cloud_range=(-50, -50, -5.0, 50, 50, 3.0), out_channel=128):
    """The LiDAR-producer entries of projects/configs/coocc_nusc/coocc_multi_r50_256x704.py:121-134 (0.125 m voxels, <= 10 points per
    voxel, <= 120 k voxels at test time, SparseLiDAREnc8x on [800,800,64])."""
    return dict(pts_voxel_layer=dict(max_num_points=10, point_cloud_range=list(point_cloud_range), voxel_size=[0.125] * 3,
                                     max_voxels=(90000, 120000)),
                pts_voxel_encoder=dict(type='HardSimpleVFE', num_features=5),
                pts_middle_encoder=dict(type='SparseLiDAREnc8x', input_channel=4, base_channel=16, out_channel=out_channel,
                                        norm_cfg=dict(type='SyncBN', requires_grad=True), sparse_shape_xyz=[800, 800, 64]))


def lidar_points(n=280000, seed=8):
    """A synthetic 10-sweep nuScenes-like cloud [n,4] (x, y, z, intensity): range ~ sqrt(U) * 50 m, ground-hugging z."""
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(n, generator=g) ** 0.5 * 50
    th = torch.rand(n, generator=g) * 6.2832
    return torch.stack([r * torch.cos(th), r * torch.sin(th), torch.randn(n, generator=g) * 0.8 - 1.5, torch.rand(n, generator=g)], 1)
