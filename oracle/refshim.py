"""Import the UNMODIFIED reference hot-path modules from /root/reference on CPU.

TEST INFRASTRUCTURE ONLY, and only usable in the build container (the GPU box has
no /root/reference).  mmcv / mmdet / mmdet3d / cv2 are not installed here, so this
installs the few names the hot-path files need into ``sys.modules`` (registries,
``build_norm_layer``/``build_conv_layer``/``ConvModule``, ``BaseModule``, the fp16
decorators) and injects the C restatements of the three CUDA ops as ``mmdet3d.ops``
(oracle/native.py).  The reference files are then loaded *by path*; none of their
source is copied.  Used by oracle/gen_golden.py to produce tests/golden/*.npz and
by tests that validate oracle/ref_cpu.py against the real reference when present.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

from . import native

REF = os.environ.get("COOCC_REFERENCE", "/root/reference")
PLUGIN = os.path.join(REF, "projects", "mmdet3d_plugin")


def available():
    return os.path.isdir(PLUGIN)


class Registry:
    """Name -> class map with mmcv's ``register_module`` / ``build(cfg)`` semantics."""

    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **default_args):
        cfg = dict(cfg)
        for k, v in default_args.items():
            cfg.setdefault(k, v)
        t = cfg.pop("type")
        cls = self.module_dict[t] if isinstance(t, str) else t
        return cls(**cfg)


_NORMS = {"BN": ("bn", None), "BN1d": ("bn", nn.BatchNorm1d), "BN2d": ("bn", nn.BatchNorm2d),
          "BN3d": ("bn", nn.BatchNorm3d), "SyncBN": ("bn", nn.SyncBatchNorm), "GN": ("gn", nn.GroupNorm)}


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    abbr, cls = _NORMS[t]
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    if t == "GN":
        layer = nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        layer = (cls or nn.BatchNorm2d)(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return abbr + str(postfix), layer


_CONVS = {"Conv1d": nn.Conv1d, "Conv2d": nn.Conv2d, "Conv3d": nn.Conv3d, "Conv": nn.Conv2d}


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg) if cfg is not None else dict(type="Conv2d")
    t = cfg.pop("type")
    return _CONVS[t](*args, **kwargs, **cfg)


def build_upsample_layer(cfg, *args, **kwargs):
    raise NotImplementedError


class ConvModule(nn.Module):
    """conv -> norm -> act with mmcv's attribute names (.conv, .bn/.gn, .activate)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias="auto", conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"),
                 inplace=True, **kw):
        super().__init__()
        if bias == "auto":
            bias = norm_cfg is None
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, groups=groups, bias=bias)
        self.norm_name = None
        if norm_cfg is not None:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        self.activate = nn.ReLU(inplace=inplace) if act_cfg is not None else None

    def forward(self, x):
        x = self.conv(x)
        if self.norm_name:
            x = getattr(self, self.norm_name)(x)
        if self.activate is not None:
            x = self.activate(x)
        return x


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


def _passthrough_decorator(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]
    return lambda f: f


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _furthest_point_sample(points_xyz, num_points):
    return _t(native.fps(points_xyz.detach().cpu().numpy(), num_points))


def _ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    return _t(native.ball_query(min_radius, max_radius, sample_num, xyz.detach().cpu().numpy(),
                                center_xyz.detach().cpu().numpy()))


class _BevPoolExt:
    @staticmethod
    def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
        return _t(native.bev_pool_forward(x.numpy(), geom_feats.numpy(), interval_lengths.numpy(),
                                          interval_starts.numpy(), int(b), int(d), int(h), int(w)))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = _mod(name)
    m.__path__ = [path] if path else []
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_installed = {}


def load_config(name):
    """Execute a reference config file (projects/configs/coocc_nusc/<name>.py; plain Python, the ``_base_`` runtime files
    are not needed for the ``model`` dict) and return its namespace."""
    path = os.path.join(REF, "projects", "configs", "coocc_nusc", name + ".py")
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    return ns


def install():
    """Install the stubs and load the reference hot-path modules.  Idempotent."""
    if _installed:
        return _installed
    assert available(), "reference checkout not present (container-only helper)"
    regs = {k: Registry(k) for k in ("DETECTORS", "NECKS", "HEADS", "BACKBONES", "FUSION_LAYERS")}

    class _Builder:
        FUSION_LAYERS = regs["FUSION_LAYERS"]
        BACKBONES = regs["BACKBONES"]
        NECKS = regs["NECKS"]
        HEADS = regs["HEADS"]
        DETECTORS = regs["DETECTORS"]
        build_fusion_layer = staticmethod(lambda cfg: regs["FUSION_LAYERS"].build(cfg))
        build_backbone = staticmethod(lambda cfg: regs["BACKBONES"].build(cfg))
        build_neck = staticmethod(lambda cfg: regs["NECKS"].build(cfg))
        build_head = staticmethod(lambda cfg: regs["HEADS"].build(cfg))

    _pkg("mmcv")
    _mod("mmcv.cnn", build_norm_layer=build_norm_layer, build_conv_layer=build_conv_layer,
         build_upsample_layer=build_upsample_layer, ConvModule=ConvModule)
    _mod("mmcv.runner", BaseModule=BaseModule, auto_fp16=_passthrough_decorator,
         force_fp32=_passthrough_decorator)
    _pkg("mmdet")
    _mod("mmdet.models", DETECTORS=regs["DETECTORS"], NECKS=regs["NECKS"], HEADS=regs["HEADS"],
         BACKBONES=regs["BACKBONES"])
    _mod("mmdet.core", reduce_mean=lambda x: x)
    _pkg("mmdet.models.backbones")
    _mod("mmdet.models.backbones.resnet", BasicBlock=type("BasicBlock", (nn.Module,), {}))
    _pkg("mmdet3d")
    builder = _mod("mmdet3d.models.builder", **{k: v for k, v in vars(_Builder).items() if not k.startswith("__")})
    m3 = _pkg("mmdet3d.models")
    m3.builder = builder
    _mod("mmdet3d.models.detectors", CenterPoint=nn.Module)
    ops = _pkg("mmdet3d.ops")
    ops.furthest_point_sample = _furthest_point_sample
    ops.ball_query = _ball_query
    ops.gather_points = None  # imported at bifuser_n.py:11, never called
    # the real Python half of bev_pool (argsort + intervals) runs unmodified on top of the C kernel
    bp = _pkg("mmdet3d.ops.bev_pool", os.path.join(REF, "mmdetection3d", "mmdet3d", "ops", "bev_pool"))
    _mod("mmdet3d.ops.bev_pool.bev_pool_ext", bev_pool_forward=_BevPoolExt.bev_pool_forward)
    bp.bev_pool_ext = sys.modules["mmdet3d.ops.bev_pool.bev_pool_ext"]
    bpm = _load("mmdet3d.ops.bev_pool.bev_pool", os.path.join(bp.__path__[0], "bev_pool.py"))
    bp.bev_pool = bpm.bev_pool
    _mod("mmdet3d.ops.voxel_pooling", voxel_pooling=None)
    _mod("cv2", imwrite=lambda *a, **k: True)

    # projects.mmdet3d_plugin skeleton (the real __init__ files import the whole plugin)
    _pkg("projects")
    _pkg("projects.mmdet3d_plugin")
    utils = _pkg("projects.mmdet3d_plugin.utils", os.path.join(PLUGIN, "utils"))
    ct = _load("projects.mmdet3d_plugin.utils.coordinate_transform", os.path.join(PLUGIN, "utils", "coordinate_transform.py"))
    nm = _load("projects.mmdet3d_plugin.utils.nerf_mlp", os.path.join(PLUGIN, "utils", "nerf_mlp.py"))
    _load("projects.mmdet3d_plugin.utils.nusc_param", os.path.join(PLUGIN, "utils", "nusc_param.py"))
    _mod("projects.mmdet3d_plugin.utils.projection", Projector=None)
    rr = _load("projects.mmdet3d_plugin.utils.render_ray", os.path.join(PLUGIN, "utils", "render_ray.py"))
    _mod("projects.mmdet3d_plugin.utils.gaussian", generate_guassian_depth_target=None)
    # the real (pure-torch) loss files: OccHead.loss runs unmodified on top of them
    semk = _load("projects.mmdet3d_plugin.utils.semkitti", os.path.join(PLUGIN, "utils", "semkitti.py"))
    utils.__dict__.update(
        coarse_to_fine_coordinates=ct.coarse_to_fine_coordinates,
        project_points_on_img=ct.project_points_on_img, per_class_iu=None, fast_hist_crop=None,
        MLP=nm.MLP, VanillaNeRFRadianceField=nm.VanillaNeRFRadianceField,
        save_rendered_img=None, compute_psnr=lambda *a, **k: 0.0, semkitti=semk)
    for n in ("render_rays", "sample_along_camera_ray", "get_ray_direction_with_intrinsics", "get_rays",
              "sample_along_rays", "grid_generation", "unproject_image_to_rect", "compute_alpha_weights",
              "construct_ray_warps"):
        setattr(utils, n, getattr(rr, n, None))

    cooc = os.path.join(PLUGIN, "coocc")
    _pkg("projects.mmdet3d_plugin.coocc")
    for sub in ("fuser", "backbones", "necks", "dense_heads", "image2bev", "detectors"):
        _pkg("projects.mmdet3d_plugin.coocc." + sub, os.path.join(cooc, sub))
    lov = _load("projects.mmdet3d_plugin.coocc.dense_heads.lovasz_softmax", os.path.join(cooc, "dense_heads", "lovasz_softmax.py"))

    class BEVDepth(nn.Module):
        """Stand-in for P/coocc/detectors/bevdepth.py:16-34: only the child building."""

        def __init__(self, img_view_transformer=None, pts_bbox_head=None, **kwargs):
            super().__init__()
            self.img_view_transformer = regs["NECKS"].build(img_view_transformer) if img_view_transformer else None
            self.pts_bbox_head = regs["HEADS"].build(pts_bbox_head) if pts_bbox_head else None

    _mod("projects.mmdet3d_plugin.coocc.detectors.bevdepth", BEVDepth=BEVDepth)

    P = "projects.mmdet3d_plugin.coocc."
    out = dict(
        registries=regs,
        bifuser_n=_load(P + "fuser.bifuser_n", os.path.join(cooc, "fuser", "bifuser_n.py")),
        resnet3d=_load(P + "backbones.resnet3d", os.path.join(cooc, "backbones", "resnet3d.py")),
        fpn3d=_load(P + "necks.fpn3d", os.path.join(cooc, "necks", "fpn3d.py")),
        occ_head=_load(P + "dense_heads.occ_head", os.path.join(cooc, "dense_heads", "occ_head.py")),
        lss_bevdepth=_load(P + "image2bev.ViewTransformerLSSBEVDepth",
                           os.path.join(cooc, "image2bev", "ViewTransformerLSSBEVDepth.py")),
        lss_voxel=_load(P + "image2bev.ViewTransformerLSSVoxel",
                        os.path.join(cooc, "image2bev", "ViewTransformerLSSVoxel.py")),
        coocc_ray=_load(P + "detectors.coocc_ray", os.path.join(cooc, "detectors", "coocc_ray.py")),
        nerf_mlp=nm, render_ray=rr, coordinate_transform=ct, semkitti=semk, lovasz=lov,
    )
    _installed.update(out)
    return out
