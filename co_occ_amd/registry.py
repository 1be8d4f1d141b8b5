"""Registry surface of the reference (mmcv ``Registry`` name lookup).

``dict(type='BiFuser_N', ...)`` configs resolve through ``FUSION_LAYERS.build(cfg)`` exactly as
``mmdet3d.models.builder`` does (M/models/builder.py:10-24,97-99).  When mmdet / mmdet3d are
importable, :func:`register_into_mmdet` additionally registers the classes into the real
registries (``force=True``), which is what makes ``projects/configs/coocc_nusc/*`` build our
modules unchanged.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError("%s is already registered in %s" % (key, self.name))
            self._module_dict[key] = cls
            return cls
        return _register(module) if module is not None else _register

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError("cfg must be a dict with a 'type' key, got %r" % (cfg,))
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        t = args.pop("type")
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError("%s is not in the %s registry" % (t, self.name))
        return cls(**args)


DETECTORS = Registry("detector")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
HEADS = Registry("head")
FUSION_LAYERS = Registry("fusion_layer")


def build_fusion_layer(cfg):
    return FUSION_LAYERS.build(cfg)


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg, **kw):
    return HEADS.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None, **overrides):
    """``overrides`` are merged into the config (e.g. ``external_encoders=True``, ``img_backbone=<module>``)."""
    return DETECTORS.build(dict(cfg, **overrides))


def register_into_mmdet(detectors=False):
    """Register our classes under the reference names into the real mmdet / mmdet3d registries (``force=True``).

    Default: the hot-path MODULES only (BiFuser_N, CustomResNet3D, FPN3D, OccHead, ViewTransformerLiftSplatShootVoxel and
    the LiDAR producer) -- the reference's own ``COOCC_Ray`` then builds them from its unchanged configs and keeps its
    encoders, losses and metrics.  ``detectors=True`` additionally replaces ``COOCC_Ray`` / ``COOCC_Ray_L`` with ours (HIP
    render block, on-device metrics; its image encoder is built back through these same registries).  Returns False when
    mmdet / mmdet3d are not importable."""
    try:
        from mmdet.models import builder as mb
        from mmdet3d.models import builder as m3b
    except Exception:
        return False
    pairs = [(BACKBONES, m3b.BACKBONES), (NECKS, m3b.NECKS), (HEADS, m3b.HEADS), (FUSION_LAYERS, m3b.FUSION_LAYERS)]
    if detectors:
        pairs.append((DETECTORS, mb.DETECTORS))
    from . import lidar
    pairs += [(lidar.VOXEL_ENCODERS, m3b.VOXEL_ENCODERS), (lidar.MIDDLE_ENCODERS, m3b.MIDDLE_ENCODERS)]
    for ours, theirs in pairs:
        for k, cls in ours.module_dict.items():
            theirs.register_module(name=k, force=True, module=cls)
    return True
