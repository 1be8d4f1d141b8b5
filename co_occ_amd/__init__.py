"""co_occ_amd -- MI355X (gfx950) native implementation of Co-Occ's fused-voxel hot path
(GSFusion KNN lookup -> voxel scatter/pool -> 3D-conv occupancy decoder -> volume-rendering
regulariser) behind the reference's mmdet3d_plugin module / registry surface.

Compute lives in libcoocc_hip.so (hand-written HIP, C ABI in include/coocc_hip.h); this package
is the Python host side mirroring the reference's modules.  There is no CPU fallback.
"""
from . import registry
from .registry import (BACKBONES, DETECTORS, FUSION_LAYERS, HEADS, NECKS, build_backbone, build_detector,
                       build_fusion_layer, build_head, build_neck, register_into_mmdet)
from . import ops
from .ops import ball_query, bev_pool, furthest_point_sample
from .fuser import BiFuser_N
from .backbone import CustomResNet3D
from .neck import FPN3D
from .head import OccHead
from .view_transformer import ViewTransformerLiftSplatShootVoxel
from .render import MLP, raw2outputs, render_block, sample_along_camera_ray, volume_sampling
from .detector import COOCC_Ray, COOCC_Ray_L
from .view_transformer import get_frustum
from . import losses
from .core import invalidate_packs
from .calibration import calibrate
from . import apis, evaluation
from . import lidar
from .lidar import HardSimpleVFE, SparseLiDAREnc4x, SparseLiDAREnc8x, Voxelization
from .evaluation import SemanticEvaluator, cm_to_ious, evaluation_semantic

register_into_mmdet()

__all__ = ["BiFuser_N", "CustomResNet3D", "FPN3D", "OccHead", "ViewTransformerLiftSplatShootVoxel", "COOCC_Ray", "MLP",
           "bev_pool", "furthest_point_sample", "ball_query", "registry", "ops"]
