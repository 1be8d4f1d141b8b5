"""``COOCC_Ray`` -- the hot path of P/coocc/detectors/coocc_ray.py between "the encoders have
produced per-modality voxel features" and "losses / metrics" (SURVEY.md section 1):

    occ_fuser (BiFuser_N) -> semantic_encoder (CustomResNet3D) -> semantic_neck (FPN3D)
    -> pts_bbox_head (OccHead) -> inline volume-render block (sigma_head / rgb_head)

Child modules are built from the same config dicts through the registries
(coocc_ray.py:80-83, bevdepth.py:21-34) and keep the reference attribute names, so a
reference checkpoint's ``occ_fuser.* / semantic_encoder.* / semantic_neck.* / pts_bbox_head.* /
sigma_head.* / rgb_head.*`` entries load with ``load_state_dict(strict=False)``.  The image and
LiDAR encoders upstream of the path are out of scope: their config entries are accepted and
ignored, and the path is fed their outputs.
"""
import torch
from torch import nn

from . import registry
from .core import to_rows
from .registry import DETECTORS
from .render import MLP, render_block, render_losses


@DETECTORS.register_module()
class COOCC_Ray(nn.Module):
    def __init__(self, voxel_size=None, n_voxels=None, loss_cfg=None, aabb=None, near_far_range=None,
                 N_samples=40, N_rand=4096, depth_supervise=False, use_nerf_mask=True, nerf_sample_view=3,
                 nerf_mode='volume', squeeze_scale=4, rgb_supervise=True, nerf_density=False,
                 rendering_test=False, disable_loss_depth=False, empty_idx=0, scale=16, white_bkgd=True,
                 occ_fuser=None, occ_encoder_backbone=None, occ_encoder_neck=None, density_encoder=None,
                 color_encoder=None, semantic_encoder=None, density_neck=None, color_neck=None,
                 semantic_neck=None, loss_norm=False, use_rendering=False, loss_voxel_ce_weight=1.0,
                 loss_voxel_sem_scal_weight=1.0, loss_voxel_geo_scal_weight=1.0, loss_voxel_lovasz_weight=1.0,
                 test_rendering=False, img_view_transformer=None, pts_bbox_head=None, pts_voxel_layer=None,
                 pts_voxel_encoder=None, pts_middle_encoder=None, **kwargs):
        super().__init__()
        self.ignored_cfg_keys = sorted(kwargs)      # img_backbone, img_neck, train/test_cfg ...
        # LiDAR-side producer (coocc_ray.py:215-234; bevdepth.py builds these three from the config)
        from . import lidar
        self.pts_voxel_layer = lidar.Voxelization(**pts_voxel_layer) if pts_voxel_layer else None
        self.pts_voxel_encoder = lidar.VOXEL_ENCODERS.build(pts_voxel_encoder) if pts_voxel_encoder else None
        self.pts_middle_encoder = lidar.MIDDLE_ENCODERS.build(pts_middle_encoder) if pts_middle_encoder else None
        self.empty_idx, self.scale = empty_idx, scale
        self.voxel_size, self.n_voxels, self.aabb = voxel_size, n_voxels, aabb
        self.near_far_range, self.N_samples, self.N_rand = near_far_range, N_samples, N_rand
        self.white_bkgd, self.loss_norm = white_bkgd, loss_norm
        self.use_rendering, self.test_rendering = use_rendering, test_rendering
        self.img_view_transformer = registry.build_neck(img_view_transformer) if img_view_transformer else None
        self.pts_bbox_head = registry.build_head(pts_bbox_head) if pts_bbox_head else None
        self.occ_fuser = registry.build_fusion_layer(occ_fuser) if occ_fuser is not None else None
        self.semantic_encoder = registry.build_backbone(semantic_encoder)
        self.semantic_neck = registry.build_neck(semantic_neck)
        if use_rendering:                                      # coocc_ray.py:111-113
            self.sigma_head = MLP(input_dim=128, output_dim=1, net_depth=1, skip_layer=None)
            self.rgb_head = MLP(input_dim=128, output_dim=3, net_depth=3, skip_layer=None)

    def extract_pts_feat(self, pts):
        """coocc_ray.py:215-234 (batch size 1): points [N,F] (or a one-element list) -> (pts_voxel_feats [1,C,X,Y,Z],
        [sparse features])."""
        if self.pts_middle_encoder is None:
            raise NotImplementedError("COOCC_Ray was built without pts_voxel_layer / pts_voxel_encoder / pts_middle_encoder")
        if isinstance(pts, (list, tuple)):
            assert len(pts) == 1, "batch size 1 (hard-coded upstream)"
            pts = pts[0]
        voxels, coors, num_points = self.pts_voxel_layer(pts)
        feats = self.pts_voxel_encoder(voxels, num_points, coors)
        enc = self.pts_middle_encoder(feats, coors, 1)
        return enc['x'], enc['pts_feats']

    def fuse(self, img_voxel_feats, pts_voxel_feats, search=None):
        """coocc_ray.py:252-256."""
        if self.occ_fuser is not None:
            return self.occ_fuser(img_voxel_feats, pts_voxel_feats, search=search)
        assert (img_voxel_feats is None) or (pts_voxel_feats is None)
        return img_voxel_feats if pts_voxel_feats is None else pts_voxel_feats

    def search(self, img_voxel_feats, pts_voxel_feats):
        """Index-search stage of the fuser for one sample (``BiFuser_N.search``): may be issued for sample i+1 on
        another stream / host thread while sample i runs ``forward_hot_path(..., search=...)``."""
        return self.occ_fuser.search(img_voxel_feats, pts_voxel_feats)

    def forward_hot_path(self, img_voxel_feats, pts_voxel_feats, gemo=None, img_feats=None, transform=None,
                         render=None, dense_fine=True, search=None):
        """simple_test (coocc_ray.py:520-627) minus encoders and metrics.

        img_voxel_feats / pts_voxel_feats: [1,C,X,Y,Z]; gemo: [1,N,D,fH,fW,3] (get_geometry);
        img_feats: [[1,N,512,fH,fW]]; transform: img_inputs[1:] (rots, trans, intrins, post_rots,
        post_trans, bda, ..., (H_img, W_img))."""
        voxel_feats = self.fuse(img_voxel_feats, pts_voxel_feats, search)
        mid = self.semantic_encoder.forward_rows(voxel_feats)
        sem = self.semantic_neck.forward_rows(mid)
        output = self.pts_bbox_head(voxel_feats=sem, img_feats=img_feats, transform=transform)
        res = dict(voxel_feats=voxel_feats, pred_c=output['output_voxels'][0], pred_f=None,
                   output_voxels_fine=output['output_voxels_fine'], output_coords_fine=output['output_coords_fine'])
        if output['output_voxels_fine'] is not None and dense_fine:
            cf = self.pts_bbox_head.cascade_ratio
            pc = res['pred_c']
            size = [pc.shape[2] * cf, pc.shape[3] * cf, pc.shape[4] * cf]     # == gt_occ size (coocc_ray.py:549)
            res['pred_f'] = self.pts_bbox_head.scatter_fine(output['output_voxels_fine'][0],
                                                            output['output_coords_fine'][0], size)
        do_render = (self.use_rendering and self.test_rendering) if render is None else render
        if do_render:
            rgbs, depths, maps = render_block(self.sigma_head, self.rgb_head, to_rows(voxel_feats), gemo, 16)
            res.update(rgbs=rgbs, depths=depths, render_maps=maps)
        return res

    def forward_train_hot_path(self, img_voxel_feats, pts_voxel_feats, gemo=None, img_feats=None, transform=None,
                               coarse_lin=None, render=True, generator=None):
        """Differentiable counterpart of ``forward_hot_path`` (frozen-statistics BN, as in ``co_occ_amd.autograd``):
        K1-K5 run as in inference (indices are not differentiated, as upstream), everything after them is an autograd
        Function over the HIP kernels.  Returns rows / tensors with ``grad_fn``:
        ``voxel_rows`` [V,C] (con_enc output), ``levels`` [(rows, geom)], ``out_voxel_rows`` [V,128], ``logit_rows``
        [V,ncls], ``fine_logits`` [8n,ncls] + ``fine_xyz`` [3,8n], ``rgbs`` / ``depths`` of the render block.
        ``coarse_lin``: int32 rows of the coarse voxels whose children the fine branch evaluates; default: the
        foreground voxels (argmax != empty), randomly thinned to ``fine_topk // ratio^3`` of them (the reference draws
        its training-time top-k among fine points, occ_head.py:204-205; here whole coarse voxels are drawn).
        The semantic losses (CE / lovasz / scal) are out of scope: apply them to the returned logits."""
        from . import autograd as ag
        B, C, X, Y, Z = img_voxel_feats.shape
        assert B == 1, "batch size 1 per GPU (coocc_ray.py:365)"
        V = X * Y * Z
        with torch.no_grad():
            sr = self.occ_fuser.search(img_voxel_feats.detach(), pts_voxel_feats.detach())
        rows = lambda t: t.float().permute(0, 2, 3, 4, 1).reshape(V, C)
        cat4 = ag.fuser_fuse_train(self.occ_fuser, rows(img_voxel_feats).contiguous(), rows(pts_voxel_feats).contiguous(), sr)
        ce = self.occ_fuser.con_enc
        vf, geom = ag.conv3d_rows(cat4, ce[0].weight, (1, X, Y, Z), bias=ce[0].bias, bn=ce[1], relu=True)
        vf, geom = ag.conv3d_rows(vf, ce[3].weight, geom, bias=ce[3].bias, bn=ce[4], relu=True)
        levels = ag.trunk_forward_train(None, self.semantic_encoder, self.semantic_neck, vf, geom)
        head = self.pts_bbox_head
        out_rows, logit_rows = ag.occhead_coarse_train(head, levels)
        res = dict(voxel_rows=vf, levels=levels, out_voxel_rows=out_rows, logit_rows=logit_rows)
        if head.cascade_ratio != 1 and (head.sample_from_img or head.sample_from_voxel):
            if coarse_lin is None:
                fg = torch.nonzero(logit_rows.detach().argmax(1) != head.empty_idx).flatten()
                cap = max(1, int(head.fine_topk) // head.cascade_ratio ** 3)
                if fg.numel() > cap:
                    sel = torch.randperm(fg.numel(), generator=generator, device=fg.device if generator is None else generator.device)[:cap]
                    fg = fg[sel.to(fg.device)].sort().values
                coarse_lin = fg.int()
            if coarse_lin.numel():
                res["fine_logits"], res["fine_xyz"] = ag.fine_branch_train(head, out_rows, (1, X, Y, Z), coarse_lin.contiguous(),
                                                                           img_feats, transform)
        if render and self.use_rendering:
            res["rgbs"], res["depths"] = ag.render_block_train(self.sigma_head, self.rgb_head, vf, (X, Y, Z), gemo)
        return res

    def render_losses(self, rgbs, depths, rgb_gt, depth_gt, D):
        """coocc_ray.py:423-433."""
        return render_losses(rgbs, depths, rgb_gt, depth_gt, D)

    def simple_test(self, img_metas=None, img=None, gt_depths=None, points=None, rescale=False, points_occ=None,
                    gt_occ=None, visible_mask=None, precomputed=None):
        """Reference signature (coocc_ray.py:520).  The encoders are out of scope, so the
        per-modality voxel features must be supplied via ``precomputed=dict(img_voxel_feats=...,
        pts_voxel_feats=..., gemo=..., img_feats=...)``; ``img[1:]`` is the transform."""
        if precomputed is None:
            raise NotImplementedError("COOCC_Ray: image/LiDAR encoders are outside the MI355X hot path; pass precomputed=")
        transform = img[1:] if img is not None else precomputed.get("transform")
        out = self.forward_hot_path(precomputed["img_voxel_feats"], precomputed["pts_voxel_feats"],
                                    precomputed.get("gemo"), precomputed.get("img_feats"), transform)
        out.update(output_voxels=out["pred_c"], target_voxels=gt_occ)
        if gt_occ is not None:   # coocc_ray.py:540-554, one kernel per prediction, histograms stay on the device
            from .evaluation import semantic_histograms, split_histograms
            C = out["pred_c"].shape[1]
            sc, ssc, occ = split_histograms(semantic_histograms(out["pred_c"], gt_occ, visible_mask, self.empty_idx), C)
            out.update(SC_metric=sc, SSC_metric=ssc, SSC_occ_metric=occ if visible_mask is not None else None)
            if out.get("pred_f") is not None:
                sc, ssc, occ = split_histograms(semantic_histograms(out["pred_f"], gt_occ, visible_mask, self.empty_idx), C)
                out.update(SC_metric=sc, SSC_metric_fine=ssc, SSC_occ_metric_fine=occ if visible_mask is not None else None)
        return out

    def evaluation_semantic(self, pred, gt, eval_type, visible_mask=None):
        """coocc_ray.py:659-684 on the device (no .cpu().numpy())."""
        from .evaluation import evaluation_semantic
        return evaluation_semantic(pred, gt, eval_type, visible_mask, self.empty_idx)

    def forward_test(self, img_metas=None, img_inputs=None, **kwargs):
        return self.simple_test(img_metas, img_inputs, **kwargs)

    def forward(self, return_loss=False, **kwargs):
        if return_loss:
            raise NotImplementedError("forward_train is not assembled: the differentiable pieces (trunk, head, fine branch, render block, "
                                      "render losses) are in co_occ_amd.autograd; the semantic losses are out of scope (SURVEY.md 8)")
        return self.forward_test(**kwargs)
