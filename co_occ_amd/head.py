"""``OccHead`` forward -- mirror of P/coocc/dense_heads/occ_head.py:17-265 (losses are out
of scope).  Same kwargs, ``forward`` signature/result dict and state_dict
keys (``occ_convs.{i}.{0,1}.*``, ``occ_pred_conv.{0,1,3}.*``, ``voxel_soft_weights.{0,1,3}.*``,
``img_mlp_0.*``, ``img_mlp.*``, ``fine_mlp.*``).
"""
import collections
import threading

import numpy as np
import torch
from torch import nn

from . import _lib
from . import core as core_mod
from ._lib import TIMER, call, host_f32, host_i32, ptr
from .backbone import build_bn
from .core import PackCache, PackedConv, Rows, conv_rows, linear_rows, to_rows
from .registry import HEADS

_I32, _F32, _I64 = torch.int32, torch.float32, torch.int64
_FINE_LOCK = threading.Lock()       # guards OccHead._fine_info (insert + evict vs lookup from other host threads)
# fine branch: the three Linear layers and two GroupNorms in one launch (coocc_fine_mlp); 0 = layer-by-layer path
FUSED_FINE_MLP = __import__("os").environ.get("COOCC_FUSED_FINE_MLP", "1") != "0"
# ... and the two Linear layers that precede a resampling applied before it, on the (much smaller) source grids
# (coocc_fine_mlp_pre); 0 = sample the 128-channel sources as the reference does
FINE_LINEAR_FIRST = __import__("os").environ.get("COOCC_FINE_LINEAR_FIRST", "1") != "0"
# the whole fine branch (two resamplings + MLP chain) in one launch (csrc/fine_fused.hip): 0 = three kernels, 1 = one launch for
# cascade ratio 4 (OpenOccupancy: 64 children per coarse voxel, 5 GB of intermediates saved, +6 % samples/s), 2 = also for ratio 2
# (measured at configs[1]: 383 us against 334 us for the three kernels -- the wave-serial phases at 2 waves per SIMD hide less gather
# latency than the samplers' own launches at 4+; kept off there)
FINE_FUSED = int(__import__("os").environ.get("COOCC_FINE_FUSED", "1"))
# ratio 2 on the split-f16 engine (csrc/fine2_h2.hip, round 5): lanes = points, voxel samples kept in registers, both Linear layers
# as three f16 MFMAs per k16 step; 0 = the three kernels (fp32-MFMA chain).
FINE2_H2 = __import__("os").environ.get("COOCC_FINE2_H2", "1") != "0"
# ... with the image samples made INSIDE that launch (0.33 -> 0.17 ms; 0 = the grouped sampler's own launch feeds the kernel).
# Rounds 5-6 kept it off: bit-exact alone, a few hundred of 640 k rows different next to other streams' split-f16 GEMMs.  Cause
# (round 6, profiles/r6_pk_opsel_probe.txt): 132 packed-fp32 instructions with op_sel[src1] = 1 in the sampling code, the form gfx950
# mis-reads in lanes 48-63 beside a 128-bit-operand MFMA; the kernel is COOCC_SCALAR_FP32 now and bit-stable under those co-runners
# (tests/test_gpu_corunner.py, tools/debug/fine2_corunner.py).
FINE2_IMG_INSIDE = __import__("os").environ.get("COOCC_FINE2_IMG_INSIDE", "1") == "1"
# occ_pred_conv[0] and the voxel half of fine_mlp[0] (both 128 -> 64 on out_voxel_feats) as one 128 -> 128 GEMM (ReLU on the first
# 64 columns): one read of the rows and one launch instead of two; needs the strided-Q consumer (fine2_h2)
MERGED_PRED_Q = __import__("os").environ.get("COOCC_MERGED_PRED_Q", "1") != "0"
# pred_f of simple_test written output-major in one pass when the fine points are the head's own (coocc_scatter_fine_grouped);
# 0 = fill + scatter by coordinates (coocc_scatter_fine[_dev])
SCATTER_GROUPED = __import__("os").environ.get("COOCC_SCATTER_GROUPED", "1") != "0"


def _conv3d(conv_cfg, cin, cout, k, pad):
    cfg = dict(conv_cfg or dict(type='Conv3d'))
    t = cfg.pop("type", "Conv3d")
    if t != "Conv3d":
        raise NotImplementedError("OccHead: conv type %r" % t)
    return nn.Conv3d(cin, cout, kernel_size=k, stride=1, padding=pad, **cfg)


@HEADS.register_module()
class OccHead(nn.Module):
    def __init__(self, in_channels, out_channel, num_level=1, num_img_level=1, soft_weights=False,
                 loss_weight_cfg=None, conv_cfg=dict(type='Conv3d', bias=False),
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), fine_topk=20000,
                 point_cloud_range=[-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], final_occ_size=[256, 256, 20],
                 empty_idx=0, visible_loss=False, balance_cls_weight=True, cascade_ratio=1,
                 sample_from_voxel=False, sample_from_img=False, train_cfg=None, test_cfg=None,
                 padding_mode='border', data_type='nus'):
        super().__init__()
        if type(in_channels) is not list:
            in_channels = [in_channels]
        if data_type != 'nus':
            raise NotImplementedError("OccHead: only the nuScenes projection branch is implemented")
        self.in_channels, self.out_channel, self.num_level = in_channels, out_channel, num_level
        self.fine_topk = fine_topk
        self.point_cloud_range = torch.tensor(np.array(point_cloud_range)).float()
        self.final_occ_size = final_occ_size
        self.cascade_ratio = cascade_ratio
        self.sample_from_voxel, self.sample_from_img = sample_from_voxel, sample_from_img
        self.padding_mode, self.data_type = padding_mode, data_type
        self.empty_idx, self.soft_weights = empty_idx, soft_weights
        self.num_img_level, self.num_point_sampling_feat = num_img_level, num_level
        lw = loss_weight_cfg or {}                              # occ_head.py:84-99
        self.loss_weight_cfg = loss_weight_cfg
        self.loss_voxel_ce_weight = lw.get('loss_voxel_ce_weight', 1.0)
        self.loss_voxel_sem_scal_weight = lw.get('loss_voxel_sem_scal_weight', 1.0)
        self.loss_voxel_geo_scal_weight = lw.get('loss_voxel_geo_scal_weight', 1.0)
        self.loss_voxel_lovasz_weight = lw.get('loss_voxel_lovasz_weight', 1.0)
        from .losses import nusc_class_weights
        self.class_weights = nusc_class_weights() if balance_cls_weight else torch.ones(17) / 17     # :134-144
        self.visible_loss = visible_loss

        if cascade_ratio != 1 and (sample_from_voxel or sample_from_img):
            fine_in = 128 if sample_from_voxel else 0
            if sample_from_img:
                self.img_mlp_0 = nn.Sequential(nn.Conv2d(512, 128, 1, 1, 0), nn.GroupNorm(16, 128), nn.ReLU(inplace=True))
                self.img_mlp = nn.Sequential(nn.Linear(128, 64), nn.GroupNorm(16, 64), nn.ReLU(inplace=True))
                fine_in += 64
            self.fine_mlp = nn.Sequential(nn.Linear(fine_in, 64), nn.GroupNorm(16, 64), nn.ReLU(inplace=True),
                                          nn.Linear(64, out_channel))
        self.occ_convs = nn.ModuleList()
        for i in range(num_level):
            mid = in_channels[i] // 2
            self.occ_convs.append(nn.Sequential(_conv3d(conv_cfg, in_channels[i], mid, 3, 1), build_bn(norm_cfg, mid),
                                                nn.ReLU(inplace=True)))
        self.occ_pred_conv = nn.Sequential(_conv3d(conv_cfg, mid, mid // 2, 1, 0), build_bn(norm_cfg, mid // 2),
                                           nn.ReLU(inplace=True), _conv3d(conv_cfg, mid // 2, out_channel, 1, 0))
        if soft_weights:
            self.voxel_soft_weights = nn.Sequential(_conv3d(conv_cfg, mid, mid // 2, 1, 0), build_bn(norm_cfg, mid // 2),
                                                    nn.ReLU(inplace=True), _conv3d(conv_cfg, mid // 2, num_level, 1, 0))
        self._packs = PackCache(self)

    # ---------------------------------------------------------------- packing
    def _packed(self):
        def seq2(m):
            return (PackedConv(m[0].weight, bn=m[1], bias=m[0].bias, ksize=1),
                    PackedConv(m[3].weight, bias=m[3].bias, ksize=1))

        def build():
            d = dict(occ=[PackedConv(m[0].weight, bn=m[1], bias=m[0].bias, ksize=3, pad=1) for m in self.occ_convs],
                     pred=seq2(self.occ_pred_conv))
            if self.soft_weights:
                d["soft"] = seq2(self.voxel_soft_weights)
            if hasattr(self, "fine_mlp"):
                d["fine0"] = PackedConv(self.fine_mlp[0].weight, bias=self.fine_mlp[0].bias)
                d["fine3"] = PackedConv(self.fine_mlp[3].weight, bias=self.fine_mlp[3].bias)
            if hasattr(self, "img_mlp"):
                d["img0"] = PackedConv(self.img_mlp_0[0].weight.flatten(1), bias=self.img_mlp_0[0].bias)
                d["img"] = PackedConv(self.img_mlp[0].weight, bias=self.img_mlp[0].bias)
                if self.sample_from_voxel and self.fine_mlp[0].weight.shape[1] == 192:
                    d["img_nb"] = PackedConv(self.img_mlp[0].weight)                                  # bias added after sampling
                    d["f0_vox_nb"] = PackedConv(self.fine_mlp[0].weight[:, :128].contiguous())
                    m = self.occ_pred_conv
                    if m[0].weight.shape[0] == 64 and m[0].weight.shape[1] == 128:
                        # occ_pred_conv[0] (128 -> 64, BN, ReLU) and the voxel half of fine_mlp[0] (128 -> 64, no bias: applied before
                        # the trilinear resampling) read the same rows: ONE 128 -> 128 GEMM, ReLU on columns [0, 64) only
                        from .core import fold_bn
                        pq = PackedConv(torch.cat([m[0].weight.flatten(1), self.fine_mlp[0].weight[:, :128]], 0).contiguous())
                        sc, bi = fold_bn(m[1], m[0].bias)
                        dev = m[0].weight.device
                        pq.scale = torch.cat([sc.float().cpu(), torch.ones(64)]).to(dev).contiguous()
                        pq.bias = torch.cat([bi.float().cpu(), torch.zeros(64)]).to(dev).contiguous()
                        pq.ksize, pq.pad = 1, 0
                        d["pred_q"] = pq
            if hasattr(self, "img_mlp") and hasattr(self, "fine_mlp"):     # the fused fine-branch kernel loads 16-byte vectors
                d["mlp_aligned"] = all(q.data_ptr() % 16 == 0 for m in (self.img_mlp, self.fine_mlp) for q in m.parameters())
                if ("img_nb" in d and self.out_channel <= 32 and self.img_mlp[1].num_groups == 16 and self.fine_mlp[1].num_groups == 16
                        and self.fine_mlp[0].weight.is_cuda):
                    # operands of the ratio-2 one-launch kernel on the split-f16 engine (csrc/fine2_h2.hip): both Linear layers of
                    # the per-point chain as f16 hi / lo fragments in the kernel's k-slot order + the six GroupNorm / bias vectors
                    dev = self.fine_mlp[0].weight.device
                    wp = torch.empty(24576, device=dev, dtype=torch.uint8)
                    cn = torch.empty(416, device=dev, dtype=_F32)
                    li, gi, l0, g0, l3 = self.img_mlp[0], self.img_mlp[1], self.fine_mlp[0], self.fine_mlp[1], self.fine_mlp[3]
                    q = lambda t: ptr(t.detach().float().contiguous())
                    call("coocc_fine2_pack", q(l0.weight), q(l3.weight), self.out_channel, q(li.bias), q(gi.weight), q(gi.bias),
                         q(l0.bias), q(g0.weight), q(g0.bias), q(l3.bias), ptr(wp), ptr(cn))
                    d["fine2"] = (wp, cn)
            return d
        return self._packs.get_modules((self,), build)

    # ---------------------------------------------------------------- C3
    def forward_coarse_rows(self, voxel_feats):
        """occ_head.py:149-171 on Rows: returns (out_voxel_feats Rows, occ logits Rows)."""
        from . import core
        p = self._packed()
        # level 0's output is read by the soft-weight branch's first 1x1x1 layer (a split-f16 GEMM at configs[1]'s size: H2 twin)
        # levels 1.. (10 000 / 1 250 / 169 rows: a fraction of the chip each) on a side stream beside level 0 and the soft-weight branch
        feats = [to_rows(f) for f in voxel_feats]
        lbr = core.Fork(2, feats[0].t.device, enable=len(feats) > 1 and not self.training)
        with lbr:
            small = [conv_rows(f, p["occ"][i], relu=True) for i, f in enumerate(feats) if i > 0]
        o0 = conv_rows(feats[0], p["occ"][0], relu=True, twin_for=((p["soft"][0],) if self.soft_weights else ()))
        occs = [o0] + small
        wlogit = None
        if self.soft_weights:
            h = conv_rows(o0, p["soft"][0], relu=True)
            wlogit = conv_rows(h, p["soft"][1], relu=False).t
        lbr.join(*[o.t for o in small])
        L = len(occs)
        levels = (_lib.c_void_p * L)(*[o.t.data_ptr() for o in occs])
        dims = host_i32([v for o in occs for v in (o.X, o.Y, o.Z)])
        out = Rows(torch.empty_like(o0.t), o0.B, o0.X, o0.Y, o0.Z, o0.C)
        tw = None
        merged = (MERGED_PRED_Q and "pred_q" in p and core.CONV_DTYPE == "f32" and self.cascade_ratio == 2 and self._fine2_h2_ok()
                  and not self.training)
        first = p["pred_q"] if merged else p["pred"][0]
        if core.CONV_ENGINE == "h2" and core.CONV_DTYPE == "f32" and o0.C % 32 == 0 and core.takes_h2(out, (first,)):
            tw = out.h2 = torch.empty_like(o0.t)
        call("coocc_occhead_mix_ex", levels, dims, L, ptr(wlogit), ptr(out.t), o0.B, o0.C, ptr(tw))
        if merged:
            hq = conv_rows(out, first, relu=64)               # [V, 128] = relu(bn(pred0 x)) | Q = W_f0[:, :128] x
            h = Rows(hq.t, hq.B, hq.X, hq.Y, hq.Z, 64, 0)
            out.aux = dict(q=hq.t[:, 64:])                    # rows 128 floats apart: csrc/fine2_h2.hip takes the stride
        else:
            h = conv_rows(out, p["pred"][0], relu=True)
        occ = conv_rows(h, p["pred"][1], relu=False)
        return out, occ

    def level_readers(self):
        """Per input level, the layers that read it (the neck's fpn_convs write H2 twins for the split-f16 ones)."""
        return [(pc,) for pc in self._packed()["occ"]]

    def forward_coarse_voxel(self, voxel_feats):
        out, occ = self.forward_coarse_rows(voxel_feats)
        return {'out_voxel_feats': [out.as_ncdhw()], 'occ': [occ.as_ncdhw()]}

    # ---------------------------------------------------------------- C4
    def _image_rows(self, p, img_feats, transform, grid_rows, dev):
        """The part of the fine branch that depends on the image features only (occ_head.py:64, :205-212): NCHW -> rows, the 1x1
        Conv2d + GroupNorm of ``img_mlp_0``, the projection matrices.  Returns (g [N*Hf*Wf, 128], params, (N, Hf, Wf))."""
        f = img_feats[0]                                    # [B,N,512,fH,fW]
        _, N_i, C_i, Hf, Wf = f.shape
        rows = torch.empty(N_i * Hf * Wf, C_i, device=dev, dtype=_F32)
        call("coocc_ncdhw_to_ndhwc", ptr(f[0].float().contiguous()), ptr(rows), N_i, C_i, Hf * Wf, C_i, 0)
        g = linear_rows(rows, p["img0"])                    # Conv2d 1x1 (occ_head.py:64)
        gn = self.img_mlp_0[1]
        call("coocc_groupnorm_nhwc", ptr(g), N_i, Hf * Wf, g.shape[1], gn.num_groups, ptr(gn.weight.detach()),
             ptr(gn.bias.detach()), float(gn.eps), 1)
        params = self._projection_params(transform, grid_rows, dev)
        return g, params, (N_i, Hf, Wf)

    def fork_image_branch(self, img_feats, transform, grid_rows):
        """``_image_rows`` forked onto a side stream (``core.Fork``) BEFORE the encoder / neck / coarse head are issued: 4 small
        launches (33-96 workgroups, ~80 us) that otherwise sit in the middle of the sample's launch chain.  ``grid_rows``: any
        Rows on the head's level-0 grid (only its dimensions are read).  Returns a handle for ``forward(image_branch=)``, or None."""
        if not (self.sample_from_img and img_feats is not None and self.cascade_ratio != 1 and not self.training and core_mod.BRANCHES):
            return None
        dev = img_feats[0].device
        br = core_mod.Fork(1, dev)
        if not br.on:
            return None
        with br:
            out = self._image_rows(self._packed(), img_feats, transform, grid_rows, dev)
        return br, out, (grid_rows.X, grid_rows.Y, grid_rows.Z)

    def _fine(self, ovf, occ, img_feats, transform, static=False, image_branch=None):
        """occ_head.py:180-237, eval branch, B == 1.  ``static``: nothing is read back -- the foreground count stays on
        the device (third return value), every buffer has its worst-case size (all V coarse voxels foreground) and the
        launches take the count from the device (``*_dev`` entry points), so the branch can be captured into a hipGraph;
        rows past count * ratio^3 of the returned logits / coordinates are undefined."""
        p = self._packed()
        dev = ovf.t.device
        V, r = ovf.V, self.cascade_ratio
        assert ovf.B == 1, "OccHead fine branch: B == 1 (as the render block asserts, coocc_ray.py:571)"
        flags = torch.empty(V, device=dev, dtype=torch.uint8)
        call("coocc_argmax_flags", ptr(occ.t), V, occ.C, occ.stride, self.empty_idx, ptr(flags))
        lin = torch.empty(V, device=dev, dtype=_I32)
        cnt = torch.empty(1, device=dev, dtype=_I32)
        ws = torch.empty(V // 1024 + 2, device=dev, dtype=_I32)
        # fgmap: coarse voxel -> its ordinal in lin (or -1), from the same pass: scatter_fine looks the children's parents up in it
        fgmap = torch.empty(V, device=dev, dtype=_I32) if SCATTER_GROUPED else None
        call("coocc_compact_flags_ex", ptr(flags), V, ptr(lin), ptr(cnt), ptr(fgmap), ptr(ws), ws.numel() * 4)
        # everything that does not depend on the foreground count is enqueued BEFORE the host reads it: the
        # image-feature branch (1x1 conv + GroupNorm) and the camera matrices (a dozen tiny torch launches)
        # then run under the device->host round trip instead of after it
        use_img = self.sample_from_img and img_feats is not None
        if use_img and image_branch is not None:
            br, (g, params, (N_i, Hf, Wf)), grid = image_branch  # issued on a side stream before the encoder (fork_image_branch)
            br.join(g, params)
            if grid != (ovf.X, ovf.Y, ovf.Z):                    # the projection matrices carry the level-0 grid: made for another one
                params = self._projection_params(transform, ovf, dev)
        elif use_img:
            g, params, (N_i, Hf, Wf) = self._image_rows(p, img_feats, transform, ovf, dev)
        if static:
            return self._fine_static(p, ovf, lin, cnt, g if use_img else None, params if use_img else None, (N_i, Hf, Wf) if use_img else None,
                                     fgmap=fgmap)
        n = int(_lib.host_read(cnt)[0])
        core_mod.check_h2_overflow()          # the stage's one host read: every conv up to the coarse head has finished
        assert n > 0, 'no foreground in coarse voxel'
        nf = n * r ** 3
        fine_xyz = torch.empty(3, nf, device=dev, dtype=_I64)
        # what scatter_fine needs to write pred_f output-major (coocc_scatter_fine_grouped): the coarse list these points descend from
        self._note_fine(dict(xyz=fine_xyz, lin=lin, n=n, cnt=None, coarse=(ovf.X, ovf.Y, ovf.Z), map=fgmap))
        cvox = 128 if self.sample_from_voxel else 0
        # one launch for Linear+GN+ReLU -> cat -> Linear+GN+ReLU -> Linear when both samples feed the MLPs
        fused = (FUSED_FINE_MLP and use_img and self.sample_from_voxel and ovf.C == 128 and g.shape[1] == 128
                 and self.out_channel <= 32 and self.img_mlp[1].num_groups == 16 and self.fine_mlp[1].num_groups == 16
                 and p.get("mlp_aligned", False))
        if fused and FINE_LINEAR_FIRST and "img_nb" in p:
            # Linear(128->64) of img_mlp on the 6 x Hf x Wf feature map and the voxel half of fine_mlp[0] on the V coarse
            # voxels instead of on the 8 V fine points: a Linear commutes with the interpolation that follows it
            P = linear_rows(g, p["img_nb"])
            fused_one = self._fused_fine_ok(ovf, N_i)
            Q = ovf.aux.get("q") if (fused_one and ovf.aux and self._fine2_h2_ok() and FINE_FUSED < 2) else None
            if Q is None:
                Q = linear_rows(ovf.t, p["f0_vox_nb"], in_coff=ovf.coff, in_C=128)
            if fused_one:
                logits = torch.empty(nf, self.out_channel, device=dev, dtype=_F32)
                self._fine_fused(p, ovf, Q, P, params, (N_i, Hf, Wf), lin, n, None, fine_xyz, logits)
                return logits, fine_xyz
            vq = torch.empty(nf, 64, device=dev, dtype=_F32)
            call("coocc_fine_sample_voxel", ptr(Q), 64, ovf.X, ovf.Y, ovf.Z, ptr(lin), n, r,
                 host_i32(self.final_occ_size), ptr(fine_xyz), ptr(vq), 64)
            samp = torch.empty(nf, 64, device=dev, dtype=_F32)
            call("coocc_fine_sample_img", ptr(P), N_i, 64, Hf, Wf, ptr(params), ptr(fine_xyz), nf, ptr(samp), 64, r if r in (2, 4) else 0)
            li, gi, l0, g0, l3 = self.img_mlp[0], self.img_mlp[1], self.fine_mlp[0], self.fine_mlp[1], self.fine_mlp[3]
            logits = torch.empty(nf, self.out_channel, device=dev, dtype=_F32)
            d = lambda t: ptr(t.detach())
            with TIMER.region("k_fine_mlp<pre>", 2.0 * nf * 64 * (64 + self.out_channel)):
                call("coocc_fine_mlp_pre", ptr(samp), 64, ptr(vq), 64, nf, d(li.bias), d(gi.weight), d(gi.bias), float(gi.eps),
                     d(l0.weight), d(l0.bias), d(g0.weight), d(g0.bias), float(g0.eps), d(l3.weight), d(l3.bias),
                     self.out_channel, ptr(logits))
            return logits, fine_xyz
        cat = torch.empty(nf, cvox if fused else cvox + (64 if use_img else 0), device=dev, dtype=_F32)
        vox_feat = cat if self.sample_from_voxel else torch.empty(nf, ovf.C, device=dev, dtype=_F32)
        # fine coordinates are always produced by this kernel (they are an output of the head)
        call("coocc_fine_sample_voxel", ptr(ovf.t), ovf.C, ovf.X, ovf.Y, ovf.Z, ptr(lin), n, r,
             host_i32(self.final_occ_size), ptr(fine_xyz), ptr(vox_feat), vox_feat.shape[1])
        if use_img:
            samp = torch.empty(nf, g.shape[1], device=dev, dtype=_F32)
            call("coocc_fine_sample_img", ptr(g), N_i, g.shape[1], Hf, Wf, ptr(params), ptr(fine_xyz), nf, ptr(samp),
                 samp.shape[1], r if r in (2, 4) else 0)
            if fused:
                li, gi, l0, g0, l3 = self.img_mlp[0], self.img_mlp[1], self.fine_mlp[0], self.fine_mlp[1], self.fine_mlp[3]
                logits = torch.empty(nf, self.out_channel, device=dev, dtype=_F32)
                d = lambda t: ptr(t.detach())
                with TIMER.region("k_fine_mlp", 2.0 * nf * 64 * (128 + 192 + self.out_channel)):
                    call("coocc_fine_mlp", ptr(samp), samp.shape[1], ptr(cat), cat.shape[1], nf,
                         d(li.weight), d(li.bias), d(gi.weight), d(gi.bias), float(gi.eps),
                         d(l0.weight), d(l0.bias), d(g0.weight), d(g0.bias), float(g0.eps),
                         d(l3.weight), d(l3.bias), self.out_channel, ptr(logits))
                return logits, fine_xyz
            linear_rows(samp, p["img"], out=cat, out_coff=cvox)
            gn = self.img_mlp[1]
            sub = cat[:, cvox:]
            call("coocc_groupnorm_rows", _lib.c_void_p(sub.data_ptr()), nf, 64, cat.shape[1], gn.num_groups,
                 ptr(gn.weight.detach()), ptr(gn.bias.detach()), float(gn.eps), 1)
        h = linear_rows(cat, p["fine0"])
        gn = self.fine_mlp[1]
        call("coocc_groupnorm_rows", ptr(h), nf, h.shape[1], h.shape[1], gn.num_groups, ptr(gn.weight.detach()),
             ptr(gn.bias.detach()), float(gn.eps), 1)
        return linear_rows(h, p["fine3"]), fine_xyz

    def _fine_static(self, p, ovf, lin, cnt, g, params, img_dims, fgmap=None):
        """The fused (Linear-first) fine branch with the foreground count on the device; see ``_fine(static=True)``."""
        dev = ovf.t.device
        V, r = ovf.V, self.cascade_ratio
        ok = (FUSED_FINE_MLP and FINE_LINEAR_FIRST and g is not None and self.sample_from_voxel and ovf.C == 128 and g.shape[1] == 128
              and self.out_channel <= 32 and self.img_mlp[1].num_groups == 16 and self.fine_mlp[1].num_groups == 16
              and p.get("mlp_aligned", False) and "img_nb" in p and r in (2, 4))
        if not ok:
            raise _lib.CooccArgError("OccHead static fine branch: only the fused Linear-first configuration (cascade 2 / 4, image + "
                                  "voxel samples, 128-channel features) has device-count kernels")
        N_i, Hf, Wf = img_dims
        nf = V * r ** 3
        fine_xyz = torch.empty(3 * nf, device=dev, dtype=_I64)
        self._note_fine(dict(xyz=fine_xyz, lin=lin, n=V, cnt=cnt, coarse=(ovf.X, ovf.Y, ovf.Z), map=fgmap))
        P = linear_rows(g, p["img_nb"])
        fused = self._fused_fine_ok(ovf, N_i)
        Q = ovf.aux.get("q") if (fused and ovf.aux and self._fine2_h2_ok() and FINE_FUSED < 2) else None   # made with occ_pred_conv[0]
        if Q is None:
            Q = linear_rows(ovf.t, p["f0_vox_nb"], in_coff=ovf.coff, in_C=128)
        if fused:
            logits = torch.empty(nf, self.out_channel, device=dev, dtype=_F32)
            self._fine_fused(p, ovf, Q, P, params, (N_i, Hf, Wf), lin, V, cnt, fine_xyz, logits)
            return logits, fine_xyz, cnt
        vq = torch.empty(nf, 64, device=dev, dtype=_F32)
        call("coocc_fine_sample_voxel_dev", ptr(Q), 64, ovf.X, ovf.Y, ovf.Z, ptr(lin), V, ptr(cnt), r,
             host_i32(self.final_occ_size), ptr(fine_xyz), ptr(vq), 64)
        samp = torch.empty(nf, 64, device=dev, dtype=_F32)
        call("coocc_fine_sample_img_dev", ptr(P), N_i, 64, Hf, Wf, ptr(params), ptr(fine_xyz), nf, ptr(cnt), ptr(samp), 64, r)
        li, gi, l0, g0, l3 = self.img_mlp[0], self.img_mlp[1], self.fine_mlp[0], self.fine_mlp[1], self.fine_mlp[3]
        logits = torch.empty(nf, self.out_channel, device=dev, dtype=_F32)
        d = lambda t: ptr(t.detach())
        call("coocc_fine_mlp_pre_dev", ptr(samp), 64, ptr(vq), 64, nf, ptr(cnt), r ** 3, d(li.bias), d(gi.weight), d(gi.bias),
             float(gi.eps), d(l0.weight), d(l0.bias), d(g0.weight), d(g0.bias), float(g0.eps), d(l3.weight), d(l3.bias),
             self.out_channel, ptr(logits))
        return logits, fine_xyz, cnt

    def _fine2_h2_ok(self):
        """Ratio 2 on the split-f16 engine: the points-as-lanes one-launch kernel (csrc/fine2_h2.hip)."""
        return (FINE2_H2 and self.cascade_ratio == 2 and core_mod.CONV_ENGINE == "h2" and core_mod.CONV_DTYPE == "f32"
                and "fine2" in self._packed())

    def _fused_fine_ok(self, ovf, ncam):
        """One-launch fine branch: ratio 2 (csrc/fine2_h2.hip on the split-f16 engine; csrc/fine_fused.hip with COOCC_FINE_FUSED=2) |
        4 (csrc/fine_fused.hip), final grid = ratio x coarse grid, <= 8 cameras."""
        r = self.cascade_ratio
        return (((FINE_FUSED >= 1 and r == 4) or (r == 2 and (FINE_FUSED >= 2 or self._fine2_h2_ok()))) and ncam <= 8
                and self.out_channel <= 32 and tuple(int(v) for v in self.final_occ_size) == (r * ovf.X, r * ovf.Y, r * ovf.Z))

    def _fine_fused(self, p, ovf, Q, P, params, img_dims, lin, n_cap, cnt, fine_xyz, logits):
        N_i, Hf, Wf = img_dims
        gi, l0, g0, l3, li = self.img_mlp[1], self.fine_mlp[0], self.fine_mlp[1], self.fine_mlp[3], self.img_mlp[0]
        d = lambda t: ptr(t.detach())
        nf = n_cap * self.cascade_ratio ** 3
        if self._fine2_h2_ok() and FINE_FUSED < 2:
            wp, cn = p["fine2"]
            samp = None
            if not FINE2_IMG_INSIDE:
                samp = torch.empty(nf, 64, device=logits.device, dtype=_F32)
                call("coocc_fine_sample_img_lin", ptr(P), N_i, 64, Hf, Wf, ptr(params), ptr(lin), ovf.Y, ovf.Z, int(n_cap),
                     ptr(cnt, _I32) if cnt is not None else None, ptr(samp), 64, 2)
            with TIMER.region("k_fine2_h2", 2.0 * nf * 64 * (64 + self.out_channel)):
                call("coocc_fine2_h2", ptr(Q, strided=True), Q.stride(0), ovf.X, ovf.Y, ovf.Z, ptr(P), N_i, Hf, Wf, ptr(params), ptr(lin), int(n_cap),
                     ptr(cnt, _I32) if cnt is not None else None, host_i32(self.final_occ_size), ptr(wp), ptr(cn), float(gi.eps),
                     float(g0.eps), self.out_channel, ptr(fine_xyz), ptr(logits), ptr(samp))
            return
        with TIMER.region("k_fine_fused", 2.0 * nf * 64 * (64 + self.out_channel)):
            call("coocc_fine_fused", ptr(Q), ovf.X, ovf.Y, ovf.Z, ptr(P), N_i, Hf, Wf, ptr(params), ptr(lin), int(n_cap),
                 ptr(cnt, _I32) if cnt is not None else None, self.cascade_ratio, host_i32(self.final_occ_size), d(li.bias), d(gi.weight),
                 d(gi.bias), float(gi.eps), d(l0.weight), d(l0.bias), d(g0.weight), d(g0.bias), float(g0.eps), d(l3.weight), d(l3.bias),
                 self.out_channel, ptr(fine_xyz), ptr(logits))

    def _projection_params(self, transform, ovf, dev):
        """Per-sample matrices of project_points_on_img (coordinate_transform.py:25-65), b = 0, packed by one
        device kernel (no torch.inverse: it synchronises with the host)."""
        rots, trans, intrins, post_rots, post_trans, bda = [t[0].float().contiguous() for t in transform[:6]]
        r = self.cascade_ratio
        W_occ, H_occ, D_occ = ovf.X * r, ovf.Y * r, ovf.Z * r
        pr = self.point_cloud_range                      # CPU tensor: fp32 arithmetic as the reference's
        voxel_size = (pr[3:] - pr[:3]) / torch.tensor([W_occ - 1, H_occ - 1, D_occ - 1], dtype=_F32)
        W_img = float(transform[-1][1][0]) if torch.is_tensor(transform[-1][1]) else float(transform[-1][1])
        H_img = float(transform[-1][0][0]) if torch.is_tensor(transform[-1][0]) else float(transform[-1][0])
        ncam = rots.shape[0]
        params = torch.empty(17 + 27 * ncam, device=dev, dtype=_F32)
        call("coocc_projection_params", ptr(rots), ptr(trans), ptr(intrins), ptr(post_rots), ptr(post_trans), ptr(bda), ncam,
             host_f32(voxel_size.tolist() + pr[:3].tolist() + [W_img - 1, H_img - 1]), ptr(params))
        return params

    # ---------------------------------------------------------------- forward
    def draw_fine_voxels(self, logit_rows, generator=None):
        """Coarse voxels whose ``ratio^3`` children the fine branch evaluates: foreground = argmax != empty; in training
        at most ``fine_topk`` of them, drawn at random (coordinate_transform.py:17-21 permutes the COARSE columns and keeps
        the first topk).  logit_rows [V,ncls] -> int32 rows."""
        fg = torch.nonzero(logit_rows.argmax(1) != self.empty_idx).flatten()
        # torch.nonzero read the count back: every kernel up to the coarse head (and the previous step's backward pass) has
        # finished -- the training path's point to look at the split-f16 engine's range guard
        core_mod.check_h2_overflow()
        if self.training and fg.numel() >= int(self.fine_topk):
            dev = fg.device if generator is None else generator.device
            sel = torch.randperm(fg.numel(), generator=generator, device=dev)[:int(self.fine_topk)]
            fg = fg[sel.to(fg.device)]
        return fg.int()

    def _forward_train(self, voxel_feats, img_feats, transform, generator=None):
        """Training branch of occ_head.py:173-245: batch-statistics BN, autograd through every HIP kernel."""
        from . import autograd as ag
        levels = [ag.rows_from_ncdhw(f) for f in voxel_feats]
        geom = levels[0][1]
        assert geom[0] == 1, "OccHead training path: batch size 1 per GPU (samples_per_gpu=1 upstream)"
        out_rows, logit_rows = ag.occhead_coarse_train(self, levels)
        res = {'output_voxels': [ag.ncdhw_from_rows(logit_rows, geom)], 'output_voxels_fine': None, 'output_coords_fine': None,
               'output_points': None}
        self.last_out_voxel_rows = out_rows
        if self.cascade_ratio != 1 and (self.sample_from_img or self.sample_from_voxel):
            lin = self.draw_fine_voxels(logit_rows.detach(), generator)
            assert lin.numel() > 0, 'no foreground in coarse voxel'
            fine, xyz = ag.fine_branch_train(self, out_rows, geom, lin.contiguous(), img_feats, transform)
            res['output_voxels_fine'], res['output_coords_fine'] = [fine], [xyz]
        return res

    def forward(self, voxel_feats, img_feats=None, img_metas=None, pts_feats=None, target_points=None,
                transform=None, points=None, **kwargs):
        assert type(voxel_feats) is list and len(voxel_feats) == self.num_level
        if target_points:
            raise NotImplementedError("forward_lidarseg is not on the hot path")
        if self.training:
            return self._forward_train(voxel_feats, img_feats, transform, kwargs.get("generator"))
        ovf, occ = self.forward_coarse_rows(voxel_feats)
        res = {'output_voxels': [occ.as_ncdhw()], 'output_voxels_fine': None, 'output_coords_fine': None,
               'output_points': None}
        if self.cascade_ratio != 1 and (self.sample_from_img or self.sample_from_voxel):
            if kwargs.get("static"):
                # capacity-sized outputs + the number of foreground coarse voxels on the device (hipGraph capture)
                fine, xyz, cnt = self._fine(ovf, occ, img_feats, transform, static=True, image_branch=kwargs.get("image_branch"))
                res['fine_count'] = cnt
            else:
                fine, xyz = self._fine(ovf, occ, img_feats, transform, image_branch=kwargs.get("image_branch"))
            res['output_voxels_fine'], res['output_coords_fine'] = [fine], [xyz]
        self.last_out_voxel_feats = ovf
        return res

    def _note_fine(self, rec):
        """Remember what ``scatter_fine`` needs for the fine points a ``_fine`` call produced, keyed by the address of their
        coordinate tensor (kept alive by the record, so the address cannot be reused while the record exists); the last few only."""
        with _FINE_LOCK:                # several host threads (serving helper / issuer, the eager pipeline) share one head
            d = self.__dict__.setdefault("_fine_info", collections.OrderedDict())
            d[rec["xyz"].data_ptr()] = rec
            while len(d) > 4:
                d.popitem(last=False)
            self._last_fine = rec

    def scatter_fine(self, fine_pred, fine_coord, out_size, count_dev=None):
        """``pred_f`` of simple_test (coocc_ray.py:546-550): [1,ncls,Xf,Yf,Zf].  ``count_dev``: the capacity-sized outputs of
        the static fine branch + the device-side count of foreground coarse voxels."""
        ncls = fine_pred.shape[1]
        grid = torch.empty(1, ncls, *out_size, device=fine_pred.device, dtype=_F32)
        # the record of THESE coordinates (several samples may be in flight through one head: host threads of the eager pipeline)
        with _FINE_LOCK:
            lf = getattr(self, "_fine_info", {}).get(fine_coord.data_ptr())
        r = self.cascade_ratio
        if (SCATTER_GROUPED and lf is not None and lf["xyz"].data_ptr() == fine_coord.data_ptr() and (lf["cnt"] is None) == (count_dev is None)
                and tuple(out_size) == tuple(v * r for v in lf["coarse"]) and fine_pred.shape[0] == lf["n"] * r ** 3):
            # the head's own points (the ratio^3 children of its foreground list): one output-major pass, no fill + scatter
            Xc, Yc, Zc = lf["coarse"]
            fgmap = lf.get("map")            # written by the compaction pass; else built here from the list
            ws = fgmap if fgmap is not None else torch.empty(Xc * Yc * Zc, device=fine_pred.device, dtype=_I32)
            call("coocc_scatter_fine_grouped", ptr(fine_pred), ncls, fine_pred.shape[1], ptr(lf["lin"]) if fgmap is None else None, lf["n"],
                 ptr(lf["cnt"]) if lf["cnt"] is not None else None, r, Xc, Yc, Zc, ptr(grid), float(self.empty_idx), ptr(ws))
            return grid
        if count_dev is not None:
            call("coocc_scatter_fine_dev", ptr(fine_pred), fine_pred.shape[0], ptr(count_dev), self.cascade_ratio ** 3, ncls,
                 fine_pred.shape[1], ptr(fine_coord), ptr(grid), out_size[0], out_size[1], out_size[2], float(self.empty_idx))
            return grid
        call("coocc_scatter_fine", ptr(fine_pred), fine_pred.shape[0], ncls, fine_pred.shape[1], ptr(fine_coord), ptr(grid),
             out_size[0], out_size[1], out_size[2], float(self.empty_idx))
        return grid

    # ---------------------------------------------------------------- losses (occ_head.py:265-337)
    def _loss_terms(self, logits, target, tag, class_weights):
        from . import losses as L
        return {
            'loss_voxel_ce_%s' % tag: self.loss_voxel_ce_weight * L.ce_ssc_loss(logits, target, class_weights, ignore_index=255),
            'loss_voxel_sem_scal_%s' % tag: self.loss_voxel_sem_scal_weight * L.sem_scal_loss(logits, target, ignore_index=255),
            'loss_voxel_geo_scal_%s' % tag: self.loss_voxel_geo_scal_weight * L.geo_scal_loss(logits, target, ignore_index=255,
                                                                                             non_empty_idx=self.empty_idx),
            'loss_voxel_lovasz_%s' % tag: self.loss_voxel_lovasz_weight * L.lovasz_softmax(torch.softmax(logits, dim=1), target,
                                                                                           ignore=255)}

    def loss_voxel(self, output_voxels, target_voxels, tag):
        """occ_head.py:265-293: labels majority-pooled to the logits' grid, then the four terms (class-weighted CE)."""
        from .losses import pool_labels
        B, C, H, W, D = output_voxels.shape
        target = pool_labels(target_voxels, H, W, D, self.empty_idx, num_cls=self.out_channel)
        return self._loss_terms(output_voxels, target, tag, self.class_weights.to(output_voxels))

    def loss_point(self, fine_coord, fine_output, target_voxels, tag):
        """occ_head.py:295-310: fine logits [N,ncls] against the labels at their coordinates (unweighted CE)."""
        gt = target_voxels[:, fine_coord[0, :], fine_coord[1, :], fine_coord[2, :]].long()[0]
        return self._loss_terms(fine_output, gt, tag, None)

    def loss(self, output_voxels=None, output_coords_fine=None, output_voxels_fine=None, target_voxels=None,
             target_points=None, img_metas=None, visible_mask=None, **kwargs):
        """occ_head.py:312-337."""
        if target_points:
            raise NotImplementedError("forward_lidarseg is not on the hot path")
        loss_dict = {}
        for index, ov in enumerate(output_voxels):
            loss_dict.update(self.loss_voxel(ov, target_voxels, tag='c_%d' % index))
        if self.cascade_ratio != 1 and (self.sample_from_voxel or self.sample_from_img) and output_voxels_fine:
            acc = {}
            for fine_coord, fine_output in zip(output_coords_fine, output_voxels_fine):
                for k, v in self.loss_point(fine_coord, fine_output, target_voxels, tag='fine').items():
                    acc[k] = acc[k] + v if k in acc else v
            for k, v in acc.items():
                loss_dict[k] = v / len(output_coords_fine)
        return loss_dict
