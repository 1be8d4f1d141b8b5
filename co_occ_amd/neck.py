"""``FPN3D`` -- mirror of P/coocc/necks/fpn3d.py.  Same kwargs / forward / state_dict keys
(``lateral_convs.{i}.0.{conv,bn}.*``, ``fpn_convs.{i}.0.{conv,bn}.*``).  1x1x1 laterals and
3x3x3 output convs are implicit-GEMM launches; the top-down path is one fused
trilinear-upsample-add kernel per level (fpn3d.py:88-92).
"""
from torch import nn

from ._lib import call, ptr
from .backbone import _eval_only, build_bn
from .core import PackCache, PackedConv, conv_rows, to_rows
from .registry import NECKS


class ConvModule(nn.Module):
    """Parameter container with mmcv ConvModule's attribute names (conv, bn, activate)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, norm_cfg=None, bias=False):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, padding=padding, bias=bias)
        self.bn = build_bn(norm_cfg, out_channels)
        self.activate = nn.ReLU(inplace=True)


@NECKS.register_module()
class FPN3D(nn.Module):
    def __init__(self, in_channels=[80, 160, 320, 640], out_channels=256,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), conv_cfg=dict(type='Conv3d'),
                 act_cfg=dict(type='ReLU'), with_cp=False, upsample_cfg=dict(mode='trilinear'), init_cfg=None):
        super().__init__()
        if upsample_cfg.get("mode", "trilinear") != "trilinear":
            raise NotImplementedError("FPN3D: only trilinear upsampling is implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.with_cp, self.upsample_cfg, self.fp16_enabled = with_cp, upsample_cfg, False
        self.num_out = len(in_channels)
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.num_out):
            self.lateral_convs.append(nn.Sequential(ConvModule(in_channels[i], out_channels, 1, 0, norm_cfg)))
            self.fpn_convs.append(nn.Sequential(ConvModule(out_channels, out_channels, 3, 1, norm_cfg)))
        self._packs = PackCache(self)

    def _packed(self):
        def build():
            return dict(
                lat=[PackedConv(m[0].conv.weight, bn=m[0].bn, bias=m[0].conv.bias, ksize=1) for m in self.lateral_convs],
                out=[PackedConv(m[0].conv.weight, bn=m[0].bn, bias=m[0].conv.bias, ksize=3, pad=1) for m in self.fpn_convs])
        return self._packs.get_modules((self,), build)

    def lateral_packs(self):
        """The 1x1x1 laterals, per input level: what the backbone's last block of each stage is read by (H2 twins)."""
        return [(pc,) for pc in self._packed()["lat"]]

    def forward_rows(self, inputs, readers=None):
        """``readers``: per output level, the layers that read it next (OccHead's occ_convs)."""
        import torch
        from . import core
        assert len(inputs) == len(self.in_channels)
        _eval_only(self)
        p = self._packed()
        top = self.num_out - 1
        # only the top lateral reaches its fpn_conv unchanged; the others are updated in place by the top-down pass below,
        # which then writes the H2 twin of the UPDATED rows itself
        # levels 1.. (10 000 / 1 250 / 169 rows at configs[1]) on a side stream beside level 0's 80 000-row launches (core.Fork)
        rows_in = [to_rows(x) for x in inputs]
        dev = rows_in[0].t.device
        lbr = core.Fork(3, dev, enable=self.num_out > 1)
        with lbr:
            small = [conv_rows(x, p["lat"][i], relu=True, twin_for=((p["out"][i],) if i == top else ()))
                     for i, x in enumerate(rows_in) if i > 0]
        lat = [conv_rows(rows_in[0], p["lat"][0], relu=True, twin_for=((p["out"][0],) if top == 0 else ()))] + small
        lbr.join(*[t for r in small for t in (r.t, r.h2) if t is not None])
        for i in range(self.num_out - 1, 0, -1):
            c, f = lat[i], lat[i - 1]
            f.h16 = f.h2 = None   # the rows change in place: the 16-bit copies the lateral conv's epilogue wrote are stale
            tw = None
            if core.CONV_ENGINE == "h2" and core.CONV_DTYPE == "f32" and f.C % 32 == 0 and core.takes_h2(f, (p["out"][i - 1],)):
                tw = f.h2 = torch.empty(f.B * f.V, f.C, device=f.t.device, dtype=torch.float32)
            call("coocc_upsample_add_trilinear_ex", ptr(c.t), ptr(f.t), f.B, f.C, c.X, c.Y, c.Z, f.X, f.Y, f.Z, ptr(tw))
        obr = core.Fork(3, dev, enable=self.num_out > 1)
        with obr:
            small = [conv_rows(x, p["out"][i], relu=True, twin_for=(readers[i] if readers is not None else ()))
                     for i, x in enumerate(lat) if i > 0]
        out0 = conv_rows(lat[0], p["out"][0], relu=True, twin_for=(readers[0] if readers is not None else ()))
        obr.join(*[t for r in small for t in (r.t, r.h2) if t is not None])
        return [out0] + small

    def forward(self, inputs):
        """list of [B,C_i,...] -> list of [B,out,...] (fpn3d.py:70-108); training mode: batch-statistics BN + autograd."""
        if self.training:
            from . import autograd as ag
            feats = [ag.rows_from_ncdhw(x) for x in inputs]
            return [ag.ncdhw_from_rows(r, g) for r, g in ag.neck_forward_train(self, feats)]
        return [r.as_ncdhw() for r in self.forward_rows(inputs)]
