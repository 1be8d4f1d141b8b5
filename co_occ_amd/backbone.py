"""``CustomResNet3D`` -- mirror of P/coocc/backbones/resnet3d.py (BasicBlock depths 10/18/34,
shortcut type B, BN-family norm in eval mode).  Same ctor kwargs, ``forward(x) -> list`` and
state_dict keys (``input_proj.{0,1}.*``, ``layers.{s}.{b}.{conv1,bn1,conv2,bn2,downsample.{0,1}}.*``).
Every conv (+folded BN, ReLU, residual add) is one implicit-GEMM launch on channels-last rows.
"""
from torch import nn

from .core import PackCache, PackedConv, conv_rows, to_rows
from .registry import BACKBONES

_NORMS = {"BN": nn.BatchNorm3d, "BN3d": nn.BatchNorm3d, "SyncBN": nn.SyncBatchNorm}


def _eval_only(module):
    """The ``forward_rows`` fast paths fold the BN running statistics and return tensors without ``grad_fn``: they are
    inference paths.  Under ``model.train()`` the reference normalises with batch statistics and is differentiable --
    use ``forward`` (which dispatches to co_occ_amd.autograd) or call ``.eval()`` first."""
    if module.training:
        raise RuntimeError("%s.forward_rows is the eval-mode (folded-BN, no autograd) path but the module is in training "
                           "mode; call .eval(), or use forward() / co_occ_amd.autograd for training" % type(module).__name__)


def build_bn(norm_cfg, num_features):
    """BN-family only: the hot-path configs use SyncBN/BN3d (coocc_multi_r50_256x704.py:141-160);
    GroupNorm volumes are not implemented in the HIP path."""
    cfg = dict(norm_cfg or dict(type="BN3d"))
    t = cfg.pop("type")
    cfg.pop("requires_grad", None)
    if t not in _NORMS:
        raise NotImplementedError("norm type %r: only BN-family norms are folded into the HIP convs" % t)
    return _NORMS[t](num_features, **cfg)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, downsample=None, norm_cfg=None):
        super().__init__()
        self.conv1 = nn.Conv3d(in_planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = build_bn(norm_cfg, planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = build_bn(norm_cfg, planes)
        self.downsample = downsample
        self.stride = stride

    def packed(self):
        d = dict(c1=PackedConv(self.conv1.weight, bn=self.bn1, ksize=3, stride=self.stride, pad=1),
                 c2=PackedConv(self.conv2.weight, bn=self.bn2, ksize=3, pad=1))
        if self.downsample is not None:
            d["ds"] = PackedConv(self.downsample[0].weight, bn=self.downsample[1], ksize=1, stride=self.stride)
        return d

    def run(self, x, p, readers=()):
        """resnet3d.py:47-64 on Rows.  ``readers``: the layers that read the block's output next (the next block's conv1 /
        downsample, an FPN lateral): a split-f16 layer among them gets its H2 operand from conv2's epilogue."""
        out = conv_rows(x, p["c1"], relu=True, twin_for=(p["c2"],))
        res = conv_rows(x, p["ds"], relu=False) if "ds" in p else x
        return conv_rows(out, p["c2"], relu=True, res=res, twin_for=readers)


@BACKBONES.register_module()
class CustomResNet3D(nn.Module):
    def __init__(self, depth, block_inplanes=[64, 128, 256, 512], block_strides=[1, 2, 2, 2],
                 out_indices=(0, 1, 2, 3), n_input_channels=3, shortcut_type='B',
                 norm_cfg=dict(type='BN3d', requires_grad=True), widen_factor=1.0):
        super().__init__()
        layer_metas = {10: [1, 1, 1, 1], 18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}
        if depth not in layer_metas:
            raise NotImplementedError("CustomResNet3D depth %s: Bottleneck depths are not on the coocc_nusc path" % depth)
        if shortcut_type != 'B':
            raise NotImplementedError("shortcut_type 'A' is not used by any shipped config")
        block_inplanes = [int(x * widen_factor) for x in block_inplanes]
        self.in_planes = block_inplanes[0]
        self.out_indices = out_indices
        self.input_proj = nn.Sequential(
            nn.Conv3d(n_input_channels, self.in_planes, kernel_size=(1, 1, 1), stride=(1, 1, 1), bias=False),
            build_bn(norm_cfg, self.in_planes), nn.ReLU(inplace=True))
        self.layers = nn.ModuleList()
        for i in range(len(block_inplanes)):
            self.layers.append(self._make_layer(block_inplanes[i], layer_metas[depth][i], block_strides[i], norm_cfg))
        for m in self.modules():                       # resnet3d.py:161-169
            if isinstance(m, nn.Conv3d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm3d, nn.SyncBatchNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._packs = PackCache(self)

    def _make_layer(self, planes, blocks, stride, norm_cfg):
        downsample = None
        if stride != 1 or self.in_planes != planes:
            downsample = nn.Sequential(nn.Conv3d(self.in_planes, planes, 1, stride=stride, bias=False),
                                       build_bn(norm_cfg, planes))
        layers = [BasicBlock(self.in_planes, planes, stride, downsample, norm_cfg)]
        self.in_planes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.in_planes, planes, norm_cfg=norm_cfg))
        return nn.Sequential(*layers)

    def _packed(self):
        def build():
            return dict(proj=PackedConv(self.input_proj[0].weight, bn=self.input_proj[1], ksize=1),
                        blocks=[[b.packed() for b in layer] for layer in self.layers])
        return self._packs.get_modules((self,), build)

    def forward_rows(self, x, readers=None):
        """``readers``: per output level, the layers outside this module that read it next (FPN3D's laterals)."""
        _eval_only(self)
        p = self._packed()
        flat = [bp for layer in p["blocks"] for bp in layer]
        x = conv_rows(to_rows(x), p["proj"], relu=True, twin_for=(flat[0]["c1"], flat[0].get("ds")))
        res = []
        n = 0
        for i, layer in enumerate(self.layers):
            for blk, bp in zip(layer, p["blocks"][i]):
                n += 1
                nxt = flat[n] if n < len(flat) else {}
                rd = [nxt.get("c1"), nxt.get("ds")]
                if readers is not None and i in self.out_indices and bp is p["blocks"][i][-1]:
                    rd += list(readers[self.out_indices.index(i)])
                x = blk.run(x, bp, tuple(rd))
            if i in self.out_indices:
                res.append(x)
        return res

    def forward(self, x):
        """[B,C,X,Y,Z] -> list of [B,C_i,X_i,Y_i,Z_i] (resnet3d.py:196-205).  In training mode every BatchNorm uses batch
        (SyncBN: all-rank) statistics and the outputs carry ``grad_fn`` (co_occ_amd.autograd), as upstream under
        ``model.train()``; in eval mode BN is folded into the GEMM epilogue."""
        if self.training:
            from . import autograd as ag
            rows, geom = ag.rows_from_ncdhw(x)
            return [ag.ncdhw_from_rows(r, g) for r, g in ag.backbone_forward_train(self, rows, geom)]
        return [r.as_ncdhw() for r in self.forward_rows(x)]
