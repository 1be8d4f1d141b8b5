"""LiDAR-side producer of ``pts_voxel_feats``: ``Voxelization`` (hard mode,
mmdet3d/ops/voxel/voxelize.py:13-138), ``HardSimpleVFE`` (mmdet3d/models/voxel_encoders/voxel_encoder.py:14-45) and
``SparseLiDAREnc8x`` / ``SparseLiDAREnc4x`` (P/coocc/voxel_encoder/sparse_lidar_enc.py:66-190).

spconv 2.3.6 is an un-vendored dependency of the reference; its SubMConv3d / SparseConv3d semantics are restated:
a rule book is a [27, M] table of input rows and every sparse convolution (+ folded eval-mode BN, ReLU, residual) is one
launch of the row-table GEMM that GSFusion already uses (``coocc_conv_fwd`` with ``gather``).  Weights keep spconv 2.x's
[Cout, kd, kh, kw, Cin] layout and the reference's state_dict keys.  Batch size 1 (the reference hard-codes it)."""
import ctypes

import torch
from torch import nn

import os

from . import _lib, core
from ._lib import ConvDesc, call, host_f32, ptr
from .backbone import build_bn
from .core import PackCache, PackedConv, Rows, TILE_HINT, fold_bn, workspace
from .registry import Registry

_F32, _I32 = torch.float32, torch.int32
# rule-book GEMMs with Cin % 32 == 0 (16 of the 18 layers of SparseLiDAREnc8x) on the split-f16 engine (k_gemm_h2w<TABLE>, the
# kernel of the GSFusion gather GEMMs): same fp32 accuracy, operands written by the producing layer's epilogue.  0 = fp32 MFMA.
LIDAR_H2 = os.environ.get("COOCC_LIDAR_H2", "1") != "0"
VOXEL_LAYERS = Registry("voxel_layer")
VOXEL_ENCODERS = Registry("voxel_encoder")
MIDDLE_ENCODERS = Registry("middle_encoder")


# ----------------------------------------------------------------------------- voxelisation + VFE
class Voxelization(nn.Module):
    """Hard voxelisation: forward(points [N,F]) -> (voxels [M,max_points,F], coors [M,3] (z,y,x) int32, num_points [M])."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size, self.point_cloud_range = list(voxel_size), list(point_cloud_range)
        self.max_num_points = max_num_points
        self.max_voxels = tuple(max_voxels) if isinstance(max_voxels, (tuple, list)) else (max_voxels, max_voxels)
        self._ws = {}

    def forward(self, points):
        if not points.is_cuda:
            raise _lib.CooccError("Voxelization runs on the GPU only (no CPU fallback)")
        pts = points.float().contiguous()
        n, F = pts.shape
        mv = self.max_voxels[0] if self.training else self.max_voxels[1]
        dev = pts.device
        voxels = torch.empty(mv, self.max_num_points, F, device=dev, dtype=_F32)
        coors = torch.empty(mv, 3, device=dev, dtype=_I32)
        num = torch.empty(mv, device=dev, dtype=_I32)
        count = torch.empty(1, device=dev, dtype=_I32)
        need = int(_lib.load().coocc_voxelize_ws(n))
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        if key not in self._ws or self._ws[key].numel() < need:
            self._ws[key] = torch.empty(need, device=dev, dtype=torch.uint8)
        ws = self._ws[key]
        call("coocc_voxelize_hard", ptr(pts), n, F, host_f32(self.point_cloud_range), host_f32(self.voxel_size),
             self.max_num_points, mv, ptr(voxels), ptr(coors), ptr(num), ptr(count), ptr(ws), ws.numel())
        m = int(count.item())
        return voxels[:m], coors[:m], num[:m]


VOXEL_LAYERS.register_module(name="Voxelization", module=Voxelization)


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features

    def forward(self, features, num_points, coors=None):
        M, P, F = features.shape
        nf = min(self.num_features, F)
        out = torch.empty(M, nf, device=features.device, dtype=_F32)
        call("coocc_vfe_mean", ptr(features.float().contiguous()), ptr(num_points.int().contiguous()), M, P, F, nf, ptr(out), nf)
        return out


# ----------------------------------------------------------------------------- sparse tensors and rule books
class SparseRows:
    """Active voxels of one resolution: feats [M,C], coors [M,3] (z,y,x) int32, spatial shape (D,H,W); rule books cached
    by indice key as spconv does."""

    def __init__(self, feats, coors, shape, books=None):
        self.feats, self.coors, self.shape = feats, coors, tuple(shape)
        self.books = books if books is not None else {}
        self._map = None

    def index_map(self):
        if self._map is None:
            D, H, W = self.shape
            self._map = torch.empty(D * H * W, device=self.coors.device, dtype=_I32)
            call("coocc_sparse_index_map", ptr(self.coors), self.coors.shape[0], D, H, W, ptr(self._map))
        return self._map

    def subm_table(self, key, ksize=3):
        if key not in self.books:
            D, H, W = self.shape
            M = self.coors.shape[0]
            t = torch.empty(ksize ** 3, M, device=self.coors.device, dtype=_I32)
            call("coocc_sparse_conv_table", ptr(self.coors), M, D, H, W, ksize, 1, ksize // 2, ptr(self.index_map()), ptr(t))
            self.books[key] = t
        return self.books[key]

    def downsample(self, ksize, stride, pad):
        """Active set + rule book of SparseConv3d(ksize, stride, pad): (coors_out, shape_out, table [k^3, Mo])."""
        D, H, W = self.shape
        Do, Ho, Wo = ((v + 2 * pad - ksize) // stride + 1 for v in (D, H, W))
        dev = self.coors.device
        flags = torch.empty(Do * Ho * Wo, device=dev, dtype=torch.uint8)
        call("coocc_sparse_down_flags", ptr(self.coors), self.coors.shape[0], ksize, stride, pad, Do, Ho, Wo, ptr(flags))
        lin = torch.empty(Do * Ho * Wo, device=dev, dtype=_I32)
        cnt = torch.empty(1, device=dev, dtype=_I32)
        ws = torch.empty(Do * Ho * Wo // 1024 + 2, device=dev, dtype=_I32)
        call("coocc_compact_flags", ptr(flags), Do * Ho * Wo, ptr(lin), ptr(cnt), ptr(ws), ws.numel() * 4)
        Mo = int(cnt.item())
        coors = torch.empty(Mo, 3, device=dev, dtype=_I32)
        call("coocc_sparse_lin_to_coors", ptr(lin[:Mo]), Mo, Do, Ho, Wo, ptr(coors), None)
        table = torch.empty(ksize ** 3, Mo, device=dev, dtype=_I32)
        call("coocc_sparse_conv_table", ptr(coors), Mo, D, H, W, ksize, stride, pad, ptr(self.index_map()), ptr(table))
        return coors, (Do, Ho, Wo), table


def _pad4(n):
    return (n + 3) // 4 * 4


def sparse_conv(feats, Cin, pc, table, relu=True, res=None, out=None, out_rows=None, feats_h2=None, twin=False):
    """out[o] = epi(sum_t W_t . feats[table[t][o]]): one row-table GEMM launch (BN folded into pc.scale/bias).
    ``feats_h2``: the H2 (split-f16) copy of ``feats`` when the producer wrote one; ``twin``: also return the H2 copy of ``out``
    (the next rule-book GEMM's operand).  Returns out, or (out, out_h2) with ``twin``."""
    taps, Mo = table.shape
    dev = feats.device
    if out is None:
        out = torch.empty(Mo, pc.Cout, device=dev, dtype=_F32)
    h2 = LIDAR_H2 and core.CONV_ENGINE == "h2" and Cin % 32 == 0 and feats.shape[1] == Cin and out_rows is None
    out_h2 = torch.empty(Mo, pc.Cout, device=dev, dtype=_F32) if (twin and h2 and pc.Cout % 32 == 0) else None
    if Mo == 0:
        return (out, out_h2) if twin else out
    ws = workspace(dev)
    d = ConvDesc()
    d.in_, d.w, d.out = ptr(feats), ptr(pc.w), ptr(out)
    d.scale, d.bias = ptr(pc.scale), ptr(pc.bias)
    d.res = ptr(res)
    d.gather = ptr(table, _I32)
    d.out_rows = ptr(out_rows, _I32) if out_rows is not None else None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = Mo, Cin, pc.Cout, taps
    d.in_stride, d.out_stride = feats.shape[1], out.shape[1]
    d.res_stride = res.shape[1] if res is not None else 0
    d.B = d.Yi = d.Zi = d.Xo = d.Yo = d.Zo = 1
    d.Xi = feats.shape[0]          # number of input rows (lets coocc_conv_fwd pick the pipelined row-table kernel)
    d.ksize, d.stride, d.pad = 1, 1, 0
    d.relu, d.res_mode, d.splitk, d.tile_hint = int(relu), (1 if res is not None else 0), 1, TILE_HINT
    kname = "k_conv"
    if h2:
        if feats_h2 is None:
            feats_h2 = torch.empty(feats.shape[0], Cin, device=dev, dtype=_F32)
            call("coocc_rows_to_h2", ptr(feats), feats.shape[1], feats.shape[0], Cin, 1.0, ptr(feats_h2))
        d.in_, d.in_stride, d.w = ptr(feats_h2), Cin, ptr(pc.h2_pack())
        d.mfma_dtype, d.alpha, kname = 3, 1.0, "k_gemm_h2w"
        if out_h2 is not None:
            d.out_h2_twin = ptr(out_h2)
    with _lib.TIMER.region("%s<sparse table %d->%d>" % (kname, Cin, pc.Cout), 2.0 * Mo * Cin * pc.Cout * taps):
        _lib.conv_fwd(d, pc.w.device)
    return (out, out_h2) if twin else out


# ----------------------------------------------------------------------------- training path (round 5)
# The reference trains this encoder (projects/configs/coocc_nusc/coocc_multi_r50_256x704.py:127-134: norm_cfg SyncBN,
# requires_grad=True; sparse_lidar_enc.py:17-62 puts a BatchNorm1d behind every sparse convolution).  Under ``train()`` every
# sparse convolution is a torch.autograd.Function over the same rule books:
#   forward  y[o]       = sum_t W_t . x[table[t][o]] (+ b)                       row-table GEMM (coocc_conv_fwd, as inference)
#   dgrad    dx[i]      = sum_t W_t^T . dy[inverse[t][i]]                        the same GEMM over the TRANSPOSED rule book
#                         (SubMConv3d: inverse[t] = table[26 - t] -- the active set is its own mirror image; SparseConv3d:
#                         inverse[t][table[t][o]] = o, built once per rule book)
#   wgrad    dW_t[n][c] = sum_o dy[o][n] . x[table[t][o]][c]                     coocc_conv_wgrad with the forward table
# BatchNorm1d runs on the [M, C] rows of the ACTIVE voxels with batch statistics (torch's batch_norm: running statistics are
# updated exactly as nn.BatchNorm1d does), GroupNorm / the dense scatter through the row Functions of autograd.py.
def _inverse_table(table, n_in):
    """[taps, Mo] forward rule book (input row of output o for tap t, or -1) -> [taps, n_in]: the output row that reads input
    i through tap t, or -1 (for a fixed tap the map o -> i is injective)."""
    taps, Mo = table.shape
    inv = torch.full((taps, n_in), -1, device=table.device, dtype=_I32)
    m = table >= 0
    t_idx = torch.arange(taps, device=table.device).view(-1, 1).expand(taps, Mo)[m]
    o_idx = torch.arange(Mo, device=table.device, dtype=_I32).view(1, -1).expand(taps, Mo)[m]
    inv[t_idx, table[m].long()] = o_idx
    return inv


def _table_launch(x2d, in_C, w_packed, out2d, Cout, table, bias=None, tag="sparse"):
    from .autograd import _conv_launch
    n_in = x2d.shape[0]
    _conv_launch(x2d, in_C, w_packed, out2d, Cout, table.shape[0], (1, n_in, 1, 1), (1, out2d.shape[0], 1, 1), 1, 1, 0, None, bias,
                 None, False, table=table, tag=tag)


class SparseConvFn(torch.autograd.Function):
    """y = sparse_conv(x; W, table) (+ bias): SubMConv3d / SparseConv3d of spconv 2.x on rows, differentiable in x, W, b.
    ``weight``: [Cout, k, k, k, Cin] (the reference's state_dict layout); x: [n_in, pad4(Cin)]."""

    @staticmethod
    def forward(ctx, x, weight, bias, table, table_bwd):
        from .autograd import pack_weights_dev
        Cout, Cin = weight.shape[0], weight.shape[-1]
        taps, Mo = table.shape
        Cp = x.shape[1]
        w3 = weight.detach().float().reshape(Cout, taps, Cin)
        if Cp != Cin:
            w3 = torch.cat([w3, w3.new_zeros(Cout, taps, Cp - Cin)], 2)
        w3 = w3.contiguous()
        out = torch.empty(Mo, Cout, device=x.device, dtype=_F32)
        if Mo:
            _table_launch(x, Cp, pack_weights_dev(w3, Cout, Cp, taps, 1), out, Cout, table,
                          bias=bias.detach().float().contiguous() if bias is not None else None, tag="sparse_fwd")
        ctx.save_for_backward(x, w3, table, table_bwd)
        ctx.cfg = (Cin, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .autograd import pack_weights_dev, _pad4 as pad4
        x, w3, table, table_bwd = ctx.saved_tensors
        Cin, has_bias = ctx.cfg
        Cout, taps, Cp = w3.shape
        Mo, n_in = table.shape[1], x.shape[0]
        dev = x.device
        dout = dout.float().contiguous()
        Co4 = pad4(Cout)
        dacc = dout
        if Co4 != Cout:
            dacc = torch.zeros(Mo, Co4, device=dev, dtype=_F32)
            dacc[:, :Cout] = dout
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros(n_in, Cp, device=dev, dtype=_F32)
            if Mo and n_in:
                wp = pack_weights_dev(w3.permute(0, 2, 1).contiguous(), Cout, Cp, taps, 3)        # W'[c][n][t] = w[n][c][t]
                _table_launch(dacc, Co4, wp, dx, Cp, table_bwd, tag="sparse_dgrad")
        if ctx.needs_input_grad[1]:
            dw3 = torch.zeros(Cout, Cp, taps, device=dev, dtype=_F32)
            if Mo:
                ws = workspace(dev)
                with _lib.TIMER.region("k_wgrad<sparse table>", 2.0 * Mo * Cp * Cout * taps):
                    call("coocc_conv_wgrad", ptr(x), n_in, Cp, ptr(dacc), Co4, ptr(table), Mo, Cp, Cout, taps, ptr(dw3), 0, ptr(ws),
                         ws.numel())
            k = round(taps ** (1.0 / 3))
            dw = dw3[:, :Cin, :].permute(0, 2, 1).reshape(Cout, k, k, k, Cin).contiguous()
        if has_bias and ctx.needs_input_grad[2]:
            db = dout.sum(0)
        return dx, dw, db, None, None


def sparse_conv_train(x, conv, table, table_bwd):
    return SparseConvFn.apply(x.contiguous(), conv.weight, conv.bias, table, table_bwd)


class _SpConv(nn.Module):
    """Parameter holder with spconv 2.x's weight layout [Cout, k, k, k, Cin] (key ``weight``)."""

    def __init__(self, cin, cout, ksize=3, bias=False):
        super().__init__()
        self.cin, self.cout, self.ksize = cin, cout, ksize
        self.weight = nn.Parameter(torch.empty(cout, ksize, ksize, ksize, cin))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def packed(self, bn=None, cin_to=None, cout_to=None):
        """``cin_to`` / ``cout_to``: zero-pad the input / output channels (the padded outputs are exact zeros: zero weights, zero
        bias; only for layers without a folded BN) -- how a 16-channel layer feeds the split-f16 engine's 32-channel rows."""
        cin_p = max(_pad4(self.cin), cin_to or 0)
        w = self.weight.detach().reshape(self.cout, self.ksize ** 3, self.cin)
        if cin_p != self.cin:       # rows are padded to a multiple of 4 channels for the 16-byte gathers
            w = torch.cat([w, w.new_zeros(self.cout, self.ksize ** 3, cin_p - self.cin)], 2)
        bias = self.bias
        if cout_to and cout_to > self.cout:
            assert bn is None
            w = torch.cat([w, w.new_zeros(cout_to - self.cout, self.ksize ** 3, cin_p)], 0)
            b = self.bias.detach() if self.bias is not None else w.new_zeros(self.cout)
            bias = torch.cat([b, b.new_zeros(cout_to - self.cout)])
        return PackedConv(w.reshape(w.shape[0], -1).contiguous(), bn=bn, bias=bias, tap_major=True, taps=self.ksize ** 3)


def _gn_rows(x, gn, relu=True, C=None):
    """In-place GroupNorm (+ ReLU) of the first ``C`` channels of the rows (default: all of them)."""
    call("coocc_groupnorm_rows", ptr(x), x.shape[0], C or x.shape[1], x.shape[1], gn.num_groups, ptr(gn.weight.detach()),
         ptr(gn.bias.detach()), float(gn.eps), int(relu))
    return x


class _PostActBlock(nn.Sequential):
    """post_act_block(conv_type='spconv'): SparseConv3d(k3, s2, p1, no bias) + BN1d + ReLU (keys 0.weight, 1.*)."""

    def __init__(self, cin, cout, norm_cfg):
        super().__init__(_SpConv(cin, cout, 3), _bn1d(norm_cfg, cout), nn.ReLU(inplace=True))


def _bn1d(norm_cfg, c):
    """The norm layer mmcv's ``build_norm_layer`` makes for ``norm_cfg`` on [rows, C] features: ``SyncBN`` ->
    ``nn.SyncBatchNorm`` (all-rank statistics once ``torch.distributed`` is initialised, per-rank otherwise -- torch's own
    rule), every other BN-family type -> ``nn.BatchNorm1d``.  Same state_dict keys either way."""
    cfg = dict(norm_cfg or dict(type="BN1d"))
    t = cfg.pop("type", "BN1d")
    cfg.pop("requires_grad", None)
    return (nn.SyncBatchNorm if t == "SyncBN" else nn.BatchNorm1d)(c, **cfg)


class SparseBasicBlock(nn.Module):
    """sparse_lidar_enc.py:40-61: SubM3 + BN + ReLU + SubM3 + BN, + identity, ReLU (keys net.{0,1,3,4}.*)."""

    def __init__(self, planes, norm_cfg):
        super().__init__()
        self.net = nn.Sequential(_SpConv(planes, planes, 3), _bn1d(norm_cfg, planes), nn.ReLU(inplace=True),
                                 _SpConv(planes, planes, 3), _bn1d(norm_cfg, planes))


class _SparseEncoderBase(nn.Module):
    def __init__(self, input_channel, norm_cfg, base_channel, out_channel, sparse_shape_xyz, widths, first_down):
        super().__init__()
        self.sparse_shape_xyz = list(sparse_shape_xyz)
        self.input_channel, self.out_channel = input_channel, out_channel
        self.conv_input = nn.Sequential(_SpConv(input_channel, base_channel, 3, bias=True), nn.GroupNorm(16, base_channel),
                                        nn.ReLU(inplace=True))
        stages, cin = [], base_channel
        for i, (c, down) in enumerate(zip(widths, first_down)):
            mods = ([_PostActBlock(cin, c, norm_cfg)] if down else []) + [SparseBasicBlock(c, norm_cfg), SparseBasicBlock(c, norm_cfg)]
            stages.append(nn.Sequential(*mods))
            cin = c
        self.conv1, self.conv2, self.conv3 = stages
        self.conv_out = nn.Sequential(_SpConv(cin, out_channel, 3, bias=True), nn.GroupNorm(16, out_channel), nn.ReLU(inplace=True))
        self._packs = PackCache(self)

    def _packed(self):
        def build():
            # split-f16 engine: a base width below 32 (16 upstream) is carried as 32-channel rows whose upper channels are exact
            # zeros -- conv_input's pack has zero weight rows / bias for them (its 32-column tile computes them anyway), GroupNorm
            # touches the real channels only, and the first SparseConv3d reads zero-padded input channels -- so that layer runs
            # the split-f16 rule-book kernel like the rest instead of the fp32-MFMA one (0.30 -> 0.13 ms at 300 k rows)
            base = self.conv_input[0].cout
            wide = 32 if (LIDAR_H2 and core.CONV_ENGINE == "h2" and base % 32 != 0 and base < 32
                          and isinstance(self.conv1[0], _PostActBlock)) else None     # (the 4x encoder's first blocks run at the base width)
            d = dict(inp=self.conv_input[0].packed(cout_to=wide), out=self.conv_out[0].packed(), stages=[], base=base)
            for st in (self.conv1, self.conv2, self.conv3):
                ps = []
                for m in st:
                    if isinstance(m, _PostActBlock):
                        ps.append(("down", m[0].packed(bn=m[1], cin_to=wide if m[0].cin == base else None)))
                    else:
                        ps.append(("block", m.net[0].packed(bn=m.net[1]), m.net[3].packed(bn=m.net[4])))
                d["stages"].append(ps)
            return d
        return self._packs.get_modules((self,), build)

    def _forward_train(self, voxel_features, coors):
        """sparse_lidar_enc.py:125-176 with gradients: BatchNorm1d on the active rows with BATCH statistics (as
        ``model.train()`` does upstream; eval-mode BN under ``eval()`` + grad), every convolution a ``SparseConvFn``."""
        from . import autograd as ag
        dev = voxel_features.device
        M, Cin = voxel_features.shape
        cin_p = _pad4(Cin)
        x = voxel_features.float()
        if cin_p != Cin:
            x = torch.cat([x, x.new_zeros(M, cin_p - Cin)], 1)
        cur = SparseRows(None, coors, self.sparse_shape_xyz[::-1])

        def subm(x_, conv, cur_, key):
            tb = cur_.subm_table(key)
            kb = key + "/bwd"
            if kb not in cur_.books:
                cur_.books[kb] = tb.flip(0).contiguous()             # the voxel at -offset: tap 26 - t of the same book
            return sparse_conv_train(x_, conv, tb, cur_.books[kb])

        def bn(x_, m):
            # the module itself: batch statistics + running-stat / num_batches_tracked updates under train() (momentum=None =
            # cumulative average, SyncBatchNorm = all-rank statistics, a user's convert_sync_batchnorm), running stats under eval()
            return m(x_)

        def gn(x_, m):
            return ag.GroupNormRowsFn.apply(x_.contiguous(), m.weight, m.bias, m.num_groups, float(m.eps), True)

        f = gn(subm(x, self.conv_input[0], cur, "in"), self.conv_input[1])
        for si, st in enumerate((self.conv1, self.conv2, self.conv3)):
            for m in st:
                if isinstance(m, _PostActBlock):
                    coors_o, shape_o, table = cur.downsample(3, 2, 1)
                    f = torch.relu(bn(sparse_conv_train(f, m[0], table, _inverse_table(table, f.shape[0])), m[1]))
                    cur = SparseRows(None, coors_o, shape_o)
                else:
                    key = "res%d" % si
                    h = torch.relu(bn(subm(f, m.net[0], cur, key), m.net[1]))
                    f = torch.relu(bn(subm(h, m.net[3], cur, key), m.net[4]) + f)
        f = gn(subm(f, self.conv_out[0], cur, "res2"), self.conv_out[1])
        cur.feats = f
        D, H, W = cur.shape
        rows = ((cur.coors[:, 2].long() * H + cur.coors[:, 1]) * D + cur.coors[:, 0]).int().contiguous()   # (x*H + y)*D + z
        dense = ag.ScatterRowsFn.apply(f, rows, W * H * D)
        vol = Rows(dense, 1, W, H, D, f.shape[1])
        return {'x': vol.as_ncdhw(), 'pts_feats': [cur]}

    def forward(self, voxel_features, coors, batch_size=1):
        """voxel_features [M,Cin], coors [M,3] (z,y,x) or [M,4] (b,z,y,x) -> dict(x=[1,C,W,H,D] dense volume (channels-last
        memory), pts_feats=[SparseRows]).  ``train()``: the differentiable path (``_forward_train``)."""
        if not voxel_features.is_cuda:
            raise _lib.CooccError("the sparse LiDAR encoder runs on the GPU only (no CPU fallback)")
        if coors.shape[1] == 4:
            assert int(batch_size) == 1, "batch size 1 (hard-coded upstream, sparse_lidar_enc.py:109)"
            coors = coors[:, 1:]
        coors = coors.int().contiguous()
        if self.training:
            return self._forward_train(voxel_features, coors)
        p = self._packed()
        dev = voxel_features.device
        M, Cin = voxel_features.shape
        cin_p = _pad4(Cin)
        x = voxel_features.float().contiguous()
        if cin_p != Cin:
            x = torch.cat([x, x.new_zeros(M, cin_p - Cin)], 1).contiguous()
        cur = SparseRows(None, coors, self.sparse_shape_xyz[::-1])
        f = _gn_rows(sparse_conv(x, cin_p, p["inp"], cur.subm_table("in"), relu=False), self.conv_input[1], C=p["base"])
        fh = None                      # H2 copy of f when the layer that made f wrote one (split-f16 engine)
        for si, ps in enumerate(p["stages"]):
            for bi, item in enumerate(ps):
                if item[0] == "down":
                    coors_o, shape_o, table = cur.downsample(3, 2, 1)
                    f, fh = sparse_conv(f, f.shape[1], item[1], table, relu=True, feats_h2=fh, twin=True)
                    cur = SparseRows(None, coors_o, shape_o)
                else:
                    tb = cur.subm_table("res%d" % si)
                    h, hh = sparse_conv(f, f.shape[1], item[1], tb, relu=True, feats_h2=fh, twin=True)
                    f, fh = sparse_conv(h, h.shape[1], item[2], tb, relu=True, res=f, feats_h2=hh, twin=True)
        # conv_out is a SubMConv3d on the last stage's active set: the rule book of that stage's blocks ("res2") is its own
        f = _gn_rows(sparse_conv(f, f.shape[1], p["out"], cur.subm_table("res%d" % (len(p["stages"]) - 1)), relu=False, feats_h2=fh),
                     self.conv_out[1])
        # (the split-f16 range guard is sticky: the flag a layer here may raise is read at the detector's next host read,
        # core.check_h2_overflow -- head.py / serving.py)
        cur.feats = f
        # dense().permute(0,1,4,3,2): [1, C, W, H, D] = (x, y, z); channels-last rows (x*H + y)*D + z
        D, H, W = cur.shape
        rows = torch.empty(f.shape[0], device=dev, dtype=_I32)
        lin = ((cur.coors[:, 0].long() * H + cur.coors[:, 1]) * W + cur.coors[:, 2]).int().contiguous()
        tmp = torch.empty_like(cur.coors)
        call("coocc_sparse_lin_to_coors", ptr(lin), f.shape[0], D, H, W, ptr(tmp), ptr(rows))
        dense = torch.zeros(W * H * D, f.shape[1], device=dev, dtype=_F32)
        dense[rows.long()] = f
        vol = Rows(dense, 1, W, H, D, f.shape[1])
        return {'x': vol.as_ncdhw(), 'pts_feats': [cur]}


@MIDDLE_ENCODERS.register_module()
class SparseLiDAREnc8x(_SparseEncoderBase):
    def __init__(self, input_channel, norm_cfg, base_channel, out_channel, sparse_shape_xyz, **kwargs):
        b = base_channel
        super().__init__(input_channel, norm_cfg, b, out_channel, sparse_shape_xyz, (2 * b, 4 * b, 8 * b), (True, True, True))


@MIDDLE_ENCODERS.register_module()
class SparseLiDAREnc4x(_SparseEncoderBase):
    def __init__(self, input_channel, norm_cfg, base_channel, out_channel, sparse_shape_xyz, **kwargs):
        b = base_channel
        super().__init__(input_channel, norm_cfg, b, out_channel, sparse_shape_xyz, (b, 2 * b, 4 * b), (False, True, True))
