"""Backward passes as ``torch.autograd.Function``s over the C-ABI kernels.

Scope: the differentiable ops of the hot path with frozen-statistics BN (eval-mode BN folded into a
per-channel scale/shift, exactly as the forward does): Conv3d / Linear (+BN, ReLU, residual) on
channels-last rows, the G1 row gather, voxel pooling / fused lift-splat, the R2 render composite and
the FPN trilinear upsample-add.  Indices (FPS / ball / top-K / assignment, voxel keys) are
non-differentiable, as upstream.  Every backward is a HIP kernel; nothing falls back to torch ops.
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, call, ptr
from .core import TILE_HINT, out_dim, workspace

_F32 = torch.float32
_tables = {}


def _pad4(n):
    return (n + 3) // 4 * 4


def tap_table(dev, B, Xi, Yi, Zi, ksize, stride, pad, dgrad):
    """[taps, M] int32 row table of a conv geometry (cached): forward reads / dgrad reads."""
    key = (dev.index, B, Xi, Yi, Zi, ksize, stride, pad, bool(dgrad))
    if key not in _tables:
        Xo, Yo, Zo = (out_dim(n, ksize, stride, pad) for n in (Xi, Yi, Zi))
        M = B * Xi * Yi * Zi if dgrad else B * Xo * Yo * Zo
        t = torch.empty(ksize ** 3, M, dtype=torch.int32, device=dev)
        call("coocc_conv_tap_table", B, Xi, Yi, Zi, Xo, Yo, Zo, ksize, stride, pad, int(bool(dgrad)), ptr(t))
        _tables[key] = t
    return _tables[key]


def pack_weights_dev(w, Cout, Cin, taps, mode):
    """Device-side fragment-major packing (modes: 0 fwd, 1 fwd tap-major, 2 dgrad flipped, 3 dgrad)."""
    lib = _lib.load()
    w = w.detach().float().contiguous()
    n = lib.coocc_conv_pack_weights_dev(None, Cout, Cin, taps, mode, None, None)
    if n < 0:
        _lib.check(int(n))
    packed = torch.empty(n, dtype=_F32, device=w.device)
    n = lib.coocc_conv_pack_weights_dev(ptr(w), Cout, Cin, taps, mode, ptr(packed), _lib.stream())
    if n < 0:
        _lib.check(int(n))
    return packed


def _conv_launch(x2d, in_C, w_packed, out2d, Cout, taps, geom_in, geom_out, ksize, stride, pad, scale, shift, res2d, relu,
                 table=None):
    d = ConvDesc()
    ws = workspace(x2d.device)
    d.in_, d.w, d.out = ptr(x2d), ptr(w_packed), ptr(out2d)
    d.scale, d.bias = ptr(scale), ptr(shift)
    d.res = ptr(res2d)
    d.gather = ptr(table, torch.int32) if table is not None else None
    d.out_rows = None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = out2d.shape[0], in_C, Cout, taps
    d.in_stride, d.out_stride = x2d.shape[1], out2d.shape[1]
    d.res_stride = res2d.shape[1] if res2d is not None else 0
    B, Xi, Yi, Zi = geom_in
    _, Xo, Yo, Zo = geom_out
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = B, Xi, Yi, Zi, Xo, Yo, Zo
    d.ksize, d.stride, d.pad = ksize, stride, pad
    d.relu, d.res_mode, d.splitk = int(relu), (1 if res2d is not None else 0), 0
    d.tile_hint = TILE_HINT
    with _lib.TIMER.region("conv_fwd(autograd)", 2.0 * d.M * in_C * Cout * taps):
        _lib.check(_lib.load().coocc_conv_fwd(ctypes.byref(d), _lib.stream()))


class ConvRowsFn(torch.autograd.Function):
    """y = relu(scale * conv(x, W) + shift + res) on channels-last rows; scale/shift are constants
    (folded eval-mode BN and/or the conv bias).  Differentiable in x, W, conv bias and res."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, res2d, scale, shift, geom, ksize, stride, pad, relu):
        B, Xi, Yi, Zi = geom
        Cout, Cin = weight.shape[0], weight.shape[1]
        taps = ksize ** 3
        assert x2d.shape == (B * Xi * Yi * Zi, Cin) and x2d.is_contiguous() and Cin % 4 == 0
        Xo, Yo, Zo = (out_dim(n, ksize, stride, pad) for n in (Xi, Yi, Zi))
        geom_out = (B, Xo, Yo, Zo)
        out = torch.empty(B * Xo * Yo * Zo, Cout, device=x2d.device, dtype=_F32)
        wp = pack_weights_dev(weight.reshape(Cout, Cin, taps), Cout, Cin, taps, 0)
        eff_shift = shift
        if bias is not None:    # y = scale * (conv + b) + shift
            eff_shift = (bias.detach() * scale if scale is not None else bias.detach()) + (shift if shift is not None else 0)
            eff_shift = eff_shift.float().contiguous()
        _conv_launch(x2d, Cin, wp, out, Cout, taps, geom, geom_out, ksize, stride, pad, scale, eff_shift, res2d, relu)
        ctx.save_for_backward(x2d, weight, out, scale if scale is not None else torch.empty(0, device=x2d.device))
        ctx.cfg = (geom, geom_out, ksize, stride, pad, relu, bias is not None, res2d is not None, scale is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2d, weight, out, scale = ctx.saved_tensors
        geom, geom_out, ksize, stride, pad, relu, has_bias, has_res, has_scale = ctx.cfg
        scale = scale if has_scale else None
        B, Xi, Yi, Zi = geom
        Cout, Cin = weight.shape[0], weight.shape[1]
        taps = ksize ** 3
        Mo, Mi = out.shape[0], x2d.shape[0]
        dev = x2d.device
        dout = dout.float().contiguous()
        need_x, need_w, need_b, need_res = ctx.needs_input_grad[:4]
        Cp = _pad4(Cout)
        dacc = torch.zeros(Mo, Cp, device=dev, dtype=_F32) if Cp != Cout else torch.empty(Mo, Cp, device=dev, dtype=_F32)
        dres = torch.empty(Mo, Cout, device=dev, dtype=_F32) if (has_res and need_res) else None
        dbias = torch.empty(Cout, device=dev, dtype=_F32) if (has_bias and need_b) else None
        ws = workspace(dev)
        call("coocc_conv_epilogue_bwd", ptr(dout), Cout, ptr(out), Cout, ptr(scale), Mo, Cout, int(relu), ptr(dacc), Cp,
             ptr(dres), Cout, 0, ptr(dbias), 0, ptr(ws), ws.numel())
        if dbias is not None and scale is not None:
            dbias = dbias * scale
        dx = dw = None
        if need_x:
            dx = torch.empty(Mi, Cin, device=dev, dtype=_F32)
            w3 = weight.reshape(Cout, Cin, taps)
            if stride == 1:
                wp = pack_weights_dev(w3, Cout, Cin, taps, 2)
                _conv_launch(dacc, Cp, wp, dx, Cin, taps, geom_out, geom, ksize, 1, ksize - 1 - pad, None, None, None, False)
            else:
                wp = pack_weights_dev(w3, Cout, Cin, taps, 3)
                tb = tap_table(dev, B, Xi, Yi, Zi, ksize, stride, pad, True)
                _conv_launch(dacc, Cp, wp, dx, Cin, taps, geom_out, geom, ksize, stride, pad, None, None, None, False,
                             table=tb)
        if need_w:
            dw = torch.empty(Cout, Cin, taps, device=dev, dtype=_F32)
            tb = tap_table(dev, B, Xi, Yi, Zi, ksize, stride, pad, False) if (taps > 1 or stride > 1) else None
            with _lib.TIMER.region("k_wgrad", 2.0 * Mo * Cin * Cout * taps):
                call("coocc_conv_wgrad", ptr(x2d), Cin, ptr(dacc), Cp, ptr(tb), Mo, Cin, Cout, taps, ptr(dw), 0, ptr(ws),
                     ws.numel())
            dw = dw.view_as(weight)
        return dx, dw, dbias, dres, None, None, None, None, None, None, None


def conv3d_rows(x2d, weight, geom, bias=None, bn=None, stride=1, pad=None, relu=True, res2d=None):
    """Differentiable Conv3d(+eval BN)(+res)(+ReLU) on rows [B*X*Y*Z, Cin] -> ([B*Xo*Yo*Zo, Cout], out geom).
    ``weight`` [Cout,Cin,k,k,k] (or [Cout,Cin] for a Linear / 1x1)."""
    from .core import fold_bn
    ksize = weight.shape[2] if weight.dim() == 5 else 1
    if pad is None:
        pad = ksize // 2
    scale = shift = None
    if bn is not None:
        s, b = fold_bn(bn, None)
        scale, shift = s.to(x2d.device).contiguous(), b.to(x2d.device).contiguous()
    out = ConvRowsFn.apply(x2d.contiguous(), weight, bias, res2d, scale, shift, tuple(geom), ksize, stride, pad, relu)
    B, X, Y, Z = geom
    return out, (B, out_dim(X, ksize, stride, pad), out_dim(Y, ksize, stride, pad), out_dim(Z, ksize, stride, pad))


def linear_rows(x2d, weight, bias=None, relu=False):
    """Differentiable nn.Linear (+ReLU) on rows through the same kernels."""
    n = x2d.shape[0]
    return ConvRowsFn.apply(x2d.contiguous(), weight, bias, None, None, None, (1, n, 1, 1), 1, 1, 0, relu)
