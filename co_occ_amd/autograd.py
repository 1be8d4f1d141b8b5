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
# 3x3x3 stride-1 convolutions of the training path (forward and dgrad) through the Winograd kernels when the layer is
# large enough (core.WINO_MIN_ROWS); wgrad stays a direct GEMM.  COOCC_TRAIN_WINO=0: direct convolutions everywhere.
TRAIN_WINO = __import__("os").environ.get("COOCC_TRAIN_WINO", "1") != "0"
# weight gradients of those layers in the Winograd domain too (COOCC_TRAIN_WINO_WGRAD=0: direct k_wgrad)
TRAIN_WINO_WGRAD = TRAIN_WINO and __import__("os").environ.get("COOCC_TRAIN_WINO_WGRAD", "1") != "0"
# Winograd forward / dgrad GEMMs of the training path on the split-f16 engine (weights transformed + split on the device every
# step, coocc_wino_pack_weights_h2_dev); 0: the fp32-MFMA kernels of rounds 1-3
TRAIN_H2 = __import__("os").environ.get("COOCC_TRAIN_H2", "1") != "0"
# ... and the dgrad GEMMs too.  The operand of a dgrad GEMM is a GRADIENT, whose magnitude is set by the loss scale, not by
# BatchNorm, and an f16 is subnormal below 6.1e-5 (with a mean-reduced loss over 640 k voxels |dy| ~ 1e-6: the hi half alone would
# keep a handful of bits).  So the gradient operand gets a per-tensor power-of-two scale chosen ON THE DEVICE: the epilogue-backward
# pass that produces dacc also collects max |dacc| (coocc_conv_epilogue_bwd_ex), a one-thread kernel turns it into {2^k, 2^-k} with
# max |dacc| 2^k in [TARGET / 2, TARGET), the operand writers multiply by the first word and the GEMM's alpha by the second --
# exact, no host read, any loss scale.  TARGET = 1024 leaves the F(4x4) input transform (x 100 / 8) inside the f16 range.
# COOCC_TRAIN_H2_DGRAD=0: dgrad on the fp32-MFMA kernels.
TRAIN_H2_DGRAD = TRAIN_H2 and __import__("os").environ.get("COOCC_TRAIN_H2_DGRAD", "1") != "0"
TRAIN_H2_GRAD_TARGET = 1024.0
# ... and the weight gradients (csrc/wgrad_h2.hip): the Winograd-domain form of the 3x3x3 stride-1 layers (Z in {2, 4, 8}) and the
# 1x1x1 / Linear layers, both operands rewritten voxel-major (KH2), the gradient one with the same device-chosen scale.
# COOCC_TRAIN_H2_WGRAD=0: the fp32-MFMA k_wgrad.
TRAIN_H2_WGRAD = TRAIN_H2 and __import__("os").environ.get("COOCC_TRAIN_H2_WGRAD", "1") != "0"
_amax_words = {}


def _amax_word(dev):
    """The zeroed words (COOCC_AMAX_WORDS) coocc_conv_epilogue_bwd_ex collects max |dacc| in (left zero by its second kernel); one
    set per stream."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _amax_words:
        _amax_words[key] = torch.zeros(2048, device=dev, dtype=torch.int32)
    return _amax_words[key]


def _second_word(t):
    p = _lib.DevPtr(t.data_ptr() + 4)
    p._keep = t
    return p

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


_live = {}


def live_taps(dev, B, Xi, Yi, Zi, ksize, stride, pad):
    """(indices of the taps that read at least one real voxel, their rows of the forward table), cached per geometry."""
    key = (dev.index, B, Xi, Yi, Zi, ksize, stride, pad)
    if key not in _live:
        tb = tap_table(dev, B, Xi, Yi, Zi, ksize, stride, pad, False)
        idx = torch.nonzero((tb >= 0).any(1)).flatten()
        _live[key] = (idx, tb[idx].contiguous())
    return _live[key]


_classes = {}


def dgrad_classes(dev, B, Xi, Yi, Zi, ksize, stride, pad):
    """Strided dgrad by residue class: input voxels with the same (x, y, z) mod stride are reached through the same
    subset of taps (stride 2, k = 3: 1, 2, 4 or 8 of the 27), so each class is its own small row-table GEMM instead
    of 27 taps of which 7/8 of the table entries are empty.  -> list of (rows int32 [Mc], taps int64 [Tc], table
    int32 [Tc, Mc]) over the non-empty classes, cached per geometry."""
    key = (dev.index, B, Xi, Yi, Zi, ksize, stride, pad)
    if key not in _classes:
        tb = tap_table(dev, B, Xi, Yi, Zi, ksize, stride, pad, True)          # [taps, Mi]: output row or -1
        m = torch.arange(B * Xi * Yi * Zi, device=dev)
        cls = (((m // (Yi * Zi)) % Xi % stride) * stride + (m // Zi) % Yi % stride) * stride + m % Zi % stride
        out = []
        for c in range(stride ** 3):
            rows = torch.nonzero(cls == c).flatten()
            if rows.numel() == 0:
                continue
            sub = tb[:, rows]
            taps = torch.nonzero((sub >= 0).any(1)).flatten()
            if taps.numel() == 0:
                continue
            out.append((rows.int().contiguous(), taps, sub[taps].contiguous()))
        _classes[key] = out
    return _classes[key]


def pack_weights_dev(w, Cout, Cin, taps, mode):
    """Device-side fragment-major packing (modes: 0 fwd, 1 fwd tap-major, 2 dgrad flipped, 3 dgrad)."""
    lib = _lib.load()
    w = w.detach().float().contiguous()
    n = lib.coocc_conv_pack_weights_dev(None, Cout, Cin, taps, mode, None, None)
    if n < 0:
        _lib.check(int(n))
    packed = torch.empty(n, dtype=_F32, device=w.device)
    n = lib.coocc_conv_pack_weights_dev(ptr(w), Cout, Cin, taps, mode, ptr(packed), _lib.stream(w.device))
    if n < 0:
        _lib.check(int(n))
    return packed


def pack_weights_h2_dev(w, Cout, Cin, taps, mode):
    """``pack_weights_dev`` for the split-f16 engine (coocc_conv_pack_weights_h2_dev)."""
    lib = _lib.load()
    n = lib.coocc_conv_pack_weights_h2_dev(None, Cout, Cin, taps, mode, None, None)
    if n < 0:
        _lib.check(int(n))
    packed = torch.empty(n, dtype=_F32, device=w.device)
    n = lib.coocc_conv_pack_weights_h2_dev(ptr(w), Cout, Cin, taps, mode, ptr(packed), _lib.stream(w.device))
    if n < 0:
        _lib.check(int(n))
    return packed


def _h2_direct(K, flops):
    """Direct-form GEMMs of the training path on the split-f16 engine: K % 32 == 0 and enough work to pay for the operand pass."""
    from . import core
    return TRAIN_H2 and core.CONV_ENGINE == "h2" and core.H2_DIRECT and K % 32 == 0 and flops >= core.H2_DIRECT_MIN_FLOPS


def _rows_h2(x2d, C, scale_dev=None):
    xh = torch.empty(x2d.shape[0], C, device=x2d.device, dtype=_F32)
    call("coocc_rows_to_h2_ex", ptr(x2d), x2d.shape[1], x2d.shape[0], C, 1.0, ptr(scale_dev), ptr(xh))
    return xh


class _DevWinoF32:
    """What core.conv_rows_wino needs of a PackedConv, with the Winograd packs transformed on the device from the live
    parameter (training re-packs every step): forward packs, or (``dgrad``) the packs of dx = conv(dy, W')."""

    def __init__(self, w5, dgrad, scale=None, bias=None):
        Cout, Cin = w5.shape[:2]
        self.Cin, self.Cout = (Cout, Cin) if dgrad else (Cin, Cout)
        self.scale, self.bias = scale, bias
        self._w_raw, self.wino_tile = True, None
        self._w5, self._dgrad, self._packs = w5, int(bool(dgrad)), {}
        self.operand_scale_dev = None         # split-f16 engine, dgrad: {scale, 1 / scale} of the gradient rows, chosen on the device

    def wino_pack(self, tile):
        if tile not in self._packs:
            lib = _lib.load()
            Cout, Cin = self._w5.shape[:2]
            n = lib.coocc_wino_pack_weights_dev(None, Cout, Cin, tile, self._dgrad, None, None)
            if n < 0:
                _lib.check(int(n))
            packed = torch.empty((tile + 2) ** 2, n // (tile + 2) ** 2, dtype=_F32, device=self._w5.device)
            n = lib.coocc_wino_pack_weights_dev(ptr(self._w5), Cout, Cin, tile, self._dgrad, ptr(packed), _lib.stream(self._w5.device))
            if n < 0:
                _lib.check(int(n))
            self._packs[tile] = packed
        return self._packs[tile]


class _DevWino(_DevWinoF32):
    def wino_h2_pack(self, tile):
        """The same packs for the split-f16 engine (csrc/gemm_h2.hip), transformed + split on the device from the live parameter:
        training's Winograd forward / dgrad GEMMs run on the f16 matrix cores like inference's (core.h2_capable)."""
        key = ("h2", tile)
        if key not in self._packs:
            lib = _lib.load()
            Cout, Cin = self._w5.shape[:2]
            n = lib.coocc_wino_pack_weights_h2_dev(None, Cout, Cin, tile, self._dgrad, None, None)
            if n < 0:
                _lib.check(int(n))
            packed = torch.empty((tile + 2) ** 2, n // (tile + 2) ** 2, dtype=_F32, device=self._w5.device)
            n = lib.coocc_wino_pack_weights_h2_dev(ptr(self._w5), Cout, Cin, tile, self._dgrad, ptr(packed), _lib.stream(self._w5.device))
            if n < 0:
                _lib.check(int(n))
            self._packs[key] = packed
        return self._packs[key]


def _wino_train(x2d, geom, w5, dgrad, out2d, scale, shift, res2d, relu, grad_scale=None):
    """Winograd path of a 3x3x3 stride-1 pad-1 convolution in training (forward or dgrad); False if not eligible.
    ``grad_scale``: dgrad on the split-f16 engine -- the device-side {scale, 1 / scale} of the gradient rows x2d."""
    from . import core
    if not TRAIN_WINO:
        return False
    B, X, Y, Z = geom
    pk = (_DevWino if ((TRAIN_H2_DGRAD and grad_scale is not None) if dgrad else TRAIN_H2) else _DevWinoF32)(w5, dgrad, scale, shift)
    pk.operand_scale_dev = grad_scale if dgrad else None
    xr = core.Rows(x2d, B, X, Y, Z, pk.Cin)
    plan = core.wino_plan(xr, pk, out2d.shape[0], 1 if res2d is not None else 0)
    if plan is None:
        return False
    core.conv_rows_wino(xr, pk, core.Rows(out2d, B, X, Y, Z, pk.Cout), relu,
                        core.Rows(res2d, B, X, Y, Z, pk.Cout) if res2d is not None else None, plan)
    return True


_ztables = {}


def _kh2(x2d, rows, rows_pad, C, scale, scale_dev, out):
    call("coocc_rows_to_kh2", ptr(x2d), C, rows, rows_pad, C, float(scale), ptr(scale_dev), ptr(out))
    return out


def _wino_wgrad(x2d, dacc, geom, Cin, Cout, dw, gscale=None):
    """Weight gradient of a 3x3x3 stride-1 pad-1 layer in the Winograd domain; False if the layer is not eligible.
    ``gscale``: the device-side {scale, 1 / scale} of dacc -> the GEMM runs on the split-f16 engine (csrc/wgrad_h2.hip)."""
    from . import core
    if not TRAIN_WINO_WGRAD:
        return False
    B, X, Y, Z = geom
    dev = x2d.device
    xr = core.Rows(x2d, B, X, Y, Z, Cin)
    pk = _DevWinoF32(torch.empty(Cout, Cin, 0, device=dev), False)      # geometry only: no packs are made (fp32 V / dM rows)
    plan = core.wino_plan(xr, pk, x2d.shape[0], 0)
    if plan is None:
        return False
    tile, pts, Tx, Ty, rows, G, _ = plan
    ws = workspace(dev)
    if ws.numel() < pts * 3 * Cin * Cout or pts * G * max(Cin, Cout) * 4 >= 0xFFFFFF00:
        return False
    if TRAIN_H2_WGRAD and gscale is not None and core.CONV_ENGINE == "h2" and Z in (2, 4, 8) and Cin % 4 == 0 and Cout % 4 == 0:
        # both operands straight into the voxel-major split-f16 form (no fp32 V / dM): V (|V| <= 100 |x|) with the forward path's
        # static scale; dM = A dY A^T amplifies by up to 15^2 on top of the device-chosen gradient scale (max |dacc| -> [512, 1024)):
        # 1/16 keeps it below the f16 range
        vs, ms = core.H2_WINO_SCALE[tile], 1.0 / 16.0
        Vk = core._wino_buffer(dev, "Vk", pts * G * Cin)
        Mk = core._wino_buffer(dev, "Mk", pts * G * Cout)
        call("coocc_wino_operand_kh2", 0, ptr(x2d), x2d.shape[1], B, X, Y, Z, Cin, tile, ptr(Vk), G, vs, None)
        call("coocc_wino_operand_kh2", 1, ptr(dacc), dacc.shape[1], B, X, Y, Z, Cout, tile, ptr(Mk), G, ms, ptr(gscale))
        with _lib.TIMER.region("k_wgrad_h2 wino%d" % tile, 2.0 * pts * rows * 3 * Cin * Cout):
            call("coocc_wino_wgrad_h2", ptr(Vk), ptr(Mk), G, Z, Cin, Cout, tile, 1.0 / (vs * ms), _second_word(gscale), ptr(dw), 0,
                 ptr(ws), ws.numel())
        return True
    V = core._wino_buffer(dev, "V", pts * G * Cin)
    dM = core._wino_buffer(dev, "M", pts * G * Cout)
    if G > rows:        # rows past the valid ones (stale from other layers) must not contribute
        V[:pts * G * Cin].view(pts, G, Cin)[:, rows:].zero_()
        dM[:pts * G * Cout].view(pts, G, Cout)[:, rows:].zero_()
    call("coocc_wino_input", ptr(x2d), x2d.shape[1], B, X, Y, Z, Cin, tile, ptr(V), G)
    call("coocc_wino_gradout", ptr(dacc), dacc.shape[1], B, X, Y, Z, Cout, tile, ptr(dM), G)
    key = (dev.index, pts * G, Z)
    if key not in _ztables:
        t = torch.empty(3, pts * G, dtype=torch.int32, device=dev)
        call("coocc_wino_ztap_table", pts * G, Z, ptr(t))
        _ztables[key] = t
    with _lib.TIMER.region("k_wgrad wino%d" % tile, 2.0 * pts * rows * 3 * Cin * Cout):
        call("coocc_wino_wgrad", ptr(V), ptr(dM), G, Z, Cin, Cout, tile, ptr(_ztables[key]), ptr(dw), 0, ptr(ws), ws.numel())
    return True


def _zrange(Zin, Zout, stride, pad):
    """First / last z tap of a 3-tap axis that reads a real voxel for at least one output (thin grids: Z = 1, 2)."""
    ok = [kz for kz in range(3) if any(0 <= zo * stride - pad + kz < Zin for zo in range(Zout))]
    return ok[0], ok[-1]


def _conv_launch(x2d, in_C, w_packed, out2d, Cout, taps, geom_in, geom_out, ksize, stride, pad, scale, shift, res2d, relu,
                 table=None, tag="conv_fwd", out_rows=None, kdims=None, h2_alpha=None, alpha_dev=None):
    """``h2_alpha``: x2d holds H2 rows (coocc_rows_to_h2, operand scale 1 / h2_alpha) and w_packed an H2 pack: split-f16 engine;
    ``alpha_dev``: device word the accumulators are also multiplied by (the inverse of a device-chosen operand scale)."""
    d = ConvDesc()
    ws = workspace(x2d.device)
    d.in_, d.w, d.out = ptr(x2d), ptr(w_packed), ptr(out2d)
    d.scale, d.bias = ptr(scale), ptr(shift)
    d.res = ptr(res2d)
    d.gather = ptr(table, torch.int32) if table is not None else None
    d.out_rows = ptr(out_rows, torch.int32) if out_rows is not None else None
    d.ws, d.ws_floats = ptr(ws), ws.numel()
    d.M, d.Cin, d.Cout, d.taps = (out_rows.shape[0] if out_rows is not None else out2d.shape[0]), in_C, Cout, taps
    d.in_stride, d.out_stride = x2d.shape[1], out2d.shape[1]
    d.res_stride = res2d.shape[1] if res2d is not None else 0
    B, Xi, Yi, Zi = geom_in
    _, Xo, Yo, Zo = geom_out
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = B, Xi, Yi, Zi, Xo, Yo, Zo
    d.ksize, d.stride, d.pad = ksize, stride, pad
    if kdims is not None:
        d.kx, d.ky, d.kz, d.px, d.py, d.pz = kdims
    d.relu, d.res_mode, d.splitk = int(relu), (1 if res2d is not None else 0), 0
    d.tile_hint = TILE_HINT
    if h2_alpha is not None:
        d.in_stride, d.mfma_dtype, d.alpha, tag = in_C, 3, float(h2_alpha), "k_gemm_h2 " + tag
        d.alpha_dev = alpha_dev
    with _lib.TIMER.region(tag, 2.0 * d.M * in_C * Cout * taps):
        _lib.conv_fwd(d, x2d.device)


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
        eff_shift = shift
        if bias is not None:    # y = scale * (conv + b) + shift
            eff_shift = (bias.detach() * scale if scale is not None else bias.detach()) + (shift if shift is not None else 0)
            eff_shift = eff_shift.float().contiguous()
        w_ = weight.detach().float().contiguous()
        if not (ksize == 3 and stride == 1 and pad == 1 and
                _wino_train(x2d, geom, w_.view(Cout, Cin, 3, 3, 3), False, out, scale, eff_shift, res2d, relu)):
            kd, wsub, nt = None, w_.reshape(Cout, Cin, taps), taps
            if ksize == 3:
                lo, hi = _zrange(Zi, Zo, stride, pad)
                if hi - lo < 2:         # z taps that only read padding are dropped (exact)
                    wsub = w_.view(Cout, Cin, 3, 3, 3)[..., lo:hi + 1].contiguous().view(Cout, Cin, -1)
                    nt, kd = wsub.shape[2], (3, 3, hi - lo + 1, pad, pad, pad - lo)
            if _h2_direct(Cin, 2.0 * out.shape[0] * Cin * Cout * nt):
                _conv_launch(_rows_h2(x2d, Cin), Cin, pack_weights_h2_dev(wsub, Cout, Cin, nt, 0), out, Cout, nt, geom, geom_out, ksize,
                             stride, pad, scale, eff_shift, res2d, relu, kdims=kd, h2_alpha=1.0)
            else:
                wp = pack_weights_dev(wsub, Cout, Cin, nt, 0)
                _conv_launch(x2d, Cin, wp, out, Cout, nt, geom, geom_out, ksize, stride, pad, scale, eff_shift, res2d, relu, kdims=kd)
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
        from . import core
        # split-f16 dgrad: max |dacc| collected by the pass that writes dacc -> {scale, 1 / scale} on the device (see TRAIN_H2_DGRAD)
        gscale = None
        if (core.CONV_ENGINE == "h2" and Cp == Cout and stride == 1 and Cin % 4 == 0 and
                ((TRAIN_H2_DGRAD and need_x and Cout % 32 == 0) or (TRAIN_H2_WGRAD and need_w))):
            gscale = torch.empty(2, device=dev, dtype=_F32)
        gscale_d = gscale if (TRAIN_H2_DGRAD and Cout % 32 == 0) else None      # the dgrad GEMM's K is Cout
        call("coocc_conv_epilogue_bwd_ex", ptr(dout), Cout, ptr(out), Cout, ptr(scale), Mo, Cout, int(relu), ptr(dacc), Cp,
             ptr(dres), Cout, 0, ptr(dbias), 0, ptr(ws), ws.numel(), ptr(_amax_word(dev)) if gscale is not None else None,
             ptr(gscale), TRAIN_H2_GRAD_TARGET)
        if dbias is not None and scale is not None:
            dbias = dbias * scale
        dx = dw = None
        if need_x:
            dx = torch.empty(Mi, Cin, device=dev, dtype=_F32)
            w3 = weight.reshape(Cout, Cin, taps)
            if (stride == 1 and ksize == 3 and pad == 1 and Cp == Cout and
                    _wino_train(dacc, geom, weight.detach().float().contiguous().view(Cout, Cin, 3, 3, 3), True, dx, None, None,
                                None, False, grad_scale=gscale_d)):
                pass
            elif stride == 1:
                kd, wsub, nt, pd = None, w3, taps, ksize - 1 - pad
                if ksize == 3:
                    lo, hi = _zrange(geom_out[3], Zi, 1, pd)        # z taps of the flipped kernel that see real dy voxels
                    if hi - lo < 2:
                        wsub = weight.detach().float().view(Cout, Cin, 3, 3, 3)[..., 2 - hi:2 - lo + 1].contiguous().view(Cout, Cin, -1)
                        nt, kd = wsub.shape[2], (3, 3, hi - lo + 1, pd, pd, pd - lo)
                if gscale_d is not None and _h2_direct(Cout, 2.0 * Mi * Cin * Cout * nt):
                    # the gradient operand scaled by gscale[0] (chosen on the device), undone by the GEMM through gscale[1]
                    _conv_launch(_rows_h2(dacc, Cout, gscale_d), Cout, pack_weights_h2_dev(wsub, Cout, Cin, nt, 2), dx, Cin,
                                 nt, geom_out, geom, ksize, 1, pd, None, None, None, False, tag="conv_dgrad", kdims=kd,
                                 h2_alpha=1.0, alpha_dev=_second_word(gscale_d))
                else:
                    wp = pack_weights_dev(wsub, Cout, Cin, nt, 2)
                    _conv_launch(dacc, Cp, wp, dx, Cin, nt, geom_out, geom, ksize, 1, pd, None, None, None, False,
                                 tag="conv_dgrad", kdims=kd)
            else:
                dx.zero_()          # voxels no output reads (and classes without taps) get a zero gradient
                for rows_c, taps_c, table_c in dgrad_classes(dev, B, Xi, Yi, Zi, ksize, stride, pad):
                    wp = pack_weights_dev(w3.detach().index_select(2, taps_c), Cout, Cin, taps_c.numel(), 3)
                    _conv_launch(dacc, Cp, wp, dx, Cin, taps_c.numel(), geom_out, geom, ksize, stride, pad, None, None, None,
                                 False, table=table_c, tag="conv_dgrad", out_rows=rows_c)
        if need_w:
            dw = torch.empty(Cout, Cin, taps, device=dev, dtype=_F32)
            if ksize == 3 and stride == 1 and pad == 1 and Cp == Cout and _wino_wgrad(x2d, dacc, geom, Cin, Cout, dw,
                                                                                        gscale if TRAIN_H2_WGRAD else None):
                pass
            elif (TRAIN_H2_WGRAD and gscale is not None and taps == 1 and stride == 1
                  and 2.0 * Mo * Cin * Cout >= core.H2_DIRECT_MIN_FLOPS):
                # 1x1x1 / Linear: both operands voxel-major (KH2), the gradient with its device-chosen scale
                Mp = -(-Mo // 16) * 16
                xk = _kh2(x2d, Mi, Mp, Cin, 1.0, None, torch.empty(Mp * Cin, device=dev, dtype=_F32))
                dk = _kh2(dacc, Mo, Mp, Cout, 1.0, gscale, torch.empty(Mp * Cout, device=dev, dtype=_F32))
                with _lib.TIMER.region("k_wgrad_h2", 2.0 * Mo * Cin * Cout):
                    call("coocc_conv_wgrad_h2", ptr(xk), ptr(dk), Mp, Cin, Cout, 1.0, _second_word(gscale), ptr(dw), 0, ptr(ws),
                         ws.numel())
            else:
                tb = tap_table(dev, B, Xi, Yi, Zi, ksize, stride, pad, False) if (taps > 1 or stride > 1) else None
                live = live_taps(dev, B, Xi, Yi, Zi, ksize, stride, pad) if tb is not None else None
                if live is not None and live[0].numel() < taps:
                    # thin grids (Z = 1, 2): taps that only ever read padding have a zero gradient -- skip their GEMMs
                    idx, tb_live = live
                    dw_live = torch.empty(Cout, Cin, idx.numel(), device=dev, dtype=_F32)
                    with _lib.TIMER.region("k_wgrad", 2.0 * Mo * Cin * Cout * idx.numel()):
                        call("coocc_conv_wgrad", ptr(x2d), Mi, Cin, ptr(dacc), Cp, ptr(tb_live), Mo, Cin, Cout, idx.numel(),
                             ptr(dw_live), 0, ptr(ws), ws.numel())
                    dw.zero_()
                    dw.index_copy_(2, idx, dw_live)
                else:
                    with _lib.TIMER.region("k_wgrad", 2.0 * Mo * Cin * Cout * taps):
                        call("coocc_conv_wgrad", ptr(x2d), Mi, Cin, ptr(dacc), Cp, ptr(tb), Mo, Cin, Cout, taps, ptr(dw), 0, ptr(ws),
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


# ----------------------------------------------------------------------------- G1 gather
class GatherRowsFn(torch.autograd.Function):
    """out[r] = src[idx[r]] (zeros where idx < 0); backward scatter-adds (bifuser_n.py:138-169 feature grads)."""

    @staticmethod
    def forward(ctx, src, idx):
        assert src.dim() == 2 and src.shape[1] % 4 == 0 and idx.dtype == torch.int32
        out = torch.empty(idx.numel(), src.shape[1], device=src.device, dtype=_F32)
        call("coocc_gather_rows", ptr(src.contiguous()), src.shape[1], ptr(idx.contiguous()), idx.numel(), src.shape[1],
             ptr(out), src.shape[1])
        ctx.save_for_backward(idx)
        ctx.nrows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        C = dout.shape[1]
        dsrc = torch.zeros(ctx.nrows, C, device=dout.device, dtype=_F32)
        call("coocc_scatter_add_rows", ptr(dout.float().contiguous()), C, ptr(idx.contiguous()), idx.numel(), C, ptr(dsrc), C)
        return dsrc, None


def gather_rows(src, idx):
    return GatherRowsFn.apply(src, idx)


class ScatterRowsFn(torch.autograd.Function):
    """out = zeros [nrows, C]; out[idx[r]] = vals[r] (idx unique, negative entries skipped); backward gathers."""

    @staticmethod
    def forward(ctx, vals, idx, nrows):
        assert vals.dim() == 2 and vals.shape[1] % 4 == 0 and idx.dtype == torch.int32
        C = vals.shape[1]
        out = torch.zeros(nrows, C, device=vals.device, dtype=_F32)
        call("coocc_scatter_add_rows", ptr(vals.float().contiguous()), C, ptr(idx.contiguous()), idx.numel(), C, ptr(out), C)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        C = dout.shape[1]
        dv = torch.empty(idx.numel(), C, device=dout.device, dtype=_F32)
        call("coocc_gather_rows", ptr(dout.float().contiguous()), C, ptr(idx.contiguous()), idx.numel(), C, ptr(dv), C)
        return dv, None, None


def fuser_fuse_train(fuser, img_rows, pts_rows, sr):
    """Differentiable G1 (bifuser_n.py:138-169) on rows: the neighbour tables of a ``BiFuser_N.search`` result are
    indices (not differentiated, as upstream); gradients reach both feature volumes and ``knn_enc``.
    img_rows / pts_rows: [V, C] channels-last rows of the two inputs -> [V, 4C] = (img | pts | fused_img | fused_pts).
    The gate product and the concat are torch elementwise / copy ops; gather, Linear and scatter are the HIP kernels."""
    V, C = img_rows.shape
    W, b = fuser.knn_enc[0].weight, fuser.knn_enc[0].bias

    def direction(src, rows, out_idx, gate_src):
        if rows is None or rows.shape[1] == 0:
            return torch.zeros(V, C, device=src.device, dtype=_F32)
        g = torch.cat([gather_rows(src, rows[k].contiguous()) for k in range(rows.shape[0])], 1)
        y = linear_rows(g, W, b, relu=True) * gather_rows(gate_src, out_idx)
        return ScatterRowsFn.apply(y, out_idx, V)
    fused_img = direction(img_rows, sr.rows, sr.lin_pts, pts_rows)
    fused_pts = direction(pts_rows, sr.rows_p, sr.lin_img, img_rows)
    return torch.cat([img_rows, pts_rows, fused_img, fused_pts], 1)


# ----------------------------------------------------------------------------- P2 pooling
class LiftSplatFn(torch.autograd.Function):
    """Fused lift (x) splat (view_transformer.lift_splat) with gradients for depth_prob and the context features."""

    @staticmethod
    def forward(ctx, vt, depth_prob, img_feat, geom_feats):
        from ._lib import host_f32
        from .ops import _pool_workspace
        BN, C, H, W = img_feat.shape
        D = depth_prob.shape[1]
        B = geom_feats.shape[0]
        X, Y, Z = (int(v) for v in vt.nx.tolist())
        dev = img_feat.device
        feat = torch.empty(BN * H * W, C, device=dev, dtype=_F32)
        call("coocc_ncdhw_to_ndhwc", ptr(img_feat.float().contiguous()), ptr(feat), BN, C, H * W, C, 0)
        lo = (vt.bx - vt.dx / 2.).tolist() + vt.dx.tolist()
        npts = BN * D * H * W
        out = torch.empty(B * X * Y * Z, C, device=dev, dtype=_F32)
        ws = _pool_workspace(dev, npts, B * X * Y * Z)
        dp = depth_prob.float().contiguous()
        g = geom_feats.reshape(-1, 3).float().contiguous()
        from .ops import pool_ws_clean, pool_ws_done
        call("coocc_lift_splat", ptr(dp), ptr(feat), ptr(g), BN, D, H, W, C, npts // B, host_f32(lo), B, X, Y, Z, ptr(out), C,
             ptr(ws), ws.numel(), pool_ws_clean(ws, npts, B * X * Y * Z))
        pool_ws_done(ws, npts, B * X * Y * Z)
        ctx.save_for_backward(dp, feat, g)
        ctx.cfg = (BN, D, H, W, C, npts // B, lo, B, X, Y, Z)
        return out.view(B, X, Y, Z, C).permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, dout):
        from ._lib import host_f32
        dp, feat, g = ctx.saved_tensors
        BN, D, H, W, C, ppb, lo, B, X, Y, Z = ctx.cfg
        dev = dp.device
        drows = dout.permute(0, 2, 3, 4, 1).reshape(-1, C).float().contiguous()
        d_depth = torch.empty_like(dp)
        d_feat = torch.empty_like(feat)
        call("coocc_lift_splat_bwd", ptr(drows), C, ptr(dp), ptr(feat), ptr(g), BN, D, H, W, C, ppb, host_f32(lo), B, X, Y, Z,
             ptr(d_depth), ptr(d_feat))
        d_img = torch.empty(BN, C, H, W, device=dev, dtype=_F32)
        call("coocc_ndhwc_to_ncdhw", ptr(d_feat), ptr(d_img), BN, C, H * W, C, 0)
        return None, d_depth, d_img, None


def lift_splat(vt, depth_prob, img_feat, geom_feats):
    return LiftSplatFn.apply(vt, depth_prob, img_feat, geom_feats)


class VoxelPoolFn(torch.autograd.Function):
    """voxel_pooling (ViewTransformerLSSVoxel.py:100-123) with the gradient of bev_pool_cuda.cu:61-84."""

    @staticmethod
    def forward(ctx, vt, geom_feats, x):
        out = vt.voxel_pooling(geom_feats, x)
        ctx.save_for_backward(geom_feats.reshape(-1, 3).float().contiguous())
        ctx.vt, ctx.shape = vt, tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        from ._lib import host_f32
        (g,) = ctx.saved_tensors
        vt = ctx.vt
        B, N, D, H, W, C = ctx.shape
        X, Y, Z = (int(v) for v in vt.nx.tolist())
        drows = dout.permute(0, 2, 3, 4, 1).reshape(-1, C).float().contiguous()
        npts = B * N * D * H * W
        dx = torch.empty(npts, C, device=dout.device, dtype=_F32)
        lo = (vt.bx - vt.dx / 2.).tolist() + vt.dx.tolist()
        call("coocc_voxel_pool_bwd", ptr(drows), C, ptr(g), npts, npts // B, C, host_f32(lo), B, X, Y, Z, ptr(dx))
        return None, None, dx.view(B, N, D, H, W, C)


def voxel_pooling(vt, geom_feats, x):
    return VoxelPoolFn.apply(vt, geom_feats, x)


# ----------------------------------------------------------------------------- R2 render
class RenderNearestFn(torch.autograd.Function):
    """table [V,4] (sigma, rgb logits per voxel) -> composited maps [N,H,W,4] (coocc_ray.py:574-617)."""

    @staticmethod
    def forward(ctx, table, gemo, grid):
        from ._lib import host_f32
        from .render import RENDER_BOUNDS
        X, Y, Z = grid
        N, D, H, W, _ = gemo.shape
        zvals = torch.linspace(0, D, D, device=table.device)
        maps = torch.empty(N, H, W, 4, device=table.device, dtype=_F32)
        t = table.float().contiguous()
        call("coocc_render_nearest", ptr(t), X, Y, Z, ptr(gemo), ptr(zvals), N, D, H, W, host_f32(RENDER_BOUNDS), 0, ptr(maps))
        ctx.save_for_backward(t, gemo, zvals)
        ctx.grid = grid
        return maps

    @staticmethod
    def backward(ctx, dmaps):
        from ._lib import host_f32
        from .render import RENDER_BOUNDS
        t, gemo, zvals = ctx.saved_tensors
        X, Y, Z = ctx.grid
        N, D, H, W, _ = gemo.shape
        dtable = torch.empty_like(t)
        call("coocc_render_nearest_bwd", ptr(t), X, Y, Z, ptr(gemo), ptr(zvals), N, D, H, W, host_f32(RENDER_BOUNDS),
             ptr(dmaps.float().contiguous()), ptr(dtable))
        return dtable, None, None


class UpsampleMapsFn(torch.autograd.Function):
    """maps [N,H,W,4] -> (rgbs [N,sH,sW,3], depths [N,sH,sW]) (coocc_ray.py:619-625)."""

    @staticmethod
    def forward(ctx, maps, scale):
        N, H, W, _ = maps.shape
        rgbs = torch.empty(N, H * scale, W * scale, 3, device=maps.device, dtype=_F32)
        depths = torch.empty(N, H * scale, W * scale, device=maps.device, dtype=_F32)
        call("coocc_upsample_maps", ptr(maps.contiguous()), N, H, W, scale, ptr(rgbs), ptr(depths))
        ctx.cfg = (N, H, W, scale)
        return rgbs, depths

    @staticmethod
    def backward(ctx, drgbs, ddepths):
        N, H, W, scale = ctx.cfg
        dmaps = torch.empty(N, H, W, 4, device=drgbs.device if drgbs is not None else ddepths.device, dtype=_F32)
        call("coocc_upsample_maps_bwd", ptr(drgbs.float().contiguous()) if drgbs is not None else None,
             ptr(ddepths.float().contiguous()) if ddepths is not None else None, N, H, W, scale, ptr(dmaps))
        return dmaps, None


class RenderLossesFn(torch.autograd.Function):
    """-> tensor [2] = (loss_depth_render, loss_rgb) (coocc_ray.py:423-433)."""

    @staticmethod
    def forward(ctx, rgbs, depths, rgb_gt, depth_gt, D):
        out = torch.empty(3, device=rgbs.device, dtype=_F32)
        r, d = rgbs.contiguous(), depths.contiguous()
        rg, dg = rgb_gt.float().contiguous(), depth_gt.float().contiguous()
        ws = workspace(r.device)
        call("coocc_render_losses", ptr(r), ptr(d), ptr(rg), ptr(dg), d.numel(), int(D), ptr(out), ptr(ws), ws.numel() * 4)
        ctx.save_for_backward(r, d, rg, dg, out)
        ctx.D = int(D)
        return out[:2].clone()

    @staticmethod
    def backward(ctx, gl):
        r, d, rg, dg, out = ctx.saved_tensors
        drgbs, ddepths = torch.empty_like(r), torch.empty_like(d)
        call("coocc_render_losses_bwd", ptr(r), ptr(d), ptr(rg), ptr(dg), d.numel(), ctx.D, ptr(out), ptr(gl.float().contiguous()),
             ptr(drgbs), ptr(ddepths))
        return drgbs, ddepths, None, None, None


def render_block_train(sigma_head, rgb_head, feats2d, grid, gemo, scale=16):
    """Differentiable render block: feats2d [X*Y*Z, C] rows -> (rgbs, depths).  The per-voxel heads run through
    ConvRowsFn (Linear+ReLU layers), so gradients reach the voxel features and both MLPs.  ``rgb_head=None``: the
    depth-only branch (coocc_ray.py:436-484) -- the colour columns of the table are zero and ``rgbs`` is meaningless."""
    def mlp(m, x):
        for l in m.hidden_layers:
            x = linear_rows(x, l.weight, l.bias, relu=True)
        return linear_rows(x, m.output_layer.weight, m.output_layer.bias, relu=False)
    sig = mlp(sigma_head, feats2d)                       # [V,1]
    rgb = mlp(rgb_head, feats2d) if rgb_head is not None else torch.zeros(sig.shape[0], 3, device=sig.device)   # [V,3]
    table = torch.cat([sig, rgb], 1)                     # plumbing: 16 B per voxel
    N, D, H, W = gemo.shape[-5:-1]
    maps = RenderNearestFn.apply(table, gemo.reshape(N, D, H, W, 3).float().contiguous(), tuple(grid))
    return UpsampleMapsFn.apply(maps, scale)


def render_losses(rgbs, depths, rgb_gt, depth_gt, D):
    out = RenderLossesFn.apply(rgbs, depths, rgb_gt, depth_gt, D)
    return dict(loss_depth_render=out[0], loss_rgb=out[1])


# ----------------------------------------------------------------------------- FPN top-down step
class UpsampleAddFn(torch.autograd.Function):
    """fine + trilinear(coarse -> fine size) on rows (fpn3d.py:88-92)."""

    @staticmethod
    def forward(ctx, coarse2d, fine2d, gc, gf):
        B, Xc, Yc, Zc = gc
        _, Xf, Yf, Zf = gf
        C = fine2d.shape[1]
        out = fine2d.clone()
        call("coocc_upsample_add_trilinear", ptr(coarse2d.contiguous()), ptr(out), B, C, Xc, Yc, Zc, Xf, Yf, Zf)
        ctx.cfg = (B, C, Xc, Yc, Zc, Xf, Yf, Zf)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, Xc, Yc, Zc, Xf, Yf, Zf = ctx.cfg
        dout = dout.float().contiguous()
        dc = torch.empty(B * Xc * Yc * Zc, C, device=dout.device, dtype=_F32)
        call("coocc_upsample_trilinear_bwd", ptr(dout), ptr(dc), B, C, Xc, Yc, Zc, Xf, Yf, Zf, 0)
        return dc, dout, None, None


def upsample_add(coarse2d, fine2d, gc, gf):
    return UpsampleAddFn.apply(coarse2d, fine2d, tuple(gc), tuple(gf))


# ----------------------------------------------------------------------------- decoder trunk (C0 + C1 + C2)
def rows_from_ncdhw(x):
    """[B,C,X,Y,Z] -> ([B*X*Y*Z, C] contiguous rows, (B,X,Y,Z)); differentiable (torch permute / copy)."""
    B, C, X, Y, Z = x.shape
    return x.float().permute(0, 2, 3, 4, 1).reshape(B * X * Y * Z, C).contiguous(), (B, X, Y, Z)


def ncdhw_from_rows(rows, geom):
    B, X, Y, Z = geom
    return rows.view(B, X, Y, Z, rows.shape[1]).permute(0, 4, 1, 2, 3)


def conv_bn_rows(x2d, weight, geom, bn=None, bias=None, stride=1, pad=None, relu=True, res2d=None):
    """Conv3d -> BatchNorm (+res) -> ReLU on rows with the norm layer's OWN mode: a BN in training mode normalises with
    batch (or SyncBN all-reduced) statistics and updates its running stats, as nn.BatchNorm3d / nn.SyncBatchNorm do under
    model.train() upstream; a BN in eval mode is folded into the GEMM epilogue (frozen statistics)."""
    if bn is not None and bn.training:
        return conv3d_bn_train_rows(x2d, weight, geom, bn, stride=stride, pad=pad, relu=relu, res2d=res2d, bias=bias)
    return conv3d_rows(x2d, weight, geom, bias=bias, bn=bn, stride=stride, pad=pad, relu=relu, res2d=res2d)


def _cm(x, geom, m, relu=True):
    """mmcv-style ConvModule (conv, bn) on rows."""
    return conv_bn_rows(x, m.conv.weight, geom, bias=m.conv.bias, bn=m.bn, relu=relu)


def con_enc_train(con_enc, x2d, geom):
    """bifuser_n.py:23-30: Sequential(Conv3d, BN, ReLU, Conv3d, BN, ReLU) on rows."""
    x2d, geom = conv_bn_rows(x2d, con_enc[0].weight, geom, bias=con_enc[0].bias, bn=con_enc[1], relu=True)
    return conv_bn_rows(x2d, con_enc[3].weight, geom, bias=con_enc[3].bias, bn=con_enc[4], relu=True)


def backbone_forward_train(backbone, x2d, geom):
    """CustomResNet3D (resnet3d.py:196-205) on rows -> [(rows, geom)] per out index."""
    x, g = conv_bn_rows(x2d, backbone.input_proj[0].weight, geom, bn=backbone.input_proj[1], relu=True)
    feats = []
    for i, layer in enumerate(backbone.layers):
        for blk in layer:
            out, go = conv_bn_rows(x, blk.conv1.weight, g, bn=blk.bn1, stride=blk.stride, relu=True)
            if blk.downsample is not None:
                res, _ = conv_bn_rows(x, blk.downsample[0].weight, g, bn=blk.downsample[1], stride=blk.stride, pad=0, relu=False)
            else:
                res = x
            x, g = conv_bn_rows(out, blk.conv2.weight, go, bn=blk.bn2, relu=True, res2d=res)
        if i in backbone.out_indices:
            feats.append((x, g))
    return feats


def neck_forward_train(neck, feats):
    """FPN3D (fpn3d.py:70-108) on [(rows, geom)]."""
    lat = [_cm(x, g, neck.lateral_convs[i][0]) for i, (x, g) in enumerate(feats)]
    for i in range(len(lat) - 1, 0, -1):
        (c, gc), (f, gf) = lat[i], lat[i - 1]
        lat[i - 1] = (upsample_add(c, f, gc, gf), gf)
    return [_cm(x, g, neck.fpn_convs[i][0]) for i, (x, g) in enumerate(lat)]


def trunk_forward_train(con_enc, backbone, neck, x2d, geom):
    """Differentiable con_enc -> CustomResNet3D -> FPN3D on rows (bifuser_n.py:23-30, resnet3d.py:196-205,
    fpn3d.py:70-108); every BN follows its own training flag (conv_bn_rows).  x2d: [B*X*Y*Z, 4C] fused rows.  Returns
    [(rows, geom)] per level.  The modules are the inference modules: parameters (and state_dict keys) are shared."""
    if con_enc is not None:
        x2d, geom = con_enc_train(con_enc, x2d, geom)
    feats = backbone_forward_train(backbone, x2d, geom)
    return feats if neck is None else neck_forward_train(neck, feats)


# ----------------------------------------------------------------------------- OccHead coarse mix (C3)
class OccHeadMixFn(torch.autograd.Function):
    """out[v] = sum_l softmax(wlogit[v])_l * trilinear(level_l -> level-0 size)[v]   (occ_head.py:155-166)."""

    @staticmethod
    def forward(ctx, wlogit, geoms, *levels):
        from ._lib import host_i32
        L = len(levels)
        B, X0, Y0, Z0 = geoms[0]
        C = levels[0].shape[1]
        lv = [t.contiguous() for t in levels]
        arr = (_lib.c_void_p * L)(*[t.data_ptr() for t in lv])
        dims = host_i32([v for g in geoms for v in g[1:]])
        out = torch.empty(B * X0 * Y0 * Z0, C, device=lv[0].device, dtype=_F32)
        wl = wlogit.contiguous() if wlogit is not None else None
        call("coocc_occhead_mix", arr, dims, L, ptr(wl), ptr(out), B, C)
        ctx.save_for_backward(*( [wl] if wl is not None else [] ), *lv)
        ctx.cfg = (geoms, wl is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        from ._lib import host_i32
        geoms, has_w = ctx.cfg
        saved = ctx.saved_tensors
        wl = saved[0] if has_w else None
        lv = saved[1:] if has_w else saved
        L = len(lv)
        B, X0, Y0, Z0 = geoms[0]
        C = lv[0].shape[1]
        dev = dout.device
        dout = dout.float().contiguous()
        V0 = B * X0 * Y0 * Z0
        g = [torch.empty(V0, C, device=dev, dtype=_F32) for _ in range(L)]
        dwl = torch.empty(V0, L, device=dev, dtype=_F32) if has_w else None
        arr = (_lib.c_void_p * L)(*[t.data_ptr() for t in lv])
        garr = (_lib.c_void_p * L)(*[t.data_ptr() for t in g])
        dims = host_i32([v for gg in geoms for v in gg[1:]])
        call("coocc_occhead_mix_bwd", arr, dims, L, ptr(wl), ptr(dout), garr, ptr(dwl), B, C)
        grads = [g[0]]
        for l in range(1, L):
            _, Xl, Yl, Zl = geoms[l]
            if (Xl, Yl, Zl) == (X0, Y0, Z0):
                grads.append(g[l])
                continue
            dl = torch.empty(B * Xl * Yl * Zl, C, device=dev, dtype=_F32)
            call("coocc_upsample_trilinear_bwd", ptr(g[l]), ptr(dl), B, C, Xl, Yl, Zl, X0, Y0, Z0, 0)
            grads.append(dl)
        return (dwl, None) + tuple(grads)


def occhead_coarse_train(head, feats):
    """Differentiable OccHead.forward_coarse_voxel on rows: feats = [(rows, geom)] per level (from
    trunk_forward_train) -> (out_voxel_feats rows, occupancy logits rows [V0, num_cls])."""
    occs = []
    for i, (x, g) in enumerate(feats):
        m = head.occ_convs[i]
        o, go = conv_bn_rows(x, m[0].weight, g, bias=m[0].bias, bn=m[1], relu=True)
        occs.append((o, go))
    wlogit = None
    g0 = occs[0][1]
    if head.soft_weights:
        sw = head.voxel_soft_weights
        h, _ = conv_bn_rows(occs[0][0], sw[0].weight, g0, bias=sw[0].bias, bn=sw[1], relu=True)
        wlogit, _ = conv_bn_rows(h, sw[3].weight, g0, bias=sw[3].bias, relu=False)
    out = OccHeadMixFn.apply(wlogit, tuple(g for _, g in occs), *[o for o, _ in occs])
    pc = head.occ_pred_conv
    h, _ = conv_bn_rows(out, pc[0].weight, g0, bias=pc[0].bias, bn=pc[1], relu=True)
    occ, _ = conv_bn_rows(h, pc[3].weight, g0, bias=pc[3].bias, relu=False)
    return out, occ


# ----------------------------------------------------------------------------- fine branch (C4)
class FineSampleVoxelFn(torch.autograd.Function):
    """Trilinear grid_sample of out_voxel_feats at the fine children of the selected coarse voxels (occ_head.py:205-214).
    vol2d [X*Y*Z, C] rows, coarse_lin int32 [n] -> (feat [r^3*n, C], fine_xyz int64 [3, r^3*n])."""

    @staticmethod
    def forward(ctx, vol2d, coarse_lin, geom, ratio, final_size):
        from ._lib import host_i32
        _, X, Y, Z = geom
        C = vol2d.shape[1]
        n = coarse_lin.numel()
        nf = n * ratio ** 3
        fine_xyz = torch.empty(3, nf, device=vol2d.device, dtype=torch.int64)
        feat = torch.empty(nf, C, device=vol2d.device, dtype=_F32)
        call("coocc_fine_sample_voxel", ptr(vol2d.contiguous()), C, X, Y, Z, ptr(coarse_lin), n, ratio, host_i32(final_size),
             ptr(fine_xyz), ptr(feat), C)
        ctx.save_for_backward(fine_xyz)
        ctx.cfg = (X, Y, Z, C, tuple(final_size))
        ctx.mark_non_differentiable(fine_xyz)
        return feat, fine_xyz

    @staticmethod
    def backward(ctx, dfeat, _):
        from ._lib import host_i32
        (fine_xyz,) = ctx.saved_tensors
        X, Y, Z, C, final_size = ctx.cfg
        dvol = torch.empty(X * Y * Z, C, device=dfeat.device, dtype=_F32)
        call("coocc_fine_sample_voxel_bwd", ptr(dfeat.float().contiguous()), C, C, X, Y, Z, ptr(fine_xyz), fine_xyz.shape[1],
             host_i32(final_size), ptr(dvol))
        return dvol, None, None, None, None


class GroupNormRowsFn(torch.autograd.Function):
    """nn.GroupNorm on rows [n, C] followed by ReLU (occ_head.py:70-83), differentiable in x, gamma, beta."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        y = x.float().contiguous().clone()
        n, C = y.shape
        call("coocc_groupnorm_rows", ptr(y), n, C, C, groups, ptr(gamma.detach().float().contiguous()),
             ptr(beta.detach().float().contiguous()), float(eps), int(relu))
        ctx.save_for_backward(x.float().contiguous(), y, gamma.detach().float().contiguous())
        ctx.cfg = (groups, float(eps), int(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma = ctx.saved_tensors
        groups, eps, relu = ctx.cfg
        n, C = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, device=x.device, dtype=_F32)
        dbeta = torch.empty(C, device=x.device, dtype=_F32)
        call("coocc_groupnorm_rows_bwd", ptr(x), ptr(y), ptr(dy.float().contiguous()), n, C, C, groups, ptr(gamma), eps, relu,
             ptr(dx), ptr(dgamma), ptr(dbeta))
        return dx, dgamma, dbeta, None, None, None


class FineSampleImgFn(torch.autograd.Function):
    """Camera branch of C4: project fine points into the cameras and sum the bilinear samples of img rows
    [ncam*Hf*Wf, Ci] (occ_head.py:217-234); gradient w.r.t. the image features."""

    @staticmethod
    def forward(ctx, img_rows, params, fine_xyz, ncam, Hf, Wf, ratio):
        Ci = img_rows.shape[1]
        nf = fine_xyz.shape[1]
        samp = torch.empty(nf, Ci, device=img_rows.device, dtype=_F32)
        call("coocc_fine_sample_img", ptr(img_rows.contiguous()), ncam, Ci, Hf, Wf, ptr(params), ptr(fine_xyz), nf, ptr(samp), Ci,
             ratio if ratio in (2, 4) else 0)
        ctx.save_for_backward(params, fine_xyz)
        ctx.cfg = (ncam, Ci, Hf, Wf)
        return samp

    @staticmethod
    def backward(ctx, dsamp):
        params, fine_xyz = ctx.saved_tensors
        ncam, Ci, Hf, Wf = ctx.cfg
        dimg = torch.empty(ncam * Hf * Wf, Ci, device=dsamp.device, dtype=_F32)
        call("coocc_fine_sample_img_bwd", ptr(dsamp.float().contiguous()), Ci, ncam, Ci, Hf, Wf, ptr(params), ptr(fine_xyz),
             fine_xyz.shape[1], ptr(dimg))
        return dimg, None, None, None, None, None, None


class GroupNormNHWCFn(torch.autograd.Function):
    """nn.GroupNorm over [N, HW, C] image rows followed by ReLU (occ_head.py:64-68)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, N, HW, groups, eps, relu):
        y = x.float().contiguous().clone()
        C = y.shape[1]
        call("coocc_groupnorm_nhwc", ptr(y), N, HW, C, groups, ptr(gamma.detach().float().contiguous()),
             ptr(beta.detach().float().contiguous()), float(eps), int(relu))
        ctx.save_for_backward(x.float().contiguous(), y, gamma.detach().float().contiguous())
        ctx.cfg = (N, HW, groups, float(eps), int(relu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma = ctx.saved_tensors
        N, HW, groups, eps, relu = ctx.cfg
        C = x.shape[1]
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, device=x.device, dtype=_F32)
        dbeta = torch.empty(C, device=x.device, dtype=_F32)
        call("coocc_groupnorm_nhwc_bwd", ptr(x), ptr(y), ptr(dy.float().contiguous()), N, HW, C, groups, ptr(gamma), eps, relu,
             ptr(dx), ptr(dgamma), ptr(dbeta))
        return dx, dgamma, dbeta, None, None, None, None, None


def fine_branch_train(head, out_voxel_rows, geom, coarse_lin, img_feats=None, transform=None):
    """Differentiable fine branch (occ_head.py:180-237): the selected coarse voxels' children sample out_voxel_feats
    trilinearly and, with sample_from_img, the image features through the cameras; fine_mlp = Linear -> GroupNorm ->
    ReLU -> Linear.  The selection itself (argmax != empty, or the training-time random top-k) is an index and not
    differentiated.  img_feats: [[1,N,Cimg,Hf,Wf]] (a leaf may require grad); transform: img_inputs[1:]."""
    r = head.cascade_ratio
    feat, fine_xyz = FineSampleVoxelFn.apply(out_voxel_rows, coarse_lin, tuple(geom), r,
                                             tuple(int(v) for v in head.final_occ_size))
    parts = [feat] if head.sample_from_voxel else []
    if head.sample_from_img and img_feats is not None:
        f = img_feats[0]
        _, N_i, C_i, Hf, Wf = f.shape
        rows = f[0].permute(0, 2, 3, 1).reshape(N_i * Hf * Wf, C_i)            # NHWC rows (a view + copy: plumbing)
        m0 = head.img_mlp_0
        g = linear_rows(rows, m0[0].weight.flatten(1), m0[0].bias, relu=False)
        g = GroupNormNHWCFn.apply(g, m0[1].weight, m0[1].bias, N_i, Hf * Wf, m0[1].num_groups, m0[1].eps, True)

        class _G:       # geometry holder for the parameter block
            X, Y, Z = geom[1], geom[2], geom[3]
        params = head._projection_params(transform, _G, out_voxel_rows.device)
        samp = FineSampleImgFn.apply(g, params, fine_xyz, N_i, Hf, Wf, r)
        m1 = head.img_mlp
        s_ = linear_rows(samp, m1[0].weight, m1[0].bias, relu=False)
        parts.append(GroupNormRowsFn.apply(s_, m1[1].weight, m1[1].bias, m1[1].num_groups, m1[1].eps, True))
    x = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
    h = linear_rows(x, head.fine_mlp[0].weight, head.fine_mlp[0].bias, relu=False)
    gn = head.fine_mlp[1]
    h = GroupNormRowsFn.apply(h, gn.weight, gn.bias, gn.num_groups, gn.eps, True)
    return linear_rows(h, head.fine_mlp[3].weight, head.fine_mlp[3].bias, relu=False), fine_xyz


# ----------------------------------------------------------------------------- BatchNorm with batch statistics
def _sync_group(bn, sync):
    """The process group a training-mode BN synchronises over, or None.  sync=None: follow the module type
    (nn.SyncBatchNorm, what norm_cfg type='SyncBN' builds upstream) whenever torch.distributed is initialised."""
    import torch.distributed as dist
    if sync is None:
        sync = isinstance(bn, torch.nn.SyncBatchNorm)
    if not sync or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    return getattr(bn, "process_group", None) or dist.group.WORLD


class BatchNormRowsFn(torch.autograd.Function):
    """Training-mode BatchNorm3d / SyncBN on rows [M, C] (+ residual, ReLU): batch mean / biased variance, running
    statistics updated in place like torch (momentum, unbiased variance).  With a process group the statistics are
    those of all ranks' rows (one all-reduce of 2C+1 floats in the forward, one of 2C in the backward, as
    torch.nn.SyncBatchNorm does); the returned dgamma / dbeta are this rank's share, which DDP then reduces."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, bn, relu, group=None):
        x = x.float().contiguous()
        M, C = x.shape
        dev = x.device
        mean = torch.empty(C, device=dev, dtype=_F32)
        var = torch.empty(C, device=dev, dtype=_F32)
        ws = workspace(dev)
        call("coocc_bn_stats", ptr(x), C, M, C, ptr(mean), ptr(var), ptr(ws), ws.numel() * 4)
        count = float(M)
        if group is not None:
            import torch.distributed as dist
            m64 = mean.double()
            pack = torch.cat([m64 * M, (var.double() + m64 * m64) * M, torch.tensor([float(M)], device=dev, dtype=torch.float64)])
            dist.all_reduce(pack, group=group)
            count = float(pack[-1].item())
            gm = pack[:C] / count
            mean = gm.float()
            var = (pack[C:2 * C] / count - gm * gm).clamp_(min=0).float()
        y = torch.empty_like(x)
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        call("coocc_bn_apply", ptr(x), M, C, ptr(mean), ptr(var), ptr(g), ptr(b), float(bn.eps),
             ptr(res.float().contiguous()) if res is not None else None, int(relu), ptr(y))
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked += 1          # torch increments first; momentum=None = cumulative moving average
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (count / max(count - 1, 1)), alpha=mom)
        ctx.save_for_backward(x, y, mean, var, g)
        ctx.cfg = (float(bn.eps), int(relu), res is not None, group, count)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, var, g = ctx.saved_tensors
        eps, relu, has_res, group, count = ctx.cfg
        M, C = x.shape
        dev = x.device
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        sums = torch.empty(2, C, device=dev, dtype=_F32)        # [dgamma | dbeta] of this rank's rows
        ws = workspace(dev)
        call("coocc_bn_backward_sums", ptr(x), ptr(y), ptr(dy), M, C, ptr(mean), ptr(var), eps, relu, ptr(sums[0]), ptr(sums[1]),
             ptr(ws), ws.numel() * 4)
        tot = sums
        if group is not None:
            import torch.distributed as dist
            tot = sums.clone()
            dist.all_reduce(tot, group=group)
        call("coocc_bn_backward_dx", ptr(x), ptr(y), ptr(dy), M, C, ptr(mean), ptr(var), ptr(g), eps, relu, ptr(tot[0]), ptr(tot[1]),
             float(count), ptr(dx), ptr(dres))
        return dx, sums[0], sums[1], dres, None, None, None


def conv3d_bn_train_rows(x2d, weight, geom, bn, stride=1, pad=None, relu=True, res2d=None, bias=None, sync=None):
    """Training-mode Conv3d -> BatchNorm (batch statistics) (+res) -> ReLU on rows: the conv runs without epilogue,
    BatchNormRowsFn normalises.  Use conv3d_rows(bn=...) for frozen statistics."""
    y, g = conv3d_rows(x2d, weight, geom, bias=bias, bn=None, stride=stride, pad=pad, relu=False)
    return BatchNormRowsFn.apply(y, bn.weight, bn.bias, res2d, bn, relu, _sync_group(bn, sync)), g
