"""The split-f16 engine's round-4 plumbing (csrc/gemm_h2.hip, csrc/h2_rows.h):

* H2 twins written by producer epilogues (GEMM, Winograd output transform, trilinear upsample-add, OccHead mix) are
  bit-identical to a conversion pass over the fp32 rows they accompany -- so the consumer computes the same bits whether its
  operand came from the producer or from ``coocc_rows_to_h2``;
* the in-kernel split-K reduction (arrival counters, last workgroup sums the slabs in slice order; ``COOCC_INKERNEL_REDUCE=1`` --
  measured slower than the two-launch form and off by default) is bit-identical to the two-launch form
  (``k_conv_reduce``, which writes the H2 twin itself) and leaves its counters at zero;
* the f16 range guard: activations that leave the operand range raise instead of silently producing inf / NaN, and the
  documented escape hatch (``COOCC_CONV_ENGINE=f32``) computes the same layer correctly.
"""
import pytest
import torch
import torch.nn.functional as F

from co_occ_amd import _lib, core
from co_occ_amd._lib import call, ptr
from test_gpu_conv import bn_like, rows_of
from util import assert_close

pytestmark = pytest.mark.gpu


def h2_of(t):
    """Conversion pass over fp32 rows [n, C] -> H2 rows (as an int32 view for exact comparison)."""
    out = torch.empty_like(t)
    call("coocc_rows_to_h2", ptr(t), t.shape[1], t.shape[0], t.shape[1], 1.0, ptr(out))
    return out.view(torch.int32)


def consumer(C, dev, ksize=1, stride=1):
    g = torch.Generator().manual_seed(C)
    w = torch.randn(64, C, ksize, ksize, ksize, generator=g) * 0.05
    return core.PackedConv(w.to(dev), ksize=ksize, stride=stride, pad=ksize // 2)


@pytest.mark.parametrize("Cin,Cout,grid,k,stride,use_res", [
    (64, 128, (40, 40, 8), 3, 1, True),       # Winograd path: the output transform writes the twin
    (128, 256, (50, 50, 4), 3, 2, False),     # strided direct split-f16 GEMM (k_gemm_h2w), split-K -> in-kernel reduce + twin
    (256, 256, (25, 25, 2), 3, 1, True),      # stride-1 small grid (k_gemm_h2z direct), split-K, residual
    (128, 128, (30, 30, 8), 1, 1, False),     # 1x1x1
])
def test_producer_h2_twin_equals_conversion_pass(dev, monkeypatch, Cin, Cout, grid, k, stride, use_res):
    monkeypatch.setattr(core, "H2_DIRECT_MIN_FLOPS", 0.0)
    g = torch.Generator().manual_seed(Cin + Cout + k)
    X, Y, Z = grid
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
    bn = bn_like(Cout, g)
    ref = bn(F.conv3d(x, w, stride=stride, padding=k // 2))
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
    ref = F.relu(ref).detach()
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=k, stride=stride, pad=k // 2)      # (moves bn to the device)
    nxt = consumer(Cout, dev)
    assert core.route(1, *ref.shape[2:], nxt) == "h2"
    rr = rows_of(res, dev) if use_res else None
    out = core.conv_rows(rows_of(x, dev), pc, relu=True, res=rr, twin_for=(nxt,))
    assert out.h2 is not None, "the producer did not write the twin its split-f16 consumer asked for"
    assert torch.equal(out.h2.view(torch.int32), h2_of(out.t))
    assert_close(out.as_ncdhw().cpu(), ref, what="producer with twin")
    plain = core.conv_rows(rows_of(x, dev), pc, relu=True, res=rr)
    assert plain.h2 is None and torch.equal(plain.t, out.t)
    # the consumer's result does not depend on where its operand came from
    a = core.conv_rows(out, nxt, relu=False)
    b = core.conv_rows(plain, nxt, relu=False)
    assert plain.h2 is not None            # ... and the conversion it had to run is cached on the Rows for the next reader
    assert torch.equal(a.t, b.t)
    core.check_h2_overflow()


def test_twin_survives_the_ncdhw_view_and_dies_with_an_inplace_write(dev, monkeypatch):
    """Modules hand each other zero-copy [B,C,X,Y,Z] views; ``to_rows`` finds the Rows (and twin) behind an untouched view and
    refuses it once torch has written to the tensor in place."""
    monkeypatch.setattr(core, "H2_DIRECT_MIN_FLOPS", 0.0)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 64, 24, 24, 4, generator=g)
    pc = core.PackedConv((torch.randn(64, 64, 1, 1, 1, generator=g) * 0.1).to(dev), ksize=1)
    nxt = consumer(64, dev)
    out = core.conv_rows(rows_of(x, dev), pc, relu=True, twin_for=(nxt,))
    v = out.as_ncdhw()
    assert core.to_rows(v) is out
    v.mul_(2.0)
    r2 = core.to_rows(v)
    assert r2 is not out and r2.h2 is None
    assert torch.equal(core.h2_rows(r2).view(torch.int32), h2_of(r2.t))


@pytest.mark.parametrize("Cin,Cout,grid,k,stride", [(256, 512, (25, 25, 2), 3, 2), (512, 512, (13, 13, 1), 3, 1), (256, 256, (50, 50, 4), 1, 1),
                                                      (128, 256, (50, 50, 4), 3, 2)])
def test_inkernel_splitk_reduction_equals_two_launch_form(dev, monkeypatch, Cin, Cout, grid, k, stride):
    monkeypatch.setattr(core, "H2_DIRECT_MIN_FLOPS", 0.0)
    g = torch.Generator().manual_seed(Cin + Cout)
    X, Y, Z = grid
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
    bn = bn_like(Cout, g)
    ref = F.relu(bn(F.conv3d(x, w, stride=stride, padding=k // 2))).detach()
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=k, stride=stride, pad=k // 2)
    xr = rows_of(x, dev)
    outs = []
    for inkernel in (True, False, True):
        monkeypatch.setattr(core, "INKERNEL_REDUCE", inkernel)
        outs.append(core.conv_rows(xr, pc, relu=True).t.clone())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert int(core.tile_sem(dev).abs().sum()) == 0, "arrival counters must be left at zero"
    assert_close(core.Rows(outs[0], 1, *ref.shape[2:], Cout).as_ncdhw().cpu(), ref, what="in-kernel split-K")


def test_f16_range_guard_raises_and_the_f32_engine_is_the_way_out(dev, monkeypatch):
    """Activations x 1e4 through an F(4x4) Winograd layer leave the f16 operand range (the transform amplifies by up to 100 at
    scale 1/8): the guard must flag it; the same layer under COOCC_CONV_ENGINE=f32 is computed correctly."""
    g = torch.Generator().manual_seed(1)
    Cin, Cout, (X, Y, Z) = 64, 128, (40, 40, 8)
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    pc = core.PackedConv(w.to(dev), ksize=3, pad=1)
    assert core.CONV_ENGINE == "h2"
    core.check_h2_overflow()                      # clean start
    out = core.conv_rows(rows_of(x, dev), pc, relu=False)
    torch.cuda.synchronize()
    core.check_h2_overflow()                      # ordinary activations: no flag
    assert torch.isfinite(out.t).all()
    big = x * 3.0e4
    core.conv_rows(rows_of(big, dev), pc, relu=False)
    torch.cuda.synchronize()
    with pytest.raises(_lib.CooccError, match="COOCC_CONV_ENGINE=f32"):
        core.check_h2_overflow()
    core.check_h2_overflow()                      # the check cleared the flag
    # the direct (non-Winograd) writers: conversion pass and a GEMM epilogue's twin
    t = torch.full((256, 64), 7.0e4, device=dev)
    h2_of(t)
    torch.cuda.synchronize()
    with pytest.raises(_lib.CooccError):
        core.check_h2_overflow()
    monkeypatch.setattr(core, "CONV_ENGINE", "f32")
    pc32 = core.PackedConv(w.to(dev), ksize=3, pad=1)
    out32 = core.conv_rows(rows_of(big, dev), pc32, relu=False)
    torch.cuda.synchronize()
    core.check_h2_overflow()
    assert_close(out32.as_ncdhw().cpu(), F.conv3d(big, w, padding=1), what="f32 engine on out-of-range activations")


def test_weight_pack_refuses_out_of_range_weights(dev):
    w = torch.zeros(32, 32, 1, 1, 1)
    w[0, 0] = 1.0e5
    pc = core.PackedConv(w.to(dev), ksize=1)
    with pytest.raises(_lib.CooccError, match="f16"):
        pc.h2_pack()


@pytest.mark.parametrize("V,depth", [(10007, 3), (8192, 1), (20000, 4)])
def test_fused_render_heads_equal_layerwise_path_and_torch(dev, monkeypatch, V, depth):
    """csrc/mlp_h2.hip (both render heads in one launch, hidden activations in LDS) against the layer-by-layer GEMM launches it
    replaces and against torch's fp64 evaluation of the same MLPs (nerf_mlp.py:14-105); ragged last tile, 1..4 hidden layers,
    the depth-only form, the in-kernel sigmoid."""
    import numpy as np
    from co_occ_amd import render as R
    import co_occ_amd.synth as synth
    g = torch.Generator().manual_seed(V + depth)
    x = torch.randn(V, 128, generator=g)
    x *= (torch.rand(V, 1, generator=g) < 0.8).float() * 3.0
    sig, rgb = R.MLP(128, 1, net_depth=1, skip_layer=None), R.MLP(128, 3, net_depth=depth, skip_layer=None)
    sig.load_state_dict(synth.random_state_dict(sig.state_dict(), 5))
    rgb.load_state_dict(synth.random_state_dict(rgb.state_dict(), 6))
    with torch.no_grad():
        def ref(m, v):
            v = v.double()
            for l in m.hidden_layers:
                v = torch.relu(torch.nn.functional.linear(v, l.weight.double(), l.bias.double()))
            return torch.nn.functional.linear(v, m.output_layer.weight.double(), m.output_layer.bias.double())
        want = torch.cat([ref(sig, x), ref(rgb, x)], 1)
    sig, rgb = sig.to(dev).eval(), rgb.to(dev).eval()
    vf = core.Rows(x.to(dev), 1, V, 1, 1, 128)
    with torch.no_grad():
        fused = R.voxel_table(sig, rgb, vf)
        assert R._cached_fused(sig, rgb) is not None
        act = R.voxel_table(sig, rgb, vf, activate=True)
        donly = R.voxel_table(sig, None, vf)
        monkeypatch.setattr(R, "FUSED_HEADS", False)
        layer = R.voxel_table(sig, rgb, vf)
        act_l = R.voxel_table(sig, rgb, vf, activate=True)
    torch.cuda.synchronize()
    core.check_h2_overflow()
    assert_close(fused.cpu(), want, what="fused heads vs torch fp64")
    assert_close(layer.cpu(), want, what="layer-by-layer heads vs torch fp64")
    assert_close(fused.cpu(), layer.cpu(), tol=2e-5, what="fused vs layer-by-layer")
    assert torch.equal(act[:, 0], fused[:, 0])
    assert_close(act.cpu(), act_l.cpu(), tol=2e-5, what="activated tables")
    assert torch.equal(donly[:, 0], fused[:, 0]) and float(donly[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", [0, 2])
def test_device_made_direct_h2_pack_equals_host_layout(dev, mode):
    """coocc_conv_pack_weights_h2_dev (training re-packs from the live parameter every step) writes exactly the pack the host
    builds for inference (core.PackedConv._h2_layout): forward (mode 0) and the stride-1 dgrad form (mode 2: roles of Cin / Cout
    swapped, taps flipped)."""
    from co_occ_amd.autograd import pack_weights_h2_dev
    Cout, Cin, taps = 96, 64, 27
    g = torch.Generator().manual_seed(5 + mode)
    w = torch.randn(Cout, Cin, taps, generator=g) * 0.1
    got = pack_weights_h2_dev(w.to(dev), Cout, Cin, taps, mode).cpu().view(torch.int16).flatten()
    ref_w = w if mode == 0 else w.permute(1, 0, 2).flip(2)          # [N, K, taps]
    N, K = ref_w.shape[:2]
    npad = -(-N // 128) * 128
    wp = torch.zeros(npad, K, taps, dtype=torch.float64)
    wp[:N] = ref_w.double()
    ref = core.PackedConv._h2_layout(wp).view(torch.int16).flatten()
    assert got.numel() == ref.numel()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("gmag", [1e-6, 1.0, 3e3])
def test_training_layers_on_the_split_f16_engine_match_torch_at_any_gradient_scale(dev, monkeypatch, gmag):
    """Strided, 1x1x1, direct stride-1 and Winograd 3x3x3 layers through ConvRowsFn: forward on the split-f16 kernels with
    device-made packs, stride-1 dgrad on the same engine with the gradient operand scaled by a power of two chosen on the device
    from max |dacc| (coocc_conv_epilogue_bwd_ex), the 1x1x1 and Winograd-domain (Z = 2, 4, 8) weight gradients on csrc/wgrad_h2.hip
    with voxel-major operands -- against torch autograd, at upstream-gradient magnitudes from 1e-6 (an unscaled
    f16 operand would be subnormal) to 3e3 (an unscaled Winograd-transformed operand would overflow)."""
    from co_occ_amd import autograd as ag
    monkeypatch.setattr(core, "H2_DIRECT_MIN_FLOPS", 0.0)
    monkeypatch.setattr(core, "WINO_MIN_ROWS", 0)
    assert ag.TRAIN_H2 and ag.TRAIN_H2_DGRAD and ag.TRAIN_H2_WGRAD
    g = torch.Generator().manual_seed(11)
    r2d = lambda t: t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).contiguous()       # [B,C,X,Y,Z] -> rows [B*X*Y*Z, C]
    B, X, Y, Cin, Cout = 1, 12, 12, 64, 96
    for k, stride, wino, Z in ((3, 2, False, 4), (1, 1, False, 4), (3, 1, False, 4), (3, 1, True, 4), (3, 1, True, 8), (3, 1, True, 2)):
        x = torch.randn(B, Cin, X, Y, Z, generator=g)
        w = torch.randn(Cout, Cin, k, k, k, generator=g) * 0.05
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.relu(F.conv3d(xr, wr, stride=stride, padding=k // 2))
        up = torch.randn(yr.shape, generator=g) * gmag
        yr.backward(up)
        xd = r2d(x).to(dev).requires_grad_(True)
        wd = w.to(dev).requires_grad_(True)
        monkeypatch.setattr(ag, "TRAIN_WINO", wino)
        yd, _ = ag.conv3d_rows(xd, wd, (B, X, Y, Z), stride=stride, pad=k // 2, relu=True)
        yd.backward(r2d(up).to(dev))
        torch.cuda.synchronize()
        core.check_h2_overflow()
        what = "k%d s%d wino=%d Z=%d |dy|~%g" % (k, stride, wino, Z, gmag)
        assert_close(yd.detach().cpu(), r2d(yr.detach()), what="h2 train fwd " + what)
        assert_close(xd.grad.cpu(), r2d(xr.grad), what="h2 train dgrad " + what)
        assert_close(wd.grad.cpu(), wr.grad, what="h2 train wgrad " + what)


@pytest.mark.parametrize("tile,dgrad", [(4, 0), (4, 1), (2, 0)])
def test_device_made_winograd_h2_pack_matches_host_pack(dev, tile, dgrad):
    """coocc_wino_pack_weights_h2_dev (training: U = G g G^T in fp64 on the device, one thread per 16-byte unit of the pack) against
    the host pack of the same weights (core.PackedConv.wino_h2_pack; for dgrad: of the flipped, transposed kernel).  Both split
    the fp64 products; the two fp64 evaluation orders may differ in the last bit, so the decoded values hi + lo 2^-11 are compared
    (1e-7 of the largest weight), and the padding columns must be zero."""
    Cout, Cin = 96, 64
    g = torch.Generator().manual_seed(17 + tile + dgrad)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.1
    lib = _lib.load()
    n = lib.coocc_wino_pack_weights_h2_dev(None, Cout, Cin, tile, dgrad, None, None)
    assert n > 0
    got = torch.empty(n, dtype=torch.float32, device=dev)
    wd = w.to(dev).contiguous()
    assert lib.coocc_wino_pack_weights_h2_dev(ptr(wd), Cout, Cin, tile, dgrad, ptr(got), _lib.stream(dev)) == n
    ref_w = w if not dgrad else w.permute(1, 0, 2, 3, 4).flip(2, 3, 4).contiguous()      # dx = conv(dy, W'): W'[c][n][a][b][z] = w[n][c][2-a][2-b][2-z]
    ref = core.PackedConv(ref_w.to(dev), ksize=3, stride=1, pad=1).wino_h2_pack(tile)

    def decode(t):              # [..., 2 planes, 64 lanes, 8] f16 -> hi + lo / 2048 per (unit, lane, e)
        h = t.cpu().view(torch.float16).double().view(-1, 2, 512)
        return h[:, 0] + h[:, 1] / 2048.0
    a, b = decode(got), decode(ref.contiguous().view(-1))
    assert a.shape == b.shape
    assert float((a - b).abs().max()) <= 1e-7 * float(b.abs().max())
