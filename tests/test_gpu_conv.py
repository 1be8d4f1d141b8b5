"""fp32-MFMA implicit GEMM (C0-C3, G1, linear layers) and the resampling kernels vs torch CPU
fp32 references of the same ops (floating point: scale-relative 1e-4, see tests/util.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from co_occ_amd import core
from co_occ_amd._lib import call, host_i32, ptr
from util import assert_close

pytestmark = pytest.mark.gpu


def rows_of(x, dev):
    """[B,C,X,Y,Z] cpu tensor -> Rows on dev (through the HIP transpose)."""
    return core.to_rows(x.to(dev))


def bn_like(C, g):
    bn = torch.nn.BatchNorm3d(C).eval()
    bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
    bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(C, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(C, generator=g) * 0.1)
    return bn


CASES = [
    # Cin, Cout, grid, ksize, stride, relu, residual, splitk
    (16, 32, (9, 7, 4), 3, 1, True, False, 0),       # small, M tail, Cout<=32 config
    (32, 64, (12, 10, 4), 3, 1, True, True, 0),      # Cout<=64 config, residual
    (64, 128, (40, 40, 8), 3, 1, True, False, 0),    # 128x128 tiles, M = 12800
    (128, 256, (20, 20, 4), 3, 2, False, False, 0),  # stride 2, two N tiles, small M (64x128 tiles)
    (64, 128, (13, 13, 1), 3, 2, True, True, 4),     # forced split-K, odd dims, residual
    (256, 256, (7, 7, 2), 3, 1, True, False, 0),     # auto split-K path (few blocks, long K)
    (128, 17, (10, 10, 8), 1, 1, False, False, 0),   # 1x1x1 classifier, Cout = 17
    (96, 4, (10, 6, 2), 1, 1, False, False, 0),      # Cout = 4
    (64, 128, (25, 25, 2), 1, 2, False, False, 0),   # 1x1x1 stride-2 shortcut
    (36, 40, (6, 5, 3), 3, 1, True, False, 0),       # Cin not a multiple of 32
    (96, 160, (13, 13, 1), 3, 1, True, True, 0),     # Z = 1: only the centre z tap is ever in range (3x3x1 pack)
    (64, 96, (9, 11, 2), 3, 2, True, False, 0),      # Z = 2 -> 1: z taps 1..2 (3x3x2 pack, pz = 0)
]


@pytest.mark.parametrize("Cin,Cout,grid,k,stride,relu,use_res,splitk", CASES)
def test_conv3d_bn_relu_residual(dev, Cin, Cout, grid, k, stride, relu, use_res, splitk):
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    X, Y, Z = grid
    B = 2 if X < 20 else 1
    x = torch.randn(B, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
    bn = bn_like(Cout, g)
    pad = k // 2
    ref = bn(F.conv3d(x, w, stride=stride, padding=pad))
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=k, stride=stride, pad=pad)
    out = core.conv_rows(rows_of(x, dev), pc, relu=relu, res=rows_of(res, dev) if use_res else None, splitk=splitk)
    assert (out.X, out.Y, out.Z) == tuple(ref.shape[2:])
    assert_close(out.as_ncdhw().cpu(), ref.detach(), what="conv")


@pytest.mark.parametrize("grid,stride", [((13, 13, 1), 1), ((9, 11, 2), 2), ((12, 9, 1), 2)])
def test_conv_ztrim_matches_full_taps(dev, grid, stride, monkeypatch):
    """Dropping the z taps that only read padding gives the result of the full 27-tap kernel."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, *grid, generator=g)
    w = torch.randn(96, 64, 3, 3, 3, generator=g) * 0.04
    pc = core.PackedConv(w.to(dev), bn=bn_like(96, g).to(dev), ksize=3, stride=stride, pad=1)
    outs = []
    for z in (True, False):
        monkeypatch.setattr(core, "ZTRIM", z)
        outs.append(core.conv_rows(rows_of(x, dev), pc, relu=False).t.cpu())
    assert len(pc._ztrim) == 1
    assert_close(outs[0], outs[1], what="ztrim vs full")


@pytest.mark.parametrize("hint", [128, 160])
def test_conv_large_tile_variants(dev, hint):
    """Both M-tile variants of the large configuration (128x128 / 160x128) on a grid whose row count
    is a multiple of neither, with residual + ReLU."""
    g = torch.Generator().manual_seed(hint)
    x = torch.randn(1, 64, 37, 29, 8, generator=g)          # M = 8584 >= 8192
    w = torch.randn(160, 64, 3, 3, 3, generator=g) * 0.03    # two N tiles, ragged Cout
    bn = bn_like(160, g)
    res = torch.randn(1, 160, 37, 29, 8, generator=g)
    ref = F.relu(bn(F.conv3d(x, w, padding=1)) + res)
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=3, pad=1)
    old = core.TILE_HINT
    core.TILE_HINT = hint
    try:
        out = core.conv_rows(rows_of(x, dev), pc, relu=True, res=rows_of(res, dev))
    finally:
        core.TILE_HINT = old
    assert_close(out.as_ncdhw().cpu(), ref.detach(), what="tile %d" % hint)


def test_conv_mfma_layout_asymmetric(dev):
    """A = identity-like probe with an asymmetric weight catches row/col swaps in the D map."""
    Cin = Cout = 64
    x = torch.zeros(1, Cin, 4, 4, 4)
    for c in range(Cin):
        x[0, c, c % 4, (c // 4) % 4, c // 16] = 1.0 + c
    w = torch.arange(Cout * Cin, dtype=torch.float32).view(Cout, Cin, 1, 1, 1) / 997.0
    pc = core.PackedConv(w.to(dev), ksize=1)
    out = core.conv_rows(rows_of(x, dev), pc, relu=False)
    assert_close(out.as_ncdhw().cpu(), F.conv3d(x, w), tol=1e-6)


def test_linear_rows_bias_relu_offsets(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1000, 192, generator=g)
    lin = torch.nn.Linear(128, 64)
    pc = core.PackedConv(lin.weight.to(dev), bias=lin.bias.to(dev))
    out = torch.zeros(1000, 80, device=dev)
    core.linear_rows(x.to(dev), pc, relu=True, out=out, out_coff=16, in_coff=64)
    ref = F.relu(lin(x[:, 64:]))
    assert_close(out[:, 16:].cpu(), ref.detach())
    assert float(out[:, :16].abs().max()) == 0.0


def test_gather_conv_is_gsfusion_g1(dev):
    """rows table + tap-major Linear(C*K -> C) + ReLU + gate + scatter (bifuser_n.py:138-169)."""
    g = torch.Generator().manual_seed(9)
    V, C, K, M = 5000, 16, 3, 1700
    cat4 = torch.randn(V, 4 * C, generator=g)
    lin = torch.nn.Linear(C * K, C)
    gather = torch.randint(0, V, (K, M), generator=g).int()
    gather[1, ::7] = -1                                  # out-of-range row -> zeros
    out_rows = torch.randperm(V, generator=g)[:M].int()
    feat = torch.cat([torch.where(gather[k][:, None] >= 0, cat4[gather[k].clamp(min=0).long(), :C], torch.zeros(M, C)) for k in range(K)], 1)
    want = cat4.clone()
    want[out_rows.long(), 2 * C:3 * C] = F.relu(lin(feat)).detach() * cat4[out_rows.long(), C:2 * C]
    pc = core.PackedConv(lin.weight.to(dev), bias=lin.bias.to(dev), tap_major=True, taps=K)
    d = cat4.to(dev)
    core.gather_conv_rows(d, 0, pc, gather.to(dev), out_rows.to(dev), d, 2 * C, C, C)
    assert_close(d.cpu(), want)


@pytest.mark.parametrize("cs,fs", [((5, 4, 2), (10, 8, 4)), ((13, 13, 1), (25, 25, 2)), ((7, 7, 3), (7, 7, 3))])
def test_upsample_add_trilinear(dev, cs, fs):
    g = torch.Generator().manual_seed(sum(cs))
    c = torch.randn(2, 8, *cs, generator=g)
    f = torch.randn(2, 8, *fs, generator=g)
    want = f + F.interpolate(c, size=fs, mode="trilinear", align_corners=False)
    rc, rf = rows_of(c, dev), rows_of(f, dev)
    call("coocc_upsample_add_trilinear", ptr(rc.t), ptr(rf.t), 2, 8, *cs, *fs)
    assert_close(rf.as_ncdhw().cpu(), want, tol=1e-5)


def test_occhead_mix_softmax_levels(dev):
    g = torch.Generator().manual_seed(4)
    sizes = [(12, 10, 4), (6, 5, 2), (3, 3, 1), (2, 2, 1)]
    lv = [torch.randn(1, 16, *s, generator=g) for s in sizes]
    logit = torch.randn(1, 4, *sizes[0], generator=g)
    w = torch.softmax(logit, 1)
    want = 0
    for f, wi in zip(lv, torch.unbind(w, 1)):
        want = want + F.interpolate(f, size=list(sizes[0]), mode="trilinear", align_corners=False) * wi.unsqueeze(1)
    rows = [rows_of(t, dev) for t in lv]
    wl = rows_of(logit, dev)
    out = torch.empty_like(rows[0].t)
    import ctypes
    levels = (ctypes.c_void_p * 4)(*[r.t.data_ptr() for r in rows])
    call("coocc_occhead_mix", levels, host_i32([v for s in sizes for v in s]), 4, ptr(wl.t), ptr(out), 1, 16)
    assert_close(out.view(1, *sizes[0], 16).permute(0, 4, 1, 2, 3).cpu(), want, tol=1e-5)


@pytest.mark.parametrize("cs,fs,C", [((50, 50, 4), (100, 100, 8), 128), ((25, 25, 2), (50, 50, 4), 128), ((13, 13, 1), (25, 25, 2), 128),
                                     ((5, 4, 2), (10, 8, 4), 32), ((3, 7, 4), (6, 14, 8), 8), ((7, 7, 4), (7, 7, 8), 16)])
def test_interp_column_forms_equal_the_per_voxel_kernels_upsample_add(dev, monkeypatch, cs, fs, C):
    """fpn3d.py:88-92 in its z-column form (one thread = the voxels of one (x, y) column: the four coarse corner columns are
    loaded once per column instead of eight taps per voxel) == the per-voxel kernel (COOCC_INTERP_COLUMN=0), bit for bit, rows
    and H2 twin; the r50 pyramid's three shapes, a batch of two, odd x / y ratios, an x-y ratio of one; and torch."""
    g = torch.Generator().manual_seed(sum(cs) + C)
    c = torch.randn(2, C, *cs, generator=g)
    f = torch.randn(2, C, *fs, generator=g)
    outs = []
    for col in ("1", "0"):                       # bit 0: the upsample-add form
        monkeypatch.setenv("COOCC_INTERP_COLUMN", col)
        rc, rf = rows_of(c, dev), rows_of(f, dev)
        tw = torch.zeros(rf.t.numel(), device=dev) if C % 32 == 0 else None
        call("coocc_upsample_add_trilinear_ex", ptr(rc.t), ptr(rf.t), 2, C, *cs, *fs, ptr(tw))
        outs.append((rf.t.clone(), tw))
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1]) and bool((outs[0][1] != 0).any())
    want = f + F.interpolate(c, size=fs, mode="trilinear", align_corners=False)
    assert_close(rf.as_ncdhw().cpu(), want, tol=1e-5)


@pytest.mark.parametrize("sizes,C,B", [([(100, 100, 8), (50, 50, 4), (25, 25, 2), (13, 13, 1)], 128, 1),
                                       ([(12, 10, 8), (6, 5, 4), (3, 3, 2), (2, 2, 1)], 32, 2),
                                       ([(9, 7, 8), (9, 7, 4), (5, 4, 2), (3, 2, 1)], 16, 1)])
def test_interp_column_forms_equal_the_per_voxel_kernels_mix(dev, monkeypatch, sizes, C, B):
    """occ_head.py:155-166 in its half-z-column form (COOCC_INTERP_COLUMN bit 1, the default since round 6: csrc/interp.hip) == the
    per-voxel kernel, bit for bit (output rows and H2 twin): the r50 head's shapes, a batch of two, a level that shares x-y with the
    output grid; and torch.  (Beside other streams' matrix-core work: tests/test_gpu_corunner.py.)"""
    import ctypes
    torch.cuda.synchronize()
    g = torch.Generator().manual_seed(C + B)
    lv = [torch.randn(B, C, *s, generator=g) for s in sizes]
    logit = torch.randn(B, 4, *sizes[0], generator=g) * 3
    rows = [rows_of(t, dev) for t in lv]
    wl = rows_of(logit, dev)
    levels = (ctypes.c_void_p * 4)(*[r.t.data_ptr() for r in rows])
    outs = []
    for col in ("2", "0"):                       # bit 1: the mix form (off by default: see csrc/interp.hip)
        monkeypatch.setenv("COOCC_INTERP_COLUMN", col)
        out = torch.zeros_like(rows[0].t)
        tw = torch.zeros(out.numel(), device=dev) if C % 32 == 0 else None
        call("coocc_occhead_mix_ex", levels, host_i32([v for s in sizes for v in s]), 4, ptr(wl.t), ptr(out), B, C, ptr(tw))
        outs.append((out, tw))
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1]) and bool((outs[0][1] != 0).any())
    w = torch.softmax(logit, 1)
    want = 0
    for f, wi in zip(lv, torch.unbind(w, 1)):
        want = want + F.interpolate(f, size=list(sizes[0]), mode="trilinear", align_corners=False) * wi.unsqueeze(1)
    assert_close(outs[0][0].view(B, *sizes[0], C).permute(0, 4, 1, 2, 3).cpu(), want, tol=1e-5)


@pytest.mark.parametrize("wino,tol", [(0, 1e-5), (1, 1e-4)])
def test_conv_linearity_at_full_grid(dev, monkeypatch, wino, tol):
    """Size-independent property at the BASELINE grid (100x100x8, C=128): conv(a*x + y) ==
    a*conv(x) + conv(y) for the bias-free, ReLU-free 3x3x3 layer; and a checksum of the
    1-channel all-ones response equals the analytic tap count.  Direct path and the Winograd
    F(4x4,3x3) path the production dispatch picks at this size."""
    monkeypatch.setattr(core, "WINO", wino)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 128, 100, 100, 8, generator=g).to(dev)
    y = torch.randn(1, 128, 100, 100, 8, generator=g).to(dev)
    w = (torch.randn(128, 128, 3, 3, 3, generator=g) * 0.02).to(dev)
    pc = core.PackedConv(w, ksize=3, pad=1)
    cx = core.conv_rows(core.to_rows(x), pc, relu=False).t
    cy = core.conv_rows(core.to_rows(y), pc, relu=False).t
    cz = core.conv_rows(core.to_rows(2.5 * x + y), pc, relu=False).t
    assert_close(cz.cpu(), (2.5 * cx + cy).cpu(), tol=tol)
    ones = torch.ones(1, 4, 100, 100, 8, device=dev)
    w1 = torch.ones(4, 4, 3, 3, 3, device=dev)
    resp = core.conv_rows(core.to_rows(ones), core.PackedConv(w1, ksize=3, pad=1), relu=False).as_ncdhw()
    taps = (100 * 3 - 2) * (100 * 3 - 2) * (8 * 3 - 2)          # sum over voxels of valid taps = prod (3n-2)
    assert float(resp[0, 0].double().sum().item()) == pytest.approx(4.0 * taps, rel=1e-6 if not wino else 1e-5)


WINO_CASES = [
    # Cin, Cout, grid, B, relu, residual
    (16, 32, (9, 7, 4), 2, True, False),       # odd X, Y (partial tiles), two batches
    (64, 128, (12, 10, 4), 1, True, True),     # residual + ReLU in the output transform
    (36, 40, (6, 5, 3), 1, False, False),      # Cin not a multiple of 32, Z = 3 (group rows = lcm(640, 3))
    (128, 17, (8, 8, 2), 1, False, False),     # Cout % 4 != 0
    (128, 128, (40, 40, 8), 1, True, False),   # 160-row tiles, M = 16 * 3200
]


@pytest.mark.parametrize("tile", [2, 3, 4])
@pytest.mark.parametrize("Cin,Cout,grid,B,relu,use_res", WINO_CASES + [(512, 128, (16, 12, 4), 1, True, False)])
def test_conv3d_winograd_path(dev, monkeypatch, Cin, Cout, grid, B, relu, use_res, tile):
    """The Winograd F(2x2,3x3) + direct-z path (csrc/winograd.hip) against torch's direct fp32 conv, and against
    our own direct path (both inside the 1e-4 bound)."""
    g = torch.Generator().manual_seed(Cin * 77 + Cout)
    X, Y, Z = grid
    x = torch.randn(B, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    bn = bn_like(Cout, g)
    ref = bn(F.conv3d(x, w, padding=1))
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=3, stride=1, pad=1)
    rr = rows_of(res, dev) if use_res else None
    monkeypatch.setattr(core, "WINO", 1)
    monkeypatch.setattr(core, "WINO_MIN_ROWS", 0)
    monkeypatch.setattr(core, "WINO_TILE", tile)
    assert core.wino_eligible(rows_of(x, dev), pc, B * X * Y * Z, 1 if use_res else 0)
    out = core.conv_rows(rows_of(x, dev), pc, relu=relu, res=rr)
    assert_close(out.as_ncdhw().cpu(), ref.detach(), what="winograd conv")
    monkeypatch.setattr(core, "WINO", 0)
    direct = core.conv_rows(rows_of(x, dev), pc, relu=relu, res=rr)
    assert_close(out.as_ncdhw().cpu(), direct.as_ncdhw().cpu(), what="winograd vs direct")


@pytest.mark.parametrize("Cin,Cout,grid,tile", [(64, 160, (96, 90, 8), 2), (128, 128, (100, 100, 8), 4), (32, 128, (90, 93, 5), 4)])
def test_conv3d_winograd_persistent_kernel(dev, monkeypatch, Cin, Cout, grid, tile):
    """Full-size Winograd layers: short K and more than 768 GEMM tiles, i.e. the launches that take the persistent
    k_conv2p kernel (every workgroup walks several tiles; chunk 0 of the next tile rides in the last prefetch slot).
    Checked against our direct path (itself checked against torch at the smaller sizes above) and, for one case,
    against torch's fp32 conv."""
    g = torch.Generator().manual_seed(Cin + Cout)
    X, Y, Z = grid
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    bn = bn_like(Cout, g)
    ref = F.relu(bn(F.conv3d(x, w, padding=1))).detach() if Cin == 32 else None
    # the persistent kernel belongs to the exact-fp32 MFMA engine (COOCC_CONV_ENGINE=f32, the documented way out of the
    # split-f16 engine's operand range): under the default engine these layers run k_gemm_h2z and k_conv2p would never launch
    monkeypatch.setattr(core, "CONV_ENGINE", "f32")
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=3, stride=1, pad=1)
    monkeypatch.setattr(core, "WINO_TILE", tile)
    monkeypatch.setattr(core, "WINO", 1)
    xr = rows_of(x, dev)
    plan = core.wino_plan(xr, pc, X * Y * Z, 0)
    assert plan is not None and plan[0] == tile and not core.h2_capable(pc)
    assert (plan[1] * plan[5] // 128) * -(-Cout // 128) > 768 and 3 * -(-Cin // 32) <= 24     # the k_conv2p dispatch rule
    core.TIMER.enabled, core.TIMER.only = 1, None
    core.TIMER.reset()
    out = core.conv_rows(xr, pc, relu=True)
    torch.cuda.synchronize()
    names = set(core.TIMER.summary())
    core.TIMER.enabled = False
    core.TIMER.reset()
    assert any(n.startswith("k_conv2p") for n in names), "the persistent kernel was not dispatched: %s" % sorted(names)
    monkeypatch.setattr(core, "WINO", 0)
    direct = core.conv_rows(xr, pc, relu=True)
    assert_close(out.as_ncdhw().cpu(), direct.as_ncdhw().cpu(), what="persistent winograd GEMM vs direct")
    if ref is not None:
        assert_close(out.as_ncdhw().cpu(), ref, what="persistent winograd GEMM vs torch")


def test_conv_anisotropic_taps_and_weight_groups(dev):
    """coocc_conv_desc kx/ky/kz + wgroup_rows: a (1,1,3) conv along z with a different weight set per row group."""
    import ctypes
    from co_occ_amd import _lib
    g = torch.Generator().manual_seed(4)
    groups, cols, Z, Cin, Cout = 3, 80, 8, 32, 48          # 640 rows per group
    x = torch.randn(groups, cols, Z, Cin, generator=g)
    w = torch.randn(groups, Cout, Cin, 3, generator=g) / (3 * Cin) ** 0.5
    ref = torch.stack([F.conv1d(x[i].permute(0, 2, 1), w[i], padding=1).permute(0, 2, 1) for i in range(groups)])
    lib = _lib.load()
    n = lib.coocc_conv_pack_weights(ctypes.c_void_p(w[0].contiguous().data_ptr()), Cout, Cin, 3, 0, None)
    packed = torch.empty(groups, n)
    for i in range(groups):
        wi = w[i].contiguous()
        lib.coocc_conv_pack_weights(ctypes.c_void_p(wi.data_ptr()), Cout, Cin, 3, 0, ctypes.c_void_p(packed[i].data_ptr()))
    xd, pd = x.reshape(-1, Cin).contiguous().to(dev), packed.to(dev)
    out = torch.empty(groups * cols * Z, Cout, device=dev)
    d = _lib.ConvDesc()
    d.in_, d.w, d.out = ptr(xd), ptr(pd), ptr(out)
    d.M, d.Cin, d.Cout, d.taps = out.shape[0], Cin, Cout, 3
    d.in_stride, d.out_stride = Cin, Cout
    d.B, d.Xi, d.Yi, d.Zi, d.Xo, d.Yo, d.Zo = groups * cols, 1, 1, Z, 1, 1, Z
    d.ksize, d.stride, d.pad = 3, 1, 1
    d.kx, d.ky, d.kz, d.px, d.py, d.pz = 1, 1, 3, 0, 0, 1
    d.wgroup_rows = cols * Z
    d.splitk = 1
    _lib.check(lib.coocc_conv_fwd(ctypes.byref(d), _lib.stream()))
    assert_close(out.cpu().view(groups, cols, Z, Cout), ref, what="grouped z-conv")


def test_conv_large_volume_beyond_4gb_offsets(dev, monkeypatch):
    """200x200x16 x 512 channels: the Winograd V buffer is 5.2 GB (F(2x2)), past 32-bit buffer offsets -- the GEMM
    bases its descriptor at each tile's first row.  Winograd (both tile sizes) vs the direct path, and a spot check
    of both against torch on cropped neighbourhoods."""
    g = torch.Generator().manual_seed(12)
    Cin, Cout, (X, Y, Z) = 512, 128, (200, 200, 16)
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
    pc = core.PackedConv(w.to(dev), ksize=3, pad=1)
    xr = rows_of(x, dev)
    monkeypatch.setattr(core, "WINO", 0)
    direct = core.conv_rows(xr, pc, relu=False).as_ncdhw()
    monkeypatch.setattr(core, "WINO", 1)
    monkeypatch.setattr(core, "WINO_MIN_ROWS", 0)
    for tile in (2, 4):
        monkeypatch.setattr(core, "WINO_TILE", tile)
        plan = core.wino_plan(xr, pc, X * Y * Z, 0)
        assert plan is not None and plan[0] == tile
        if tile == 2:
            assert plan[1] * plan[5] * Cin * 4 > 2 ** 32          # V really is past 4 GB
        out = core.conv_rows(xr, pc, relu=False).as_ncdhw()
        assert_close(out.cpu(), direct.cpu(), what="winograd F(%d) vs direct at 200x200x16" % tile)
    for (cx, cy, cz) in ((0, 0, 0), (199, 199, 15), (100, 57, 8), (3, 198, 1)):
        x0, y0, z0 = max(cx - 1, 0), max(cy - 1, 0), max(cz - 1, 0)
        crop = x[:, :, x0:cx + 2, y0:cy + 2, z0:cz + 2]
        ref = F.conv3d(crop, w, padding=1)[0, :, cx - x0, cy - y0, cz - z0]
        assert_close(direct[0, :, cx, cy, cz].cpu(), ref, what="direct vs torch at %s" % ((cx, cy, cz),))


@pytest.mark.parametrize("use_res", [False, True])
def test_pointwise_conv_persistent_kernel_ragged(dev, use_res):
    """1x1x1 stride-1 layers with more than 768 tiles take k_conv2p<false, true> (persistent, full epilogue): ragged row
    count (last tile partly outside M), ragged Cout (last N tile partly outside), BN scale/shift, residual, ReLU."""
    g = torch.Generator().manual_seed(9)
    X, Y, Z, Cin, Cout = 103, 97, 10, 64, 160          # M = 99910 = 780 * 128 + 70
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, 1, generator=g) * 0.1
    bn = bn_like(Cout, g)
    res = torch.randn(1, Cout, X, Y, Z, generator=g) if use_res else None
    ref = bn(F.conv3d(x, w))
    if use_res:
        ref = ref + res
    ref = F.relu(ref).detach()
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=1, stride=1, pad=0)
    assert core.conv_kernel_name(X * Y * Z, Cout, False, 0, 2, True) == "k_conv2p<1x1>"
    out = core.conv_rows(rows_of(x, dev), pc, relu=True, res=rows_of(res, dev) if use_res else None)
    assert_close(out.as_ncdhw().cpu(), ref, what="persistent 1x1 conv")


def test_sparse_tap_sum_raises_the_sticky_fault_word_on_a_corrupted_map(dev):
    """an ordinal past the rows P was sized for (a stale / corrupted voxel -> ordinal map) is skipped by the kernel --
    no GPU memory fault, no device printf -- and recorded in the host-mapped fault word; ``core.check_h2_overflow`` raises."""
    import pytest as _pt
    from co_occ_amd import _lib
    from co_occ_amd._lib import call, ptr
    X = Y = Z = 6
    Co, Np = 8, 5
    vmap = torch.full((X * Y * Z,), -1, dtype=torch.int32, device=dev)
    vmap[[3, 40, 77, 100, 200]] = torch.arange(Np, dtype=torch.int32, device=dev)
    P = torch.randn(Np, 27 * Co, device=dev)
    S = torch.empty(X * Y * Z, Co, device=dev)
    call("coocc_sparse_tap_sum", ptr(P), ptr(vmap), 1, X, Y, Z, Co, None, ptr(S), Co, Np)
    torch.cuda.synchronize()
    core.check_h2_overflow()                                  # a valid map: no fault
    good = S.clone()
    vmap[120] = 9999                                          # no valid caller produces this
    call("coocc_sparse_tap_sum", ptr(P), ptr(vmap), 1, X, Y, Z, Co, None, ptr(S), Co, Np)
    torch.cuda.synchronize()
    with _pt.raises(_lib.CooccError, match="fault code 1"):
        core.check_h2_overflow()
    core.check_h2_overflow()                                  # cleared by the read
    # the bad entry was skipped: voxels outside its 3x3x3 neighbourhood are untouched
    far = torch.ones(X * Y * Z, dtype=torch.bool, device=dev)
    z, y, x = 120 % Z, (120 // Z) % Y, 120 // (Z * Y)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                ux, uy, uz = x + dx, y + dy, z + dz
                if 0 <= ux < X and 0 <= uy < Y and 0 <= uz < Z:
                    far[(ux * Y + uy) * Z + uz] = False
    assert torch.equal(S[far], good[far]) and torch.isfinite(S).all()
