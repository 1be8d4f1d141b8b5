"""Accuracy (vs an fp64 CPU conv) and time of the split-f16 ("h2") Winograd convolutions next to the fp32-MFMA engine.
    python tools/h2_check.py [acc] [time]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from co_occ_amd import core  # noqa: E402

dev = torch.device("cuda:0")


def rows_of(x):
    return core.to_rows(x.to(dev))


def bn_like(C, g):
    bn = torch.nn.BatchNorm3d(C).eval()
    bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
    bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(C, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(C, generator=g) * 0.1)
    return bn


def run(x, pc, engine, tile, relu=True):
    core.CONV_ENGINE, core.WINO_TILE, core.WINO, core.WINO_MIN_ROWS = engine, tile, 1, 0
    return core.conv_rows(x, pc, relu=relu)


def acc():
    for (Cin, Cout, grid, ascale) in [(128, 128, (24, 20, 8), 1.0), (256, 256, (18, 22, 4), 1.0), (256, 128, (16, 16, 8), 40.0), (64, 160, (20, 17, 3), 1.0)]:
        g = torch.Generator().manual_seed(Cin + Cout)
        X, Y, Z = grid
        x = torch.relu(torch.randn(1, Cin, X, Y, Z, generator=g)) * torch.exp(torch.randn(1, Cin, X, Y, Z, generator=g)) * ascale
        w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
        bn = bn_like(Cout, g)
        ref = F.conv3d(x.double(), w.double(), padding=1)
        sc, bi = core.fold_bn(bn)
        ref = ref * sc.double().view(1, -1, 1, 1, 1) + bi.double().view(1, -1, 1, 1, 1)
        rms = ref.pow(2).mean().sqrt()
        pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=3, stride=1, pad=1)
        xr = rows_of(x)
        line = "Cin %3d Cout %3d grid %-12s act x%-4g:" % (Cin, Cout, grid, ascale)
        for tile in (2, 4):
            for eng in ("f32", "h2"):
                out = run(xr, pc, eng, tile, relu=False).as_ncdhw().cpu().double()
                e = (out - ref)
                line += "  F(%d) %-3s rms %.2e max %.2e |" % (tile, eng, float(e.pow(2).mean().sqrt() / rms), float(e.abs().max() / rms))
        core.WINO = 0
        core.CONV_ENGINE = "f32"
        out = core.conv_rows(xr, pc, relu=False).as_ncdhw().cpu().double()
        e = out - ref
        line += "  direct f32 rms %.2e max %.2e" % (float(e.pow(2).mean().sqrt() / rms), float(e.abs().max() / rms))
        print(line, flush=True)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def bench():
    core.TIMER.enabled = 0
    for (name, Cin, Cout, grid, tile) in [("enc.l0", 128, 128, (100, 100, 8), 4), ("fpn.out0", 256, 256, (100, 100, 8), 4),
                                          ("head.occ0", 256, 128, (100, 100, 8), 4), ("con_enc.3", 256, 128, (100, 100, 8), 2),
                                          ("con_enc.0 dense", 256, 256, (100, 100, 8), 2), ("con_enc.3 F4", 256, 128, (100, 100, 8), 4),
                                          ("enc.l1", 256, 256, (50, 50, 4), 4), ("openocc l0", 128, 128, (128, 128, 10), 4)]:
        g = torch.Generator().manual_seed(1)
        X, Y, Z = grid
        x = torch.randn(1, Cin, X, Y, Z, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
        pc = core.PackedConv(w.to(dev), bn=bn_like(Cout, g).to(dev), ksize=3, stride=1, pad=1)
        xr = rows_of(x)
        out = core.Rows(torch.empty(X * Y * Z, Cout, device=dev), 1, X, Y, Z, Cout)
        line = "%-16s %3d->%3d %-14s F(%d):" % (name, Cin, Cout, grid, tile)
        res = {}
        for eng in ("f32", "h2"):
            core.CONV_ENGINE, core.WINO_TILE, core.WINO, core.WINO_MIN_ROWS = eng, tile, 1, 0
            plan = core.wino_plan(xr, pc, X * Y * Z, 0)
            t_all = timeit(lambda: core.conv_rows_wino(xr, pc, out, True, None, plan))
            # GEMM alone: time with the transforms stubbed out is not possible from here; use the region timer instead
            core.TIMER.enabled, core.TIMER.only = 1, None
            core.TIMER.reset()
            for _ in range(10):
                core.conv_rows_wino(xr, pc, out, True, None, plan)
            torch.cuda.synchronize()
            sm = core.TIMER.summary()
            core.TIMER.enabled = 0
            parts = {k.split(" wino")[0]: v["ms"] / v["launches"] for k, v in sm.items() if not k.startswith("coocc_")}
            gk = [k for k in parts if k.startswith("k_conv") or k.startswith("k_gemm")][0]
            tile_, pts, Tx, Ty, rows, G, hint = plan
            fl = 2.0 * pts * rows * Cin * Cout * 3
            res[eng] = out.t.clone()
            line += "  %s: layer %.3f ms  in %.3f gemm %.3f (%.0f TF/s exec) out %.3f |" % (
                eng, t_all, parts.get("k_wino_in", 0), parts[gk], fl / parts[gk] / 1e9, parts.get("k_wino_out", 0))
        d = float((res["f32"] - res["h2"]).abs().max() / res["f32"].abs().max())
        print(line + "  |f32-h2| %.1e" % d, flush=True)


def direct():
    """Stride-1 3x3x3 layers outside the Winograd path (small grids): h2 direct kernel vs the fp32-MFMA kernel, error vs fp64."""
    core.TIMER.enabled = 0
    core.WINO = 1
    core.WINO_MIN_ROWS = 8192
    for (name, Cin, Cout, grid, use_res) in [("enc.l2", 512, 512, (25, 25, 2), True), ("enc.l3", 1024, 1024, (13, 13, 1), True),
                                            ("fpn.out2", 256, 256, (25, 25, 2), False), ("head.occ3", 256, 128, (13, 13, 1), False),
                                            ("odd", 96, 160, (11, 9, 3), False)]:
        g = torch.Generator().manual_seed(3)
        X, Y, Z = grid
        x = torch.relu(torch.randn(1, Cin, X, Y, Z, generator=g)) * torch.exp(torch.randn(1, Cin, X, Y, Z, generator=g))
        w = torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (2.0 / (Cin * 27)) ** 0.5
        bn = bn_like(Cout, g)
        res = torch.randn(1, Cout, X, Y, Z, generator=g) if use_res else None
        ref = F.conv3d(x.double(), w.double(), padding=1)
        sc, bi = core.fold_bn(bn)
        ref = ref * sc.double().view(1, -1, 1, 1, 1) + bi.double().view(1, -1, 1, 1, 1)
        if use_res:
            ref = ref + res.double()
        ref = torch.relu(ref)
        rms = ref.pow(2).mean().sqrt()
        pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=3, stride=1, pad=1)
        xr = rows_of(x)
        rr = rows_of(res) if use_res else None
        line = "%-10s %4d->%4d %-12s:" % (name, Cin, Cout, grid)
        for eng in ("f32", "h2"):
            core.CONV_ENGINE = eng
            out = core.conv_rows(xr, pc, relu=True, res=rr)
            e = out.as_ncdhw().cpu().double() - ref
            t = timeit(lambda: core.conv_rows(xr, pc, relu=True, res=rr))
            line += "  %-3s rms %.2e max %.2e  %.3f ms |" % (eng, float(e.pow(2).mean().sqrt() / rms), float(e.abs().max() / rms), t)
        print(line, flush=True)


def general():
    """Strided 3x3x3 and 1x1x1 layers on the general h2 kernel (k_gemm_h2w) vs the fp32-MFMA kernels, error vs fp64."""
    core.TIMER.enabled = 0
    core.WINO, core.WINO_MIN_ROWS = 1, 8192
    for (name, Cin, Cout, grid, k, stride) in [("enc.l1.0 s2", 128, 256, (100, 100, 8), 3, 2), ("enc.l2.0 s2", 256, 512, (50, 50, 4), 3, 2),
                                               ("enc.l3.0 s2", 512, 1024, (25, 25, 2), 3, 2), ("lateral0 1x1", 128, 256, (100, 100, 8), 1, 1),
                                               ("proj 1x1", 128, 128, (100, 100, 8), 1, 1), ("ds 1x1 s2", 128, 256, (100, 100, 8), 1, 2),
                                               ("odd s2", 96, 160, (11, 9, 5), 3, 2)]:
        g = torch.Generator().manual_seed(5)
        X, Y, Z = grid
        big = X * Y * Z > 20000
        x = torch.relu(torch.randn(1, Cin, X, Y, Z, generator=g)) * torch.exp(torch.randn(1, Cin, X, Y, Z, generator=g))
        w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
        bn = bn_like(Cout, g)
        pad = 1 if k == 3 else 0
        ref = F.conv3d(x.double(), w.double(), padding=pad, stride=stride) if not big else F.conv3d(x.to(dev), w.to(dev), padding=pad, stride=stride).cpu().double()
        sc, bi = core.fold_bn(bn)
        ref = torch.relu(ref * sc.double().view(1, -1, 1, 1, 1) + bi.double().view(1, -1, 1, 1, 1))
        rms = ref.pow(2).mean().sqrt()
        pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=k, stride=stride, pad=pad)
        xr = rows_of(x)
        line = "%-13s %4d->%4d %-14s%s:" % (name, Cin, Cout, grid, " (ref: torch fp32 on the GPU)" if big else "")
        for eng in ("f32", "h2"):
            core.CONV_ENGINE = eng
            out = core.conv_rows(xr, pc, relu=True)
            e = out.as_ncdhw().cpu().double() - ref
            t = timeit(lambda: core.conv_rows(xr, pc, relu=True))
            line += "  %-3s rms %.2e max %.2e  %.3f ms |" % (eng, float(e.pow(2).mean().sqrt() / rms), float(e.abs().max() / rms), t)
        print(line, flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["acc", "time"]
    with torch.no_grad():
        if "acc" in what:
            acc()
        if "time" in what:
            bench()
        if "direct" in what:
            direct()
        if "general" in what:
            general()
