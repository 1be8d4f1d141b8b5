#!/usr/bin/env python
"""Per-kernel micro-benchmarks at the r50 workload's shapes (HIP events on the launch stream).
    python tools/kbench.py [fps] [knn] [conv] [render] [pool]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import co_occ_amd as pkg  # noqa: E402
import co_occ_amd.synth as synth  # noqa: E402
from co_occ_amd import core, fuser  # noqa: E402
from co_occ_amd._lib import call, ptr  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def voxel_lists(grid=(100, 100, 8)):
    img, pts = synth.voxel_inputs(grid, C=4, seed=1234)
    li = torch.nonzero(img.sum(1).flatten())[:, 0].int().to(dev)
    lp = torch.nonzero(pts.sum(1).flatten())[:, 0].int().to(dev)
    return li, lp


def bench_fps():
    for grid in ((100, 100, 8), (200, 200, 16)):
        li, lp = voxel_lists(grid)
        for name, lin in (("img", li), ("pts", lp)):
            X, Y, Z = grid
            xyz = torch.stack([lin // (Y * Z), (lin // Z) % Y, lin % Z], 1).float().contiguous()
            t_v = timeit(lambda: fuser._fps_voxels(lin, grid, 2048))
            t_g = timeit(lambda: pkg.furthest_point_sample(xyz[None], 2048), n=2, warm=1)
            a = fuser._fps_voxels(lin, grid, 2048)
            b = pkg.furthest_point_sample(xyz[None], 2048)
            print("fps %s %-4s n=%6d  voxel-bucket %.3f ms (%.2f us/iter)  generic %.3f ms  equal=%s" % (
                "x".join(map(str, grid)), name, lin.numel(), t_v, 1e3 * t_v / 2047, t_g, bool(torch.equal(a, b))))


def bench_knn():
    li, lp = voxel_lists()
    grid = (100, 100, 8)
    X, Y, Z = grid
    for qn, q, k in (("pts->img", lp, li), ("img->pts", li, lp)):
        qx = torch.stack([q // (Y * Z), (q // Z) % Y, q % Z], 1).float().contiguous()
        kx = torch.stack([k // (Y * Z), (k // Z) % Y, k % Z], 1).float().contiguous()
        rq = qx[fuser._fps_voxels(q, grid, 2048)[0].long()].contiguous()
        val = torch.empty(2048, 2, device=dev)
        idx = torch.empty(2048, 2, device=dev, dtype=torch.int32)
        t1 = timeit(lambda: call("coocc_knn_topk", 2048, kx.shape[0], 2, ptr(rq), ptr(kx), ptr(val), ptr(idx)))
        grp = torch.empty(2048, 200, device=dev, dtype=torch.int32)
        t2 = timeit(lambda: call("coocc_ball_query", 1, qx.shape[0], 2048, 0.0, 6.0, 200, ptr(rq), ptr(qx), ptr(grp)))
        # the grid forms the fuser uses (index maps + sorted offset table / window walk), incl. building both maps
        rep = fuser._fps_voxels(q, grid, 2048)[0].contiguous()
        maps = torch.empty(2, X * Y * Z, device=dev, dtype=torch.int32)
        off = fuser.offset_table(Z, dev)

        def grid_forms():
            call("coocc_voxel_index_map", ptr(q), q.numel(), X * Y * Z, ptr(maps[0]))
            call("coocc_voxel_index_map", ptr(k), k.numel(), X * Y * Z, ptr(maps[1]))
            call("coocc_knn_topk_voxels", 2048, kx.shape[0], 2, X, Y, Z, ptr(rep), ptr(q), ptr(maps[1]), ptr(off), off.numel(), ptr(rq),
                 ptr(kx), ptr(val), ptr(idx))
            call("coocc_ball_query_voxels", 2048, 0.0, 6.0, 200, X, Y, Z, ptr(rep), ptr(q), ptr(maps[0]), ptr(grp))
        t3 = timeit(grid_forms)
        print("%s: brute force topk %.3f ms  ball_query %.3f ms | grid forms (2 index maps + topk + ball query) %.3f ms" % (qn, t1, t2, t3))


def bench_conv():
    shapes = [  # (name, Cin, Cout, grid, k, stride)
        ("con_enc.0", 512, 256, (100, 100, 8), 3, 1), ("con_enc.3", 256, 128, (100, 100, 8), 3, 1),
        ("enc.l0.conv", 128, 128, (100, 100, 8), 3, 1), ("enc.l1.conv1", 128, 256, (100, 100, 8), 3, 2),
        ("enc.l1.conv2", 256, 256, (50, 50, 4), 3, 1), ("enc.l2.conv2", 512, 512, (25, 25, 2), 3, 1),
        ("enc.l3.conv2", 1024, 1024, (13, 13, 1), 3, 1), ("fpn.out0", 256, 256, (100, 100, 8), 3, 1),
        ("head.occ0", 256, 128, (100, 100, 8), 3, 1), ("lat0 1x1", 128, 256, (100, 100, 8), 1, 1),
        ("pred 1x1", 128, 64, (100, 100, 8), 1, 1), ("cls 1x1", 64, 17, (100, 100, 8), 1, 1),
    ]
    tot = 0
    for name, ci, co, g, k, st in shapes:
        x = core.Rows(torch.randn(g[0] * g[1] * g[2], ci, device=dev), 1, g[0], g[1], g[2], ci)
        pc = core.PackedConv(torch.randn(co, ci, k, k, k, device=dev) * 0.02, ksize=k, stride=st, pad=k // 2)
        if name.startswith("con_enc"):
            pc.wino_tile = 2          # as BiFuser_N packs them (fuser.py): F(2x2) keeps the fine logits inside 1e-4
        t = timeit(lambda: core.conv_rows(x, pc, relu=True), n=3, warm=1)
        M = (core.out_dim(g[0], k, st, k // 2) * core.out_dim(g[1], k, st, k // 2) * core.out_dim(g[2], k, st, k // 2))
        fl = 2.0 * M * ci * co * k ** 3
        plan = core.wino_plan(x, pc, M, 0) if (k == 3 and st == 1) else None
        path = "winograd F(%dx%d): transforms + grouped k_conv2" % (plan[0], plan[0]) if plan else "direct " + core.conv_kernel_name(M, co, False, 0, k ** 3 * -(-ci // 32))
        print("%-14s %4d->%4d %-11s k%d s%d  %8.3f ms  %6.1f TFLOP/s direct-equivalent  [%s]" % (
            name, ci, co, "x".join(map(str, g)), k, st, t, fl / t / 1e9, path))



def bench_convbf16():
    """The reduced-precision convolution path at the OpenOccupancy decoder's shapes: k_conv_bf16w (operands bf16 in memory,
    global_load_lds staging; includes the coocc_rows_to_bf16 pass) and k_conv_bf16 (fp32 operands rounded inside the K loop),
    against the 2.5 PFLOP/s dense bf16-MFMA peak."""
    shapes = [("con_enc.0", 512, 256, (128, 128, 10), 3, 1), ("con_enc.3", 256, 128, (128, 128, 10), 3, 1),
              ("enc.l0.conv", 128, 128, (128, 128, 10), 3, 1), ("enc.l1.conv1", 128, 256, (128, 128, 10), 3, 2),
              ("enc.l1.conv2", 256, 256, (64, 64, 5), 3, 1), ("enc.l2.conv2", 512, 512, (32, 32, 3), 3, 1),
              ("enc.l3.conv2", 1024, 1024, (16, 16, 2), 3, 1), ("fpn.out0", 256, 256, (128, 128, 10), 3, 1),
              ("lat0 1x1", 128, 256, (128, 128, 10), 1, 1)]
    core.CONV_DTYPE = "bf16"
    try:
        for name, ci, co, g, k, st in shapes:
            x = core.Rows(torch.randn(g[0] * g[1] * g[2], ci, device=dev), 1, g[0], g[1], g[2], ci)
            pc = core.PackedConv(torch.randn(co, ci, k, k, k, device=dev) * 0.02, ksize=k, stride=st, pad=k // 2)
            M = (core.out_dim(g[0], k, st, k // 2) * core.out_dim(g[1], k, st, k // 2) * core.out_dim(g[2], k, st, k // 2))
            fl = 2.0 * M * ci * co * k ** 3
            ts = {}
            for pre in (True, False):
                core.BF16_PRECONVERT = pre
                ts[pre] = timeit(lambda: core.conv_rows(x, pc, relu=True), n=5, warm=2)
            xb = torch.empty(x.V * ci, device=dev, dtype=torch.bfloat16)
            tc = timeit(lambda: call("coocc_rows_to_bf16", x.data(), x.stride, x.V, ci, ptr(xb)), n=5, warm=1)
            print("%-14s %4d->%4d %-11s k%d s%d  bf16w %7.3f ms (conversion pass %.3f) = %6.1f TFLOP/s = %.3f of peak | k_conv_bf16 %7.3f ms = %6.1f TFLOP/s" % (
                name, ci, co, "x".join(map(str, g)), k, st, ts[True], tc, fl / ts[True] / 1e9, fl / ts[True] / 1e9 / 2500.0,
                ts[False], fl / ts[False] / 1e9))
    finally:
        core.CONV_DTYPE, core.BF16_PRECONVERT = "f32", True


def bench_render():
    """R2 at r50 / r101 sizes: ray kernel + x16 upsample vs their algorithmic HBM bytes."""
    from co_occ_amd import render as R
    g = torch.Generator().manual_seed(3)
    table = torch.randn(80000, 4, generator=g).to(dev)
    for name, (fH, fW) in (("r50", (16, 44)), ("r101", (56, 100))):
        N, D = 6, 112
        gemo = (torch.rand(N, D, fH, fW, 3, generator=g) * torch.tensor([120., 120., 10.]) - torch.tensor([60., 60., 5.5])).to(dev)
        zv = torch.linspace(0, D, D, device=dev)
        maps = torch.empty(N, fH, fW, 4, device=dev)
        rgbs = torch.empty(N, fH * 16, fW * 16, 3, device=dev)
        dep = torch.empty(N, fH * 16, fW * 16, device=dev)
        from co_occ_amd._lib import host_f32
        b = host_f32(R.RENDER_BOUNDS)
        call("coocc_render_activate_table", ptr(table), 80000)
        t1 = timeit(lambda: call("coocc_render_nearest", ptr(table), 100, 100, 8, ptr(gemo), ptr(zv), N, D, fH, fW, b, 1, ptr(maps)), n=20)
        t2 = timeit(lambda: call("coocc_upsample_maps", ptr(maps), N, fH, fW, 16, ptr(rgbs), ptr(dep)), n=20)
        by1 = 12.0 * N * D * fH * fW + 16.0 * 80000 + 16.0 * N * fH * fW
        by2 = 16.0 * N * fH * fW + 16.0 * N * fH * fW * 256
        from co_occ_amd import render as Rm
        class _VF:      # the table is passed in directly: stub the per-voxel heads
            pass
        def pair(chunks):
            Rm.RENDER_CHUNKS = chunks
            saved = Rm.voxel_table
            Rm.voxel_table = lambda a, b, vf: table.clone()
            try:
                vfr = core.Rows(torch.empty(80000, 4, device=dev), 1, 100, 100, 8, 4)
                return timeit(lambda: Rm.render_block(None, None, vfr, gemo[None], 16), n=20)
            finally:
                Rm.voxel_table = saved
        tp = {c: pair(c) for c in (1, 2, 3, 6)}
        print("render %-4s pair via render_block (activate + rays + upsample, incl. table clone): %s  -> best %.0f GB/s = %.3f of 8 TB/s" % (
            name, "  ".join("chunks=%d %.3f ms" % kv for kv in tp.items()), (by1 + by2) / min(tp.values()) / 1e6,
            (by1 + by2) / min(tp.values()) / 1e6 / 8000))
        print("render %-4s rays %.3f ms (%.0f GB/s of %.1f MB)  upsample %.3f ms (%.0f GB/s of %.1f MB)  total %.0f GB/s = %.3f of 8 TB/s" % (
            name, t1, by1 / t1 / 1e6, by1 / 1e6, t2, by2 / t2 / 1e6, by2 / 1e6, (by1 + by2) / (t1 + t2) / 1e6, (by1 + by2) / (t1 + t2) / 1e6 / 8000))


def bench_pool():
    """P2 at r50 / r101: materialised volume + voxel_pool vs fused lift (x) splat."""
    cfg = dict(xbound=[-50, 50, 1.0], ybound=[-50, 50, 1.0], zbound=[-5.0, 3.0, 1.0], dbound=[2.0, 58.0, 0.5])
    g = torch.Generator().manual_seed(4)
    for name, size, (fH, fW) in (("r50", (256, 704), (16, 44)), ("r101", (896, 1600), (56, 100))):
        N, D, C = 6, 112, 128
        rig = synth.camera_rig(N, size, seed=7)
        vt = pkg.ViewTransformerLiftSplatShootVoxel(grid_config=cfg, data_config=dict(input_size=size), downsample=16,
                                                    numC_Trans=C).to(dev)
        cams = tuple(rig[k].to(dev) for k in ("rots", "trans", "intrins", "post_rots", "post_trans", "bda"))
        depth = torch.softmax(torch.randn(N, D, fH, fW, generator=g), 1).to(dev)
        feat = torch.randn(N, C, fH, fW, generator=g).to(dev)
        geom = vt.get_geometry(*cams)
        kept = float(((geom[..., 0].abs() < 50) & (geom[..., 1].abs() < 50) & (geom[..., 2] >= -5) & (geom[..., 2] < 3)).float().mean())
        t_f = timeit(lambda: vt.lift_splat(depth, feat, cams=cams), n=10)
        t_g = timeit(lambda: vt.lift_splat(depth, feat, geom), n=10)
        npts = N * D * fH * fW
        by_f = 4.0 * npts + 4.0 * N * fH * fW * C + 4.0 * 80000 * C
        line = "pool %-4s kept %.2f  fused(cams) %.3f ms  fused(geom tensor) %.3f ms  = %.0f GB/s of %.1f MB algorithmic" % (
            name, kept, t_f, t_g, by_f / t_f / 1e6, by_f / 1e6)
        if name == "r50":
            def materialised():
                vol = depth.view(1, N, 1, D, fH, fW) * feat.view(1, N, C, 1, fH, fW)
                return vt.voxel_pooling(geom, vol.permute(0, 1, 3, 4, 5, 2))
            t_m = timeit(materialised, n=5)
            a, b = materialised(), vt.lift_splat(depth, feat, geom)
            b2 = vt.lift_splat(depth, feat, geom)
            line += "  | materialised volume + voxel_pool (ascending point id) %.3f ms, max |diff| %.1e of %.1f, run-to-run equal=%s" % (
                t_m, float((a - b).abs().max()), float(a.abs().max()), bool(torch.equal(b, b2)))
        print(line)


def bench_poolprof():
    """lift_splat only (for rocprofv3 --kernel-trace --stats / --pmc): COOCC_POOLPROF = r50 | r101 (default), geometry in-kernel
    (the form the step runs)."""
    which = os.environ.get("COOCC_POOLPROF", "r101")
    cfg = dict(xbound=[-50, 50, 1.0], ybound=[-50, 50, 1.0], zbound=[-5.0, 3.0, 1.0], dbound=[2.0, 58.0, 0.5])
    g = torch.Generator().manual_seed(4)
    N, D, C = 6, 112, 128
    size, (fH, fW) = ((256, 704), (16, 44)) if which == "r50" else ((896, 1600), (56, 100))
    rig = synth.camera_rig(N, size, seed=7)
    vt = pkg.ViewTransformerLiftSplatShootVoxel(grid_config=cfg, data_config=dict(input_size=size), downsample=16, numC_Trans=C).to(dev)
    cams = tuple(rig[k].to(dev) for k in ("rots", "trans", "intrins", "post_rots", "post_trans", "bda"))
    depth = torch.softmax(torch.randn(N, D, fH, fW, generator=g), 1).to(dev)
    feat = torch.randn(N, C, fH, fW, generator=g).to(dev)
    print("lift_splat %s %.3f ms" % (which, timeit(lambda: vt.lift_splat(depth, feat, cams=cams), n=10)))


def bench_bwd():
    """Conv backward at the r50 decoder shapes: forward, dgrad, wgrad (each 2*M*Cin*Cout*taps flop)."""
    from co_occ_amd import autograd as ag
    from co_occ_amd import _lib
    g = torch.Generator().manual_seed(6)
    for name, Cin, Cout, grid, k, stride in (("enc.l0", 128, 128, (100, 100, 8), 3, 1), ("fpn.out0", 256, 256, (100, 100, 8), 3, 1),
                                              ("con_enc.0", 512, 256, (100, 100, 8), 3, 1), ("enc.l1.conv1", 128, 256, (100, 100, 8), 3, 2),
                                              ("lat0 1x1", 128, 256, (100, 100, 8), 1, 1)):
        X, Y, Z = grid
        x = torch.randn(X * Y * Z, Cin, generator=g).to(dev).requires_grad_()
        w = (torch.randn(Cout, Cin, k, k, k, generator=g) / (Cin * k ** 3) ** 0.5).to(dev).requires_grad_()
        out, geom = ag.conv3d_rows(x, w, (1, X, Y, Z), stride=stride, relu=True)
        go = torch.randn_like(out)
        flop = 2.0 * out.shape[0] * Cin * Cout * k ** 3
        t_f = timeit(lambda: ag.conv3d_rows(x, w, (1, X, Y, Z), stride=stride, relu=True), n=5)
        _lib.TIMER.enabled = 2
        _lib.TIMER.reset()
        for _ in range(3):
            x.grad = w.grad = None
            out.backward(go, retain_graph=True)
        torch.cuda.synchronize()
        rows = _lib.TIMER.summary()
        _lib.TIMER.enabled = False
        parts = []
        for key in ("conv_dgrad", "k_wgrad", "coocc_conv_epilogue_bwd"):
            if key in rows:
                ms = rows[key]["ms"] / rows[key]["launches"]
                parts.append("%s %.3f ms%s" % (key.replace("coocc_conv_", ""), ms, " (%.0f TF)" % (flop / ms / 1e9) if key != "coocc_conv_epilogue_bwd" else ""))
        print("bwd %-13s %4d->%4d k%d s%d  fwd(+pack) %.3f ms | backward: %s" % (name, Cin, Cout, k, stride, t_f, "  ".join(parts)))


def bench_trunk():
    """Training step of the decoder trunk at the r50 workload: con_enc -> ResNet3D-18 -> FPN3D forward + backward
    (frozen-statistics BN), per-kernel-family times from the C-ABI timer."""
    from co_occ_amd import autograd as ag
    from co_occ_amd import _lib
    cfg = synth.model_cfg()
    model = pkg.build_detector(cfg).to(dev).eval()
    X, Y, Z = 100, 100, 8
    C = model.occ_fuser.in_channels
    x = torch.randn(X * Y * Z, 4 * C, device=dev).requires_grad_()
    def step():
        for p in model.parameters():
            p.grad = None
        x.grad = None
        outs = ag.trunk_forward_train(model.occ_fuser.con_enc, model.semantic_encoder, model.semantic_neck, x, (1, X, Y, Z))
        loss = sum(o.square().mean() for o, _ in outs)
        loss.backward()
    t = timeit(step, n=3, warm=1)
    _lib.TIMER.enabled = 2
    _lib.TIMER.reset()
    step()
    torch.cuda.synchronize()
    rows = _lib.TIMER.summary()
    _lib.TIMER.enabled = False
    print("trunk train step (fwd+bwd, r50 100x100x8): %.1f ms" % t)
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:16]:
        print("   %-44s %3d launches %8.2f ms%s" % (k, r["launches"], r["ms"], "  %.0f TFLOP/s" % (r["work"] / r["ms"] / 1e9) if r["work"] else ""))


def bench_trainfull():
    """Whole differentiable path at the r50 workload, forward + backward (frozen-statistics BN): con_enc -> ResNet3D-18 ->
    FPN3D -> OccHead coarse + fine (training-time top-k = 20000 coarse voxels, occ_head.py:187-196) + render block with
    its two losses.  The semantic losses (CE / lovasz / scal) are out of scope: squared-mean surrogates close the graph."""
    from co_occ_amd import autograd as ag
    from co_occ_amd import _lib
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
    model, _ = B.build_model("r50", dev)
    s = B.make_inputs("r50", 1234, dev, model)
    X, Y, Z = 100, 100, 8
    C = model.occ_fuser.in_channels
    x = torch.randn(X * Y * Z, 4 * C, device=dev).requires_grad_()
    g = torch.Generator().manual_seed(1)
    coarse = torch.randperm(X * Y * Z, generator=g)[:20000].sort().values.int().to(dev)
    N, D, fH, fW = s["gemo"].shape[1:5]
    rgb_gt = torch.rand(N * fH * 16 * fW * 16, 3, device=dev)
    depth_gt = torch.rand(N * fH * 16 * fW * 16, device=dev) * 50
    fuser, head = model.occ_fuser, model.pts_bbox_head
    def step():
        for p in model.parameters():
            p.grad = None
        x.grad = None
        ce = fuser.con_enc
        vf, geom = ag.conv3d_rows(x, ce[0].weight, (1, X, Y, Z), bias=ce[0].bias, bn=ce[1], relu=True)
        vf, geom = ag.conv3d_rows(vf, ce[3].weight, geom, bias=ce[3].bias, bn=ce[4], relu=True)
        levels = ag.trunk_forward_train(None, model.semantic_encoder, model.semantic_neck, vf, geom)
        out_rows, logits = ag.occhead_coarse_train(head, levels)
        fine, _ = ag.fine_branch_train(head, out_rows, (1, X, Y, Z), coarse, s["img_feats"], s["transform"])
        rgbs, depths = ag.render_block_train(model.sigma_head, model.rgb_head, vf, (X, Y, Z), s["gemo"])
        L = ag.render_losses(rgbs.reshape(-1, 3), depths.reshape(-1), rgb_gt, depth_gt, D)
        loss = logits.square().mean() + fine.square().mean() + L["loss_rgb"] + L["loss_depth_render"]
        loss.backward()
    t = timeit(step, n=3, warm=1)
    _lib.TIMER.enabled = 2
    _lib.TIMER.reset()
    step()
    torch.cuda.synchronize()
    rows = _lib.TIMER.summary()
    _lib.TIMER.enabled = False
    print("full-model train step (fwd+bwd, r50: trunk + coarse head + fine branch (20000 coarse voxels) + render + losses): %.1f ms" % t)
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:12]:
        print("   %-44s %3d launches %8.2f ms%s" % (k, r["launches"], r["ms"], "  %.0f TFLOP/s" % (r["work"] / r["ms"] / 1e9) if r["work"] else ""))


def bench_r3():
    """R3, the library renderer (P/utils/render_ray.py; dead code upstream) at the config's ray budget: N_rand = 4096 rays x
    N_samples = 64 (coocc_multi_r50_256x704.py:82-84), trilinear volume_sampling of the 128-channel 100x100x8 volume, heads per
    sample, raw2outputs.  Algorithmic bytes: N_rays*N_samples*(12 + 8*C*4) worst case for the gather."""
    from co_occ_amd import render as R
    g = torch.Generator().manual_seed(2)
    C, NR, NS = 128, 4096, 64
    feats = torch.randn(1, C, 8, 100, 100, generator=g).to(dev)          # [1,C,D,W,H]
    o = torch.tensor([0.0, 0.0, -1.0]).expand(NR, 3)
    d = torch.randn(NR, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    pts, z = R.sample_along_camera_ray(o.to(dev), d.to(dev), (0.2, 60.0), NS, det=True)
    aabb = (torch.tensor([-50.0, -50.0, -5.0]), torch.tensor([50.0, 50.0, 3.0]))
    t1 = timeit(lambda: R.volume_sampling(pts, feats, aabb), n=10)
    f, m = R.volume_sampling(pts, feats, aabb)
    raw = torch.randn(NR, NS, 4, generator=g).to(dev)
    t2 = timeit(lambda: R.raw2outputs(raw, z), n=10)
    by = NR * NS * (12.0 + 8 * C * 4) + NR * NS * C * 4
    print("R3  %d rays x %d samples: volume_sampling %.3f ms (%.0f GB/s of the worst-case gather + output bytes, %.0f%% of points in the box)"
          "  raw2outputs %.3f ms" % (NR, NS, t1, by / t1 / 1e6, 100.0 * float(m.float().mean()), t2))


def bench_lidar():
    """LiDAR producer at nuScenes scale: ~280 k points (10 sweeps) -> 0.125 m voxels on [800,800,64] -> 8x sparse encoder."""
    from co_occ_amd import lidar as L
    g = torch.Generator().manual_seed(8)
    n = 280000
    r = torch.rand(n, generator=g) ** 0.5 * 50
    th = torch.rand(n, generator=g) * 6.2832
    pts = torch.stack([r * torch.cos(th), r * torch.sin(th), torch.randn(n, generator=g) * 0.8 - 1.5, torch.rand(n, generator=g)], 1).to(dev)
    vox = L.Voxelization([0.125] * 3, [-50, -50, -5, 50, 50, 3], 10, (90000, 120000)).eval()
    vfe = L.HardSimpleVFE(5)
    enc = L.SparseLiDAREnc8x(4, dict(type="BN1d"), 16, 128, [800, 800, 64]).to(dev).eval()
    def run():
        v, c, k = vox(pts)
        return enc(vfe(v, k, c), c, 1), v.shape[0]
    (out, m) = run()
    t_v = timeit(lambda: vox(pts), n=5)
    t_all = timeit(lambda: run(), n=5)
    from co_occ_amd import _lib
    _lib.TIMER.enabled = 2
    _lib.TIMER.reset()
    run()
    torch.cuda.synchronize()
    rows = _lib.TIMER.summary()
    _lib.TIMER.enabled = False
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[:14]:
        print("   %-32s %3d launches %8.3f ms  %6.1f TFLOP/s (table entries counted dense)" % (k, v["launches"], v["ms"], v["work"] / max(v["ms"], 1e-9) / 1e9))
    print("lidar  %d points -> %d voxels: voxelise %.3f ms, voxelise + VFE + SparseLiDAREnc8x %.3f ms -> %s, %d active at 1/8" % (
        n, m, t_v, t_all, tuple(out["x"].shape), out["pts_feats"][0].coors.shape[0]))


def bench_eval():
    from co_occ_amd import evaluation as ev
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(1, 17, 100, 100, 8, generator=g).to(dev)
    gt = torch.randint(0, 17, (1, 200, 200, 16), generator=g).to(torch.uint8).to(dev)
    vis = torch.randint(0, 2, (1, 200, 200, 16), generator=g).to(torch.uint8).to(dev)
    out = torch.zeros(4 + 2 * 17 * 17, dtype=torch.int64, device=dev)
    t = timeit(lambda: ev.semantic_histograms(pred, gt, vis, out=out, accumulate=True), n=20)
    def upstream():
        up = torch.nn.functional.interpolate(pred, size=[200, 200, 16], mode='trilinear', align_corners=False)
        return torch.argmax(up[0], 0).cpu().numpy(), gt[0].cpu().numpy()
    t_u = timeit(upstream, n=5)
    print("eval  coarse->200x200x16: on-device SC+SSC+OCC %.3f ms ; upstream's interpolate+argmax+.cpu() alone %.3f ms (before 3 bincounts)" % (t, t_u))


def bench_fpsdbg():
    import ctypes
    from co_occ_amd import _lib
    lib = _lib.load()
    grid = tuple(int(v) for v in os.environ.get("COOCC_FPSDBG_GRID", "100,100,8").split(","))
    li, lp = voxel_lists(grid)
    dbg = torch.zeros(16 * 8, device=dev, dtype=torch.int64)
    lib.coocc_fps_voxels_set_debug(ctypes.c_void_p(dbg.data_ptr()))
    for name, lin in (("img", li), ("pts", lp)):
        dbg.zero_()
        fuser._fps_voxels(lin, grid, 2048)
        torch.cuda.synchronize()
        d = dbg.view(16, 8).cpu()
        print(name, "per-iteration cycles by wave [test+refresh, wave max, barrier, tail | dirty buckets total]")
        for w in range(16):
            if int(d[w].sum()):
                print("  wave %2d: %s  dirty=%d" % (w, " ".join("%7.0f" % (float(v) / 2047) for v in d[w, :4]), int(d[w, 5])))
    lib.coocc_fps_voxels_set_debug(ctypes.c_void_p(0))


def bench_finemlp():
    """coocc_fine_mlp alone at the r50 worst case (every coarse voxel foreground: 640 000 fine points)."""
    from co_occ_amd._lib import call, ptr
    nf, ncls = 640000, 17
    g = torch.Generator().manual_seed(3)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    samp, vox = mk(nf, 128), mk(nf, 128)
    ws = [mk(64, 128) * 0.1, mk(64), mk(64), mk(64), mk(64, 192) * 0.1, mk(64), mk(64), mk(64), mk(ncls, 64) * 0.1, mk(ncls)]
    out = torch.empty(nf, ncls, device=dev)
    def run():
        call("coocc_fine_mlp", ptr(samp), 128, ptr(vox), 128, nf, ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(ws[3]), 1e-5,
             ptr(ws[4]), ptr(ws[5]), ptr(ws[6]), ptr(ws[7]), 1e-5, ptr(ws[8]), ptr(ws[9]), ncls, ptr(out))
    t = timeit(run, n=20)
    fl = 2.0 * nf * 64 * (128 + 192 + 32)
    print("fine_mlp  %d points: %.3f ms  %.1f TFLOP/s executed (%.0f GB/s of samples)" % (nf, t, fl / t / 1e9, nf * 1024 / t / 1e6))


if __name__ == "__main__":
    which = sys.argv[1:] or ["fps", "knn", "conv"]
    with torch.no_grad():
        for w in which:
            with torch.set_grad_enabled(w in ("bwd", "trunk", "trainfull")):
                globals()["bench_" + w]()
