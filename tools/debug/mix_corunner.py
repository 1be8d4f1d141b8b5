"""Debug: does the half-z-column OccHead mix (COOCC_INTERP_COLUMN bit 1, csrc/interp.hip) differ from its own result when other work
shares the GPU?  Inside the serving loop it did (profiles/r5_interp_column_bisect.txt: pred_c of the pipelined loop != eager calls in
3 of 3 runs, the per-voxel kernel 0 of 3).  Here the kernel is called through the C ABI on FIXED buffers (inputs made once, one
preallocated output per call: no allocator, no graph, no pipeline), stream s0; stream s1 loops a co-runner.  Every output is compared
with the kernel's own result computed alone.

    python tools/debug/mix_corunner.py            # both forms x every co-runner
    python tools/debug/mix_corunner.py --bisect 2 # the column form next to ONE split-f16 kernel family at a time
    COOCC_MIX_COL_VAR=1 python tools/debug/mix_corunner.py --bisect 2   # the same source with its stores reversed: 0 of 20 (round 5)
    COOCC_MIX_COL_VAR=2 python tools/debug/mix_corunner.py --bisect 2   # s_nop fences after every division / accumulator update (not run yet)

Read it next to tools/debug/fine2_corunner.py (the one-launch ratio-2 fine branch): if the column form differs here
too, the defect is at kernel level (co-residency), not a race between the slots' graphs."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import core
from co_occ_amd._lib import call, ptr, host_i32

dev = torch.device("cuda:0")
C, sizes = 128, [(100, 100, 8), (50, 50, 4), (25, 25, 2), (13, 13, 1)]
g = torch.Generator().manual_seed(7)
levels = [torch.randn(s[0] * s[1] * s[2], C, generator=g).to(dev) for s in sizes]
wlogit = (torch.randn(sizes[0][0] * sizes[0][1] * sizes[0][2], 4, generator=g) * 3).to(dev)
arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in levels])
dims = host_i32([v for s in sizes for v in s])
N = 20
outs = [torch.empty_like(levels[0]) for _ in range(N)]
twins = [torch.empty_like(levels[0]) for _ in range(N)]


def mix(i):
    call("coocc_occhead_mix_ex", arr, dims, 4, ptr(wlogit), ptr(outs[i]), 1, C, ptr(twins[i]))


bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
s = bench.make_inputs("r50", 4013, dev, model)
with torch.no_grad():
    vf = model.fuse(model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"]), s["pts"])
    enc, neck, head = model.semantic_encoder, model.semantic_neck, model.pts_bbox_head
    a, b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    big = torch.randn(64 << 20, device=dev)
    torch.cuda.synchronize()

    def co_conv(engine):
        def f():
            keep = core.CONV_ENGINE
            core.CONV_ENGINE = engine
            try:
                for _ in range(2):
                    neck.forward_rows(enc.forward_rows(vf, readers=neck.lateral_packs()), readers=head.level_readers())
            finally:
                core.CONV_ENGINE = keep
        return f

    def co_mm():
        for _ in range(6):
            torch.mm(a, b)

    def co_elem():
        for _ in range(40):
            big.mul_(1.0001)

    # --bisect: one kernel family of the split-f16 engine at a time as the co-runner (not run on the GPU yet: written at the end of
    # round 5 when the budget was spent).  x: 80 000 rows x 128 channels; WINO / CONV_ENGINE are read per call by conv_rows.
    gb = torch.Generator().manual_seed(11)
    xb = core.to_rows(torch.randn(1, 128, 100, 100, 8, generator=gb).to(dev))
    pc3 = core.PackedConv((torch.randn(128, 128, 3, 3, 3, generator=gb) * 0.02).to(dev), ksize=3, pad=1)
    pc1 = core.PackedConv((torch.randn(128, 128, 1, 1, 1, generator=gb) * 0.05).to(dev), ksize=1, pad=0)

    def co_layer(pc, engine, wino, n=4):
        def f():
            keep = core.CONV_ENGINE, core.WINO
            core.CONV_ENGINE, core.WINO = engine, wino
            try:
                for _ in range(n):
                    core.conv_rows(xb, pc, relu=False)
            finally:
                core.CONV_ENGINE, core.WINO = keep
        return f

    def co_rows_to_h2():
        for _ in range(20):
            core.rows_to_h2(xb.t, 128, 0, name="corunner_h2")

    bisect = [("nothing", None),
              ("k_rows_to_h2 only (fp32 rows -> H2 rows, no MFMA, no LDS)", co_rows_to_h2),
              ("1x1x1 128 -> 128, split-f16 engine (k_gemm_h2p / k_gemm_h2w)", co_layer(pc1, "h2", 0)),
              ("3x3x3 128 -> 128 direct, split-f16 engine (k_gemm_h2z<3,true>: global_load_lds A image)", co_layer(pc3, "h2", 0)),
              ("3x3x3 128 -> 128 Winograd, split-f16 engine (k_wino_in_h2 + k_gemm_h2z<3,false> + k_wino_out)", co_layer(pc3, "h2", 1)),
              ("3x3x3 128 -> 128 Winograd, fp32-MFMA engine (k_wino_in + k_conv + k_wino_out)", co_layer(pc3, "f32", 1)),
              ("3x3x3 128 -> 128 direct, fp32-MFMA engine", co_layer(pc3, "f32", 0))]
    full = [("nothing", None), ("decoder convolutions, split-f16 engine (global_load_lds GEMMs)", co_conv("h2")),
            ("decoder convolutions, fp32-MFMA engine", co_conv("f32")), ("torch.mm 4096^3", co_mm),
            ("torch elementwise over 256 MB", co_elem)]
    argv = [a for a in sys.argv[1:] if a != "--bisect"]
    corunners = bisect if "--bisect" in sys.argv else full

    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    for mask in (argv or ["2", "6", "0"]):       # 2 = half-column mix, 6 = the same with s_waitcnt vmcnt(0) after every load batch, 0 = per-voxel
        os.environ["COOCC_INTERP_COLUMN"] = mask
        with torch.cuda.stream(s0):
            mix(0)
        torch.cuda.synchronize()
        ref, ref_tw = outs[0].clone(), twins[0].clone()
        for name, co in corunners:
            for o, t in zip(outs, twins):
                o.fill_(float("nan")); t.fill_(float("nan"))
            torch.cuda.synchronize()
            for i in range(N):
                if co is not None:
                    with torch.cuda.stream(s1):
                        co()
                with torch.cuda.stream(s0):
                    mix(i)
            torch.cuda.synchronize()
            bad = sum(int(not (torch.equal(o, ref) and torch.equal(t, ref_tw))) for o, t in zip(outs, twins))
            rows = sorted({int(r) for o in outs for r in torch.nonzero((o != ref).any(1))[:8, 0].tolist()})[:8]
            print("COOCC_INTERP_COLUMN=%s  co-runner %-70s: %2d of %2d calls differ%s" %
                  (mask, name, bad, N, ("  first rows " + str(rows)) if rows else ""), flush=True)
