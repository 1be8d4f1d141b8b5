"""Debug: which co-running work makes k_fine2_h2's image samples differ?  One stream runs the head (fine2 on), the other a loop of
unrelated kernels: torch matmuls, torch elementwise ops, one of this package's convolutions, or the head with fine2 off."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import head as H, core

dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
samples = [bench.make_inputs("r50", 4000 + 13 * i, dev, model) for i in range(2)]
head = model.pts_bbox_head
ins = []
with torch.no_grad():
    for s in samples:
        vol = model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"])
        vf = model.fuse(vol, s["pts"])
        neck = model.semantic_neck
        mid = model.semantic_encoder.forward_rows(vf, readers=neck.lateral_packs())
        sem = neck.forward_rows(mid, readers=head.level_readers())
        ins.append((sem, s["img_feats"], s["transform"]))
    torch.cuda.synchronize()

    def run(i, fine2=True):
        H.FINE2_H2 = fine2
        sem, imf, tr = ins[i]
        r = head(voxel_feats=sem, img_feats=imf, transform=tr, static=True)
        return r["output_voxels_fine"][0], r["fine_count"]

    ref = []
    for i in range(2):
        lg, c = run(i)
        torch.cuda.synchronize()
        ref.append(lg[:int(c.item()) * 8].clone())
    a, b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    big = torch.randn(64 << 20, device=dev)
    enc = model.semantic_encoder
    s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def co_mm():
        for _ in range(6):
            torch.mm(a, b)

    def co_elem():
        for _ in range(40):
            big.mul_(1.0001)

    def co_conv():
        sem, imf, tr = ins[1]
        for _ in range(3):
            neck.forward_rows(enc.forward_rows(vf_keep, readers=neck.lateral_packs()), readers=head.level_readers())

    def co_head3():
        run(1, fine2=False)

    vf_keep = model.fuse(model.img_view_transformer.lift_splat(samples[1]["depth"], samples[1]["ctx"], cams=samples[1]["cams"]), samples[1]["pts"])
    torch.cuda.synchronize()
    def with_engine(engine, wino):
        def f():
            keep = core.CONV_ENGINE, core.WINO
            core.CONV_ENGINE, core.WINO = engine, int(wino)
            try:
                co_conv()
            finally:
                core.CONV_ENGINE, core.WINO = keep
        return f

    for name, co, inside in (("nothing", None, True), ("decoder convolutions (h2 engine)", co_conv, True),
                             ("nothing; image samples from the grouped sampler's launch", None, False),
                             ("decoder convolutions (h2 engine); image samples from the grouped sampler's launch", co_conv, False)):
        H.FINE2_IMG_INSIDE = inside
        ref = []
        for i in range(2):
            lg, c = run(i)
            torch.cuda.synchronize()
            ref.append(lg[:int(c.item()) * 8].clone())
        outs = []
        for it in range(20):
            if co is not None:
                with torch.cuda.stream(s1):
                    co()
            with torch.cuda.stream(s0):
                outs.append(run(0)[0])
        torch.cuda.synchronize()
        bad = sum(int(not torch.equal(o[:ref[0].shape[0]], ref[0])) for o in outs)
        print("co-runner %-90s: %2d of %2d fine2 calls differ" % (name, bad, len(outs)), flush=True)
