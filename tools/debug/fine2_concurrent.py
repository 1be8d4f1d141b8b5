"""Debug: k_fine2_h2 on two streams at once against its single-stream result (the pipelined loop's failure, without the pipeline)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import head as H

dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
samples = [bench.make_inputs("r50", 4000 + 13 * i, dev, model) for i in range(4)]
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

    def run(i):
        sem, imf, tr = ins[i]
        r = head(voxel_feats=sem, img_feats=imf, transform=tr, static=True)
        return r["output_voxels_fine"][0], r["fine_count"]

    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    H.FINE2_H2 = True
    orig_ff = H.OccHead._fine_fused
    for variant in sys.argv[1:] or ["persistent"]:
        # variants: persistent | full (one tile per workgroup) | full+sync (host synchronisation in front of the kernel)
        os.environ["COOCC_FINE2_GRID"] = "full" if variant.startswith("full") else "persistent"
        os.environ["COOCC_FINE2_DBG"] = "1" if "+inv" in variant else "0"
        if variant.endswith("+sync"):
            def synced(self, *a, **k):
                torch.cuda.synchronize()
                return orig_ff(self, *a, **k)
            H.OccHead._fine_fused = synced
        else:
            H.OccHead._fine_fused = orig_ff
        ref = []
        for i in range(4):
            lg, c = run(i)
            torch.cuda.synchronize()
            n = int(c.item()) * 8
            ref.append(lg[:n].clone())
        for nstreams in (1, 2, 3):
            outs = []
            for it in range(12):
                for si in range(nstreams):
                    with torch.cuda.stream(streams[si]):
                        i = (it + si) % 4
                        lg, c = run(i)
                        outs.append((i, lg, si))
            torch.cuda.synchronize()
            bad = 0
            kids = set()
            for i, lg, si in outs:
                n = ref[i].shape[0]
                d = (lg[:n] - ref[i]).abs()
                rows = (d > 0).any(1).nonzero().flatten()
                if rows.numel():
                    bad += 1
                    kids.update((rows // (n // 8)).tolist())
            print("variant %-12s streams %d: %2d of %2d calls differ, children %s" % (variant, nstreams, bad, len(outs), sorted(kids)), flush=True)
