"""Debug: is k_fine2_h2 deterministic run to run, eager vs device-count form, and vs the three-kernel path?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import head as H

dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
samples = [bench.make_inputs("r50", 4000 + 13 * i, dev, model) for i in range(3)]
head = model.pts_bbox_head
with torch.no_grad():
    for si, s in enumerate(samples):
        vol = model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"])
        vf = model.fuse(vol, s["pts"])
        neck = model.semantic_neck
        mid = model.semantic_encoder.forward_rows(vf, readers=neck.lateral_packs())
        sem = neck.forward_rows(mid, readers=head.level_readers())
        outs = {}
        for tag, on, static in (("three", False, False), ("f2_a", True, False), ("f2_b", True, False), ("f2_static", True, True), ("f2_static2", True, True)):
            H.FINE2_H2 = on
            r = head(voxel_feats=sem, img_feats=s["img_feats"], transform=s["transform"], static=static)
            if static:
                n = int(r["fine_count"].item()) * 8
                outs[tag] = (r["output_voxels_fine"][0][:n].clone(), r["output_coords_fine"][0].reshape(-1)[:3 * n].view(3, n).clone())
            else:
                outs[tag] = (r["output_voxels_fine"][0].clone(), r["output_coords_fine"][0].clone())
        torch.cuda.synchronize()
        a = outs["f2_a"]
        print("sample", si, "points", a[0].shape[0])
        for tag in ("f2_b", "f2_static", "f2_static2", "three"):
            b = outs[tag]
            d = (a[0] - b[0]).abs()
            bad = (d > 0).any(1).nonzero().flatten()
            print("  f2_a vs %-10s xyz equal %s  logits equal %s  max diff %.3e  rows differing %d %s" % (
                tag, torch.equal(a[1], b[1]), torch.equal(a[0], b[0]), float(d.max()), bad.numel(), bad[:8].tolist()))
