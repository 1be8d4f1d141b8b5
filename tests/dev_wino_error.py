"""Developer script (not collected): end-to-end error of the conv paths vs the oracle on the 50x50x8 scene of
test_hot_path_end_to_end_vs_oracle.  python tests/dev_wino_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import co_occ_amd as pkg, co_occ_amd.synth as synth
from co_occ_amd import core
from oracle import ref_cpu
from util import rel_err
dev = torch.device("cuda:0")
grid, C, knum = (50, 50, 8), 128, 2
cfg = synth.model_cfg(C=C, knum=knum, final_occ_size=(100, 100, 16), point_cloud_range=(-25, -25, -5.0, 25, 25, 3.0))
model = pkg.build_detector(cfg)
sd = synth.random_state_dict(model.state_dict(), seed=5)
model.load_state_dict(sd)
model = model.to(dev).eval()
img, pts = synth.voxel_inputs(grid, C=C, seed=77)
rig = synth.camera_rig(6, (64, 176), seed=77)
img_feats = [synth.image_feats(6, (4, 11), 512, seed=77)]
tr = synth.rig_transform(rig)
sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
vf = ref_cpu.bifuser_forward(sub("occ_fuser."), img, pts, knum)
sem = ref_cpu.fpn3d_forward(sub("semantic_neck."), ref_cpu.resnet3d_forward(sub("semantic_encoder."), vf))
h = ref_cpu.occhead_forward(sub("pts_bbox_head."), sem, img_feats, tr, 2, (100, 100, 16), (-25, -25, -5.0, 25, 25, 3.0))
def packs():
    f = model.occ_fuser._packed()
    e = model.semantic_encoder._packed()
    n = model.semantic_neck._packed()
    hd = model.pts_bbox_head
    out = dict(con0=f["c0"], con3=f["c1"] if "c1" in f else None)
    out["enc"] = [b[k] for layer in e["blocks"] for b in layer for k in ("c1", "c2")]
    out["fpn"] = n["out"]
    return out, f, hd

P, f, hd = packs()
print("fuser pack keys", list(f.keys()))
def set_tiles(con, enc, fpn):
    for k, v in f.items():
        if hasattr(v, "wino_tile"):
            v.wino_tile = (3 if k == "c0" else 2) if con == "c0=3" else con
    for pc in P["enc"]:
        pc.wino_tile = enc
    for pc in P["fpn"]:
        pc.wino_tile = fpn

for name, wino, tile, tiles in (("direct", 0, 4, None), ("F2 all", 1, 2, None), ("F4 all", 1, 4, None),
                                ("con F2, rest F4", 1, 4, (2, 4, 4)), ("con F4, rest F2", 1, 2, (4, 2, 2)),
                                ("con+enc F2, fpn/head F4", 1, 4, (2, 2, 4)), ("con.. F4, fpn F2", 1, 4, (4, 4, 2)), ("con F3, rest F4", 1, 4, (3, 4, 4)),
                                ("con.0 F3, con.3 F2, rest F4", 1, 4, ("c0=3", 4, 4)), ("F3 all", 1, 3, None)):
    core.WINO, core.WINO_TILE, core.WINO_MIN_ROWS = wino, tile, 0
    set_tiles(*(tiles or (None, None, None)))
    with torch.no_grad():
        out = model.forward_hot_path(img.to(dev), pts.to(dev), None, [img_feats[0].to(dev)],
                                     tuple(t.to(dev) if torch.is_tensor(t) else t for t in tr), render=False)
    a = set(map(tuple, out["output_coords_fine"][0].cpu().t().tolist()))
    b = set(map(tuple, h["fine_coord"].t().tolist()))
    fe = rel_err(out["output_voxels_fine"][0].cpu(), h["fine_output"]) if a == b else float("nan")
    print("%-26s voxel_feats %.2e  pred_c %.2e  fine %.2e  coord diff %d" % (
        name, rel_err(out["voxel_feats"].cpu(), vf), rel_err(out["pred_c"].cpu(), h["output_voxels"]), fe, len(a ^ b)))
