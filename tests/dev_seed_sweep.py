"""Developer script (not collected): scale-relative error of the shipped conv dispatch vs the direct path against the
oracle over several seeds of the 50x50x8 end-to-end scene.  python tests/dev_seed_sweep.py"""
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
for seed in (5, 6, 7, 8, 9, 10):
    model = pkg.build_detector(cfg)
    sd = synth.random_state_dict(model.state_dict(), seed=seed)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    img, pts = synth.voxel_inputs(grid, C=C, seed=70 + seed)
    rig = synth.camera_rig(6, (64, 176), seed=70 + seed)
    img_feats = [synth.image_feats(6, (4, 11), 512, seed=70 + seed)]
    tr = synth.rig_transform(rig)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    vf = ref_cpu.bifuser_forward(sub("occ_fuser."), img, pts, knum)
    sem = ref_cpu.fpn3d_forward(sub("semantic_neck."), ref_cpu.resnet3d_forward(sub("semantic_encoder."), vf))
    h = ref_cpu.occhead_forward(sub("pts_bbox_head."), sem, img_feats, tr, 2, (100, 100, 16), (-25, -25, -5.0, 25, 25, 3.0))
    line = "seed %2d" % seed
    for name, wino in (("direct", 0), ("shipped", 1)):
        core.WINO, core.WINO_MIN_ROWS = wino, 0
        with torch.no_grad():
            out = model.forward_hot_path(img.to(dev), pts.to(dev), None, [img_feats[0].to(dev)],
                                         tuple(t.to(dev) if torch.is_tensor(t) else t for t in tr), render=False)
        a = set(map(tuple, out["output_coords_fine"][0].cpu().t().tolist()))
        b = set(map(tuple, h["fine_coord"].t().tolist()))
        fe = rel_err(out["output_voxels_fine"][0].cpu(), h["fine_output"]) if a == b else float("nan")
        line += " | %s vf %.1e pred_c %.1e fine %.1e (coords differ %d)" % (name, rel_err(out["voxel_feats"].cpu(), vf),
                                                                            rel_err(out["pred_c"].cpu(), h["output_voxels"]), fe, len(a ^ b))
    print(line, flush=True)
