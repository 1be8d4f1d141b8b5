"""Debug: where do the pipelined loop's fine logits differ from the eager call's when csrc/fine2_h2.hip is on?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import apis

dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
model.test_rendering = True
samples = [bench.make_inputs("r50", 4000 + 13 * i, dev, model) for i in range(8)]
g = torch.Generator().manual_seed(5)
gts = [torch.randint(0, 17, (1, 200, 200, 16), generator=g).to(dev) for _ in range(8)]
kws = [dict(bench.simple_test_kwargs(s), gt_occ=gg) for s, gg in zip(samples, gts)]


def grab(out):
    return dict(pred_f=out["pred_f"].clone(), pred_c=out["pred_c"].clone(), fine=out["output_voxels_fine"][0].clone(),
                xyz=out["output_coords_fine"][0].reshape(3, -1).clone())


with torch.no_grad():
    model.graph_simple_test = False
    ref = [grab(model.simple_test(**kw)) for kw in kws]
    ref2 = [grab(model.simple_test(**kw)) for kw in kws]
    print("eager vs eager:", [bool(torch.equal(a["fine"], b["fine"]) and torch.equal(a["pred_f"], b["pred_f"])) for a, b in zip(ref, ref2)])
    data = [dict(precomputed=kw["precomputed"], gt_occ=kw["gt_occ"]) for kw in kws] * 2
    for slots, ds in ((4, 2), (4, 1), (1, 1), (6, 3)):
        print("== slots %d dense_streams %d" % (slots, ds))
        for i, (d, res) in enumerate(apis.pipelined_test(model, iter(data), slots=slots, dense_streams=ds)):
            got = grab(res)
            want = ref[i % 8]
            eq = {k: bool(got[k].shape == want[k].shape and torch.equal(got[k], want[k])) for k in got}
            if all(eq.values()):
                continue
            msg = "  sample %2d: %s" % (i, eq)
            if got["fine"].shape == want["fine"].shape:
                dd = (got["fine"] - want["fine"]).abs()
                bad = (dd > 0).any(1).nonzero().flatten()
                n = got["fine"].shape[0] // 8
                msg += "  rows differing %d of %d, max %.3e, coarse idx (row %% n) %s, child %s" % (
                    bad.numel(), got["fine"].shape[0], float(dd.max()), (bad % n)[:12].tolist(), (bad // n)[:12].tolist())
                nanrows = torch.isnan(got["fine"]).any(1).sum().item()
                msg += " nan rows %d" % nanrows
            else:
                msg += "  shapes %s vs %s" % (tuple(got["fine"].shape), tuple(want["fine"].shape))
            print(msg)
