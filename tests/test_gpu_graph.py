"""The captured dense stage (co_occ_amd.graph.DenseGraph: one hipGraphLaunch per sample, every data-dependent count read on
the device) against the eager path on the same samples, bit for bit."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _eager(model, s, dev):
    import bench
    with torch.no_grad():
        img = bench.pool(model, s)
        sr = model.search(img, s["pts"])
        out = model.forward_hot_path(img, s["pts"], s["gemo"], s["img_feats"], s["transform"], render=True, search=sr)
    torch.cuda.synchronize()
    return {k: out[k].clone() for k in ("pred_c", "pred_f", "rgbs", "depths")}, sr.counts


def test_dense_graph_replay_equals_eager(dev):
    import bench
    from co_occ_amd import graph as cg
    bench.CFGNAME[0] = "r50"
    model, _ = bench.build_model("r50", dev)
    a = bench.make_inputs("r50", 1234, dev, model)
    b = bench.make_inputs("r50", 99, dev, model)
    # the graph binds the per-sample tensors it reads besides the slot: give sample b the rig of sample a
    for k in ("gemo", "img_feats", "transform", "cams"):
        b[k] = a[k]
    want_a, counts_a = _eager(model, a, dev)
    want_b, counts_b = _eager(model, b, dev)
    assert counts_a != counts_b
    X, Y, Z = a["pts"].shape[2:]
    slot = cg.make_slot(model, (X, Y, Z), dev)
    stream = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream(dev)

    def search(s):
        with torch.no_grad():
            sr = cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
        sr.done_main.wait(stream)
        sr.done_side.wait(stream)
        return sr

    sr = search(a)
    g = cg.DenseGraph(model, slot, a, stream).capture()
    assert g.fits(sr.counts)
    for s, want, counts in ((a, want_a, counts_a), (b, want_b, counts_b), (a, want_a, counts_a)):
        cur.wait_stream(stream)          # the slot is rewritten only after the previous replay has finished with it
        sr = search(s)
        assert sr.counts == counts and g.fits(sr.counts)
        with torch.cuda.stream(stream):
            out = g.replay()
        stream.synchronize()
        n = int(out["fine_count"].item())
        assert n > 0
        for k in ("pred_c", "pred_f", "rgbs", "depths"):
            assert torch.equal(out[k], want[k]), "%s differs between the graph replay and the eager path" % k


def test_native_search_equals_python_search(dev):
    """coocc_fuser_search (the whole index-search stage issued from C++, csrc/search.hip) against BiFuser_N.search: voxel lists,
    counts, neighbour ordinals and row tables bit for bit, concat rows included."""
    import bench
    from co_occ_amd import graph as cg
    bench.CFGNAME[0] = "r50"
    model, _ = bench.build_model("r50", dev)
    s = bench.make_inputs("r50", 4321, dev, model)
    X, Y, Z = s["pts"].shape[2:]
    f = model.occ_fuser
    with torch.no_grad():
        a, b = cg.make_slot(model, (X, Y, Z), dev), cg.make_slot(model, (X, Y, Z), dev)
        for slot in (a, b):
            model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"], out=slot.img_rows())
        sa = f.search(a.img_rows().as_ncdhw(), s["pts"], slot=a)
        sb = f.search_native(s["pts"], b)
    torch.cuda.synchronize()
    assert sa.counts == sb.counts and min(sa.counts) > 2048
    assert torch.equal(sa.lin_img, sb.lin_img) and torch.equal(sa.lin_pts, sb.lin_pts)
    assert torch.equal(sa.near_img, sb.near_img) and torch.equal(sa.near_pts, sb.near_pts)
    assert torch.equal(sa.rows, sb.rows) and torch.equal(sa.rows_p, sb.rows_p)
    C = f.in_channels
    assert torch.equal(a.cat4[:, :2 * C], b.cat4[:, :2 * C])
    assert torch.equal(a.counts, b.counts)
