"""``co_occ_amd.serving`` behind the reference's calls: ``COOCC_Ray.simple_test`` with its dense stage as one captured hipGraph
launch, and the pipelined test loop, against the eager path -- 8 DIFFERENT frames (inputs, camera rigs and image features all
change), bit for bit."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

KEYS = ("pred_c", "pred_f", "rgbs", "depths", "output_voxels")


def _setup(dev, n=8):
    import bench
    bench.CFGNAME[0] = "r50"
    model, _ = bench.build_model("r50", dev)
    model.test_rendering = True
    samples = [bench.make_inputs("r50", 4000 + 13 * i, dev, model) for i in range(n)]
    g = torch.Generator().manual_seed(5)
    gts = [torch.randint(0, 17, (1, 200, 200, 16), generator=g).to(dev) for _ in range(n)]
    return bench, model, samples, gts


def _grab(out):
    d = {k: out[k].clone() for k in KEYS}
    d["fine"] = out["output_voxels_fine"][0].clone()
    d["fine_xyz"] = out["output_coords_fine"][0].reshape(3, -1).clone()
    for k in ("SC_metric", "SSC_metric", "SSC_metric_fine"):
        d[k] = out[k].copy()
    return d


def _same(a, b, tag):
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), "%s: %s differs" % (tag, k)
        else:
            assert (a[k] == b[k]).all(), "%s: %s differs" % (tag, k)


def test_simple_test_with_graphs_equals_eager_on_eight_different_frames(dev):
    bench, model, samples, gts = _setup(dev)
    kws = [dict(bench.simple_test_kwargs(s), gt_occ=g) for s, g in zip(samples, gts)]
    with torch.no_grad():
        model.graph_simple_test = False
        ref = [_grab(model.simple_test(**kw)) for kw in kws]
        assert model._pipe1 is None
        model.graph_simple_test = True
        got = [_grab(model.simple_test(**kw)) for kw in kws]
        again = _grab(model.simple_test(**kws[2]))              # an earlier frame again, after the slot has seen five others
    assert model._pipe1 is not None and model.graph_unavailable is None, model.graph_unavailable
    assert model._pipe1[1].fallbacks == 0
    for i, (a, b) in enumerate(zip(ref, got)):
        _same(a, b, "frame %d" % i)
    _same(ref[2], again, "frame 2 again")
    assert not torch.equal(ref[0]["pred_c"], ref[1]["pred_c"]) and not torch.equal(ref[0]["rgbs"], ref[3]["rgbs"])
    assert int(ref[0]["SSC_metric"].sum()) > 0


def test_pipelined_test_loop_equals_per_sample_calls(dev):
    """``apis.pipelined_test`` = custom_single_gpu_test's loop with several samples in flight: same results, same order."""
    from co_occ_amd import apis
    bench, model, samples, gts = _setup(dev)
    kws = [dict(bench.simple_test_kwargs(s), gt_occ=g) for s, g in zip(samples, gts)]
    with torch.no_grad():
        model.graph_simple_test = False
        ref = [_grab(model.simple_test(**kw)) for kw in kws]
    data = [dict(precomputed=kw["precomputed"], gt_occ=kw["gt_occ"]) for kw in kws] * 2          # 16 samples through 4 slots
    n = 0
    for i, (d, res) in enumerate(apis.pipelined_test(model, iter(data), slots=4, dense_streams=2)):
        assert d is data[i]
        got = {k: res[k].clone() for k in KEYS}
        for k in ("SC_metric", "SSC_metric", "SSC_metric_fine"):
            got[k] = res[k]
        want = {k: ref[i % 8][k] for k in got}
        _same(want, got, "sample %d" % i)
        n += 1
    assert n == 16


def test_scatter_form_con_enc0_survives_three_graphs_in_flight(dev, monkeypatch):
    """Regression (round 5): grids above ``fuser.DENSE_C0_MAX_VOXELS`` run ``con_enc.0`` split by channel support, whose captured
    chain resets a voxel -> ordinal map.  As a ``hipMemsetAsync`` NODE that reset left stale bytes when several graphs replayed
    concurrently (garbage ordinals in ``k_sparse_tap_sum``: ``bench.py --config openocc / stress200`` aborted); it is a fill
    kernel now.  Forced here on the 100x100x8 grid: 24 samples through 6 slots / 3 dense streams, bit-equal to eager calls."""
    from co_occ_amd import apis, fuser
    monkeypatch.setattr(fuser, "DENSE_C0_MAX_VOXELS", 1000)
    bench, model, samples, gts = _setup(dev)
    kws = [dict(bench.simple_test_kwargs(s), gt_occ=g) for s, g in zip(samples, gts)]
    with torch.no_grad():
        model.graph_simple_test = False
        ref = [_grab(model.simple_test(**kw)) for kw in kws]
    data = [dict(precomputed=kw["precomputed"], gt_occ=kw["gt_occ"]) for kw in kws] * 3
    n = 0
    for i, (d, res) in enumerate(apis.pipelined_test(model, iter(data), slots=6, dense_streams=3)):
        got = {k: res[k].clone() for k in KEYS}
        _same({k: ref[i % 8][k] for k in got}, got, "sample %d" % i)
        n += 1
    assert n == 24


def test_serving_takes_a_pooled_camera_volume_too(dev):
    """``precomputed=dict(img_voxel_feats=...)`` (the camera volume already pooled) through the captured form."""
    bench, model, samples, gts = _setup(dev, n=2)
    with torch.no_grad():
        outs = []
        for flag in (False, True):
            model.graph_simple_test = flag
            o = []
            for s in samples:
                vol = model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"]).contiguous()
                r = model.simple_test(img=None, precomputed=dict(img_voxel_feats=vol, pts_voxel_feats=s["pts"], img_feats=s["img_feats"],
                                                                transform=s["transform"], gemo=s["gemo"]))
                o.append({k: r[k].clone() for k in KEYS})
            outs.append(o)
    assert model._pipe1 is not None, model.graph_unavailable
    for a, b in zip(*outs):
        _same(a, b, "pooled volume")


def test_pipeline_takes_raw_lidar_points_and_matches_the_volume_it_would_have_been_given(dev):
    """Frames that carry raw LiDAR points: the producer (Voxelization -> HardSimpleVFE -> SparseLiDAREnc8x, coocc_ray.py:215-234)
    runs in the search stage; the outputs equal those of frames that carry the producer's volume."""
    import bench
    from co_occ_amd import synth
    bench.CFGNAME[0] = "r50"
    model, _ = bench.build_model("r50", dev, with_lidar=True)
    samples = [bench.make_inputs("r50", 4100 + 7 * i, dev, model) for i in range(3)]
    pts = [synth.lidar_points(n=60000, seed=20 + i).to(dev) for i in range(3)]
    with torch.no_grad():
        vols = [model.extract_pts_feat(p)[0] for p in pts]
    base = [dict(bench.frame_of(s)) for s in samples]
    fr_pts, fr_vol = [], []
    for f, p, v in zip(base, pts, vols):
        f.pop("pts", None)
        fr_pts.append(dict(f, points=p))
        fr_vol.append(dict(f, pts=v))
    outs = []
    for frames in (fr_vol, fr_pts):
        pipe = model.serving(frames[0], slots=2, dense_streams=1)
        res = []
        for f in frames:
            o = pipe.submit(f, copy=True).result()
            res.append({k: o[k].clone() for k in ("pred_c", "rgbs", "depths")})
        pipe.drain()
        pipe.close()
        outs.append(res)
    for i, (a, b) in enumerate(zip(*outs)):
        for k in a:
            assert torch.equal(a[k], b[k]), "frame %d: %s differs" % (i, k)


def test_simple_test_recaptures_when_the_weights_change(dev):
    """the captured dense stage bakes pointers to the weight packs of capture time.  After an in-place update
    (an optimizer step between ``train()`` and ``eval()``), ``load_state_dict`` or ``invalidate_packs`` the next ``simple_test``
    must serve the NEW weights: re-capture, then equal the eager path bit for bit."""
    import co_occ_amd
    bench, model, samples, gts = _setup(dev, n=2)
    kws = [dict(bench.simple_test_kwargs(s), gt_occ=g) for s, g in zip(samples, gts)]
    with torch.no_grad():
        model.graph_simple_test = True
        before = _grab(model.simple_test(**kws[0]))
        pipe = model._pipe1[1]
        assert pipe.recaptures == 0
        again = _grab(model.simple_test(**kws[0]))
        assert pipe.recaptures == 0, "an unchanged model must not re-capture (a buffer is rewritten by the forward pass?)"
        _same(before, again, "unchanged weights")
        # 1. in-place parameter updates, as an optimizer does them (version counters move)
        for name, p in model.named_parameters():
            if name.endswith("weight") and p.dim() >= 2 and ("con_enc" in name or "occ_convs" in name or "sigma_head" in name):
                p.mul_(1.03)
        got = _grab(model.simple_test(**kws[0]))
        assert model._pipe1[1] is pipe and pipe.recaptures == 1
        model.graph_simple_test = False
        want = _grab(model.simple_test(**kws[0]))
        _same(want, got, "after in-place weight update")
        assert not torch.equal(before["pred_c"], got["pred_c"])
        # 2. load_state_dict drops the pipeline altogether
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        for k in sd:
            if k.endswith("fine_mlp.0.weight") or "lateral_convs.0" in k and k.endswith("weight"):
                sd[k] = sd[k] * 0.9
        model.load_state_dict(sd)
        assert model._pipe1 is None
        model.graph_simple_test = True
        got2 = _grab(model.simple_test(**kws[1]))
        model.graph_simple_test = False
        want2 = _grab(model.simple_test(**kws[1]))
        _same(want2, got2, "after load_state_dict")
        # 3. writes that bypass the version counters + invalidate_packs
        model.graph_simple_test = True
        pipe = model._pipe1[1]
        for name, p in model.named_parameters():
            if "occ_pred_conv" in name and name.endswith("weight"):
                p.data.mul_(1.1)
        co_occ_amd.invalidate_packs(model)
        got3 = _grab(model.simple_test(**kws[1]))
        assert pipe.recaptures == 1
        model.graph_simple_test = False
        want3 = _grab(model.simple_test(**kws[1]))
        _same(want3, got3, "after .data surgery + invalidate_packs")
        # 4. train() drops the captured eval-mode graph
        model.graph_simple_test = True
        model.simple_test(**kws[0])
        assert model._pipe1 is not None
        model.train()
        assert model._pipe1 is None
        model.eval()


def test_pipelined_test_routes_an_odd_ground_truth_grid_through_simple_test_and_trims_the_fine_outputs(dev):
    """a sample whose ``gt_occ`` is not cascade_ratio x the coarse grid takes the eager decode (as
    ``simple_test`` does); fine outputs come back at their exact size; breaking out of the loop closes the pipeline."""
    from co_occ_amd import apis
    bench, model, samples, gts = _setup(dev, n=3)
    odd = torch.randint(0, 17, (1, 200, 200, 32), generator=torch.Generator().manual_seed(9)).to(dev)     # a taller label grid
    data = [dict(precomputed=bench.simple_test_kwargs(s)["precomputed"], gt_occ=g) for s, g in zip(samples, gts)]
    data[1]["gt_occ"] = odd
    with torch.no_grad():
        model.graph_simple_test = False
        ref = [model.simple_test(img=None, **d) for d in data]
        ref = [dict(_grab(r), nfine=r["output_voxels_fine"][0].shape[0]) for r in ref]
    stats = {}
    for i, (d, res) in enumerate(apis.pipelined_test(model, iter(data), slots=3, dense_streams=1, stats=stats)):
        got = dict(_grab(res), nfine=res["output_voxels_fine"][0].shape[0])
        assert got["nfine"] == ref[i]["nfine"] and got["fine"].shape == ref[i]["fine"].shape
        assert got.pop("nfine") == ref[i]["nfine"]
        _same({k: v for k, v in ref[i].items() if k != "nfine"}, got, "sample %d" % i)
    assert stats == dict(fallbacks=0, recaptures=0, eager_samples=1), stats
    gen = apis.pipelined_test(model, iter(data * 3), slots=3, dense_streams=1, stats=stats)
    next(gen)
    gen.close()                       # the consumer breaks out: the finally clause drains and closes the pipeline
    assert stats["eager_samples"] >= 0


def test_calibrate_reports_both_con_enc_presets_against_the_exact_fp32_kernels(dev):
    """``co_occ_amd.calibrate(model, frames)``: both tile presets inside the tolerance on the synthetic
    checkpoint, (2, 2) closer to the exact-fp32 anchor than (4, 4) at the fuser's output, the model's configuration restored."""
    import co_occ_amd
    from co_occ_amd import calibration, core
    bench, model, samples, gts = _setup(dev, n=2)
    frames = [bench.simple_test_kwargs(s) for s in samples]
    with torch.no_grad():
        before = model.simple_test(**frames[0])["pred_c"].clone()
    rep = co_occ_amd.calibrate(model, frames, reps=1)
    print(calibration.format_report(rep))
    p44, p22 = rep["presets"][(4, 4)], rep["presets"][(2, 2)]
    assert p44["within_tol"] and p22["within_tol"], rep
    assert p22["voxel_feats"][1] < p44["voxel_feats"][1]
    assert min(p44["label_agreement"], p22["label_agreement"]) > 0.999
    assert rep["recommended"] in ((4, 4), (2, 2))
    assert core.CONV_ENGINE == "h2" and model.occ_fuser.conenc_tiles is None
    with torch.no_grad():
        after = model.simple_test(**frames[0])["pred_c"]
    assert torch.equal(before, after)


def test_simple_test_falls_back_to_eager_when_the_library_refuses_the_captured_form(dev, monkeypatch):
    """an argument / shape refusal (``CooccArgError``: COOCC_EINVAL, COOCC_ENOMEM, a configuration without
    device-count kernels) while the dense stage is captured leaves the EAGER launches as the route -- warned once, recorded in
    ``graph_unavailable``, not retried, same results; only the range guard and HIP failures propagate."""
    import warnings
    from co_occ_amd import _lib, head
    bench, model, samples, gts = _setup(dev, n=2)
    kws = [dict(bench.simple_test_kwargs(s), gt_occ=g) for s, g in zip(samples, gts)]
    with torch.no_grad():
        model.graph_simple_test = False
        ref = [_grab(model.simple_test(**kw)) for kw in kws]
        monkeypatch.setattr(head, "FUSED_FINE_MLP", False)          # the static fine branch refuses: no device-count kernels
        model.graph_simple_test = True
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = [_grab(model.simple_test(**kw)) for kw in kws]
    assert model._pipe1 is None and model.graph_unavailable is not None and "CooccArgError" in model.graph_unavailable[1]
    assert sum("hipGraph" in str(x.message) for x in w) == 1        # warned once, not retried for the second frame
    assert issubclass(_lib.CooccArgError, _lib.CooccError) and not issubclass(_lib.CooccRangeError, _lib.CooccArgError)
    monkeypatch.setattr(head, "FUSED_FINE_MLP", True)
    for i, (a, b) in enumerate(zip(ref, got)):
        for k in ("pred_c", "rgbs", "depths", "SC_metric"):
            if torch.is_tensor(a[k]):
                assert torch.equal(a[k], b[k]), (i, k)
