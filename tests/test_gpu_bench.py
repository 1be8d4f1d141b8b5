"""bench.py's contract, end to end on one GPU: the one-JSON-line output with ``roofline`` / ``cpu_baseline``-less fast settings,
and the N > 1 path (one process per rank under torch.distributed.run, two dense streams per rank, ticketed collectives) with both
ranks on cuda:0 over gloo -- the control flow the driver's scaling run takes, minus RCCL."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def _check_contract(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup and d["unit"] == "samples/s" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"].startswith("f32")      # "f32 (split-f16 products, fp32 accumulate)" under the default engine
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "roofline_render_r101" in d and "roofline_pool" in d
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.05 * d["value"]


def test_default_line_is_the_graph_pipeline_with_the_contract_fields():
    """The driver's command: the dense stage of every sample is one hipGraph launch; the line says so and carries the roofline
    objects (per-kernel HIP events from the eager pass after the timed region)."""
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--also", "r101"])
    _check_contract(d, 12, 2)
    g = d["graph"]
    assert g["slots"] == d["config"]["samples_in_flight"] == 6 and g["eager_fallbacks"] == 0 and g["dense_stage_ms"] > 0
    assert "hipGraph" in d["config"]["pipeline"] and d["config"]["conv_engine"] == "h2" and "stream_probe" not in d
    assert "measured" in d["roofline"] and isinstance(d["env_knobs"], dict)
    # ``--also CONFIG``: a short child-process run of a second workload inside the same JSON line (the default command appends
    # north_star's stress workload, stress200_r101, this way: VERDICT r5 item 7); here the cheaper r101 frames
    a = d["also"]["r101"]
    assert "error" not in a, a
    assert a["value"] > 0 and a["steps"] == 10 and a["windows"] == 2 and len(a["window_ms_per_step"]) == 2
    assert a["config"]["render_maps"] == "6x896x1600" and "also" not in a


def test_eager_line_reports_its_stream_probe():
    """--graph 0: every launch issued from Python; the number of samples in flight is chosen from untimed bursts and reported."""
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--graph", "0"])
    _check_contract(d, 8, 2)
    p = d["stream_probe"]
    assert p["chosen"] == d["config"]["samples_in_flight"] and p["chosen"] in (1, 2)
    assert set(p["samples_per_s"]) == {"1", "2"} and all(v > 0 for v in p["samples_per_s"].values())
    faster = max(p["samples_per_s"], key=lambda k: p["samples_per_s"][k])
    assert int(faster) == p["chosen"]


def test_two_streams_line_carries_the_alone_figures():
    d = _run([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--streams", "2", "--no-cpu-baseline", "--graph", "0"])
    _check_contract(d, 6, 2)
    assert d["config"]["samples_in_flight"] == 2 and "stream_probe" not in d
    assert 0 < d["roofline"]["frac_alone"] < 1 and "roofline_isolated" in d


@pytest.mark.parametrize("graph", [0, 1])
def test_two_ranks_two_streams_each_on_one_gpu(graph):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2", "--same-device", "--backend", "gloo", "--steps", "6",
              "--warmup", "2", "--streams", "2", "--slots", "4", "--graph", str(graph), "--no-cpu-baseline", "--no-kernel-timing"])
    assert d["n_gpus"] == 2 and d["world_size_seen_by_backend"] == 2 and d["backend"] == "gloo"
    assert d["config"]["samples_in_flight"] == (4 if graph else 2)
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 0.05 * d["value"]      # whole-job rate: both ranks' samples


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "process group has 1 rank" in (p.stderr + p.stdout)


def test_two_stream_pipeline_outputs_equal_sequential_calls():
    """What bench.py times is what the parity tests check: every sample that goes through the two-stream pipeline (pooling +
    search prefetched on their own streams by helper threads, dense stages of two samples in flight on two streams / host
    threads) produces bit for bit the logits, fine coordinates and rendered maps of a plain sequential call."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, _ = bench.build_model("r50", dev)
    samples = [bench.make_inputs("r50", 77 + i, dev, model) for i in range(2)]
    keys = ("pred_c", "pred_f", "rgbs", "depths", "voxel_feats")

    def grab(out):
        d = {k: (out[k].t if hasattr(out[k], "t") and not torch.is_tensor(out[k]) else out[k]).clone() for k in keys}
        d["fine_xyz"] = out["output_coords_fine"][0].clone()
        return d
    with torch.no_grad():
        ref = [grab(bench.step(model, s, 1)) for s in samples]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    pipe = bench.Pipeline(model, samples, dev, streams, prefetch=True, world=1)
    got = {}
    pipe.run(4)                                         # sizes every stream's scratch
    pipe.run(10, collect=lambda i, out: got.__setitem__(i, grab(out)))
    torch.cuda.synchronize()
    assert sorted(got) == list(range(10))
    for i, g in got.items():
        for k, v in g.items():
            assert torch.equal(v, ref[i % 2][k]), (i, k)
    # the two samples really differ (the comparison above is not vacuous)
    assert not torch.equal(ref[0]["pred_c"], ref[1]["pred_c"])


def test_graph_pipeline_outputs_equal_sequential_calls():
    """The default serving loop of bench.py -- the PRODUCT's ``COOCC_Ray.serving()`` (co_occ_amd.serving.ServingPipeline: native search
    driver on prefetch streams, the dense stage of every sample one hipGraph launch on one of several dense streams, every frame's
    dense-stage inputs copied into its slot) -- produces bit for bit the outputs of plain sequential eager calls."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model, _ = bench.build_model("r50", dev)
    samples = [bench.make_inputs("r50", 177 + i, dev, model) for i in range(5)]       # 5 frames over 4 slots: frames move between slots
    keys = ("pred_c", "pred_f", "rgbs", "depths")
    with torch.no_grad():
        ref = [{k: bench.step(model, s, 1)[k].clone() for k in keys} for s in samples]
    torch.cuda.synchronize()
    frames = [bench.frame_of(s) for s in samples]
    gp = model.serving(frames[0], slots=4, dense_streams=2)
    got = {}
    gp.run(frames, 6)
    gp.run(frames, 11, collect=lambda i, out: got.__setitem__(i, {k: out[k].clone() for k in keys}))
    torch.cuda.synchronize()
    assert sorted(got) == list(range(11)) and gp.fallbacks == 0
    for i, g in got.items():
        for k, v in g.items():
            assert torch.equal(v, ref[i % 5][k]), (i, k)
    assert not torch.equal(ref[0]["pred_c"], ref[1]["pred_c"]) and not torch.equal(ref[0]["rgbs"], ref[1]["rgbs"])
    gp.close()


def _torchrun(nproc, extra, timeout=900):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", str(nproc)] + extra, timeout=timeout)


def test_ray_shard_mode_two_ranks_on_one_gpu():
    """--shard rays (configs[4]'s per-camera render shard): ONE scene over 2 ranks (gloo, both on cuda:0): strong scaling, the
    sharded share of the step is reported."""
    d = _torchrun(2, ["--same-device", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--shard", "rays", "--no-cpu-baseline",
                      "--no-kernel-timing"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "ray-shard x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.05 * d["value"]           # scenes/s of the whole job, not x world
    assert 0 < d["ray_shard"]["sharded_fraction_of_step"] < 0.2


def test_eight_ranks_on_one_gpu_over_gloo():
    """The driver's N = 8 command line with every rank on cuda:0 over gloo: what depends on the RANK COUNT
    -- thread pinning of 8 ranks on the host's cores (dist.pin_rank_threads), the automatic one-search-helper-per-rank setting,
    the ordered (ticketed) all-gathers across helper threads, 8 x the slots' workspaces in one device's memory -- has run once at
    N = 8 before an 8-GPU node does it over RCCL."""
    d = _torchrun(8, ["--same-device", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--windows", "1", "--slots", "3",
                      "--no-cpu-baseline", "--no-kernel-timing"], timeout=1200)
    assert d["n_gpus"] == 8 and d["world_size_seen_by_backend"] == 8 and d["backend"] == "gloo" and d["scaling"] == "weak"
    assert "dp8" in d["config"]["parallelism"] and d["graph"]["searches_ahead"] == 1 and d["graph"]["eager_fallbacks"] == 0
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 0.05 * d["value"]       # whole-job rate: all eight ranks' samples


def test_rccl_two_ranks_when_two_gpus_are_present():
    """The N = 2 path over RCCL (backend nccl, one rank per GPU): both bench modes.  Skipped on single-GPU boxes -- so that the
    first multi-GPU box that runs the suite is not also the first RCCL run of this code."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    d = _torchrun(2, ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-timing"])
    assert d["n_gpus"] == 2 and d["backend"] == "nccl" and d["world_size_seen_by_backend"] == 2
    d = _torchrun(2, ["--steps", "4", "--warmup", "1", "--shard", "rays", "--no-cpu-baseline", "--no-kernel-timing"])
    assert d["backend"] == "nccl" and d["scaling"] == "strong"
