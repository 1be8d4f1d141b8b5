#!/usr/bin/env python
"""bench.py -- Co-Occ fused-voxel hot path on MI355X, BASELINE.json's metric:
samples/s (6-cam frame + sweep -> occupancy logits + rendered rgb/depth).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched through torch.distributed.run)

A step is one pass of the hot path over one synthetic sample already resident in HBM:
fused lift (x) splat pooling of the sample's depth distribution and context features (P2) ->
BiFuser_N (KNN + gather/encode/scatter + con_enc) -> CustomResNet3D -> FPN3D -> OccHead
(coarse + cascade fine, scattered into the 200x200x16 grid) -> 6-camera volume render
(+ for N>1 one RCCL all-gather of the rendered maps).  Workload = the reference config
coocc_multi_r50_256x704 (configs[1]): fused grid 100x100x8 x 128 ch, final occupancy grid
200x200x16, 6 cams with 16x44 feature maps -> 6x256x704 maps, knum=2, random weights, fp32.
Weak scaling: every rank processes its own samples (the reference's samples_per_gpu=1 DP).

The serving loop is the PRODUCT's (co_occ_amd.serving.ServingPipeline, reached through ``COOCC_Ray.serving()``): six sample
slots; pooling + index search of the next samples prefetched eagerly by helper threads (one C-ABI call per search); the dense
stage of a sample is ONE captured hipGraph launch on one of three dense streams; a new frame's dense-stage inputs are copied
into its slot's static tensors.  ``--api simple_test`` times the reference's own per-sample call instead (synchronous, one
slot: the search is then on the critical path); ``--graph 0`` is the eager two-stream pipeline of round 2 (class Pipeline).
Outputs are checked bit for bit against sequential eager calls in tests/test_gpu_serving.py / tests/test_gpu_bench.py.

``--train``: one step = forward + backward of the differentiable hot path under ``model.train()`` (its own JSON line, roofline =
the weight-gradient GEMMs); ``--with-lidar``: frames carry a raw point cloud and the LiDAR producer (voxelisation -> VFE -> sparse
encoder) runs inside the step on the prefetch stream.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the Winograd-domain split-f16 GEMM k_gemm_h2z, priced against
the dense f16 MFMA peak; per-kernel HIP events from an eager pass right after the timed region, *_alone = nothing else on the
GPU), `roofline_pool`, `roofline_render[_r101]` and, at N=1, `cpu_baseline` (the oracle's CPU restatement timed on this host).
"""
import argparse
import json
import os
import sys
import time

# The serving loop keeps 2 dense + up to 4 search streams (+ their side streams) busy; with ROCm's default of 4 hardware queues
# per process several of them share one in-order queue and a search stage waits behind whole dense graphs (search latency
# 10.5 -> 7 ms, 143 -> 158 samples/s measured with 16).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import co_occ_amd as pkg  # noqa: E402
import co_occ_amd.dist as cdist  # noqa: E402
import co_occ_amd.synth as synth  # noqa: E402
from co_occ_amd import _lib, core  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, fp32-input MFMA (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA (spec; 2:1-sparse figures are not used)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16 MFMA: the same rate as bf16 on gfx950 (v_mfma_f32_32x32x16_{f16,bf16})
HBM_PEAK_GBS = 8000.0          # HBM3E spec
TRAFFIC_FILE = "r6_traffic.json"


def make_inputs(cfgname, seed, dev, model):
    """One synthetic sample, resident in HBM: the lifted image-branch outputs (softmax depth [N,D,fH,fW] + context features
    [N,C,fH,fW], what DepthNet hands to Lift-Splat), the LiDAR-sweep voxel volume, the 2D features the fine branch samples,
    the camera matrices.  ``img`` (the pooled camera volume) is kept too for --no-pool runs and the parity tests."""
    c = synth.CONFIGS[cfgname]
    img, pts = synth.voxel_inputs(c["grid"], C=c["C"], seed=seed, p_img=c.get("p_img", 0.65), p_pts=c.get("p_pts", 0.12))
    fH, fW = c["fmap"]
    rig = synth.camera_rig(c["ncam"], (fH * 16, fW * 16), seed=seed)
    r = {k: v.to(dev) for k, v in rig.items() if torch.is_tensor(v)}
    vt = model.img_view_transformer
    cams = tuple(r[k] for k in ("rots", "trans", "intrins", "post_rots", "post_trans", "bda"))
    gemo = vt.get_geometry(*cams)
    depth, ctx = synth.lift_inputs(c["ncam"], vt.D, c["fmap"], c["C"], seed=seed)
    img_feats = [synth.image_feats(c["ncam"], c["fmap"], 512, seed=seed).to(dev)]
    transform = tuple(t.to(dev) if torch.is_tensor(t) else t for t in synth.rig_transform(rig))
    return dict(img=img.to(dev), pts=pts.to(dev), gemo=gemo, img_feats=img_feats, transform=transform, cams=cams,
                depth=depth.to(dev), ctx=ctx.to(dev),
                cpu=dict(img=img, pts=pts, rig=rig, img_feats=[img_feats[0].cpu()], lift=(depth, ctx)))


def build_model(cfgname, dev, with_lidar=False):
    c = synth.CONFIGS[cfgname]
    fH, fW = c["fmap"]
    X, Y, Z = c["grid"]
    if cfgname == "openocc":
        cfg = synth.model_cfg_openocc()
    else:
        cfg = synth.model_cfg(C=c["C"], knum=c["knum"], final_occ_size=(2 * X, 2 * Y, 2 * Z), input_size=(fH * 16, fW * 16))
        cfg["img_view_transformer"]["grid_config"].update(synth.pool_bounds(cfgname))
    c["bounds"] = synth.pool_bounds(cfgname)
    if with_lidar:          # the LiDAR producer of the reference config (coocc_multi_r50_256x704.py:121-134) joins the step
        cfg.update(synth.lidar_cfg())
    model = pkg.build_detector(cfg)
    sd = synth.random_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    return model.to(dev).eval(), sd


def pool(model, s):
    """P2 inside the step: fused Lift (x) Splat of the sample's depth distribution and context features, geometry from
    the camera matrices in-kernel (ViewTransformerLSSVoxel.py:135-145) -> the camera voxel volume [1,C,X,Y,Z]."""
    X, Y, Z = synth.CONFIGS[CFGNAME[0]]["grid"]
    slot0 = model.occ_fuser.concat_buffer(1, X, Y, Z, s["depth"].device)      # pooled rows land in the fuser's concat buffer
    return model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"], out=slot0)


class _Ticket:
    """Collectives must be issued in the same order on every rank: sample order, whichever host thread owns the sample."""

    def __init__(self):
        self.cv = __import__("threading").Condition()
        self.turn, self.dead = 0, False

    def reset(self):
        with self.cv:
            self.turn, self.dead = 0, False

    def abort(self):
        with self.cv:
            self.dead = True
            self.cv.notify_all()

    def __call__(self, i):
        ticket = self

        class _Turn:
            def __enter__(self_):
                with ticket.cv:
                    ticket.cv.wait_for(lambda: ticket.turn == i or ticket.dead)

            def __exit__(self_, *a):
                with ticket.cv:
                    ticket.turn += 1
                    ticket.cv.notify_all()
        return _Turn()


TICKET = _Ticket()


class _Gates:
    """Events posted by sample i for sample i+1 to wait on (host side: until the event object exists)."""

    def __init__(self):
        self.cv = __import__("threading").Condition()
        self.ev = {}
        self.dead = False

    def reset(self):
        with self.cv:
            self.ev, self.dead = {-1: None}, False

    def post(self, i, ev):
        with self.cv:
            self.ev[i] = ev
            self.ev.pop(i - 4, None)
            self.cv.notify_all()

    def wait_for(self, i):
        with self.cv:
            self.cv.wait_for(lambda: i in self.ev or self.dead, timeout=60)
            return self.ev.get(i)

    def abort(self):
        with self.cv:
            self.dead = True
            self.cv.notify_all()


GATES = _Gates()
TRACE, TRACE_T0 = [None], [0.0]          # --diag: per-sample host issue window + completion event
DIAG = {"search_host_s": [], "search_blocked_s": [], "wait_search_host_s": [], "dense_host_s": [], "dense_blocked_s": []}     # host-side issue times per sample (--diag)
STAGGER = [0]


def step(model, s, world, search=None, img=None, ticket=None):
    # the reference hard-codes the render bounds to a 100x100x8 volume (coocc_ray.py:577): smaller test grids
    # (config1) cannot be rendered there either
    if img is None:
        img = pool(model, s) if WITH_POOL[0] else s["img"]
    X, Y, Z = img.shape[2:]
    do_render = X >= 100 and Y >= 100 and Z >= 8
    if STAGGER[0] and ticket is not None:
        # S > 1: sample i+1 may enter its dense stage only after sample i has passed the gate point (end of the fuser or of the
        # encoder), so the two dense stages in flight keep a fixed phase offset instead of locking step by chance
        cur = torch.cuda.current_stream()
        prev = GATES.wait_for(ticket - 1)
        if prev is not None:
            cur.wait_event(prev)

        def passed():
            ev = torch.cuda.Event()
            ev.record(cur)
            GATES.post(ticket, ev)
        vf = model.fuse(img, s["pts"], search)
        if STAGGER[0] == 1:
            passed()
        out = model.decode(vf, s["gemo"], s["img_feats"], s["transform"], do_render, after_encoder=passed if STAGGER[0] == 2 else None)
    elif SHARD[0]:
        # config 5 (--shard rays): ONE scene over all ranks -- every rank runs K / G / C on the same sample (the conv stack
        # does not shard: a slab decomposition needs halos at 13 layers), renders its contiguous chunk of the
        # (camera, row) ray space and the 16-byte-per-ray map chunks are all-gathered (equal-padded, one collective)
        from co_occ_amd.render import render_block_sharded
        out = model.forward_hot_path(img, s["pts"], s["gemo"], s["img_feats"], s["transform"], render=False, search=search)
        if do_render:
            ev = []
            out["rgbs"], out["depths"], out["render_maps"] = render_block_sharded(model.sigma_head, model.rgb_head, out["voxel_feats"], s["gemo"], 16,
                                                                                  events=ev)
            SHARD_EV.append(ev)
        return out
    else:
        out = model.forward_hot_path(img, s["pts"], s["gemo"], s["img_feats"], s["transform"], render=do_render, search=search)
    if world > 1 and ticket is not None:
        with TICKET(ticket):
            return _gather(out)
    if world > 1:
        return _gather(out)
    return out


def _gather(out):
    if True:
        # one RCCL all-gather of the packed maps per step, issued asynchronously: the previous step's gather is
        # completed first (at most one in flight), so it overlaps the whole next dense stage instead of delaying it
        if _pending:
            _pending.pop().wait()
        if _async_ok[0]:
            try:
                _pending.append(cdist.all_gather_maps_async(out["rgbs"], out["depths"]))
                return out
            except Exception as e:        # a backend without asynchronous all-gather: blocking form from here on
                _async_ok[0] = False
                print("bench: asynchronous all-gather unavailable (%s); using the blocking form" % e, file=sys.stderr)
        out["all_rgbs"], out["all_depths"] = cdist.all_gather_maps(out["rgbs"], out["depths"])
    return out


_pending = []
_async_ok = [True]
SHARD = [False]
SHARD_EV = []
WITH_POOL = [True]
CFGNAME = ["r50"]


GATHERER = [None]        # dist.GatherThread of the serving loop at world > 1


def drain_gathers():
    out = None
    while _pending:
        out = _pending.pop().wait()
    if GATHERER[0] is not None:
        out = GATHERER[0].drain() or out
    return out


def usable_cores():
    """Host cores this process may actually use: the cgroup CPU quota when there is one (the GPU
    box exposes 256 logical CPUs but caps the container; oversubscribing OpenMP thrashes)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(sd, s, cfgname, with_pool, runs=3):
    """The oracle's literal CPU restatement of the same step (materialised lift + sort-based pooling when the step starts
    from the lifted inputs, materialised concat, gather-then-MLP render) on this host's cores: 1 warm-up run + the median of
    ``runs`` timed runs, with per-stage seconds."""
    from oracle import ref_cpu
    c = synth.CONFIGS[cfgname]
    X, Y, Z = c["grid"]
    cpu = s["cpu"]
    rig = cpu["rig"]
    fr = ref_cpu.create_frustum((c["fmap"][0] * 16, c["fmap"][1] * 16), 16, [2.0, 58.0, 0.5])
    gemo = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    sdc = {k: v.cpu() for k, v in sd.items()}
    sub = lambda p: {k[len(p):]: v for k, v in sdc.items() if k.startswith(p)}
    torch.set_num_threads(usable_cores())
    tr = synth.rig_transform(rig)
    cr = c.get("cascade_ratio", 2)
    occ = tuple(c.get("final_occ_size", (2 * X, 2 * Y, 2 * Z)))
    pcr = tuple(c.get("point_cloud_range", (-50, -50, -5.0, 50, 50, 3.0)))
    render = X >= 100 and Y >= 100 and Z >= 8

    def once():
        st = {}
        t = time.perf_counter()
        img = cpu["img"]
        if with_pool:
            depth, ctx = cpu["lift"]
            vol = (depth[:, None] * ctx[:, :, None]).permute(0, 2, 3, 4, 1)[None].contiguous()      # Lift (LSSVoxel.py:135-143)
            dx, bx, nx = ref_cpu.gen_dx_bx(*[c["bounds"][k] for k in ("xbound", "ybound", "zbound")])
            img = ref_cpu.voxel_pooling(gemo, vol, dx, bx, nx).contiguous()
        st["lift+pool"] = time.perf_counter() - t; t = time.perf_counter()
        vf = ref_cpu.bifuser_forward(sub("occ_fuser."), img, cpu["pts"], c["knum"])
        st["fuser"] = time.perf_counter() - t; t = time.perf_counter()
        sem = ref_cpu.fpn3d_forward(sub("semantic_neck."), ref_cpu.resnet3d_forward(sub("semantic_encoder."), vf))
        st["encoder+neck"] = time.perf_counter() - t; t = time.perf_counter()
        ref_cpu.occhead_forward(sub("pts_bbox_head."), sem, cpu["img_feats"], tr, cr, occ, pcr)
        st["head"] = time.perf_counter() - t; t = time.perf_counter()
        if render:
            ref_cpu.render_block(sub("sigma_head."), sub("rgb_head."), vf, gemo, True)
        st["render"] = time.perf_counter() - t
        st["total"] = sum(st.values())
        return st
    with torch.no_grad():
        once()                                             # warm-up (allocator, oneDNN primitive caches)
        runs_ = sorted((once() for _ in range(runs)), key=lambda d: d["total"])
    med = runs_[len(runs_) // 2]
    return dict(value=round(1.0 / med["total"], 5), unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample="1 sample of the %s workload (rank-0 inputs, seed 1234): 1 warm-up + median of %d runs, %.2f s each" % (
                    cfgname, runs, med["total"]),
                seconds=round(med["total"], 3), stage_seconds={k: round(v, 3) for k, v in med.items() if k != "total"},
                all_runs_seconds=[round(r["total"], 3) for r in runs_])


def _traffic(key, cfg):
    """HBM bytes per launch / call from the committed PMC passes (profiles/TRAFFIC_FILE, tools/make_traffic.py), or None."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))[key]
        return (d[cfg] if cfg in d else d)["bytes_per_launch"]
    except Exception:
        return None


def rooflines(ksum, nsteps, args, rank, table):
    """roofline objects from a KernelTimer summary: dominant conv instantiation (largest summed HIP-event time), every
    conv launch together (per MFMA dtype), the pooling call, the render pair."""
    roof = None
    extra = {}
    if ksum:
        tot = sum(v["ms"] for v in ksum.values())
        if table and rank == 0:
            for k, v in sorted(ksum.items(), key=lambda kv: -kv[1]["ms"]):
                unit = v["work"] / (v["ms"] * 1e-3) if v["ms"] else 0
                print("%-34s launches %5d  total %9.3f ms  avg %8.3f ms  %6.1f%%  work/s %.4g" % (
                    k, v["launches"], v["ms"], v["ms"] / v["launches"], 100 * v["ms"] / tot, unit), file=sys.stderr)
        # one entry per kernel symbol: the Winograd-domain launches of either tile size are the same instantiation
        # (k_conv2<BM,PF,0,true>); keep their direct-convolution-equivalent flops alongside the executed ones
        convs = {}
        for k, v in ksum.items():
            if not (k.startswith("k_conv") or k.startswith("k_gemm")):
                continue
            sym = k.split(" wino")[0] + (",wg> wino" if " wino" in k else "")
            sym = sym.replace(">,wg>", ",wg>").replace("k_conv2p,wg>", "k_conv2p")
            if k.startswith("k_gemm_h2"):
                sym = k.split()[0] + (" wino" if " wino" in k else (" direct" if "direct" in k else ""))
            c_ = convs.setdefault(sym, dict(launches=0, ms=0.0, work=0.0, equiv=0.0))
            c_["launches"] += v["launches"]; c_["ms"] += v["ms"]; c_["work"] += v["work"]
            c_["equiv"] += v["work"] * {"wino2": 2.25, "wino4": 4.0}.get(k.split()[-1], 1.0)
        dom = max(convs, key=lambda k: convs[k]["ms"]) if convs else None
        if dom:
            v = convs[dom]
            h2 = dom.startswith("k_gemm_h2")
            h1 = dom.startswith("k_gemm_h1")
            # flops the matrix cores execute: Winograd-domain for "wino"; the split-f16 engine issues THREE f16 MFMAs per
            # fp32-accurate product (hi*hi, hi*lo, lo*hi), all of which count against the f16 peak
            ach = v["work"] / (v["ms"] * 1e-3) / 1e12 * (3.0 if h2 else 1.0)
            traffic = None
            try:   # HBM bytes per launch from the committed PMC passes (cannot be collected inside this process)
                tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
                traffic = tj[dom]["bytes_per_launch"] if args.config == "r50" else None
            except Exception:
                pass
            # direct-convolution-equivalent rate of the same launches: F(m x m,3x3) needs 9 m^2/(m+2)^2 x fewer multiplies
            equiv = v["equiv"] / v["work"]
            symbol = {"k_conv2<160,wg> wino": "k_conv2<160, 2, true, 2, false>", "k_conv2p wino": "k_conv2p<true, false>",
                      "k_conv2<128,wg> wino": "k_conv2<128, 1, true, 3, false>", "k_gemm_h2z wino": "k_gemm_h2z<3, false>",
                      "k_gemm_h2z direct": "k_gemm_h2z<*, true>", "k_gemm_h2w": "k_gemm_h2w<false>"}.get(dom, dom)     # name in the rocprofv3 trace
            peak = MFMA_F16_PEAK_TFLOPS if (h2 or h1) else (MFMA_BF16_PEAK_TFLOPS if dom.startswith("k_conv_bf16") else MFMA_F32_PEAK_TFLOPS)
            roof = dict(bound="mfma", kernel=dom, symbol=symbol, achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                        frac=round(ach / peak, 4), traffic=traffic,
                        traffic_source=("profiles/%s (rocprofv3 --pmc passes of this command, replayed: HBM counters cannot be "
                                        "collected in-process)" % TRAFFIC_FILE) if traffic is not None else None,
                        launches=v["launches"],
                        avg_launch_ms=round(v["ms"] / v["launches"], 4),
                        share_of_timed_kernels=round(v["ms"] / tot, 3),
                        flops=("executed on the f16 MFMA pipe: 3 v_mfma_f32_32x32x16_f16 per fp32-accurate product (hi*hi, hi*lo, lo*hi), "
                               "peak = dense f16; fp32-equivalent %.1f TFLOP/s = %.2f x the fp32-MFMA peak of %.1f; "
                               "direct-conv-equivalent x%.2f = %.1f TFLOP/s" % (ach / 3, ach / 3 / MFMA_F32_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS,
                                                                                 equiv, ach / 3 * equiv)) if h2 else
                              "executed on the MFMA pipe (direct-conv-equivalent x%.2f = %.1f TFLOP/s)" % (equiv, ach * equiv))
            for tag, sel, pk, mul, label in (
                    ("roofline_all_convs", lambda k: k.startswith("k_conv") and not k.startswith("k_conv_bf16"), MFMA_F32_PEAK_TFLOPS, 1.0,
                     "every fp32-MFMA k_conv* launch"),
                    ("roofline_bf16_convs", lambda k: k.startswith("k_conv_bf16"), MFMA_BF16_PEAK_TFLOPS, 1.0, "every k_conv_bf16 launch"),
                    ("roofline_f16_convs", lambda k: k.startswith("k_gemm_h1"), MFMA_F16_PEAK_TFLOPS, 1.0, "every one-term f16 k_gemm_h1* launch"),
                    ("roofline_h2_gemms", lambda k: k.startswith("k_gemm_h2"), MFMA_F16_PEAK_TFLOPS, 3.0,
                     "every split-f16 k_gemm_h2* launch (3 f16 MFMAs per product counted)")):
                grp = [v2 for k2, v2 in convs.items() if sel(k2)]
                if not grp:
                    continue
                allc = mul * sum(v2["work"] for v2 in grp) / (sum(v2["ms"] for v2 in grp) * 1e-3) / 1e12
                extra[tag] = dict(bound="mfma", kernel=label, achieved=round(allc, 2), peak=pk, unit="TFLOP/s", frac=round(allc / pk, 4),
                                  ms_per_step=round(sum(v2["ms"] for v2 in grp) / nsteps, 3))
        if "k_lift_splat" in ksum:
            v = ksum["k_lift_splat"]
            ach = v["work"] / (v["ms"] * 1e-3) / 1e9
            extra["roofline_pool"] = dict(bound="hbm", kernel="coocc_lift_splat_cams (keys + binning + per-voxel sums)",
                                          achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                                          traffic=_traffic("coocc_lift_splat_cams", args.config), algorithmic_bytes=int(v["work"] / v["launches"]),
                                          avg_ms_per_step=round(v["ms"] / v["launches"], 4),
                                          note="fused Lift (x) Splat, in the timed step (on the prefetch stream)")
        rk = [ksum[k] for k in ("k_render_nearest+k_upsample_maps",) if k in ksum]
        if rk:
            ms = sum(v["ms"] for v in rk)
            by = sum(v["work"] for v in rk)
            ach = by / (ms * 1e-3) / 1e9
            rtraffic = None
            try:
                rtraffic = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))["k_render_nearest+k_upsample_maps"][args.config]["bytes_per_launch"]
            except Exception:
                pass
            extra["roofline_render"] = dict(bound="hbm", kernel="k_render_nearest+k_upsample_maps", achieved=round(ach, 1),
                                            peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=rtraffic,
                                            avg_ms_per_step=round(ms / max(1, rk[0]["launches"]), 4))

    return roof, extra


def render_r101_roofline(model, dev, iters=20):
    """The HBM-bound render pair at configs[2]'s size (6 x 56 x 100 rays x 112 samples -> 6 x 896 x 1600 maps, 184.1 MB
    algorithmic), timed with HIP events AFTER the timed region on random fused features: north_star's ">= 40 % of the HBM
    roofline on the render kernel" is defined at this size (at r50 the pair moves 24 MB in ~25 us and is launch-bound)."""
    from co_occ_amd.render import render_block
    from co_occ_amd.view_transformer import get_frustum
    rig = synth.camera_rig(6, (896, 1600), seed=7)
    mats = [rig[k].to(dev) for k in ("rots", "trans", "intrins", "post_rots", "post_trans", "bda")]
    gemo = get_frustum(*mats, (896, 1600), 16)
    vf = torch.randn(1, 128, 100, 100, 8, device=dev)
    keep = (core.TIMER.enabled, core.TIMER.only)
    with torch.no_grad():
        for _ in range(3):
            render_block(model.sigma_head, model.rgb_head, vf, gemo, 16)
        torch.cuda.synchronize()
        core.TIMER.enabled, core.TIMER.only = 1, ("k_render_nearest",)
        core.TIMER.reset()
        for _ in range(iters):
            render_block(model.sigma_head, model.rgb_head, vf, gemo, 16)
        torch.cuda.synchronize()
    v = core.TIMER.summary()["k_render_nearest+k_upsample_maps"]
    # the same pair with the geometry evaluated inside the ray kernel from the camera constants (what the timed step does):
    # SURVEY 8(d)'s second byte count (the 45 MB geometry read is gone, so are its bytes in the numerator)
    from co_occ_amd.view_transformer import camera_mats, frustum_axes
    xs, ys, ds = frustum_axes((896, 1600), 16, [2.0, 58.0, 0.5], dev)
    cg = (camera_mats(*mats).reshape(-1, 39), xs, ys, ds)
    with torch.no_grad():
        for _ in range(3):
            render_block(model.sigma_head, model.rgb_head, vf, None, 16, cam_geo=cg)
        torch.cuda.synchronize()
        core.TIMER.reset()
        for _ in range(iters):
            render_block(model.sigma_head, model.rgb_head, vf, None, 16, cam_geo=cg)
        torch.cuda.synchronize()
    v2 = core.TIMER.summary()["k_render_nearest+k_upsample_maps"]
    core.TIMER.enabled, core.TIMER.only = keep
    core.TIMER.reset()
    ach = v["work"] / (v["ms"] * 1e-3) / 1e9
    ach2 = v2["work"] / (v2["ms"] * 1e-3) / 1e9
    return dict(bound="hbm", kernel="k_render_nearest+k_upsample_maps", workload="coocc_multi_r101_896x1600 render pair",
                achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                algorithmic_bytes=int(v["work"] / v["launches"]), avg_ms=round(v["ms"] / v["launches"], 4), launches=v["launches"],
                geometry_in_kernel=dict(achieved=round(ach2, 1), frac=round(ach2 / HBM_PEAK_GBS, 4),
                                        algorithmic_bytes=int(v2["work"] / v2["launches"]), avg_ms=round(v2["ms"] / v2["launches"], 4),
                                        note="sample positions from the 39 camera constants inside the ray kernel (the form the timed "
                                             "step runs): the [N,D,H,W,3] geometry is neither written nor read"))


class Pipeline:
    """The serving loop bench.py times: sample i runs its dense stage (fuser finish -> encoder -> neck -> head -> render) on
    dense stream i mod S (one host thread per stream; S = 1: the calling thread); its pooling + index search were issued S
    samples earlier on a high-priority prefetch stream by a helper thread, so they overlap the dense stages in flight.  With S = 2
    the low-occupancy tail of one sample (the 25x25x2 / 13x13x1 encoder stages, laterals, small heads: ~1.3 ms on <= 20 % of
    the CUs) runs under the other sample's large GEMMs.  Collectives (N > 1) are issued in sample order on every rank (ticket),
    whatever thread owns the sample.  ``collect(i, out)`` (tests) is called on sample i's stream right after its launches."""

    def __init__(self, model, samples, dev, streams, prefetch=True, world=1):
        from concurrent.futures import ThreadPoolExecutor
        self.model, self.samples, self.dev, self.streams, self.world = model, samples, dev, streams, world
        S = len(streams)
        # one prefetch worker + high-priority stream per dense stream: a lone 1024-thread FPS workgroup must win a CU slot
        # against the queue of convolution workgroups
        self.tpool = ThreadPoolExecutor(S) if prefetch else None
        self.search_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(S)]

    def do_search(self, i):
        """Pooling (P2) + index search (K1-K5) of sample i on a high-priority prefetch stream."""
        s = self.samples[i % len(self.samples)]
        torch.cuda.set_device(self.dev)
        t0, b0 = time.perf_counter(), _lib.blocked_seconds()
        with torch.cuda.stream(self.search_streams[i % len(self.search_streams)]), torch.no_grad():
            img = pool(self.model, s) if WITH_POOL[0] else s["img"]
            sr = self.model.search(img, s["pts"])
        DIAG["search_host_s"].append(time.perf_counter() - t0)
        DIAG["search_blocked_s"].append(_lib.blocked_seconds() - b0)
        return img, sr

    def run(self, nsteps, S=None, collect=None):
        import threading
        streams, tpool, samples, model, world, dev = self.streams, self.tpool, self.samples, self.model, self.world, self.dev
        S = len(streams) if S is None else S
        errs = []
        futs = {}
        lock = threading.Lock()

        def submit(i):
            if tpool is not None and i < nsteps:
                with lock:
                    futs[i] = tpool.submit(self.do_search, i)

        for i in range(min(S, nsteps)):
            submit(i)

        def worker(w):
            try:
                torch.cuda.set_device(dev)
                with torch.no_grad():
                    for i in range(w, nsteps, S):
                        img, sr = (None, None)
                        t0 = time.perf_counter()
                        if tpool is not None:
                            with lock:
                                f = futs.pop(i)
                            img, sr = f.result()
                        t1, b1 = time.perf_counter(), _lib.blocked_seconds()
                        submit(i + S)
                        with torch.cuda.stream(streams[w]):
                            if img is not None and torch.is_tensor(img):
                                img.record_stream(streams[w])
                            out = step(model, samples[i % len(samples)], world, search=sr, img=img, ticket=i)
                            if collect is not None:
                                collect(i, out)
                            if TRACE[0] is not None:
                                ev = torch.cuda.Event(enable_timing=True)
                                ev.record()
                                TRACE[0].append((i, w, t0 - TRACE_T0[0], t1 - TRACE_T0[0], time.perf_counter() - TRACE_T0[0], ev))
                        DIAG["wait_search_host_s"].append(t1 - t0)
                        DIAG["dense_host_s"].append(time.perf_counter() - t1)
                        DIAG["dense_blocked_s"].append(_lib.blocked_seconds() - b1)
                streams[w].synchronize()
            except Exception as e:  # surface worker failures in the main thread
                errs.append(e)
                TICKET.abort()
                GATES.abort()

        TICKET.reset()
        GATES.reset()
        if S == 1:
            worker(0)
        else:
            ths = [threading.Thread(target=worker, args=(w,)) for w in range(S)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if errs:
            raise errs[0]
        with torch.cuda.stream(streams[0]):
            drain_gathers()          # the last step's all-gather belongs to the timed region
        for st in streams:
            st.synchronize()


def frame_of(s):
    """The frame dict ``co_occ_amd.serving.ServingPipeline.submit`` takes, from one synthetic sample (``make_inputs``)."""
    fr = dict(depth=s["depth"], ctx=s["ctx"], cams=s["cams"], img_feats=s["img_feats"], transform=s["transform"])
    if s.get("points") is not None:        # --with-lidar: raw points, the LiDAR producer runs inside the step
        fr["points"] = s["points"]
    else:
        fr["pts"] = s["pts"]
    return fr


def simple_test_kwargs(s):
    """The same sample as keyword arguments of the reference's per-sample call ``COOCC_Ray.simple_test`` (coocc_ray.py:520):
    the upstream encoders' outputs are handed over as ``precomputed`` (they are outside the hot path)."""
    return dict(img=None, points=None, precomputed=dict(depth=s["depth"], ctx=s["ctx"], cams=s["cams"], pts_voxel_feats=s["pts"],
                                                        img_feats=s["img_feats"], transform=s["transform"]))


def train_main(args, model, dev, rank, world, seen_world, backend_name):
    """``--train``: one step = forward + backward of the differentiable hot path on one resident sample (batch size 1 per GPU, as
    upstream: coocc_ray.py:365).  The semantic losses (CE / lovasz / scal, host-side torch like upstream) are outside the path:
    squared-mean surrogates close the graph.  Weak scaling: every rank trains on its own sample, gradients are NOT all-reduced
    here (the optimiser / DDP wrapper is upstream)."""
    from co_occ_amd import autograd as ag
    c = synth.CONFIGS[args.config]
    s = make_inputs(args.config, 1234 + 17 * rank, dev, model)
    model.train()
    g = torch.Generator(device=dev).manual_seed(1)
    N, D, fH, fW = s["gemo"].shape[1:5]
    rgb_gt = torch.rand(N * fH * 16 * fW * 16, 3, device=dev)
    depth_gt = torch.rand(N * fH * 16 * fW * 16, device=dev) * 50
    render = c["grid"][0] >= 100 and c["grid"][1] >= 100 and c["grid"][2] >= 8
    img = s["img"].clone().requires_grad_()
    # the index search of the NEXT step runs on a side stream under this step (the indices depend on the batch, not on the
    # weights: what a data-loader-side prefetch does in a training loop); --train-prefetch 0 keeps it inside the step
    pre = torch.cuda.Stream(device=dev) if (args.train_prefetch and model.occ_fuser is not None) else None
    ahead = [None]

    def prefetch():
        with torch.cuda.stream(pre), torch.no_grad():
            return model.occ_fuser.search(img.detach(), s["pts"])

    def step():
        for p in model.parameters():
            p.grad = None
        img.grad = None
        sr = None
        if pre is not None:
            sr = ahead[0] if ahead[0] is not None else prefetch()
            ahead[0] = prefetch()
        res = model.forward_train_hot_path(img, s["pts"], s["gemo"], s["img_feats"], s["transform"], render=render, generator=g, search=sr)
        loss = res["logit_rows"].square().mean()
        if "fine_logits" in res:
            loss = loss + res["fine_logits"].square().mean()
        if render:
            L = ag.render_losses(res["rgbs"].reshape(-1, 3), res["depths"].reshape(-1), rgb_gt, depth_gt, D)
            loss = loss + L["loss_rgb"] + L["loss_depth_render"]
        loss.backward()
    for _ in range(max(1, args.warmup)):
        step()
    core.TIMER.enabled, core.TIMER.only = (0 if args.no_kernel_timing else 1), ("k_wgrad", "k_gemm", "k_conv", "conv_fwd", "conv_dgrad")
    core.TIMER.reset()
    cdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    cdist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    core.check_h2_overflow()       # forward and backward operands of the last step stayed inside the f16 range
    core.TIMER.enabled = False
    dt = cdist.max_over_ranks(dt, dev)
    windows = [dt]                  # the training line times ONE window of --steps steps (a step is 25-30 ms)
    ksum = core.TIMER.summary()
    roof = None
    # weight gradients: the split-f16 kernels (csrc/wgrad_h2.hip: three f16 MFMAs per product) and the fp32-MFMA rest; the
    # roofline object is the group with more time in it, the other one rides along
    def wroof(keys_ok, label, peak, mult):
        wg = {k: v for k, v in ksum.items() if keys_ok(k)}
        if not wg:
            return None
        ms, work, n = sum(v["ms"] for v in wg.values()), sum(v["work"] for v in wg.values()), sum(v["launches"] for v in wg.values())
        ach = mult * work / (ms * 1e-3) / 1e12
        return dict(bound="mfma", kernel=label, achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4), traffic=None,
                    launches=n, ms_per_step=round(ms / args.steps, 3),
                    note="HIP events around every launch (GEMM + slice sum) inside the timed region; flops = those the matrix cores execute "
                         "(Winograd-domain for the 3x3x3 stride-1 layers%s)" % (", x3 MFMAs per fp32-accurate product" if mult == 3 else ""))
    r_h2 = wroof(lambda k: k.startswith("k_wgrad_h2"), "k_wgrad_h2 (Winograd-domain + 1x1x1 weight gradients, split-f16 engine)", MFMA_F16_PEAK_TFLOPS, 3)
    r_f32 = wroof(lambda k: k.startswith("k_wgrad") and not k.startswith("k_wgrad_h2"),
                  "k_wgrad (strided / small-grid weight gradients, fp32 MFMA)", MFMA_F32_PEAK_TFLOPS, 1)
    cands = [r for r in (r_h2, r_f32) if r]
    roof = max(cands, key=lambda r: r["ms_per_step"]) if cands else None
    if roof is not None and len(cands) == 2:
        roof = dict(roof, other=[r for r in cands if r is not roof][0])
    groups = {}
    for k, v in ksum.items():
        tag = "wgrad" if k.startswith("k_wgrad") else ("fwd/dgrad GEMMs, split-f16 engine" if k.startswith("k_gemm_h2") else "fwd/dgrad GEMMs, fp32 MFMA")
        gq = groups.setdefault(tag, dict(ms=0.0, launches=0))
        gq["ms"] += v["ms"]; gq["launches"] += v["launches"]
    line = dict(metric="training samples/sec (hot path forward + backward), 200x200x16 grid", value=round(world * args.steps / dt, 4), unit="samples/s",
                n_gpus=world, world_size_seen_by_backend=seen_world, backend=backend_name, steps=args.steps, warmup=args.warmup,
                ms_per_step=round(1e3 * dt / args.steps, 3),
                windows=len(windows), window_ms_per_step=[round(1e3 * w / args.steps, 3) for w in windows],
                spread_pct=round(100.0 * (max(windows) - min(windows)) / dt, 2),
                timing=("%d timed windows of %d steps, each bracketed by barrier + torch.cuda.synchronize() (max over ranks); value / "
                        "ms_per_step = the MEDIAN window" % (len(windows), args.steps)),
                higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype=(("f32 (split-f16 products, fp32 accumulate: forward%s%s GEMMs; fp32 MFMA: the rest)"
                        % (" / dgrad" if ag.TRAIN_H2_DGRAD else "", " / Winograd-domain + 1x1x1 wgrad" if ag.TRAIN_H2_WGRAD else ""))
                       if ag.TRAIN_H2 and core.CONV_ENGINE == "h2" else "f32"),
                data="synthetic",
                config=dict(workload=("coocc_multi_r50_256x704 hot path, TRAINING step" if args.config == "r50" else args.config + " training step"),
                            fused_grid="x".join(map(str, c["grid"])) + "x%d" % c["C"], cams=c["ncam"], knum=c["knum"],
                            batchnorm="batch statistics (model.train())", issue="eager (Python-issued launches)",
                            parallelism="dp%d (1 scene per GPU; gradient all-reduce belongs to the optimiser wrapper upstream)" % world,
                            step=("index search%s + G1 + con_enc + encoder + neck + coarse / fine head + render block, forward + backward; "
                                  "surrogate squared-mean losses on the logits + the two render losses"
                                  % (" (one per step, issued one step ahead on a side stream)" if pre is not None else ""))),
                roofline=roof,
                kernel_groups_ms_per_step={k: dict(ms=round(v["ms"] / args.steps, 3), launches_per_step=v["launches"] // args.steps) for k, v in groups.items()},
                env_knobs={k: v for k, v in sorted(os.environ.items()) if k.startswith("COOCC_")})
    if rank == 0:
        print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="r50", choices=sorted(synth.CONFIGS))
    ap.add_argument("--streams", type=int, default=0,
                    help="dense stages in flight.  --graph 1 (default): dense streams the captured graphs replay on (0 = 3, the "
                         "measured optimum with 6 slots: DESIGN 5).  --graph 0: S host threads / HIP streams, sample i on stream i mod S.  With 2, one "
                         "sample's low-occupancy tail (25x25x2 / 13x13x1 layers, heads, fine branch) runs under the other's GEMMs: "
                         "120.4-127.0 samples/s over 12 processes against 112.5-116.3 at S = 1 on a quiet host, but no gain and "
                         "more spread when neighbours saturate the host's CPUs (profiles/r2_streams_ab.txt).  Default 0: decide "
                         "between 1 and 2 from two short untimed bursts of each before the warm-up")
    ap.add_argument("--stagger", type=int, default=0,
                    help="with --streams > 1: 1 = sample i+1 enters its dense stage after sample i finished the fuser, 2 = after "
                         "its encoder (fixed phase offset between the dense stages in flight); 0 = free running")
    ap.add_argument("--prefetch", type=int, default=1,
                    help="issue the index search of sample i+1 (helper host thread + stream) under the dense stage of sample i")
    ap.add_argument("--reserve-cus", type=int, default=0, help="CUs set aside for the FPS chains (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL on GPUs)")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (single-GPU check of the N > 1 path, gloo)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"],
                    help="f32: exact-fp32 MFMA (the parity path).  bf16: C0-C3 run k_conv_bf16 (operands rounded to bf16, fp32 "
                         "accumulate / epilogue / storage, direct form) -- round 2's reduced-precision path of configs[4].  f16: the same layers on "
                         "v_mfma_f32_32x32x16_f16 with f16 operands in HBM (2 bytes per element), written by the producing layer's epilogue")
    ap.add_argument("--no-pool", action="store_true",
                    help="start the step from an already-pooled camera volume (round-1 definition) instead of the lifted "
                         "depth/context pair (SURVEY.md 8d: 'lifted features + sweep volume -> logits')")
    ap.add_argument("--graph", type=int, default=1,
                    help="1 (default): the dense stage of a sample is one captured hipGraph launch (co_occ_amd.serving.ServingPipeline); 0: every launch "
                         "issued from Python (Pipeline, --streams)")
    ap.add_argument("--with-lidar", action="store_true",
                    help="the LiDAR producer (hard voxelisation of a synthetic 280 k-point cloud -> HardSimpleVFE -> SparseLiDAREnc8x, "
                         "coocc_ray.py:215-234) runs inside the step on the prefetch stream instead of a ready-made LiDAR volume "
                         "(serving pipeline only; its own workload: the encoder's active set is denser than the default volume)")
    ap.add_argument("--train-prefetch", type=int, default=1, help="--train: 1 = the index search of the next step runs on a side stream under "
                    "the current step (indices depend on the batch only); 0 = inside the step")
    ap.add_argument("--ahead", type=int, default=0, help="--graph 1: searches submitted ahead of the dense stage (0: slots - dense streams; "
                                                         "with more than one rank: 1 -- one helper thread per rank, so an 8-rank node does not run 32 host threads)")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step of the hot path instead (SURVEY.md 8f rank 1: COOCC_Ray.forward_train_hot_path -- index "
                         "search, G1, con_enc, encoder, neck, coarse + fine head, render block -- forward + backward with batch-statistics "
                         "BatchNorm, surrogate scalar losses closing the graph) and print its own JSON line with a wgrad roofline")
    ap.add_argument("--api", default="serving", choices=["serving", "simple_test", "pipelined_test"],
                    help="serving (default): COOCC_Ray.serving() -- co_occ_amd.serving.ServingPipeline, --slots samples in flight.  "
                         "simple_test: the reference's per-sample call COOCC_Ray.simple_test(precomputed=...) (coocc_ray.py:520), "
                         "synchronous, dense stage = one captured hipGraph launch (one slot).  pipelined_test: the reference's test "
                         "LOOP (custom_single_gpu_test, P/coocc/apis/test.py:22-60) = co_occ_amd.apis.pipelined_test with ground truth "
                         "per sample: SC / SSC confusion matrices computed on the device and delivered to the host as numpy arrays, "
                         "every sample's result dict yielded in order")
    ap.add_argument("--windows", type=int, default=3,
                    help="timed windows of --steps steps each (barrier + synchronize on both sides of every window); the line reports the "
                         "MEDIAN window (value, ms_per_step) and all of them (window_ms_per_step, spread_pct)")
    ap.add_argument("--slots", type=int, default=6, help="--graph 1: samples in flight (1 in its dense stage + slots-1 in the prefetched search)")
    ap.add_argument("--shard", default="samples", choices=["samples", "rays"],
                    help="samples (default): one scene per GPU, weak scaling (configs[3]).  rays: ONE scene over all ranks -- K / G / C "
                         "replicated, the render rays sharded, map chunks all-gathered (configs[4]'s 'per-camera render shard'); "
                         "strong scaling: value = scenes/s of the whole job")
    ap.add_argument("--also", default="auto",
                    help="a second, SHORT measurement appended to the JSON line as \"also\": {name: {value, ms_per_step, ...}} -- north_star's literal "
                         "stress workload (--config stress200_r101: 200x200x16 fused grid + 6 x 896 x 1600 frames), 2 windows x 10 steps in a "
                         "child process after the main measurement (<= 25 s).  auto: only for the default command's workload (config r50, "
                         "serving API, one GPU, no --train / --with-lidar / --shard rays); none: never; or a config name")
    ap.add_argument("--diag", action="store_true", help="host-side issue times per sample to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel event timing table to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank, world, local = cdist.init(backend=args.backend)
    if args.same_device:
        local = 0           # control-flow check of the multi-rank path on a single GPU (use with --backend gloo)
    if world != args.gpus:  # never fall back to fewer ranks than asked for: the scaling curve would be silently wrong
        raise SystemExit("bench.py: --gpus %d but the process group has %d rank(s) (WORLD_SIZE=%s); launch with "
                         "torch.distributed.run --nproc-per-node %d or without a launcher" % (
                             args.gpus, world, os.environ.get("WORLD_SIZE"), args.gpus))
    import torch.distributed as tdist
    seen_world = tdist.get_world_size() if (tdist.is_available() and tdist.is_initialized()) else 1
    backend_name = tdist.get_backend() if seen_world > 1 else None
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    WITH_POOL[0] = not args.no_pool
    CFGNAME[0] = args.config
    core.CONV_DTYPE = args.dtype
    auto_streams = args.streams == 0
    if auto_streams:
        args.streams = 2
    STAGGER[0] = args.stagger if args.streams > 1 else 0
    SHARD[0] = args.shard == "rays"
    if SHARD[0]:
        args.graph, args.streams = 0, 1           # the sharded render has a collective inside the step: eager, one sample in flight
        auto_streams = False
    model, sd = build_model(args.config, dev, with_lidar=args.with_lidar)
    if args.train:
        return train_main(args, model, dev, rank, world, seen_world, backend_name)
    samples = [make_inputs(args.config, 1234 + (0 if SHARD[0] else 17 * rank) + i, dev, model) for i in range(max(2, args.slots if args.graph else 2))]
    if args.with_lidar:
        # raw points per sample; their volume (what the eager warm-up / roofline passes read) from the producer itself, untimed
        args.no_cpu_baseline = True            # the CPU oracle of the sparse encoder is a test-size restatement (oracle/ref_lidar.py)
        for i, s_ in enumerate(samples):
            s_["points"] = synth.lidar_points(seed=8 + i + 17 * rank).to(dev)
            with torch.no_grad():
                s_["pts"] = model.extract_pts_feat(s_["points"])[0]
    if args.reserve_cus > 0 and not args.graph:   # (eager pipeline only; with the graph loop masked streams ran 4-6x slower: profiles/r6_reserve_cus_*.txt)
        # CU partition: the FPS chains get private CUs, everything else runs on the remaining ones
        from co_occ_amd import streams as cstreams
        parts = [cstreams.partition(dev, reserved=args.reserve_cus) for _ in range(max(1, args.streams))]
        streams = [p.main for p in parts]
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    S = len(streams)
    pipe = Pipeline(model, samples, dev, streams, prefetch=bool(args.prefetch), world=world)
    run, tpool = (lambda n, timed, S=S: pipe.run(n, S)), pipe.tpool

    with torch.no_grad():
        step(model, samples[0], world)          # packs weights, sizes workspaces (untimed, extra to --warmup)
    torch.cuda.synchronize()
    # every dense / prefetch stream owns its scratch (split-K slabs, Winograd buffers, search workspaces): size them for both
    # synthetic samples before anything is timed, whatever --warmup is (a first use inside the timed region is a multi-GB
    # allocation + zero fill: seen once as 110 instead of 54 ms per step at stress200 with --warmup 1)
    run(2 * S, False)
    probe = None
    gp = None
    frames = [frame_of(x) for x in samples]
    if world > 1 and args.ahead <= 0:
        args.ahead = 1          # one search helper thread per rank: 8 ranks x (1 issuing + 1 helper) threads on a CPU-capped node
    if world > 1:               # ... each rank on its own slice of the host cores (single node: LOCAL_WORLD_SIZE ranks share them)
        cdist.pin_rank_threads(int(os.environ.get("LOCAL_RANK", rank)), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if args.graph and WITH_POOL[0] and args.api == "serving":
        if world > 1 and GATHERER[0] is None:
            # the rank's all-gathers from one thread that waits for each replay's completion on the HOST first (dist.GatherThread)
            GATHERER[0] = cdist.GatherThread(dev)
        try:
            gp = model.serving(frames[0], slots=max(2, args.slots), dense_streams=max(1, args.streams if not auto_streams else 3),
                               ahead=args.ahead, search_priority=int(os.environ.get("COOCC_SEARCH_PRIO", "0")),
                               after_done=(GATHERER[0].submit if world > 1 else None))
            gp.run(frames, 2 * gp.n)
        except Exception as e:           # configurations the static form does not cover run the eager pipeline
            print("bench: hipGraph pipeline unavailable for this configuration (%s: %s); eager pipeline" % (type(e).__name__, e), file=sys.stderr)
            gp = None
    if gp is not None:
        auto_streams, S = False, 1
        run_eager = run

        def run(n, timed, S=1):
            gp.time_dense = timed and os.environ.get("COOCC_BENCH_TIME_DENSE", "1") != "0"
            gp.run(frames, n)
            with torch.cuda.stream(gp.dense_streams[0]):
                drain_gathers()
            for ds_ in gp.dense_streams:
                ds_.synchronize()
    st_api = None
    if args.api == "simple_test":
        # the reference's per-sample call: every step is one synchronous COOCC_Ray.simple_test (one-slot captured graph)
        model.test_rendering = True
        st_kw = [simple_test_kwargs(x) for x in samples]
        auto_streams, S = False, 1
        run_eager = run

        def run(n, timed, S=1):
            with torch.no_grad():
                for i in range(n):
                    o = model.simple_test(**st_kw[i % len(st_kw)])
                    if world > 1:
                        _gather(o)
            drain_gathers()
            torch.cuda.synchronize()
        run(3, False)
        st_api = dict(graph=model._pipe1 is not None, eager_reason=(model.graph_unavailable or (None, None))[1])
    if args.api == "pipelined_test":
        # the reference's test loop: ONE generator over an endless stream of samples (the pipeline and its captured graphs are made
        # at its first sample, in the warm-up); a step = one (data, result) pair yielded, metrics on the host as numpy arrays
        from co_occ_amd import apis
        model.test_rendering = True
        c_ = synth.CONFIGS[args.config]
        gsz = c_.get("final_occ_size", [2 * g for g in c_["grid"]])
        g_ = torch.Generator().manual_seed(77 + rank)
        gts = [torch.randint(0, 17, (1,) + tuple(gsz), generator=g_).to(dev) for _ in samples]
        pt_data = [dict(simple_test_kwargs(x), gt_occ=gt) for x, gt in zip(samples, gts)]
        for d_ in pt_data:
            d_.pop("img"), d_.pop("points")

        def endless():
            i = 0
            while True:
                yield pt_data[i % len(pt_data)]
                i += 1
        pt_stats = {}
        pt_gen = apis.pipelined_test(model, endless(), slots=max(2, args.slots), dense_streams=(3 if auto_streams else max(1, args.streams)),
                                     ahead=args.ahead, stats=pt_stats)
        auto_streams, S = False, 1
        run_eager = run
        pt_seen = [0]

        def run(n, timed, S=1):
            for _ in range(n):
                d_, res = next(pt_gen)
                pt_seen[0] += int(res["SSC_metric_fine" if "SSC_metric_fine" in res else "SSC_metric"].sum() > 0)
                if world > 1:
                    _gather(res)
            drain_gathers()
            torch.cuda.synchronize()
        run(2 * max(2, args.slots), False)
        st_api = dict(graph=True, eager_reason=None, pipelined=True)
    if auto_streams:
        # One or two samples in flight?  Two win by ~8 % when the host keeps up (four Python threads share the GIL) and lose
        # that margin when neighbours saturate the box's CPUs (profiles/r2_streams_ab.txt) -- so ask the box: two untimed
        # bursts of each, the faster total decides.  The rule is the same on every rank (the decision is rank-local: collectives
        # are issued in sample order whatever the stream count).
        nprobe = max(8, min(16, args.steps))
        tt = {1: 0.0, 2: 0.0}
        for _ in range(2):
            for Sp in (1, 2):
                torch.cuda.synchronize()
                tp = time.perf_counter()
                run(nprobe, False, S=Sp)
                torch.cuda.synchronize()
                tt[Sp] += time.perf_counter() - tp
        S = 2 if tt[2] < tt[1] else 1
        probe = dict(samples_per_burst=nprobe, bursts_each=2, samples_per_s={k: round(2 * nprobe / v, 2) for k, v in tt.items()},
                     chosen=S)
        run = (lambda n, timed, S=S: pipe.run(n, S))
    run(args.warmup, False)
    graphish = gp is not None or st_api is not None       # the dense stage is a captured graph: no per-launch events inside the timed region
    if not graphish:
        core.TIMER.enabled = 0 if args.no_kernel_timing else (2 if args.kernel_table else 1)
    core.TIMER.only = ("k_conv", "k_gemm", "k_render_nearest", "k_lift_splat")      # what the roofline objects below need
    core.TIMER.reset()
    windows = []
    nwin = max(1, args.windows)
    for wi in range(nwin):
        cdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if wi == nwin - 1:
            for v in DIAG.values():
                del v[:]
            if args.diag:
                TRACE[0], TRACE_T0[0] = [], time.perf_counter()
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
        run(args.steps, True)
        cdist.barrier()
        torch.cuda.synchronize()
        windows.append(cdist.max_over_ranks(time.perf_counter() - t0, dev))
    dt = sorted(windows)[len(windows) // 2] if len(windows) % 2 else sorted(windows)[len(windows) // 2 - 1]
    if args.diag and rank == 0 and TRACE[0]:
        for i, w, a, b, c, ev in sorted(TRACE[0]):
            print("trace: sample %2d stream %d  host: search ready %.1f  issue %.1f -> %.1f ms   GPU done %.1f ms" % (
                i, w, 1e3 * b, 1e3 * b, 1e3 * c, ev0.elapsed_time(ev)), file=sys.stderr)
        TRACE[0] = None
    if args.diag and rank == 0:
        print("diag: " + json.dumps({k: dict(mean_ms=round(1e3 * sum(v) / max(len(v), 1), 3), max_ms=round(1e3 * max(v or [0]), 3))
                                     for k, v in DIAG.items()}, sort_keys=True) + " wall_ms_per_step %.3f" % (1e3 * dt / args.steps),
              file=sys.stderr)
    core.TIMER.enabled = False

    c = synth.CONFIGS[args.config]
    graph_info = None
    if graphish:
        if gp is not None:
            dense_ms = [ev[0].elapsed_time(ev[1]) for ev in gp.dense_ev]
            graph_info = dict(api="COOCC_Ray.serving() -> co_occ_amd.serving.ServingPipeline", slots=gp.n, dense_streams=gp.ndense,
                              searches_ahead=gp.ahead, eager_fallbacks=gp.fallbacks, recaptures=gp.recaptures,
                              dense_stage_ms=round(sum(dense_ms) / max(1, len(dense_ms)), 3),
                              note="dense stage = ONE hipGraphLaunch per sample (HIP events around every replay inside the timed region: "
                                   "dense_stage_ms); pooling + index search eager on prefetch streams; a new frame's image features and "
                                   "camera matrices are copied into the slot's static tensors")
        elif st_api.get("pipelined"):
            graph_info = dict(api="co_occ_amd.apis.pipelined_test(model, data_iter) -- the reference's test loop (custom_single_gpu_test) with "
                                  "samples in flight; ground truth per sample, SC / SSC confusion matrices on the device, numpy on the host",
                              slots=max(2, args.slots), samples_with_nonzero_metrics=pt_seen[0], **pt_stats)
        else:
            graph_info = dict(api="COOCC_Ray.simple_test(precomputed=...) -- the reference's per-sample call, synchronous", slots=1,
                              captured_graph=st_api["graph"], eager_reason=st_api["eager_reason"],
                              eager_fallbacks=(model._pipe1[1].fallbacks if model._pipe1 is not None else None))
        if not args.no_kernel_timing:
            # per-kernel durations cannot be taken inside a graph launch: an eager pass of the same step (one sample in flight,
            # search prefetched) with HIP events around every launch, right after the timed region
            n_e = max(8, min(20, args.steps))
            run_eager(3, False, S=1)
            core.TIMER.enabled = 2 if args.kernel_table else 1
            core.TIMER.reset()
            run_eager(n_e, False, S=1)
            torch.cuda.synchronize()
            core.TIMER.enabled = False
    roof, extra = rooflines(core.TIMER.summary(), (max(8, min(20, args.steps)) if graphish else args.steps), args, rank, args.kernel_table)
    if graphish and roof:
        roof["measured"] = ("HIP events around every launch of an eager pass of the same steps right after the timed region "
                            "(inside it the dense stage is one hipGraphLaunch per sample); the next sample's pooling + index search run "
                            "beside it as in the timed region")
        if not args.no_kernel_timing:
            # the dominant kernel's own rate: the same eager steps with NOTHING else on the GPU (no prefetch: search, then dense stage)
            pipe_alone = Pipeline(model, samples, dev, streams[:1], prefetch=False, world=1)
            pipe_alone.run(2, 1)
            core.TIMER.enabled, core.TIMER.only = 1, ("k_conv", "k_gemm")
            core.TIMER.reset()
            pipe_alone.run(6, 1)
            torch.cuda.synchronize()
            core.TIMER.enabled, core.TIMER.only = False, None
            r_al, _ = rooflines(core.TIMER.summary(), 6, args, rank, False)
            core.TIMER.reset()
            if r_al and r_al.get("kernel") == roof.get("kernel"):
                roof.update(achieved_alone=r_al["achieved"], frac_alone=r_al["frac"], avg_launch_ms_alone=r_al["avg_launch_ms"],
                            note="achieved / frac: launches that share the CUs with the prefetched search of the next sample (as in the "
                                 "timed pipeline); *_alone: the same launches with nothing else on the GPU (6 samples, no prefetch)")
    if S > 1 and not graphish and not args.no_kernel_timing:
        # Kernel durations inside the S-stream pipeline include the contention between the samples in flight (that is the
        # point of it: one sample's low-occupancy tail runs under the other's GEMMs).  A short pass of the same pipeline with ONE
        # sample in flight (the next sample's pooling + search still prefetched) gives each kernel's own rate next to it.
        n_iso = max(8, min(20, args.steps // 2))
        run(3, False, S=1)
        core.TIMER.enabled, core.TIMER.only = 1, ("k_conv", "k_gemm", "k_render_nearest", "k_lift_splat")
        core.TIMER.reset()
        run(n_iso, False, S=1)
        torch.cuda.synchronize()
        core.TIMER.enabled = False
        r_iso, e_iso = rooflines(core.TIMER.summary(), n_iso, args, rank, False)
        core.TIMER.reset()
        if roof and r_iso and r_iso.get("kernel") == roof.get("kernel"):
            # the same kernel with nothing else on the GPU, inside the headline object: a launch that shares the CUs with the
            # other sample in flight takes longer while the GPU as a whole does more work per unit time
            roof.update(achieved_alone=r_iso["achieved"], frac_alone=r_iso["frac"], avg_launch_ms_alone=r_iso["avg_launch_ms"],
                        note="achieved / frac: HIP-event durations inside the timed %d-sample pipeline (launches of two samples "
                             "share the CUs); *_alone: the same launches with one sample in flight, measured after the timed "
                             "region (roofline_isolated)" % S)
        extra["roofline_isolated"] = dict(r_iso or {}, note="separate pass after the timed region: the same pipeline with one sample in flight (--streams 1)",
                                          all_convs=e_iso.get("roofline_all_convs"), render=e_iso.get("roofline_render"),
                                          pool=e_iso.get("roofline_pool"))
    if rank == 0 and args.config == "r50" and not args.no_kernel_timing:
        extra["roofline_render_r101"] = render_r101_roofline(model, dev)
    jobs = 1 if SHARD[0] else world          # --shard rays: all ranks work on the same scene
    line = dict(metric="samples/sec (6-cam frame + sweep -> occ+render), 200x200x16 grid", value=round(jobs * args.steps / dt, 4),
                unit="samples/s", n_gpus=world, world_size_seen_by_backend=seen_world, backend=backend_name, steps=args.steps,
                warmup=args.warmup,
                ms_per_step=round(1e3 * dt / args.steps, 3),
                windows=len(windows), window_ms_per_step=[round(1e3 * w / args.steps, 3) for w in windows],
                spread_pct=round(100.0 * (max(windows) - min(windows)) / dt, 2),
                timing=("%d timed windows of %d steps, each bracketed by barrier + torch.cuda.synchronize() (max over ranks); value / "
                        "ms_per_step = the MEDIAN window" % (len(windows), args.steps)),
                higher_is_better=True, scaling=("strong" if SHARD[0] else "weak"), vs_baseline=None,
                dtype=("f32 (split-f16 products, fp32 accumulate)" if (args.dtype == "f32" and core.CONV_ENGINE == "h2") else args.dtype),
                data="synthetic",
                config=dict(workload=("coocc_multi_r50_256x704 hot path" if args.config == "r50" else args.config) +
                            (" + LiDAR producer in the step (280 k synthetic points -> hard voxelisation -> HardSimpleVFE -> SparseLiDAREnc8x)"
                             if args.with_lidar else ""),
                            fused_grid="x".join(map(str, c["grid"])) + "x%d" % c["C"],
                            occupancy_grid="x".join(str(v) for v in c.get("final_occ_size", [2 * g for g in c["grid"]])), cams=c["ncam"],
                            render_maps="%dx%dx%d" % (c["ncam"], c["fmap"][0] * 16, c["fmap"][1] * 16), knum=c["knum"],
                            parallelism=("ray-shard x%d (ONE scene: K/G/C replicated on every rank, render rays sharded, all-gather of the "
                                         "map chunks)" % world if SHARD[0] else "dp%d (1 scene per GPU, RCCL all-gather of maps)" % world),
                            samples_in_flight=(gp.n if gp is not None else S), prefetched_search=bool(tpool) and (st_api is None or bool(st_api.get("pipelined"))), weights="random",
                            pipeline=("hipGraph dense stage + eager prefetched search" if gp is not None else
                                      "apis.pipelined_test loop (metrics per sample), hipGraph dense stage + prefetched search" if (st_api or {}).get("pipelined") else
                                      "simple_test per sample (synchronous), hipGraph dense stage" if st_api is not None else
                                      "eager (Python-issued launches)"),
                            conv_engine=(core.CONV_ENGINE if args.dtype == "f32" else args.dtype),
                            step_starts_from=("lifted depth/context pair (fused lift-splat pooling inside the step)" if WITH_POOL[0]
                                              else "pooled camera volume")),
                roofline=roof)
    line.update(extra)
    if graph_info is not None:
        line["graph"] = graph_info
    if SHARD[0] and SHARD_EV:
        torch.cuda.synchronize()
        evs = SHARD_EV[-args.steps:]
        rms = sum(e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for e in evs) / len(evs)      # table build + rays, upsample
        gms = sum(e[1].elapsed_time(e[2]) for e in evs) / len(evs)                                 # the all-gather, incl. the wait for the slowest rank
        line["ray_shard"] = dict(render_ms_per_step=round(rms, 3), gather_ms_per_step=round(gms, 3),
                                 sharded_fraction_of_step=round(rms / (1e3 * dt / args.steps), 4),
                                 note="only the render block shards (table build + rays + gather + upsample); the index search and the "
                                      "3-D conv stack are replicated on every rank, so this mode cannot scale: one scene per GPU "
                                      "(--shard samples) is the scaling configuration")
    knobs = {k: v for k, v in sorted(os.environ.items()) if k.startswith("COOCC_")}
    line["env_knobs"] = knobs           # every dispatch-changing environment variable that was set for this run
    if probe is not None:
        line["stream_probe"] = dict(probe, note="untimed bursts before the warm-up: samples in flight chosen by the faster total "
                                                "(--streams 1 / 2 fixes it)")
    if world == 1 and not args.no_cpu_baseline and args.config != "openocc":   # cascade 4: ~10 M fine points, hours on the CPU
        line["cpu_baseline"] = cpu_baseline(sd, samples[0], args.config, WITH_POOL[0])
    also_cfg = None
    if args.also == "auto":
        if (world == 1 and args.config == "r50" and args.api == "serving" and args.graph and not args.with_lidar and not SHARD[0]
                and args.dtype == "f32" and not args.no_cpu_baseline):
            also_cfg = "stress200_r101"
    elif args.also not in ("none", "0", ""):
        also_cfg = args.also
    if rank == 0 and also_cfg:
        line["also"] = {also_cfg: also_line(also_cfg, dev)}
    if rank == 0:
        print(json.dumps(line))


def also_line(cfg, dev):
    """A short run of another workload in a CHILD process (its own model, graphs and workspaces; this process's GPU memory is released
    first), reduced to {value, unit, ms_per_step, window_ms_per_step, steps, config}.  Never raises: a failure is reported in the object."""
    import subprocess
    try:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", "10", "--warmup", "3", "--windows", "2",
               "--no-cpu-baseline", "--no-kernel-timing", "--also", "none"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
        wall = time.perf_counter() - t0
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], window_ms_per_step=d["window_ms_per_step"],
                    steps=d["steps"], windows=d["windows"], config=d["config"], wall_s=round(wall, 1),
                    command=" ".join(["python", "bench.py"] + cmd[2:]))
    except Exception as e:
        return dict(error="%s: %s" % (type(e).__name__, str(e)[:300]))


if __name__ == "__main__":
    main()
