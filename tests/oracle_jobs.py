"""CPU-oracle evaluations of the large parity scenes, run in a PROCESS POOL next to the GPU tests.

The wall time of ``pytest -m gpu`` was CPU-oracle time (fp32 + fp64 evaluations of the whole decoder at full size, the literal
gather-then-MLP render block at 6 x 56 x 100 x 112 samples): 1 099 s of the driver's 1 200 s limit.  Nothing about those
evaluations needs the GPU or each other, so ``conftest.py`` submits the jobs the selected tests will ask for when the session
starts (longest first, a few workers x a few threads: torch's CPU conv3d scales poorly past ~4 threads) and the tests
``get(key)`` the result -- normally already there by the time the alphabetical order reaches them.  Without a pool (one test
run by hand, or the pool disabled with COOCC_ORACLE_POOL=0) ``get`` evaluates inline: same functions, same results.

This module is test infrastructure: it is the only place in ``tests/`` that decides WHICH oracle form a large scene is judged
against.  The literal render form (coocc_ray.py:570-627: gather the 128-channel feature of every ray sample, then the MLPs) is
evaluated on a STRIDED SUBSET of rays and required to equal the per-voxel-table form there (``render_table_with_literal_check``);
the whole maps are then judged against the table form (oracle/gen_golden.py shows the two within 1.8e-7 / 9.5e-7 on the
reference's own block)."""
import os
import time

import numpy as np
import torch

import co_occ_amd as pkg
import co_occ_amd.synth as synth
from oracle import ref_cpu

SMALL_RANGE = (-25, -25, -5.0, 25, 25, 3.0)
R50_RANGE = (-50, -50, -5.0, 50, 50, 3.0)
SWEEP = [(seed, 1.0) for seed in (5, 6, 7, 8, 9, 10, 11, 12)] + [(seed, 0.85) for seed in (5, 6, 7, 8)]
LITERAL_STRIDE = (2, 3)          # every 2nd row x 3rd column of the ray grid gets the literal form (1 / 6 of the rays)


# ------------------------------------------------------------------------------------------------ scenes (shared with the tests)
def scene(grid, fmap, ncam, input_size, seed, gain, final_occ, pc_range, knum=2, C=128, want_model=True):
    cfg = synth.model_cfg(C=C, knum=knum, final_occ_size=final_occ, point_cloud_range=pc_range, input_size=input_size)
    model = pkg.build_detector(cfg)
    sd = synth.random_state_dict(model.state_dict(), seed=seed, gain=gain)
    if want_model:
        model.load_state_dict(sd)
    img, pts = synth.voxel_inputs(grid, C=C, seed=70 + seed)
    rig = synth.camera_rig(ncam, input_size, seed=70 + seed)
    img_feats = [synth.image_feats(ncam, fmap, 512, seed=70 + seed)]
    return (model if want_model else None), sd, img, pts, rig, img_feats


def sweep_scene(seed, gain, want_model=True):
    return scene((50, 50, 8), (4, 11), 6, (64, 176), seed, gain, (100, 100, 16), SMALL_RANGE, want_model=want_model)


def r50_scene(want_model=True):
    c = synth.CONFIGS["r50"]
    model, sd, img, pts, rig, img_feats = scene(c["grid"], c["fmap"], c["ncam"], (256, 704), 3, 1.0, (200, 200, 16), R50_RANGE,
                                                want_model=want_model)
    tr = synth.rig_transform(rig)
    fr = ref_cpu.create_frustum((256, 704), 16, [2.0, 58.0, 0.5])
    gemo = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    return dict(model=model, sd=sd, img=img, pts=pts, img_feats=img_feats, tr=tr, gemo=gemo)


def stress200_scene(want_model=True):
    c = synth.CONFIGS["stress200_r101"]
    X, Y, Z = c["grid"]
    H, W = c["fmap"][0] * 16, c["fmap"][1] * 16
    model, sd, img, pts, rig, img_feats = scene(c["grid"], c["fmap"], c["ncam"], (H, W), 2, 1.0, (2 * X, 2 * Y, 2 * Z),
                                                (-100, -100, -5.0, 100, 100, 11.0), want_model=want_model)
    tr = synth.rig_transform(rig)
    fr = ref_cpu.create_frustum((H, W), 16, [2.0, 58.0, 0.5])
    gemo = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    return dict(model=model, sd=sd, img=img, pts=pts, img_feats=img_feats, tr=tr, gemo=gemo, grid=(X, Y, Z), HW=(H, W))


def openocc_e2e_scene(want_model=False):
    c = synth.CONFIGS["openocc"]
    seed, gain = 5, 0.85
    model = pkg.build_detector(synth.model_cfg_openocc())
    sd = synth.random_state_dict(model.state_dict(), seed=seed, gain=gain)
    img, pts = synth.voxel_inputs(c["grid"], C=c["C"], seed=70 + seed)
    rig = synth.camera_rig(c["ncam"], c["input_size"], seed=70 + seed)
    img_feats = [synth.image_feats(c["ncam"], c["fmap"], 512, seed=70 + seed)]
    tr = synth.rig_transform(rig)
    fr = ref_cpu.create_frustum(c["input_size"], 16, [2.0, 58.0, 0.5])
    gemo = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    kw = dict(knum=2, cascade_ratio=4, final_occ_size=c["final_occ_size"], point_cloud_range=c["point_cloud_range"],
              fine_subset=_fine_subset)
    return dict(c=c, sd=sd, img=img, pts=pts, rig=rig, img_feats=img_feats, tr=tr, gemo=gemo, kw=kw)


def _fine_subset(n):
    g = torch.Generator().manual_seed(1234)
    return torch.randperm(n, generator=g)[:60000].sort().values


def openocc_trunk_scene(want_model=True, seed=9, gain=0.85):
    cfg = synth.model_cfg_openocc()
    model = pkg.build_detector(cfg)
    sd = synth.random_state_dict(model.state_dict(), seed=seed, gain=gain)
    if want_model:
        model.load_state_dict(sd)
    X, Y, Z = synth.CONFIGS["openocc"]["grid"]
    g = synth._rng(seed, "cat4")
    cat4 = torch.from_numpy(g.standard_normal((1, 512, X, Y, Z), dtype=np.float32))
    cat4 *= torch.from_numpy((g.random((1, 1, X, Y, Z)) < 0.7).astype(np.float32))
    return (model if want_model else None), sd, cat4


def r101_render_scene():
    c = synth.CONFIGS["r101"]
    seed = 4
    g = synth._rng(seed, "vf")
    X, Y, Z = c["grid"]
    vf = torch.from_numpy(g.standard_normal((1, 128, X, Y, Z), dtype=np.float32))
    rig = synth.camera_rig(6, (896, 1600), seed=seed)
    from co_occ_amd import render as R
    sig, rgb = R.MLP(128, 1, net_depth=1, skip_layer=None), R.MLP(128, 3, net_depth=3, skip_layer=None)
    ssd, rsd = synth.random_state_dict(sig.state_dict(), seed), synth.random_state_dict(rgb.state_dict(), seed + 1)
    fr = ref_cpu.create_frustum((896, 1600), 16, [2.0, 58.0, 0.5])
    gemo = ref_cpu.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"], rig["bda"])
    return dict(vf=vf, ssd=ssd, rsd=rsd, gemo=gemo)


# ------------------------------------------------------------------------------------------------ the render forms
def render_table_with_literal_check(sigma_sd, rgb_sd, voxel_feats, gemo, stride=LITERAL_STRIDE):
    """The render block in the per-voxel-table form over ALL rays (-> upsampled maps), and the LITERAL gather-then-MLP form
    (coocc_ray.py:575-616) on every ``stride``-th ray: returns (rgbs, depths, dev) with dev = (max |literal - table| of the
    low-resolution rgb map, of the depth map relative to its scale) on the subset.  The callers assert dev <= 2e-6."""
    B, N, D, H, W, _ = gemo.shape
    assert B == 1
    sh, sw = stride
    rgbs, depths, d_rgb, d_dep = [], [], 0.0, 0.0
    for i in range(N):
        r, d = ref_cpu.render_camera(sigma_sd, rgb_sd, voxel_feats[0], gemo[0, i], literal=False)
        rl, dl = ref_cpu.render_camera(sigma_sd, rgb_sd, voxel_feats[0], gemo[0, i][:, ::sh, ::sw].contiguous(), literal=True)
        d_rgb = max(d_rgb, float((rl - r[::sh, ::sw]).abs().max()))
        d_dep = max(d_dep, float((dl - d[::sh, ::sw]).abs().max()) / max(1.0, float(d.abs().max())))
        r, d = ref_cpu.upsample16(r, d)
        rgbs.append(r)
        depths.append(d)
    return torch.stack(rgbs), torch.stack(depths), (d_rgb, d_dep)


def _hot_path_table_render(sd, img, pts, gemo, img_feats, tr, **kw):
    """hot_path_forward with the render block in the table form + the literal subset check."""
    o = ref_cpu.hot_path_forward(sd, img, pts, gemo, img_feats, tr, render=False, **kw)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    o["rgbs"], o["depths"], o["literal_dev"] = render_table_with_literal_check(sub("sigma_head."), sub("rgb_head."), o["voxel_feats"], gemo)
    return o


def _slim(o, keys=("voxel_feats", "output_voxels", "fine_output", "fine_coord", "fine_coord_all", "rgbs", "depths", "literal_dev")):
    return {k: o[k] for k in keys if k in o and o[k] is not None}


def _fuser_sd(sd):
    return {k[len("occ_fuser."):]: v for k, v in sd.items() if k.startswith("occ_fuser.")}


# ------------------------------------------------------------------------------------------------ jobs
def job_sweep(seed, gain):
    _, sd, img, pts, rig, img_feats = sweep_scene(seed, gain, want_model=False)
    tr = synth.rig_transform(rig)
    kw = dict(knum=2, final_occ_size=(100, 100, 16), point_cloud_range=SMALL_RANGE, render=False)
    o32 = ref_cpu.hot_path_forward(sd, img, pts, None, img_feats, tr, **kw)
    o64 = ref_cpu.hot_path_forward(sd, img, pts, None, img_feats, tr, dtype=torch.float64, **kw)
    return dict(o32=_slim(o32), o64=_slim(o64))


def job_r50(which):
    S = r50_scene(want_model=False)
    a = (S["sd"], S["img"], S["pts"], S["gemo"], S["img_feats"], S["tr"])
    if which == "o32":
        # r50's render block is small (6 x 16 x 44 rays): the literal form over ALL rays, as the reference evaluates it
        return _slim(ref_cpu.hot_path_forward(*a, knum=2, literal_render=True))
    if which == "o64":
        return _slim(ref_cpu.hot_path_forward(*a, knum=2, dtype=torch.float64, render=False))
    f = ref_cpu.bifuser_fuse(_fuser_sd(S["sd"]), S["img"], S["pts"], 2)
    return dict(near_img=f["near_img"], near_pts=f["near_pts"])


def job_openocc_e2e(which):
    S = openocc_e2e_scene()
    a = (S["sd"], S["img"], S["pts"], S["gemo"], S["img_feats"], S["tr"])
    if which == "o32":
        return _slim(_hot_path_table_render(*a, **S["kw"]))
    if which == "o64":
        return _slim(ref_cpu.hot_path_forward(*a, dtype=torch.float64, render=False, **S["kw"]))
    if which == "o16":
        ref_cpu.CONV_OPERAND_DTYPE = torch.float16
        try:
            return _slim(ref_cpu.hot_path_forward(*a, literal_render=False, **S["kw"]))
        finally:
            ref_cpu.CONV_OPERAND_DTYPE = None
    f = ref_cpu.bifuser_fuse(_fuser_sd(S["sd"]), S["img"], S["pts"], 2)
    return dict(near_img=f["near_img"], near_pts=f["near_pts"])


def oracle_trunk(sd, cat4, dtype=None):
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    x = cat4
    if dtype is not None:
        sd, x = ref_cpu.to_dtype(dict(sd), dtype), cat4.to(dtype)
        sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    vf = ref_cpu.con_enc(sub("occ_fuser."), x.permute(0, 2, 3, 4, 1))
    sem = ref_cpu.fpn3d_forward(sub("semantic_neck."), ref_cpu.resnet3d_forward(sub("semantic_encoder."), vf))
    return vf, ref_cpu.occhead_coarse(sub("pts_bbox_head."), sem)["occ"]


def job_openocc_trunk(which):
    _, sd, cat4 = openocc_trunk_scene(want_model=False)
    if which == "o64":
        return oracle_trunk(sd, cat4, torch.float64)
    ref_cpu.CONV_OPERAND_DTYPE = {"o32": None, "obf": torch.bfloat16, "of16": torch.float16}[which]
    try:
        return oracle_trunk(sd, cat4)
    finally:
        ref_cpu.CONV_OPERAND_DTYPE = None


def job_r101_render():
    S = r101_render_scene()
    wr, wd, dev = render_table_with_literal_check(S["ssd"], S["rsd"], S["vf"], S["gemo"])
    return dict(rgbs=wr, depths=wd, literal_dev=dev)


STRESS_CROPS = 24


def job_stress200():
    """Neighbour tables of the 200x200x16 scene and the oracle's two con_enc layers on cropped 5^3 neighbourhoods at seeded
    positions (the whole decoder at this size takes the CPU tens of minutes)."""
    S = stress200_scene(want_model=False)
    X, Y, Z = S["grid"]
    fsd = _fuser_sd(S["sd"])
    fuse = ref_cpu.bifuser_fuse(fsd, S["img"], S["pts"], 2)
    allf = fuse["all_feats"]                                      # [1,X,Y,Z,4C]
    g = np.random.default_rng(3)
    pos, want = [], []
    for _ in range(STRESS_CROPS):
        cx, cy, cz = int(g.integers(0, X)), int(g.integers(0, Y)), int(g.integers(0, Z))
        x0, x1, y0, y1, z0, z1 = max(cx - 2, 0), min(cx + 3, X), max(cy - 2, 0), min(cy + 3, Y), max(cz - 2, 0), min(cz + 3, Z)
        # crops that touch the volume's border keep the zero padding there; interior crop borders are 2 voxels away from the
        # centre, i.e. outside the 5^3 receptive field of the two 3x3x3 layers
        want.append(ref_cpu.con_enc(fsd, allf[:, x0:x1, y0:y1, z0:z1])[0, :, cx - x0, cy - y0, cz - z0])
        pos.append((cx, cy, cz))
    return dict(near_img=fuse["near_img"], near_pts=fuse["near_pts"], pos=pos, want=torch.stack(want))


# key -> (function, args, measured job seconds at 4 threads: used only to order the submissions, longest first)
JOBS = {}
for _s, _g in SWEEP:
    JOBS["sweep_%d_%.2f" % (_s, _g)] = (job_sweep, (_s, _g), 15)
for _w, _c in (("o32", 20), ("o64", 59), ("fuse", 7)):
    JOBS["r50_" + _w] = (job_r50, (_w,), _c)
for _w, _c in (("o32", 55), ("o64", 85), ("o16", 29), ("fuse", 12)):
    JOBS["openocc_e2e_" + _w] = (job_openocc_e2e, (_w,), _c)
for _w, _c in (("o32", 15), ("o64", 59), ("obf", 18), ("of16", 16)):
    JOBS["openocc_trunk_" + _w] = (job_openocc_trunk, (_w,), _c)
JOBS["r101_render"] = (job_r101_render, (), 8)
JOBS["stress200"] = (job_stress200, (), 35)

# which jobs a test asks for (matched as a substring of the node id)
WANTS = {
    "test_hot_path_fp64_anchored_seed_sweep": [k for k in JOBS if k.startswith("sweep_")],
    "test_full_size_r50_hot_path": ["r50_o32", "r50_o64", "r50_fuse"],
    "test_openocc_end_to_end_vs_subsampled_oracle[f32]": ["openocc_e2e_o32", "openocc_e2e_o64", "openocc_e2e_fuse"],
    "test_openocc_end_to_end_vs_subsampled_oracle[f16]": ["openocc_e2e_o32", "openocc_e2e_o64", "openocc_e2e_fuse", "openocc_e2e_o16"],
    "test_openocc_decoder_fp32_and_bf16_at_full_size": ["openocc_trunk_o32", "openocc_trunk_o64", "openocc_trunk_obf", "openocc_trunk_of16"],
    "test_full_size_r101_render_pair_vs_oracle": ["r101_render"],
    "test_stress200_r101_end_to_end_properties": ["stress200"],
}


# ------------------------------------------------------------------------------------------------ the pool
_POOL = None
_FUT = {}
_DONE = {}
_T0 = {}


def _init_worker(threads):
    torch.set_num_threads(threads)
    try:
        os.nice(5)              # the GPU tests' host thread goes first
    except OSError:
        pass


def _run(key):
    t = time.time()
    f, a, _ = JOBS[key]
    with torch.no_grad():
        out = f(*a)
    return out, time.time() - t


def _usable_cores():
    """Host cores this process may actually use: affinity mask AND the cgroup CPU quota (the GPU box shows 256 logical CPUs but caps
    the container at 16: a pool sized for 256 is throttled as a whole, the GPU tests' host thread with it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def start(keys):
    """Submit ``keys`` (longest first) to a spawn-context process pool.  Called once by conftest.py."""
    global _POOL
    keys = [k for k in dict.fromkeys(keys) if k not in _FUT and k not in _DONE]
    if not keys or os.environ.get("COOCC_ORACLE_POOL", "1") == "0":
        return
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    ncpu = _usable_cores()
    threads = 4 if ncpu >= 8 else 2
    workers = int(os.environ.get("COOCC_ORACLE_WORKERS", max(1, min(len(keys), (ncpu - 2) // threads))))
    if _POOL is None:
        _POOL = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"), initializer=_init_worker, initargs=(threads,))
    for k in sorted(keys, key=lambda k: -JOBS[k][2]):
        _FUT[k] = _POOL.submit(_run, k)
        _T0[k] = time.time()


def get(key):
    """The result of job ``key``: from the pool when it was submitted, else evaluated here."""
    if key in _DONE:
        return _DONE[key]
    if key in _FUT:
        t = time.time()
        out, cpu_s = _FUT.pop(key).result()
        print("[oracle pool] %-24s job %.1f s, waited %.1f s" % (key, cpu_s, time.time() - t), flush=True)
    else:
        out, cpu_s = _run(key)
    _DONE[key] = out
    return out


def release(*keys):
    """Drop cached results (hundreds of MB for the rendered maps) once their last user has run."""
    for k in keys:
        _DONE.pop(k, None)


def shutdown():
    global _POOL
    if _POOL is not None:
        for f in _FUT.values():
            f.cancel()
        _POOL.shutdown(wait=False, cancel_futures=True)
        _POOL = None
    _FUT.clear()
