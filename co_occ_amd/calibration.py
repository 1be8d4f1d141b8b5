"""One-call accuracy check of the split-f16 engine's presets on a user's own checkpoint and frames.

The default engine evaluates the fp32 convolutions of the decoder with operands split into two f16 halves (csrc/gemm_h2.hip) and
``con_enc`` under Winograd F(4x4) tiles.  On synthetic weights that holds the reference's 1e-4 bound with a margin that is one seed
wide (full r50 scene: rgb maps 7.4e-5 from the CPU oracle; 2.9e-5 with F(2x2) tiles, 2.5e-5 on the exact-fp32 MFMA kernels --
DESIGN.md section 4).  A trained checkpoint has its own activation ranges, so instead of promising a margin the package measures
it: ``calibrate(model, frames)`` runs every frame

* on the exact-fp32 MFMA kernels (``COOCC_CONV_ENGINE=f32``: ``v_mfma_f32_32x32x2_f32`` everywhere, F(2x2) con_enc) -- the anchor:
  the arithmetic closest to the reference's fp32 PyTorch path that exists on the device,
* on the split-f16 engine with each ``con_enc`` tile preset ((4, 4) = default / fastest, (2, 2) = accuracy),

and reports, per preset, the distance of every output of ``simple_test`` from the anchor (max abs and scale-relative), the
fraction of voxels whose arg-max label agrees, and the dense-stage time.  ``recommended`` is the fastest preset whose rgb / depth
maps and logits stay within ``tol`` (default 1e-4, north_star's bound) of the anchor on every frame -- or ``"f32"`` if none does.
Apply it with ``model.occ_fuser.set_conenc_tiles(preset)`` (or ``COOCC_CONENC_TILES=2,2`` / ``COOCC_CONV_ENGINE=f32``).

The reference has no counterpart (its arithmetic is whatever cuDNN picks); this is the documented contract next to the operand
range guard ("Operand range of the split-f16 engine").
"""
import time

import torch

from . import core, fuser as fuser_mod

KEYS = ("pred_c", "pred_f", "rgbs", "depths", "voxel_feats")


def _run(model, frames, reps):
    outs, ms = [], []
    for kw in frames:
        torch.cuda.synchronize()
        model.simple_test(**kw)                                   # packs / workspaces of this configuration
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            o = model.simple_test(**kw)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0) / reps)
        core.check_h2_overflow()
        d = {}
        for k in KEYS:
            v = o.get(k)
            if v is not None:
                d[k] = (v.as_ncdhw() if isinstance(v, core.Rows) else v).detach().float().clone()
        outs.append(d)
    return outs, sum(ms) / len(ms)


def _distance(a, b):
    """max |a - b| and the same relative to max(1, max |b|) (the scale-relative form tests/util.py uses)."""
    e = float((a - b).abs().max())
    return e, e / max(1.0, float(b.abs().max()))


def calibrate(model, frames, presets=((4, 4), (2, 2)), tol=1e-4, reps=3, fine_tol=None):
    """``model``: a ``COOCC_Ray`` on the GPU (put into eval mode); ``frames``: a list of ``simple_test`` keyword dicts (e.g.
    ``dict(img_metas=..., img=img_inputs, points=[pts])`` or ``dict(precomputed=...)``).  Returns a dict:
    ``presets[(m0, m3)] = dict(rgbs=(abs, rel), depths=..., pred_c=..., pred_f=..., voxel_feats=..., label_agreement=f, ms=t,
    within_tol=bool)``, ``anchor_ms``, ``recommended``.  The model's tile preset and engine are restored afterwards.
    ``fine_tol`` (default 10 x tol): the bound on the FINE logits.  They pass two per-row GroupNorms over 4-channel groups, which
    amplify an upstream rounding difference by up to 1e3 (with the synthetic weights the fp32 CPU oracle
    itself sits 6e-5 .. 8e-4 from its fp64 evaluation depending on the seed), so between two fp32-accurate engines they differ by
    ~2e-4 of their scale whatever the preset -- measured with the three-kernel fp32 fine branch as with csrc/fine2_h2.hip; the
    parity tests judge them against an fp64 anchor for the same reason (tests/test_gpu_parity_full.py).  The arg-max agreement is
    reported next to the distance."""
    fine_tol = 10.0 * tol if fine_tol is None else fine_tol
    if not frames:
        raise ValueError("calibrate needs at least one frame")
    f = model.occ_fuser
    if f is None:
        raise ValueError("calibrate: the model has no occ_fuser (the presets are con_enc's Winograd tiles)")
    was_training = model.training
    model.eval()
    keep = (core.CONV_ENGINE, fuser_mod.SPLIT_C0, getattr(f, "conenc_tiles", None), model.graph_simple_test)
    report = dict(presets={}, tol=tol, fine_tol=fine_tol, frames=len(frames))
    try:
        with torch.no_grad():
            model.graph_simple_test = False                        # eager launches: every configuration packs its own weights
            core.CONV_ENGINE, fuser_mod.SPLIT_C0 = "f32", True
            f.set_conenc_tiles((2, 2))
            core.invalidate_packs(model)
            anchor, report["anchor_ms"] = _run(model, frames, reps)
            core.CONV_ENGINE, fuser_mod.SPLIT_C0 = "h2", keep[1] if keep[0] == "h2" else False
            for pre in presets:
                pre = (int(pre[0]), int(pre[1]))
                f.set_conenc_tiles(pre)
                core.invalidate_packs(model)
                outs, ms = _run(model, frames, reps)
                r = dict(ms=round(ms, 3))
                agree, nvox = 0, 0
                for a, o in zip(anchor, outs):
                    for k in KEYS:
                        if k in a and k in o:
                            e = _distance(o[k], a[k])
                            r[k] = tuple(max(x, y) for x, y in zip(r.get(k, (0.0, 0.0)), e))
                    lab = "pred_f" if "pred_f" in a else "pred_c"
                    agree += int((o[lab].argmax(1) == a[lab].argmax(1)).sum())
                    nvox += a[lab][:, 0].numel()
                r["label_agreement"] = agree / max(1, nvox)
                # north_star's bound: rendered colour / depth and occupancy logits within 1e-4 (relative to the tensor's scale
                # where that exceeds 1, as everywhere in tests/)
                r["within_tol"] = (all(r[k][1] <= tol for k in ("rgbs", "depths", "pred_c") if k in r) and
                                   ("pred_f" not in r or r["pred_f"][1] <= fine_tol))
                report["presets"][pre] = r
    finally:
        core.CONV_ENGINE, fuser_mod.SPLIT_C0 = keep[0], keep[1]
        f.set_conenc_tiles(keep[2])
        core.invalidate_packs(model)
        model.graph_simple_test = keep[3]
        if was_training:
            model.train()
    ok = [(v["ms"], k) for k, v in report["presets"].items() if v["within_tol"]]
    report["recommended"] = min(ok)[1] if ok else "f32"
    return report


def format_report(rep):
    lines = ["calibrate: %d frame(s), anchor = exact-fp32 MFMA kernels (%.2f ms / simple_test), tolerance %.1e" % (
        rep["frames"], rep["anchor_ms"], rep["tol"])]
    lines[0] += " (fine logits: %.1e)" % rep.get("fine_tol", 10 * rep["tol"])
    for pre, r in rep["presets"].items():
        dist = "  ".join("%s %.2e (rel %.2e)" % (k, r[k][0], r[k][1]) for k in KEYS if k in r)
        lines.append("  con_enc tiles %s: %.2f ms  labels agree %.6f  %s  -> %s" % (
            pre, r["ms"], r["label_agreement"], dist, "within tol" if r["within_tol"] else "OVER tol"))
    lines.append("  recommended: %s" % (rep["recommended"],))
    return "\n".join(lines)
