"""Every stage of the shipped dense path, run REPEATEDLY on fixed inputs next to a split-f16 GEMM loop on a second stream, gives the
bits it gives alone (VERDICT r5 "Next round" 5b).

Why: two kernels of this package (the half-z-column OccHead mix, ``COOCC_INTERP_COLUMN`` bit 1; the one-launch ratio-2 fine branch,
``COOCC_FINE2_IMG_INSIDE=1``) are bit-exact alone and wrong only while a ``k_gemm_h2*`` kernel of ANOTHER stream shares the chip
(``profiles/r6_corunner_defect.txt``: one 64-byte return beat of a 16-byte-per-lane load -- one dword component, lanes 48-63 -- reads
as zero).  Both are parked (off); the whole-graph soaks only replay complete graphs.  This file is the per-stage guard for what IS
shipped: no allocator reuse between the compared calls' inputs, no graph, no pipeline -- a stage, a co-runner, ``N`` calls, bits.
The two parked forms run under the same harness as ``xfail`` (non-strict: a box where they happen to pass is not an error)."""
import os

import pytest
import torch

import co_occ_amd as pkg
import co_occ_amd.synth as synth
from co_occ_amd import core
from co_occ_amd.core import to_rows

pytestmark = pytest.mark.gpu
N_CALLS = 20
_S = {}


def _scene(dev):
    """configs[1] scene + one eager pass alone (the inputs of every stage are the reference outputs of the stage before it)."""
    if _S:
        return _S
    import bench
    bench.CFGNAME[0] = "r50"
    model, _ = bench.build_model("r50", dev)
    model.test_rendering = True
    s = bench.make_inputs("r50", 4242, dev, model)
    neck, head, enc = model.semantic_neck, model.pts_bbox_head, model.semantic_encoder
    with torch.no_grad():
        vol = model.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"]).contiguous()
        vf = model.fuse(vol, s["pts"])
        mid = enc.forward_rows(vf, readers=neck.lateral_packs())
        sem = neck.forward_rows(mid, readers=head.level_readers())
        gemo = s["gemo"] if s.get("gemo") is not None else model.img_view_transformer.get_geometry(*s["cams"])
    torch.cuda.synchronize()
    # the co-runner: a 1x1x1 128 -> 128 layer over 80 000 rows on the split-f16 engine (k_gemm_h2p; its H2 input is made once) --
    # next to it the parked mix form differed in 20 of 20 calls
    gb = torch.Generator().manual_seed(11)
    xb = to_rows(torch.randn(1, 128, 100, 100, 8, generator=gb).to(dev))
    pc1 = core.PackedConv((torch.randn(128, 128, 1, 1, 1, generator=gb) * 0.05).to(dev), ksize=1, pad=0)
    pc3 = core.PackedConv((torch.randn(128, 128, 3, 3, 3, generator=gb) * 0.02).to(dev), ksize=3, pad=1)
    with torch.no_grad():
        core.conv_rows(xb, pc1, relu=False)
        core.conv_rows(xb, pc3, relu=False)
    torch.cuda.synchronize()
    _S.update(model=model, s=s, vol=vol, vf=vf, mid=mid, sem=sem, gemo=gemo, xb=xb, pc1=pc1, pc3=pc3,
              s0=torch.cuda.Stream(device=dev), s1=torch.cuda.Stream(device=dev))
    return _S


def _flat(out):
    """Every tensor a stage produced, as a flat list (Rows -> rows + twins)."""
    if torch.is_tensor(out):
        return [out]
    if isinstance(out, core.Rows):
        return [t for t in (out.t, out.h2, out.h16) if t is not None]
    if isinstance(out, dict):
        return [t for k in sorted(out) for t in _flat(out[k])]
    if isinstance(out, (list, tuple)):
        return [t for v in out for t in _flat(v)]
    return []


def _stages(S):
    m, s = S["model"], S["s"]
    neck, head, enc = m.semantic_neck, m.pts_bbox_head, m.semantic_encoder
    return {
        "lift_splat (P2 pooling)": lambda: m.img_view_transformer.lift_splat(s["depth"], s["ctx"], cams=s["cams"]),
        "fuser (K1-K5 search, G1 gather GEMMs, con_enc)": lambda: m.fuse(S["vol"], s["pts"]),
        "encoder (CustomResNet3D)": lambda: enc.forward_rows(S["vf"], readers=neck.lateral_packs()),
        "neck (FPN3D, z-column upsample-add)": lambda: neck.forward_rows(S["mid"], readers=head.level_readers()),
        "head (coarse mix + fine branch)": lambda: head(voxel_feats=S["sem"], img_feats=s["img_feats"], transform=s["transform"]),
        "decode (head + pred_f scatter + render heads / rays / upsample)": lambda: {
            k: v for k, v in m.decode(S["vf"], S["gemo"], s["img_feats"], s["transform"], render=True).items() if k != "voxel_feats"},
    }


def _run_beside(S, fn, n=N_CALLS, corunner="h2p"):
    """``fn`` alone, then n times on s0 while s1 loops the co-runner; returns (reference tensors, list of n result lists)."""
    s0, s1 = S["s0"], S["s1"]
    with torch.no_grad():
        with torch.cuda.stream(s0):
            ref = [t.clone() for t in _flat(fn())]
        torch.cuda.synchronize()
        got = []
        for _ in range(n):
            with torch.cuda.stream(s1):
                for _ in range(4):
                    core.conv_rows(S["xb"], S["pc1"] if corunner == "h2p" else S["pc3"], relu=False)
            with torch.cuda.stream(s0):
                got.append([t.clone() for t in _flat(fn())])
        torch.cuda.synchronize()
    core.check_h2_overflow()
    return ref, got


def _bits_equal(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.is_floating_point():                       # bit patterns, so that NaN == NaN and -0 != +0
        it = {4: torch.int32, 2: torch.int16, 8: torch.int64}[a.element_size()]
        return torch.equal(a.contiguous().view(it), b.contiguous().view(it))
    return torch.equal(a, b)


def _count_differing(ref, got):
    bad = 0
    for g in got:
        assert len(g) == len(ref)
        bad += int(not all(_bits_equal(a, b) for a, b in zip(ref, g)))
    return bad


STAGES = ["lift_splat (P2 pooling)", "fuser (K1-K5 search, G1 gather GEMMs, con_enc)", "encoder (CustomResNet3D)",
          "neck (FPN3D, z-column upsample-add)", "head (coarse mix + fine branch)",
          "decode (head + pred_f scatter + render heads / rays / upsample)"]


@pytest.mark.parametrize("corunner", ["h2p", "wino"])
@pytest.mark.parametrize("stage", STAGES)
def test_shipped_stage_is_bit_stable_beside_a_split_f16_gemm(dev, stage, corunner):
    """20 calls of one stage of the default dispatch on fixed inputs, a split-f16 layer looping on a second stream (``h2p``: the
    pointwise kernel k_gemm_h2p; ``wino``: k_wino_in_h2 + k_gemm_h2z + k_wino_out): every output tensor (rows, H2 twins, index
    lists) bit-equal to the stage run alone."""
    S = _scene(dev)
    ref, got = _run_beside(S, _stages(S)[stage], corunner=corunner)
    assert len(ref) > 0
    bad = _count_differing(ref, got)
    assert bad == 0, "%s: %d of %d calls differ from the stage run alone (co-runner %s)" % (stage, bad, N_CALLS, corunner)


@pytest.mark.xfail(strict=False, reason="parked: bit-exact alone, one load beat reads as zero next to k_gemm_h2* (profiles/r6_corunner_defect.txt)")
def test_parked_half_column_mix_beside_a_split_f16_gemm(dev, monkeypatch):
    S = _scene(dev)
    monkeypatch.setenv("COOCC_INTERP_COLUMN", "3")
    head = S["model"].pts_bbox_head
    fn = lambda: head.forward_coarse_rows(S["sem"])
    ref, got = _run_beside(S, fn)
    monkeypatch.setenv("COOCC_INTERP_COLUMN", "1")
    with torch.no_grad():
        alone = [t.clone() for t in _flat(fn())]          # the shipped per-voxel kernel alone: what the parked form must equal
    torch.cuda.synchronize()
    assert all(_bits_equal(a, b) for a, b in zip(alone, ref)), "the parked form differs from the shipped kernel even alone"
    assert _count_differing(ref, got) == 0


@pytest.mark.xfail(strict=False, reason="parked: bit-exact alone, differs next to k_gemm_h2* (DESIGN 3.2, profiles/r5_fine2_corunner.txt)")
def test_parked_one_launch_fine_branch_beside_a_split_f16_gemm(dev, monkeypatch):
    from co_occ_amd import head as H
    S = _scene(dev)
    s, head = S["s"], S["model"].pts_bbox_head
    monkeypatch.setattr(H, "FINE2_IMG_INSIDE", True)
    fn = lambda: head(voxel_feats=S["sem"], img_feats=s["img_feats"], transform=s["transform"])
    ref, got = _run_beside(S, fn)
    assert _count_differing(ref, got) == 0
