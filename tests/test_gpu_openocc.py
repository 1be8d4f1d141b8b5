"""configs[4]: the OpenOccupancy grid (coocc_multi_r101_openoccupancy.py: fused grid 128x128x10, occupancy 512x512x40,
cascade 4) and the reduced-precision convolution path (bf16 MFMA operands, fp32 accumulate / epilogue / storage).

* the bf16 kernel against torch's conv on bf16-ROUNDED operands (fp32 arithmetic): same rounding rule, so only the
  accumulation order differs -> tight tolerance; this is what pins the kernel itself;
* the whole C0-C3 decoder at the full 128x128x10 size in fp32 (default dispatch) against the oracle, and in bf16 against the
  oracle run in the same operand precision (ref_cpu.CONV_OPERAND_DTYPE) -- with the distance of all of them from an fp64
  evaluation reported, which is where the bf16 tolerance comes from."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import co_occ_amd as pkg
import co_occ_amd.synth as synth
from co_occ_amd import core
import oracle_jobs as J
from oracle import ref_cpu
from test_gpu_conv import CASES, bn_like, rows_of
from util import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("Cin,Cout,grid,k,stride,relu,use_res,splitk", CASES + [(512, 256, (24, 20, 8), 3, 1, True, False, 0),
                                                                             (128, 128, (50, 50, 8), 3, 1, True, True, 0)])
@pytest.mark.parametrize("preconvert", [True, False])
def test_conv_bf16_mfma_equals_conv_on_bf16_rounded_operands(dev, monkeypatch, Cin, Cout, grid, k, stride, relu, use_res, splitk,
                                                             preconvert):
    """k_conv_bf16w (operands rounded to bf16 once in memory, staged by global_load_lds; layers with Cin % 64 == 0) and
    k_conv_bf16 (operands rounded to bf16 in LDS): RNE rounding, fp32 accumulation, fp32 BN / residual / ReLU epilogue."""
    monkeypatch.setattr(core, "BF16_PRECONVERT", preconvert)
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + 1)
    X, Y, Z = grid
    B = 2 if X < 20 else 1
    x = torch.randn(B, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
    bn = bn_like(Cout, g)
    pad = k // 2
    ref = bn(F.conv3d(_bf(x), _bf(w), stride=stride, padding=pad))
    exact = bn(F.conv3d(x, w, stride=stride, padding=pad))
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref, exact = ref + res, exact + res
    if relu:
        ref, exact = F.relu(ref), F.relu(exact)
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=k, stride=stride, pad=pad)
    monkeypatch.setattr(core, "CONV_DTYPE", "bf16")
    out = core.conv_rows(rows_of(x, dev), pc, relu=relu, res=rows_of(res, dev) if use_res else None, splitk=splitk)
    got = out.as_ncdhw().cpu()
    assert_close(got, ref.detach(), tol=2e-5, what="bf16-MFMA conv vs bf16-rounded operands")
    # and it IS a reduced-precision result: ~2^-9 relative per operand
    e = rel_err(got, exact.detach())
    assert 1e-5 < e < 2e-2, e


def _trunk(dev):
    model, sd, cat4 = J.openocc_trunk_scene()
    return model.to(dev).eval(), sd, cat4


def _hip_trunk(model, cat4, dev):
    from co_occ_amd.core import conv_rows, to_rows
    f = model.occ_fuser
    p = f._packed()
    with torch.no_grad():
        x = conv_rows(to_rows(cat4.to(dev)), p["c0"], relu=True)
        vf = conv_rows(x, p["c3"], relu=True)
        sem = model.semantic_neck.forward_rows(model.semantic_encoder.forward_rows(vf))
        ovf, occ = model.pts_bbox_head.forward_coarse_rows(sem)
    return vf.as_ncdhw().cpu(), occ.as_ncdhw().cpu()


def test_openocc_decoder_fp32_and_bf16_at_full_size(dev, monkeypatch):
    """C0-C3 on the OpenOccupancy fused grid (128 x 128 x 10 x 512 -> 17 classes; 3.56 TFLOP of convolutions).
    fp32 (default dispatch: Winograd F(2x2)/F(4x4) + persistent GEMMs at this size): within 1e-4 of the oracle / the fp64 anchor.
    bf16 operands: the fused voxel features (2 layers deep) equal the oracle evaluated with bf16-rounded conv operands to 2e-3
    of the tensor scale; 25 layers deep the bf16 rounding noise (4e-3 per operand and layer) has been amplified by the
    random-weight decoder until two bf16 evaluations of the same graph are as far from each other as from the truth, so the
    logits are judged against the fp64 anchor:  err(HIP bf16, fp64) <= 1.5 x err(oracle bf16, fp64)  (both reported; that
    distance IS the price of the reduced precision -- the kernel itself is pinned layer by layer above)."""
    import os
    model, sd, cat4 = _trunk(dev)
    h32 = _hip_trunk(model, cat4, dev)
    monkeypatch.setattr(core, "CONV_DTYPE", "bf16")
    hbf = _hip_trunk(model, cat4, dev)
    # the fp16 the config names (one-term f16 MFMA, 16-bit activations written by the producing layer): same judgement against
    # the oracle evaluated with f16-rounded conv operands and the fp64 anchor
    monkeypatch.setattr(core, "CONV_DTYPE", "f16")
    hf16 = _hip_trunk(model, cat4, dev)
    # the four oracle evaluations (fp32, fp64, bf16- and f16-rounded conv operands) come from the oracle process pool
    o32, o64, obf, of16 = (J.get("openocc_trunk_" + k) for k in ("o32", "o64", "obf", "of16"))
    J.release(*("openocc_trunk_" + k for k in ("o32", "o64", "obf", "of16")))
    lines = []
    for name, i in (("voxel_feats", 0), ("coarse logits", 1)):
        scale = max(1.0, float(o64[i].abs().max()))
        e = lambda a, b: float((a.double() - b.double()).abs().max()) / scale
        r = lambda a, b: float(((a.double() - b.double()) ** 2).mean().sqrt()) / scale
        lines.append("openocc 128x128x10 %-13s |x| %.1f  hip32-fp64 %.2e  oracle32-fp64 %.2e  hip32-oracle32 %.2e | hipbf16-oraclebf16 %.2e  "
                     "hipbf16-fp64 max %.2e rms %.2e  oraclebf16-fp64 max %.2e rms %.2e" % (
                         name, scale, e(h32[i], o64[i]), e(o32[i], o64[i]), e(h32[i], o32[i]), e(hbf[i], obf[i]), e(hbf[i], o64[i]),
                         r(hbf[i], o64[i]), e(obf[i], o64[i]), r(obf[i], o64[i])))
        print(lines[-1], flush=True)
        assert e(h32[i], o64[i]) <= 1e-4 and e(h32[i], o32[i]) <= 1e-4 + e(o32[i], o64[i]), lines[-1]
        if i == 0:
            assert e(hbf[i], obf[i]) <= 2e-3, lines[-1]
        assert e(hbf[i], o64[i]) <= 1.5 * e(obf[i], o64[i]) + 1e-4 and r(hbf[i], o64[i]) <= 1.5 * r(obf[i], o64[i]) + 1e-5, lines[-1]
        lines.append("openocc 128x128x10 %-13s f16: hipf16-oraclef16 %.2e  hipf16-fp64 max %.2e rms %.2e  oraclef16-fp64 max %.2e rms %.2e" % (
            name, e(hf16[i], of16[i]), e(hf16[i], o64[i]), r(hf16[i], o64[i]), e(of16[i], o64[i]), r(of16[i], o64[i])))
        print(lines[-1], flush=True)
        if i == 0:
            assert e(hf16[i], of16[i]) <= 5e-4, lines[-1]
        assert e(hf16[i], o64[i]) <= 1.5 * e(of16[i], o64[i]) + 1e-4 and r(hf16[i], o64[i]) <= 1.5 * r(of16[i], o64[i]) + 1e-5, lines[-1]
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "r6_openocc_parity.txt"), "a") as f:
        f.write("\n".join(lines) + "\n")


_E2E = {}        # oracle evaluations of the end-to-end scene, shared by the parametrised runs below


def _e2e_scene():
    """The scene and its oracle evaluations (tests/oracle_jobs.py:job_openocc_e2e, run in the oracle process pool): fp32 with the
    render block in the per-voxel-table form over all rays + the literal form on the strided ray subset, fp64, neighbour tables."""
    if "scene" not in _E2E:
        S = J.openocc_e2e_scene()
        S.update(o32=J.get("openocc_e2e_o32"), o64=J.get("openocc_e2e_o64"), fuse=J.get("openocc_e2e_fuse"))
        _E2E["scene"] = S
    return _E2E["scene"]


def _fine_rows_at(out, coord, dev, dims):
    """HIP fine logits looked up at the oracle's (subsampled) coordinates: (rows [n,ncls] fp64, hit mask, HIP list length)."""
    Xf, Yf, Zf = dims
    hx = out["output_coords_fine"][0]
    hkey = (hx[0] * Yf + hx[1]) * Zf + hx[2]
    order = torch.argsort(hkey)
    hkey_sorted = hkey[order]
    okey = ((coord[0] * Yf + coord[1]) * Zf + coord[2]).to(dev)
    pos = torch.searchsorted(hkey_sorted, okey).clamp(max=hkey_sorted.numel() - 1)
    hit = hkey_sorted[pos] == okey
    rows = order[pos[hit]]
    return out["output_voxels_fine"][0][rows].cpu().double(), hit.cpu(), int(hx.shape[1])


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_openocc_end_to_end_vs_subsampled_oracle(dev, monkeypatch, dtype):
    """configs[4] at full size END TO END (coocc_multi_r101_openoccupancy.py: fused grid 128x128x10, cascade 4 -> 512x512x40,
    6 cameras 56x100, render 6x896x1600): ``forward_hot_path(render=True)`` against the oracle -- neighbour tables bit-exact,
    fused voxel features / coarse logits / rendered maps in full, and the cascade-4 fine branch (~1e7 points with random
    weights: hours on the CPU) on a seeded 60 k-point subsample of the oracle's fine list (every fine point is an independent
    row of the branch; the HIP rows are looked up by coordinate).

    ``f32``: the default dispatch (fp32-accurate split-f16 engine), judged against the fp32 oracle and the fp64 anchor as in
    tests/test_gpu_parity_full.py.
    ``f16`` (``bench.py --config openocc --dtype f16``, the mode configs[4] names): the C0-C3 convolutions run one-term f16
    MFMAs on f16 operands; everything else is unchanged.  Judged against the oracle evaluated with f16-ROUNDED convolution
    operands (``ref_cpu.CONV_OPERAND_DTYPE``) where the two must agree closely (the fused features, two layers deep), and by
    the anchor rule 25 layers deep:  err(HIP f16, fp64) <= 1.5 x err(oracle f16, fp64)  -- the distance of BOTH from the exact
    answer is the price of the reduced precision, printed in the table."""
    import os
    S = _e2e_scene()
    c, sd = S["c"], S["sd"]
    monkeypatch.setattr(core, "CONV_DTYPE", dtype)
    model = pkg.build_detector(synth.model_cfg_openocc())
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    with torch.no_grad():
        out = model.forward_hot_path(S["img"].to(dev), S["pts"].to(dev), S["gemo"].to(dev), [S["img_feats"][0].to(dev)],
                                     tuple(t.to(dev) if torch.is_tensor(t) else t for t in S["tr"]), render=True)
    torch.cuda.synchronize()
    core.check_h2_overflow()
    near_img, near_pts = model.occ_fuser.last_near
    fuse = S["fuse"]
    assert np.array_equal(near_img.cpu().numpy().reshape(fuse["near_img"].shape), fuse["near_img"].numpy())      # bit-exact
    assert np.array_equal(near_pts.cpu().numpy().reshape(fuse["near_pts"].shape), fuse["near_pts"].numpy())
    o32, o64 = S["o32"], S["o64"]
    lines = []
    if dtype == "f16":
        o16 = J.get("openocc_e2e_o16")          # conv operands rounded to f16 (ref_cpu.CONV_OPERAND_DTYPE), table-form render
    for k_hip, k_ref in (("voxel_feats", "voxel_feats"), ("pred_c", "output_voxels")):
        h, r32, r64 = out[k_hip].detach().cpu().double(), o32[k_ref].double(), o64[k_ref]
        scale = max(1.0, float(r64.abs().max()))
        e_h64, e_r64, e_h32 = float((h - r64).abs().max()), float((r32 - r64).abs().max()), float((h - r32).abs().max())
        rms = lambda a, b: float(((a - b) ** 2).mean().sqrt())
        if dtype == "f32":
            lines.append("openocc e2e f32 %-11s |x| %.1f  abs: hip-fp64 %.2e ref32-fp64 %.2e hip-ref32 %.2e  (of scale: %.2e / %.2e / %.2e)" % (
                k_hip, scale, e_h64, e_r64, e_h32, e_h64 / scale, e_r64 / scale, e_h32 / scale))
            assert e_h64 <= 1e-4 * scale and e_h32 <= 1e-4 * scale + e_r64, lines[-1]
        else:
            r16 = o16[k_ref].double()
            e_h16, e_1664 = float((h - r16).abs().max()), float((r16 - r64).abs().max())
            lines.append("openocc e2e f16 %-11s |x| %.1f  abs: hipf16-oraclef16 %.2e  hipf16-fp64 max %.2e rms %.2e  oraclef16-fp64 max %.2e rms %.2e" % (
                k_hip, scale, e_h16, e_h64, rms(h, r64), e_1664, rms(r16, r64)))
            if k_hip == "voxel_feats":
                assert e_h16 <= 5e-4 * scale, lines[-1]            # two layers deep: same operand rounding, different accumulation order
            assert e_h64 <= 1.5 * e_1664 + 1e-4 * scale and rms(h, r64) <= 1.5 * rms(r16, r64) + 1e-5 * scale, lines[-1]
    # fine branch: HIP rows at the oracle's subsampled coordinates
    ref = o32 if dtype == "f32" else o16
    fh, keep, n_h = _fine_rows_at(out, ref["fine_coord"], dev, c["final_occ_size"])
    n_o = int(ref["fine_coord_all"].shape[1])
    lim = 0.002 if dtype == "f32" else 0.02                      # f16 operands move more argmax near-ties of the coarse head
    assert abs(n_o - n_h) <= lim * n_o + 64 and float(keep.float().mean()) >= 1 - lim, (n_o, n_h, float(keep.float().mean()))
    # the fp64 anchor's subset is drawn from ITS list: identical whenever the coarse argmax masks agree
    same64 = o64["fine_coord"].shape == ref["fine_coord"].shape and torch.equal(o64["fine_coord"], ref["fine_coord"])
    if dtype == "f32":
        assert same64, "fp32 / fp64 oracle foreground sets differ at this seed: pick another subsample seed"
    if same64:
        fr_, f64 = ref["fine_output"].double()[keep], o64["fine_output"][keep]
        scale = max(1.0, float(f64.abs().max()))
        mh, rh = float((fh - f64).abs().max()), float(((fh - f64) ** 2).mean().sqrt())
        mr, rr = float((fr_ - f64).abs().max()), float(((fr_ - f64) ** 2).mean().sqrt())
        lines.append("openocc e2e %s fine (cascade 4): %d of %d oracle points subsampled, HIP list %d | |x| %.1f hip-fp64 max %.2e rms %.2e ; "
                     "oracle(%s)-fp64 max %.2e rms %.2e ; ratio max %.2f rms %.2f" % (dtype, int(keep.sum()), n_o, n_h, scale, mh, rh, dtype, mr, rr,
                                                                                    mh / max(mr, 1e-30), rh / max(rr, 1e-30)))
        ulp = 1.2e-7 * scale
        assert mh <= 3.0 * mr + 8 * ulp and rh <= 1.5 * rr + ulp, lines[-1]
    else:
        # f16: the f16-operand oracle's foreground list differs from the fp64 one at a few near-ties, so its subsample is another
        # one; the fine rows are then judged against the f16-operand oracle itself (same operands, fp32 arithmetic elsewhere)
        fr_ = ref["fine_output"].double()[keep]
        scale = max(1.0, float(fr_.abs().max()))
        mh, rh = float((fh - fr_).abs().max()), float(((fh - fr_) ** 2).mean().sqrt())
        lines.append("openocc e2e f16 fine (cascade 4): %d of %d points, HIP list %d | |x| %.1f hipf16-oraclef16 max %.2e rms %.2e (of scale %.2e / %.2e)" % (
            int(keep.sum()), n_o, n_h, scale, mh, rh, mh / scale, rh / scale))
        assert rh <= 2e-3 * scale, lines[-1]
    e_rgb = float((out["rgbs"].cpu() - o32["rgbs"]).abs().max())
    e_dep = rel_err(out["depths"].cpu(), o32["depths"])
    if dtype == "f32":
        lines.append("openocc e2e f32 render 6x896x1600: rgbs abs %.2e depths rel %.2e ; oracle literal vs table form on the strided rays %.2e / %.2e" % (
            (e_rgb, e_dep) + tuple(o32["literal_dev"])))
        assert max(o32["literal_dev"]) <= 2e-6, lines[-1]
        assert e_rgb <= 1e-4 and e_dep <= 1e-4, lines[-1]
    else:
        r_rgb = float((o16["rgbs"] - o32["rgbs"]).abs().max())
        r_dep = rel_err(o16["depths"], o32["depths"])
        lines.append("openocc e2e f16 render 6x896x1600: rgbs abs hip-ref32 %.2e (oraclef16-ref32 %.2e)  depths rel %.2e (%.2e)  hip-oraclef16 rgbs %.2e" % (
            e_rgb, r_rgb, e_dep, r_dep, float((out["rgbs"].cpu() - o16["rgbs"]).abs().max())))
        assert e_rgb <= 1.5 * r_rgb + 1e-4 and e_dep <= 1.5 * r_dep + 1e-4, lines[-1]
    assert tuple(out["pred_f"].shape) == (1, 17) + tuple(c["final_occ_size"])
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "r6_openocc_parity.txt"), "a") as f:
        f.write("\n".join(lines) + "\n")
    for l in lines:
        print(l, flush=True)


def _f16(x):
    return x.to(torch.float16).float()


@pytest.mark.parametrize("Cin,Cout,grid,k,stride,relu,use_res", [(128, 128, (24, 20, 10), 3, 1, True, True), (256, 128, (16, 18, 4), 3, 1, True, False),
                                                                  (128, 256, (20, 20, 8), 3, 2, True, False), (512, 512, (13, 12, 2), 3, 1, True, True),
                                                                  (1024, 1024, (7, 7, 1), 3, 1, True, True), (128, 256, (30, 20, 10), 1, 1, True, False),
                                                                  (64, 160, (11, 9, 3), 3, 1, False, False)])
def test_conv_f16_mfma_equals_conv_on_f16_rounded_operands(dev, monkeypatch, Cin, Cout, grid, k, stride, relu, use_res):
    """mfma_dtype 4 (csrc/gemm_h2.hip with TERMS = 1: operands f16 in memory, ONE v_mfma_f32_32x32x16_f16 per step, fp32
    accumulate / BN / residual / ReLU) against torch's conv on f16-ROUNDED operands: same rounding rule, only the accumulation
    order differs.  The producer-side 16-bit copy (out16 -> Rows.h16) is the RNE rounding of the fp32 output, and a second
    layer fed from it equals the layer fed from a fresh conversion."""
    g = torch.Generator().manual_seed(Cin * 1000 + Cout + 7)
    X, Y, Z = grid
    x = torch.randn(1, Cin, X, Y, Z, generator=g)
    w = torch.randn(Cout, Cin, k, k, k, generator=g) * (2.0 / (Cin * k ** 3)) ** 0.5
    bn = bn_like(Cout, g)
    pad = k // 2
    ref = bn(F.conv3d(_f16(x), _f16(w), stride=stride, padding=pad))
    exact = bn(F.conv3d(x, w, stride=stride, padding=pad))
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref, exact = ref + res, exact + res
    if relu:
        ref, exact = F.relu(ref), F.relu(exact)
    pc = core.PackedConv(w.to(dev), bn=bn.to(dev), ksize=k, stride=stride, pad=pad)
    monkeypatch.setattr(core, "CONV_DTYPE", "f16")
    out = core.conv_rows(rows_of(x, dev), pc, relu=relu, res=rows_of(res, dev) if use_res else None)
    got = out.as_ncdhw().cpu()
    assert_close(got, ref.detach(), tol=2e-5, what="f16-MFMA conv vs f16-rounded operands")
    e = rel_err(got, exact.detach())
    assert 1e-6 < e < 5e-3, e                      # it IS a reduced-precision result: ~2^-12 relative per operand
    assert out.h16 is not None and out.h16.dtype == torch.float16
    assert torch.equal(out.h16.cpu(), out.t.cpu().to(torch.float16))
    if Cout % 64 == 0:                             # a second layer reads the producer's 16-bit copy: no conversion pass
        w2 = torch.randn(64, Cout, 1, 1, 1, generator=g) * (2.0 / Cout) ** 0.5
        pc2 = core.PackedConv(w2.to(dev), bn=bn_like(64, g).to(dev), ksize=1, stride=1, pad=0)
        a = core.conv_rows(out, pc2, relu=False)
        out.h16 = None
        b = core.conv_rows(out, pc2, relu=False)
        assert torch.equal(a.t, b.t)
