"""Debug (discriminator i): WHAT do the differing rows of the half-column OccHead mix hold when it runs next to a
split-f16 GEMM co-runner?  The outputs are prefilled with NaN before every call, so

  * NaN in a differing row  -> the store never happened in that lane (EXEC lost: the s[40:41] write-after-write candidate);
  * |x| / max patterns      -> the store read its data registers after the H2-twin code overwrote them;
  * finite, unrelated       -> a level's contribution is missing / mis-weighted.

    python tools/debug/mix_values.py [COOCC_MIX_COL_VAR values ...]        (default: 0 2)
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from co_occ_amd import core
from co_occ_amd._lib import call, ptr, host_i32

dev = torch.device("cuda:0")
C, sizes = 128, [(100, 100, 8), (50, 50, 4), (25, 25, 2), (13, 13, 1)]
g = torch.Generator().manual_seed(7)
levels = [torch.randn(s[0] * s[1] * s[2], C, generator=g).to(dev) for s in sizes]
wlogit = (torch.randn(sizes[0][0] * sizes[0][1] * sizes[0][2], 4, generator=g) * 3).to(dev)
arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in levels])
dims = host_i32([v for s in sizes for v in s])
N = 20
outs = [torch.empty_like(levels[0]) for _ in range(N)]
twins = [torch.empty_like(levels[0]) for _ in range(N)]


def mix(i):
    call("coocc_occhead_mix_ex", arr, dims, 4, ptr(wlogit), ptr(outs[i]), 1, C, ptr(twins[i]))


gb = torch.Generator().manual_seed(11)
xb = core.to_rows(torch.randn(1, 128, 100, 100, 8, generator=gb).to(dev))
pc1 = core.PackedConv((torch.randn(128, 128, 1, 1, 1, generator=gb) * 0.05).to(dev), ksize=1, pad=0)


def co():
    for _ in range(4):
        core.conv_rows(xb, pc1, relu=False)


s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
os.environ["COOCC_INTERP_COLUMN"] = "0"
with torch.cuda.stream(s0):
    mix(0)
torch.cuda.synchronize()
ref, ref_tw = outs[0].clone(), twins[0].clone()          # the per-voxel kernel, alone
# the four level terms w_l * sample_l of the reference, each from the per-voxel kernel with the OTHER levels zeroed (same weights)
terms = []
for l in range(4):
    zs = [lv if k == l else torch.zeros_like(lv) for k, lv in enumerate(levels)]
    za = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in zs])
    tl = torch.empty_like(levels[0])
    call("coocc_occhead_mix_ex", za, dims, 4, ptr(wlogit), ptr(tl), 1, C, None)
    torch.cuda.synchronize()
    terms.append(tl)
os.environ["COOCC_INTERP_COLUMN"] = "2"
which = {}
for var in (sys.argv[1:] or ["0", "2"]):
    os.environ["COOCC_MIX_COL_VAR"] = var
    for o, t in zip(outs, twins):
        o.fill_(float("nan")); t.fill_(float("nan"))
    torch.cuda.synchronize()
    with torch.no_grad():
        for i in range(N):
            with torch.cuda.stream(s1):
                co()
            with torch.cuda.stream(s0):
                mix(i)
    torch.cuda.synchronize()
    bad_calls = 0
    n_nan = n_abs = n_other = n_tw = 0
    shown = 0
    for i, (o, t) in enumerate(zip(outs, twins)):
        d = (o != ref) & ~(torch.isnan(o) & torch.isnan(ref))
        d = d | torch.isnan(o)
        dt = (t.view(torch.int32) != ref_tw.view(torch.int32))
        if not (d.any() or dt.any()):
            continue
        bad_calls += 1
        n_tw += int(dt.any(1).sum())
        rows = torch.nonzero(d.any(1))[:, 0]
        for r in rows.tolist():
            w, rr = o[r], ref[r]
            m = d[r]
            is_nan = torch.isnan(w[m]).all()
            is_abs = (not is_nan) and bool(((w[m] == rr[m].abs()) | (w[m] >= rr[m].abs())).all())
            n_nan += int(is_nan)
            n_abs += int(is_abs)
            n_other += int(not is_nan and not is_abs)
            if not is_nan:
                dd = (w[m] - rr[m]).double()
                for e, cch in zip(dd.tolist(), torch.nonzero(m)[:, 0].tolist()):
                    # which level term explains the difference?  wrong - right = (x - 1) * T_l for a single l  <=>  the other terms cancel
                    best = min(range(4), key=lambda l: abs(abs(e) - abs(float(terms[l][r, cch]))))
                    ratio = e / float(terms[best][r, cch]) if float(terms[best][r, cch]) != 0 else float("nan")
                    key = (best, "minus the whole term" if abs(ratio + 1) < 1e-3 else "other")
                    which[key] = which.get(key, 0) + 1
            if shown < 6:
                shown += 1
                cs = torch.nonzero(m)[:, 0].tolist()
                print("  call %2d row %6d (column voxel %d, column %d) channels %s..%s (%d): wrong %s | right %s | twin row differs: %s" % (
                    i, r, r % 8, r // 8, cs[0], cs[-1], len(cs), [float("%.4g" % v) for v in w[m][:4].tolist()],
                    [float("%.4g" % v) for v in rr[m][:4].tolist()], bool(dt[r].any())))
    print("COOCC_MIX_COL_VAR=%s next to k_gemm_h2p/h2w (1x1x1 128->128): %d of %d calls differ; differing rows: %d all-NaN (store skipped), "
          "%d abs/max-like (data overwritten), %d other; twin rows differing %d" % (var, bad_calls, N, n_nan, n_abs, n_other, n_tw), flush=True)
    print("   differing elements by the level term closest in magnitude to (wrong - right): %s" % sorted(which.items()), flush=True)
    which = {}
