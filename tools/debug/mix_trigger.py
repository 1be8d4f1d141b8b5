"""Debug (profiles/r6_corunner_defect.txt; step 1 of the round-6 hunt): WHICH part of a split-f16 GEMM makes the half-column OccHead mix
(COOCC_INTERP_COLUMN bit 1) lose its level-0 term?  (Answer: the MFMA stream alone -- COOCC_H2_ABLATE=15 -- and nothing else; with the
kernel compiled COOCC_SCALAR_FP32 every line reads 0 of 20.)  The victim is the real kernel; the co-runner on the second stream is one
PIECE of the split-f16 layer, chosen by name:

    h2p        the pointwise kernel k_gemm_h2p (1x1x1 128 -> 128, 80 000 rows): the known trigger
    chain      k_wino_in_h2 + k_gemm_h2z<3,false> + k_wino_out (one Winograd layer)
    gemm       k_gemm_h2z<3,false> alone, on the V / M buffers of one earlier chain call
               (with COOCC_H2_ABLATE=1 / 2 / 4 / 8 / 16 / 15 in the environment: the kernel's timing ablations -- no LDS-DMA image,
               no weight loads, no fragment ds_reads, no stores, LDS-DMA from the zero page only, MFMAs only)
    wino_in    k_wino_in_h2 alone        wino_out   k_wino_out alone        to_h2   k_rows_to_h2 alone
    f32        the same 3x3x3 layer on the fp32-MFMA engine (COOCC_CONV_ENGINE=f32 semantics: core.CONV_ENGINE)
    none       no co-runner (control)

    python tools/debug/mix_trigger.py <piece> [n_calls]
prints one line:  piece, ablation, calls that differ from the per-voxel kernel run alone."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from co_occ_amd import core, _lib
from co_occ_amd._lib import call, ptr, host_i32

piece = sys.argv[1] if len(sys.argv) > 1 else "h2p"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
C, sizes = 128, [(100, 100, 8), (50, 50, 4), (25, 25, 2), (13, 13, 1)]
g = torch.Generator().manual_seed(7)
levels = [torch.randn(s[0] * s[1] * s[2], C, generator=g).to(dev) for s in sizes]
wlogit = (torch.randn(sizes[0][0] * sizes[0][1] * sizes[0][2], 4, generator=g) * 3).to(dev)
arr = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in levels])
dims = host_i32([v for s in sizes for v in s])
outs = [torch.empty_like(levels[0]) for _ in range(N)]
twins = [torch.empty_like(levels[0]) for _ in range(N)]


def mix(i):
    call("coocc_occhead_mix_ex", arr, dims, 4, ptr(wlogit), ptr(outs[i]), 1, C, ptr(twins[i]))


gb = torch.Generator().manual_seed(11)
xb = core.to_rows(torch.randn(1, 128, 100, 100, 8, generator=gb).to(dev))
pc1 = core.PackedConv((torch.randn(128, 128, 1, 1, 1, generator=gb) * 0.05).to(dev), ksize=1, pad=0)
pc3 = core.PackedConv((torch.randn(128, 128, 3, 3, 3, generator=gb) * 0.02).to(dev), ksize=3, pad=1)
s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
scratch = torch.empty(xb.B * xb.V, 128, device=dev)

# record the native calls of one Winograd layer issued on s1 (so that its V / M buffers belong to that stream)
calls = []
real_call, real_fwd = _lib.call, _lib.conv_fwd
if piece in ("gemm", "wino_in", "wino_out"):
    def rec_call(name, *a):
        calls.append((name, a))
        return real_call(name, *a)

    def rec_fwd(d, device):
        calls.append(("conv_fwd", (d, device)))
        return real_fwd(d, device)
    core.call, _lib.conv_fwd = rec_call, rec_fwd
with torch.no_grad(), torch.cuda.stream(s1):
    y1 = core.conv_rows(xb, pc1, relu=False)
    if piece == "f32":
        core.CONV_ENGINE = "f32"
    y3 = core.conv_rows(xb, pc3, relu=False)
torch.cuda.synchronize()
core.call, _lib.conv_fwd = real_call, real_fwd
names = [c[0] for c in calls]


def replay(prefix):
    for name, a in calls:
        if name.startswith(prefix):
            if name == "conv_fwd":
                real_fwd(*a)
            else:
                real_call(name, *a)


def co():
    for _ in range(4):
        if piece == "h2p":
            core.conv_rows(xb, pc1, relu=False)
        elif piece in ("chain", "f32"):
            core.conv_rows(xb, pc3, relu=False)
        elif piece == "gemm":
            replay("conv_fwd")
        elif piece == "wino_in":
            replay("coocc_wino_input")
        elif piece == "wino_out":
            replay("coocc_wino_output")
        elif piece == "to_h2":
            for _ in range(4):
                call("coocc_rows_to_h2", xb.data(), xb.stride, xb.B * xb.V, 128, 1.0, ptr(scratch))


os.environ["COOCC_INTERP_COLUMN"] = "0"
with torch.cuda.stream(s0):
    mix(0)
torch.cuda.synchronize()
ref, ref_tw = outs[0].clone(), twins[0].clone()
os.environ["COOCC_INTERP_COLUMN"] = "2"
for o, t in zip(outs, twins):
    o.fill_(float("nan")); t.fill_(float("nan"))
torch.cuda.synchronize()
with torch.no_grad():
    for i in range(N):
        if piece != "none":
            with torch.cuda.stream(s1):
                co()
        with torch.cuda.stream(s0):
            mix(i)
torch.cuda.synchronize()
bad = rows = 0
comps, lanes = {}, {}
for o, t in zip(outs, twins):
    d = o.view(torch.int32) != ref.view(torch.int32)
    if d.any() or (t.view(torch.int32) != ref_tw.view(torch.int32)).any():
        bad += 1
        rows += int(d.any(1).sum())
        for cch in torch.nonzero(d.any(0))[:, 0].tolist():
            comps[cch & 3] = comps.get(cch & 3, 0) + 1
            lanes[(cch >> 2) // 16 * 16] = lanes.get((cch >> 2) // 16 * 16, 0) + 1   # 32 threads x 4 channels per row: lane within the half-wave
print("co-runner %-9s ablate %-3s recorded %s: %2d of %d calls differ, %d rows; channel %% 4 of the differing columns %s" % (
    piece, os.environ.get("COOCC_H2_ABLATE", "-"), sorted(set(names)), bad, N, rows, sorted(comps.items())), flush=True)
