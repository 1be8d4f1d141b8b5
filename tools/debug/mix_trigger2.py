"""Debug (profiles/r6_corunner_defect.txt): the parked half-column OccHead mix next to SYNTHETIC matrix-core co-runners
(tools/proto/mfma_corunner.hip, built to /tmp/libmfma_co.so by the job script): which property of the split-f16 GEMM skeleton is the
trigger -- the MFMA instruction, the wave's VGPR allocation, the workgroup's LDS allocation?

    python tools/debug/mix_trigger2.py <kind 0..5> <vgprs 256|208|128|0> <lds bytes> [blocks] [iters] [n_calls]
"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from co_occ_amd._lib import call, ptr, host_i32

kind, nv, lds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 128
N = int(sys.argv[6]) if len(sys.argv) > 6 else 20
co_lib = ctypes.CDLL(os.environ.get("MFMA_CO_LIB", "/tmp/libmfma_co.so"))
co_lib.mfma_co_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
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


s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
os.environ["COOCC_INTERP_COLUMN"] = "0"
with torch.cuda.stream(s0):
    mix(0)
torch.cuda.synchronize()
ref, ref_tw = outs[0].clone(), twins[0].clone()
os.environ["COOCC_INTERP_COLUMN"] = "2"
for o, t in zip(outs, twins):
    o.fill_(float("nan")); t.fill_(float("nan"))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s1)
for i in range(N):
    for _ in range(4):
        rc = co_lib.mfma_co_launch(kind, nv, blocks, iters, lds, ctypes.c_void_p(s1.cuda_stream))
        assert rc == 0, rc
    with torch.cuda.stream(s0):
        mix(i)
e1.record(s1)
torch.cuda.synchronize()
bad = rows = 0
for o, t in zip(outs, twins):
    d = o.view(torch.int32) != ref.view(torch.int32)
    if d.any() or (t.view(torch.int32) != ref_tw.view(torch.int32)).any():
        bad += 1
        rows += int(d.any(1).sum())
names = ["v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_bf16", "v_fma_f32 only"]
print("synthetic co-runner %-26s vgprs %3d lds %6d blocks %d iters %d (%.0f us per launch): %2d of %d calls differ, %d rows" % (
    names[kind], nv, lds, blocks, iters, e0.elapsed_time(e1) * 1000 / (4 * N), bad, N, rows), flush=True)
