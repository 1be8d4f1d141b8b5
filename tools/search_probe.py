"""The prefetch stage alone (fused lift-splat pooling + index search into a slot), nothing else on the GPU: wall time per sample;
under rocprofv3 --kernel-trace --stats the per-kernel table of the stage."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from co_occ_amd import graph as cg
dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
s = bench.make_inputs("r50", 1234, dev, model)
X, Y, Z = s["pts"].shape[2:]
slot = cg.make_slot(model, (X, Y, Z), dev)
st = torch.cuda.Stream(device=dev)
with torch.no_grad(), torch.cuda.stream(st):
    for _ in range(3):
        cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
    torch.cuda.synchronize()
    print("pooling + search alone: %.3f ms per sample" % (1e3 * (time.perf_counter() - t0) / n))
