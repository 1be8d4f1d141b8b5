"""Isolated timing of the dense stage: eager (host-count) vs eager static (device-count) vs hipGraph replay, nothing else on the GPU."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import graph as cg

dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "r50"
bench.CFGNAME[0] = cfg
model, _ = bench.build_model(cfg, dev)
s = bench.make_inputs(cfg, 1234, dev, model)
X, Y, Z = s["pts"].shape[2:]
slot = cg.make_slot(model, (X, Y, Z), dev)
stream = torch.cuda.Stream(device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream())
    for _ in range(n):
        fn()
    b.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n, 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad(), torch.cuda.stream(stream):
    sr = cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
    torch.cuda.synchronize()
    print("counts", sr.counts)
    g = cg.DenseGraph(model, slot, s, stream).capture()
    img = slot.img_rows().as_ncdhw()

    def eager():
        vf = model.occ_fuser(img, s["pts"], search=sr)
        model.decode(vf, s["gemo"], s["img_feats"], s["transform"], True)
    print("eager (host counts)      gpu %.3f ms  wall %.3f ms" % timeit(eager))
    print("eager static (dev counts) gpu %.3f ms  wall %.3f ms" % timeit(g._run))
    print("graph replay              gpu %.3f ms  wall %.3f ms" % timeit(g.replay))
