import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from co_occ_amd import graph as cg
dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
pool = [torch.cuda.Stream(device=dev) for _ in range(8)]
def trial(name, sel):
    N = len(sel)
    graphs = []
    with torch.no_grad():
        for i, si in enumerate(sel):
            s = bench.make_inputs("r50", 1234 + i, dev, model)
            X, Y, Z = s["pts"].shape[2:]
            slot = cg.make_slot(model, (X, Y, Z), dev)
            st = pool[si]
            with torch.cuda.stream(st):
                cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
                torch.cuda.synchronize()
                graphs.append((cg.DenseGraph(model, slot, s, st).capture(), st))
    def burst(n):
        for k in range(n):
            g, st = graphs[k % N]
            with torch.cuda.stream(st):
                g.replay()
    burst(3 * N); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 60; burst(n); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-28s %d in flight: %.1f samples/s" % (name, N, n / dt), flush=True)
trial("streams 0,1,2", [0, 1, 2])
trial("streams 0,1,3", [0, 1, 3])
trial("streams 0,2,4", [0, 2, 4])
trial("streams 1,4,7", [1, 4, 7])
trial("streams 0..4 (5)", [0, 1, 2, 3, 4])
trial("streams 0..5 (6)", [0, 1, 2, 3, 4, 5])
trial("streams 0,1", [0, 1])
trial("streams 0..3", [0, 1, 2, 3])
