"""Host time of one hipGraphLaunch of the dense stage (136 kernel nodes): is the issuing thread the limit of the pipeline?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from co_occ_amd import graph as cg
dev = torch.device("cuda:0")
bench.CFGNAME[0] = "r50"
model, _ = bench.build_model("r50", dev)
graphs = []
with torch.no_grad():
    for i in range(2):
        s = bench.make_inputs("r50", 1234 + i, dev, model)
        X, Y, Z = s["pts"].shape[2:]
        slot = cg.make_slot(model, (X, Y, Z), dev)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
            torch.cuda.synchronize()
            graphs.append((cg.DenseGraph(model, slot, s, st).capture(), st))
for N in (1, 2):
    for _ in range(4):
        for g, st in graphs[:N]:
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for k in range(40):
        g, st = graphs[k % N]
        with torch.cuda.stream(st):
            a = time.perf_counter(); g.replay(); host.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.sort()
    print("%d stream(s): host time per replay median %.3f ms (min %.3f max %.3f); 40 replays issued in %.1f ms, finished after %.1f ms"
          % (N, 1e3 * host[len(host) // 2], 1e3 * host[0], 1e3 * host[-1], 1e3 * (t1 - t0), 1e3 * (t2 - t0)), flush=True)
