"""Dense-stage ceiling: N hipGraphs (one per slot) replayed round-robin on N streams with NOTHING else on the GPU (no pooling,
no index search) -> samples/s the dense stage alone allows at N graphs in flight.  Compare with bench.py's pipeline number to see
what the prefetched search stage costs."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")       # as bench.py: before HIP starts
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from co_occ_amd import graph as cg

dev = torch.device("cuda:0")
cfg = "r50"
bench.CFGNAME[0] = cfg
model, _ = bench.build_model(cfg, dev)
for N in (1, 2, 3, 4):
    graphs, streams = [], []
    with torch.no_grad():
        for i in range(N):
            s = bench.make_inputs(cfg, 1234 + i, dev, model)
            X, Y, Z = s["pts"].shape[2:]
            slot = cg.make_slot(model, (X, Y, Z), dev)
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
                torch.cuda.synchronize()
                graphs.append(cg.DenseGraph(model, slot, s, st).capture())
            streams.append(st)
    def burst(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % N]):
                graphs[k % N].replay()
    burst(3 * N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    burst(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("dense graphs in flight %d: %.1f samples/s (%.3f ms per sample)" % (N, n / dt, 1e3 * dt / n), flush=True)
    del graphs
