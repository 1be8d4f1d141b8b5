"""Dense-stage ceiling: G hipGraphs (one per slot) replayed round-robin on N streams with NOTHING else on the GPU (no pooling,
no index search) -> samples/s the dense stage alone allows.  Compare with bench.py's pipeline number to see what the loop around
it costs.

    python tools/dense_concurrency.py                 # G = N = 1, 2, 3, 4
    python tools/dense_concurrency.py 6:3 4:2 8:4     # G graphs (slots) over N streams, graph k on stream k mod N
    python tools/dense_concurrency.py 6:3:ev          # ... with the serving loop's per-replay events (a cross-stream wait before,
                                                      #     an event record after every replay)
"""
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
specs = sys.argv[1:] or ["1:1", "2:2", "3:3", "4:4"]
for spec in specs:
    parts = spec.split(":")
    G, N = int(parts[0]), int(parts[1])
    ev_mode = len(parts) > 2 and parts[2] == "ev"
    graphs = []
    streams = [torch.cuda.Stream(device=dev) for _ in range(N)]
    other = torch.cuda.Stream(device=dev)
    with torch.no_grad():
        for i in range(G):
            s = bench.make_inputs(cfg, 1234 + i, dev, model)
            X, Y, Z = s["pts"].shape[2:]
            slot = cg.make_slot(model, (X, Y, Z), dev)
            st = streams[i % N]
            with torch.cuda.stream(st):
                cg.search_into_slot(model, slot, s["depth"], s["ctx"], s["cams"], s["pts"])
                torch.cuda.synchronize()
                graphs.append(cg.DenseGraph(model, slot, s, st).capture())

    def burst(n):
        for k in range(n):
            st = streams[(k % G) % N]
            if ev_mode:
                e = torch.cuda.Event()
                e.record(other)
                st.wait_event(e)
            with torch.cuda.stream(st):
                graphs[k % G].replay()
                if ev_mode:
                    torch.cuda.Event().record()
    burst(3 * G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 60
    burst(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("dense graphs %d over %d streams%s: %.1f samples/s (%.3f ms per sample)" % (
        G, N, " + per-replay events" if ev_mode else "", n / dt, 1e3 * dt / n), flush=True)
    del graphs
