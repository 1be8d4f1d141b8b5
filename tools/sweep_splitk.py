"""split-K sweep of the small-grid direct convolutions (enc.l2 / l3 / stride-2 layers at the r50 shapes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from co_occ_amd import core
dev = torch.device("cuda", 0)

def timeit(f, n=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

shapes = [("enc.l2.conv2", 512, 512, (25, 25, 2), 3, 1), ("enc.l2.conv1", 256, 512, (50, 50, 4), 3, 2),
          ("enc.l3.conv2", 1024, 1024, (13, 13, 1), 3, 1), ("enc.l3.conv1", 512, 1024, (25, 25, 2), 3, 2),
          ("enc.l1.conv1", 128, 256, (100, 100, 8), 3, 2)]
with torch.no_grad():
    for name, ci, co, g, k, st in shapes:
        x = core.Rows(torch.randn(g[0] * g[1] * g[2], ci, device=dev), 1, g[0], g[1], g[2], ci)
        pc = core.PackedConv(torch.randn(co, ci, k, k, k, device=dev) * 0.02, ksize=k, stride=st, pad=k // 2)
        res = []
        for sk in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            try:
                res.append((sk, timeit(lambda: core.conv_rows(x, pc, relu=True, splitk=sk))))
            except Exception as ex:
                res.append((sk, float("nan")))
        print(name, "  ".join("sk%d %.3f" % r for r in res))
