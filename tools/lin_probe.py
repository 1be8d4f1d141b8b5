import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from co_occ_amd import core
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for rows, cin, cout in ((163840, 128, 64), (80000, 128, 64), (33600, 128, 64), (33600, 512, 128), (163840, 64, 17)):
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(cout, cin, generator=g) * 0.05).to(dev)
    pc = core.PackedConv(w)
    x = torch.randn(rows, cin, generator=g).to(dev)
    a = t(lambda: core.linear_rows(x, pc))
    line = "rows %6d %4d->%3d  linear_rows (fp32) %7.1f us" % (rows, cin, cout, a)
    if cin % 32 == 0:
        xh = core.rows_to_h2(x)
        b = t(lambda: core.linear_rows_h2(xh, rows, cin, pc))
        c = t(lambda: core.rows_to_h2(x))
        line += " | linear_rows_h2 %7.1f us (+ rows_to_h2 %5.1f us)" % (b, c)
    print(line, flush=True)
