"""coocc_fuser_prepare_rows alone at configs[1] (V = 80 000, C = 128): time and effective bandwidth of its 205 MB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from co_occ_amd._lib import call, ptr
dev = torch.device("cuda:0")
V, C = 80000, 128
cat4 = torch.randn(V, 4 * C, device=dev)
pts = torch.randn(V, C, device=dev) * (torch.rand(V, 1, device=dev) < 0.12)
fl = torch.empty(2, V, device=dev, dtype=torch.uint8)
def run():
    call("coocc_fuser_prepare_rows", ptr(cat4), 1, 4 * C, ptr(pts), 1, C, ptr(cat4), ptr(fl[0]), ptr(fl[1]), 1, C, V)
for _ in range(5): run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 50
print("prepare_rows alone: %.1f us, %.2f TB/s of 205 MB" % (ms * 1e3, 205e6 / ms / 1e9))
