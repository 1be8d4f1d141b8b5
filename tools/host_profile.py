"""cProfile of the host side of one sample (sequential, one stream): where the ~3.5 ms of dense-stage issue time and the
~1.2 ms of search issue time go.  python tools/host_profile.py [n]"""
import cProfile
import os
import pstats
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model, sd = bench.build_model("r50", dev)
s = bench.make_inputs("r50", 1234, dev, model)
with torch.no_grad():
    for _ in range(3):
        bench.step(model, s, 1)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        bench.step(model, s, 1)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)
