"""Which aten ops (small torch launches / copies) one sample of the hot path issues besides the C-ABI kernels."""
import os
import sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model, sd = bench.build_model("r50", dev)
s = bench.make_inputs("r50", 1234, dev, model)
with torch.no_grad():
    for _ in range(3):
        bench.step(model, s, 1)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
        for _ in range(5):
            bench.step(model, s, 1)
        torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") and e.count >= 5]
rows.sort(key=lambda e: -e.self_cpu_time_total)
for e in rows[:45]:
    st = [l for l in e.stack if "/root/repo" in l or "co_occ_amd" in l or "bench.py" in l]
    print("%-28s x%-4d %7.1f us/step   %s" % (e.key, e.count // 5, e.self_cpu_time_total / 5, (st[0].strip()[-90:] if st else "")))
