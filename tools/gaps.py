#!/usr/bin/env python
"""Idle gaps in a rocprofv3 kernel trace (last bench step): python tools/gaps.py b_kernel_trace.csv [min_us]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
# last step: from the last k_fuser_prepare to the end
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_fuser_prepare")]
seg = rows[starts[-2]:starts[-1]]
t0 = int(seg[0]["Start_Timestamp"])
busy_end = t0
tot_gap = 0
print("step wall %.3f ms, %d kernels" % ((int(rows[starts[-1]]["Start_Timestamp"]) - t0) / 1e6, len(seg)))
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        g = (s - busy_end) / 1e3
        tot_gap += g
        if g >= thr:
            print("  gap %7.1f us at +%8.3f ms before %s" % (g, (s - t0) / 1e6, r["Kernel_Name"][:60]))
    busy_end = max(busy_end, e)
print("total idle %.3f ms" % (tot_gap / 1e3))
