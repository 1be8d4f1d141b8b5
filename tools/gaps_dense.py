#!/usr/bin/env python
"""Idle time of the dense stage (main stream) in a rocprofv3 kernel trace of bench.py --prefetch 1:
    python tools/gaps_dense.py b_kernel_trace.csv [min_us]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
skip = ("k_fps", "k_ball", "k_knn", "k_assign", "k_fuser_prepare", "k_flag", "k_compact", "k_lin_to", "k_index_rows", "k_fpsv",
        "k_scan", "k_threshold")
main = [r for r in rows if not r["Kernel_Name"].startswith(skip) and "fillBuffer" not in r["Kernel_Name"]]
ends = [i for i, r in enumerate(main) if r["Kernel_Name"].startswith("k_upsample_maps")]
seg = main[ends[-3] + 1: ends[-2] + 1]
t0 = int(seg[0]["Start_Timestamp"]); busy = t0; gap = 0.0; big = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy:
        g = (s - busy) / 1e3; gap += g
        if g > thr:
            big.append((g, (s - t0) / 1e6, r["Kernel_Name"][:56]))
    busy = max(busy, e)
print("dense stage wall %.3f ms, %d kernels, idle %.3f ms; gap to previous sample %.1f us" % (
    (busy - t0) / 1e6, len(seg), gap / 1e3, (t0 - int(main[ends[-3]]["End_Timestamp"])) / 1e3))
for g in big:
    print("  gap %6.1f us at +%.3f ms before %s" % g)
