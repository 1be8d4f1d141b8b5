#!/usr/bin/env python
"""One sample's dense stage, kernel by kernel, from a rocprofv3 kernel trace of bench.py --prefetch 1:
    python tools/timeline.py b_kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
skip = ("k_fps", "k_ball", "k_knn", "k_assign", "k_fuser_prepare", "k_flag", "k_compact", "k_lin_to", "k_index_rows", "k_fpsv",
        "k_scan", "k_threshold")
main = [r for r in rows if not r["Kernel_Name"].startswith(skip)]
ends = [i for i, r in enumerate(main) if r["Kernel_Name"].startswith("k_upsample_maps")]
seg = main[ends[-3] + 1: ends[-2] + 1]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("+%7.3f ms %8.1f us  grid %-8s %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Grid_Size_X", "?") if "Grid_Size_X" in r else r.get("Grid_Size", "?"),
                                              r["Kernel_Name"].split("(")[0][:70]))
