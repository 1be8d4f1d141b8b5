"""Per-kernel table of one dense-stage graph replay from a rocprofv3 kernel trace of tools/graph_probe.py:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o gp -- python tools/graph_probe.py ; python tools/graph_trace.py DIR/gp_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the probe ends with 3 + 20 graph replays back to back: take the kernels of the last 20 replays by counting a kernel
# that runs exactly once per replay
marker = "k_occhead_mix"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"].split("(")[0]]
n = 20
start = idx[-n] if len(idx) >= n else 0
# a replay starts with the first kernel of the dense stage; walk back from the first marker to the previous marker + 1
prev = idx[-n - 1] if len(idx) > n else -1
per = (idx[-1] - idx[-n]) // (n - 1)
first = idx[-n] - (idx[-n] - prev - 1 if False else 0)
sel = rows[idx[-n - 1] + 1: idx[-1] + 1] if len(idx) > n else rows
# sel covers exactly n replays shifted by a constant offset (from just after a marker to the last marker)
agg = collections.OrderedDict()
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"].split("(")[0][:60]
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print("dense stage: %.3f ms of kernels per replay, %.3f ms wall per replay, %d launches per replay" % (tot / n / 1e6, span / n / 1e6, len(sel) // n))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s x%5.1f  %.3f ms/replay  avg %6.1f us" % (k, v[0] / n, v[1] / n / 1e6, v[1] / v[0] / 1e3))
if len(sys.argv) > 2 and sys.argv[2] == "--seq":
    # the launches of the LAST replay in issue order: name, grid (workgroups), duration
    last = rows[idx[-2] + 1: idx[-1] + 1]
    print("\nlast replay, launch by launch (shifted: starts right after the previous replay's k_occhead_mix):")
    t0 = int(last[0]["Start_Timestamp"])
    for r in last:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        print("%8.1f us  %-50s wgs %6d  %6.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, r["Kernel_Name"].split("(")[0][:50], wg, d / 1e3))
