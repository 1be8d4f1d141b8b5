#!/usr/bin/env python
"""Print a rocprofv3 *_kernel_stats.csv compactly:  python tools/kstats.py FILE [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    name = re.sub(r"\(.*", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"(radix_sort_\w+|k_\w+(<[^>]*>)?|\w+)$", name.split("<rocprim::wrapped")[0]) if "trampoline" not in name else None
    if "trampoline" in r["Name"]:
        mm = re.search(r"detail::(radix_sort_onesweep_\w+|\w+)<", r["Name"][r["Name"].find("target_arch"):])
        name = "rocprim:" + (mm.group(1) if mm else "kernel")
    print("%-48s calls %5s  avg %9.1f us  total %9.1f us  %5s%%" % (name[:48], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                   float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
