#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o s -- python $R/bench.py --train --steps 10 --warmup 2 > $O/train_traced.json 2> $O/train_traced.err
python - <<PY
import csv, json
rows = list(csv.DictReader(open("/tmp/tt/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
d = json.load(open("$O/train_traced.json"))
print("bench.py --train under rocprofv3 --kernel-trace --stats: %.2f ms/step wall (traced); kernel time %.1f ms over 12 steps + setup" % (d["ms_per_step"], tot / 1e6))
for r in rows[:45]:
    print("  %-78s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
