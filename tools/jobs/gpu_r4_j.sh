#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o s -- python $R/bench.py --train --steps 10 --warmup 2 > $O/train_traced.json 2> $O/train_traced.err
python - <<PY
import csv, json
rows = list(csv.DictReader(open("/tmp/tt/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
d = json.load(open("$O/train_traced.json"))
print("traced train step: %.2f ms/step wall; kernel time total %.1f ms over the whole process" % (d["ms_per_step"], tot / 1e6))
for r in rows[:25]:
    print("%-80s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
cd $R
timeout 600 python -m cProfile -o /tmp/train.prof bench.py --train --steps 10 --warmup 2 > $O/train_cprof.json 2> $O/train_cprof.err
python - <<PY
import pstats
p = pstats.Stats("/tmp/train.prof")
p.sort_stats("cumulative").print_stats(45)
PY
