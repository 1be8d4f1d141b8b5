#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_boundary.py tests/test_gpu_bench.py tests/test_gpu_serving.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
cd /tmp && export TMPDIR=/tmp
for side in 1 0; do
rm -rf /tmp/pp
COOCC_POOL_SIDE_STREAM=$side timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o s -- python $R/tools/kbench.py poolprof > $O/poolprof_side$side.txt 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/pp/s_kernel_stats.csv")))
print("side stream $side:", open("$O/poolprof_side$side.txt").read().strip().splitlines()[-8:][0][:80] if False else "")
for r in rows[:8]:
    print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep "lift_splat r101" $O/poolprof_side$side.txt
done
cd $R
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench1.json 2>> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench1.json"))
rr = d.get("roofline_render_r101") or {}
print("bench1", d["value"], d["ms_per_step"], "pool", (d.get("roofline_pool") or {}).get("avg_ms_per_step"), "r101", rr.get("frac"), (rr.get("geometry_in_kernel") or {}).get("frac"))
PY
