#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_serving.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
timeout 300 python tools/kbench.py pool 2>&1 | grep -v amdgpu.ids | tee $O/kbench_pool.txt
COOCC_POOL_SIDE_STREAM=0 timeout 300 python tools/kbench.py pool 2>&1 | grep -v amdgpu.ids | tee $O/kbench_pool_noside.txt
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing > $O/bench$i.json 2>> $O/bench.err; done
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_k.json 2>> $O/bench.err
timeout 300 python bench.py --config r101 --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_r101.json 2>> $O/bench.err
timeout 300 python bench.py --config stress200 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_stress200.json 2>> $O/bench.err
for f in bench1 bench2 bench_k bench_r101 bench_stress200; do python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("$f", d["value"], d["ms_per_step"], (d.get("roofline_pool") or {}).get("avg_ms_per_step"), (d.get("roofline_pool") or {}).get("frac"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $R/tools/search_probe.py > $O/search_stage.txt 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/sp/s_kernel_stats.csv")))
for r in rows[:12]:
    print("%-62s x%5.1f  %8.1f us/sample" % (r["Name"][:62], int(r["Calls"]) / 23, float(r["TotalDurationNs"]) / 23 / 1e3))
PY
tail -n 3 $O/bench.err
