#!/bin/bash
# pooling forms side by side (COOCC_POOL_SPLIT auto / one kernel / two kernels) + module tests + one bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_boundary.py tests/test_gpu_serving.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
for sp in -1 0 1; do
  echo "== COOCC_POOL_SPLIT=$sp" | tee -a $O/pool.txt
  COOCC_POOL_SPLIT=$sp timeout 300 python tools/kbench.py pool 2>&1 | tee -a $O/pool.txt | grep -i "ms" | head -n 12
done
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench1.json 2>> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench1.json"))
rr = d.get("roofline_render_r101") or {}
print("bench1", d["value"], d["ms_per_step"], "pool", (d.get("roofline_pool") or {}).get("avg_ms_per_step"), "r101", rr.get("frac"), (rr.get("geometry_in_kernel") or {}).get("frac"))
PY
