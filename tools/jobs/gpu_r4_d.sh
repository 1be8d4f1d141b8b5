#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_boundary.py tests/test_gpu_bench.py "tests/test_gpu_parity_full.py::test_full_size_r101_render_pair_vs_oracle" -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
timeout 300 python tools/kbench.py pool 2>&1 | grep -v amdgpu.ids | tee $O/kbench_pool.txt
for i in 1 2; do timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench$i.json 2>> $O/bench.err; done
COOCC_RENDER_GEO_LDS=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_geolds.json 2>> $O/bench.err
timeout 300 python bench.py --config r101 --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_r101.json 2>> $O/bench.err
for f in bench1 bench2 bench_geolds bench_r101; do python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    rr = d.get("roofline_render_r101") or {}
    print("$f", d["value"], d["ms_per_step"], "pool", (d.get("roofline_pool") or {}).get("avg_ms_per_step"), "render", (d.get("roofline_render") or {}).get("frac"), (d.get("roofline_render") or {}).get("avg_ms_per_step"), "r101", rr.get("frac"), rr.get("avg_ms"), (rr.get("geometry_in_kernel") or {}).get("frac"), (rr.get("geometry_in_kernel") or {}).get("avg_ms"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
tail -n 3 $O/bench.err
