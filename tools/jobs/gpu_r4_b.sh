#!/bin/bash
# Round 4: the whole GPU suite + the bench lines (default, simple_test API) + the dense stage kernel by kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
grep -E 'FAILED|ERROR|passed|failed' $O/pytest.txt | tail -n 40
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench2.json 2>> $O/bench.err
COOCC_FUSED_RENDER_HEADS=0 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_layerwise_heads.json 2>> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing --api simple_test > $O/bench_simple_test.json 2> $O/bench_simple_test.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $R/tools/graph_probe.py > $O/graph_probe.txt 2>&1
python $R/tools/graph_trace.py /tmp/gp/gp_kernel_trace.csv --seq > $O/dense_stage_kernels.txt 2>&1
timeout 300 python $R/tools/dense_concurrency.py 2>&1 | grep -v amdgpu.ids > $O/dense_concurrency.txt
cd $R
for f in bench bench2 bench_layerwise_heads bench_simple_test; do python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("$f", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("graph") or {}).get("dense_stage_ms"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
head -n 42 $O/dense_stage_kernels.txt
cat $O/dense_concurrency.txt
tail -n 5 $O/bench.err; tail -n 5 $O/bench_simple_test.err
cp gpurun_out/r4_*.txt $O/ 2>/dev/null
