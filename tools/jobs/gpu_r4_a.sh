#!/bin/bash
# Round 4, first GPU pass: the new plumbing (H2 twins, in-kernel split-K reduction, range guard, serving behind simple_test),
# then the bench through the product's serving API and the dense stage kernel by kernel.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_h2_engine.py tests/test_gpu_serving.py tests/test_gpu_graph.py tests/test_gpu_conv.py \
  tests/test_gpu_bench.py::test_graph_pipeline_outputs_equal_sequential_calls tests/test_gpu_modules.py tests/test_gpu_boundary.py -x -q > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
tail -15 $O/pytest.txt
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench2.json 2>> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing --api simple_test > $O/bench_simple_test.json 2> $O/bench_simple_test.err
COOCC_INKERNEL_REDUCE=0 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_reduce2launch.json 2>> $O/bench.err
[ -f tools/_bench_r3.py ] && timeout 300 python tools/_bench_r3.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_r3pipe.json 2>> $O/bench.err
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench3.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $R/tools/graph_probe.py > $O/graph_probe.txt 2>&1
python $R/tools/graph_trace.py /tmp/gp/gp_kernel_trace.csv --seq > $O/dense_stage_kernels.txt 2>&1
timeout 300 python $R/tools/dense_concurrency.py 2>&1 | grep -v amdgpu.ids > $O/dense_concurrency.txt
cd $R
for f in bench bench2 bench3 bench_simple_test bench_reduce2launch bench_r3pipe; do python - <<PY
import json
try:
    d = json.load(open("$O/$f.json"))
    print("$f", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("graph") or {}))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
head -45 $O/dense_stage_kernels.txt
cat $O/dense_concurrency.txt
tail -n 5 $O/bench.err; tail -n 5 $O/bench_simple_test.err
