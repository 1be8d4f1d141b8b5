#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_serving.py tests/test_gpu_boundary.py tests/test_gpu_openocc.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $R/tools/graph_probe.py > $O/graph_probe.txt 2>&1
python $R/tools/graph_trace.py /tmp/gp/gp_kernel_trace.csv --seq > $O/dense_stage_kernels.txt 2>&1
head -n 1 $O/dense_stage_kernels.txt
cd $R
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r50', d['value'], d['ms_per_step'])"
