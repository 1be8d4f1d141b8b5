#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_serving.py tests/test_gpu_boundary.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
for g in 1 0; do
COOCC_SCATTER_GROUPED=$g timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r50 grouped=$g', d['value'], d['ms_per_step'])"
COOCC_SCATTER_GROUPED=$g timeout 400 python bench.py --config openocc --dtype f16 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('openocc f16 grouped=$g', d['value'], d['ms_per_step'])"
done
