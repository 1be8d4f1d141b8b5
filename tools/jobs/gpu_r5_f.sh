#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
timeout 900 python tools/debug/fine2_corunner.py > $O/corunner.txt 2>&1
tail -n 6 $O/corunner.txt
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "fine" > $O/pytest_fine.txt 2>&1
tail -n 4 $O/pytest_fine.txt
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_serving.py -x -q -m gpu 2>&1 | tail -n 2; done > $O/pytest_serving.txt
cat $O/pytest_serving.txt
bash tools/dense_stage_kernels.sh $O/dense_stage.txt
head -n 1 $O/dense_stage.txt; grep -E "fine" $O/dense_stage.txt | head -n 4
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['window_ms_per_step'])"
