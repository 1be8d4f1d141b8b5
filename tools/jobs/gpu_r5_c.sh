#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5c
mkdir -p $O
cd $R
timeout 300 python tools/debug/fine2_determinism.py > $O/determinism.txt 2>&1
grep -c "logits equal True" $O/determinism.txt; grep "three" $O/determinism.txt
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu > $O/pytest_modules.txt 2>&1
tail -n 4 $O/pytest_modules.txt
timeout 900 python -m pytest tests/test_gpu_serving.py tests/test_gpu_graph.py -x -q -m gpu > $O/pytest_serving.txt 2>&1
tail -n 4 $O/pytest_serving.txt
bash tools/dense_stage_kernels.sh $O/dense_stage.txt
head -n 1 $O/dense_stage.txt; grep -E "fine" $O/dense_stage.txt | head -n 4
COOCC_MERGED_PRED_Q=0 bash tools/dense_stage_kernels.sh $O/dense_stage_nomerge.txt
head -n 1 $O/dense_stage_nomerge.txt
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['window_ms_per_step'])"
