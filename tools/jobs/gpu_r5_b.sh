#!/bin/bash
# round 5: the ratio-2 one-launch fine branch on the split-f16 engine: parity, then its time alone and inside the dense stage
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "fine" > $O/pytest_fine.txt 2>&1
tail -8 $O/pytest_fine.txt
timeout 600 python -m pytest tests/test_gpu_serving.py -x -q -m gpu > $O/pytest_serving.txt 2>&1
tail -5 $O/pytest_serving.txt
for occ in 2 3; do
  COOCC_FINE2_OCC=$occ bash tools/dense_stage_kernels.sh $O/dense_stage_occ$occ.txt
  head -1 $O/dense_stage_occ$occ.txt; grep -E "fine|k_conv<128, 64" $O/dense_stage_occ$occ.txt | head -8
done
COOCC_FINE2_H2=0 bash tools/dense_stage_kernels.sh $O/dense_stage_fine3.txt
head -1 $O/dense_stage_fine3.txt; grep -E "fine" $O/dense_stage_fine3.txt | head -8
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['window_ms_per_step'])"
