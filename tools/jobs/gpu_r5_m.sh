#!/bin/bash
# round 5: FPS bucket kernel with every slot's dirty buckets fetched in one round: parity (kNN tests) + times
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5m
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_fuser.py -x -q -m gpu > $O/pytest_knn.txt 2>&1
tail -n 4 $O/pytest_knn.txt
timeout 300 python tools/kbench.py fps 2>&1 | grep -v amdgpu.ids > $O/kbench_fps.txt
cat $O/kbench_fps.txt
COOCC_FPS_REG=0 timeout 300 python tools/kbench.py fps 2>&1 | grep -v amdgpu.ids > $O/kbench_fps_noreg.txt
cat $O/kbench_fps_noreg.txt
timeout 400 python bench.py --config stress200 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_stress200.json 2> $O/bench_stress200.err
python -c "
import json; d=json.load(open('$O/bench_stress200.json')); print('stress200', d['value'], d['ms_per_step'], d['window_ms_per_step'])"
