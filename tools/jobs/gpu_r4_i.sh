#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lidar.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
COOCC_LIDAR_H2=1 timeout 300 python tools/kbench.py lidar 2>&1 | tee $O/lidar_h2.txt | tail -n 16
COOCC_LIDAR_H2=0 timeout 300 python tools/kbench.py lidar 2>&1 | tee $O/lidar_f32.txt | tail -n 16
