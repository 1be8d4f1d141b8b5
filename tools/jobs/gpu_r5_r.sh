#!/bin/bash
# LiDAR encoder: base width carried as 32-channel rows (first SparseConv3d on the split-f16 engine)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5t
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_lidar.py tests/test_gpu_boundary.py tests/test_gpu_serving.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
