#!/bin/bash
# k_gemm_h2n<4> at four workgroups per CU (128 registers, one spilled) against three: LiDAR 128 -> 128 stage, dense stage alone
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5x
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_lidar.py -x -q -m gpu -k "narrow" 2>&1 | tail -2
timeout 300 python tools/kbench.py lidar 2>&1 | grep -E "128->128|64->128|^lidar" | cut -c1-150 | tee -a $O/kbench_lidar.txt
timeout 300 python tools/graph_probe.py 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/kbench_lidar.txt
