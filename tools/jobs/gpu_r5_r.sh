#!/bin/bash
# narrow-N rule-book GEMM (k_gemm_h2n) against k_gemm_h2w<table>: bits, tests, time
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5s
mkdir -p $O
cd $R
for g in 1 0; do
  echo "== COOCC_H2_NARROW=$g" | tee -a $O/checksum.txt
  COOCC_H2_NARROW=$g timeout 300 python tools/debug/lidar_checksum.py 2>&1 | grep -v amdgpu | tee -a $O/checksum.txt
done
timeout 900 python -m pytest tests/test_gpu_lidar.py tests/test_gpu_h2_engine.py tests/test_gpu_serving.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 300 python bench.py --with-lidar --steps 40 --warmup 3 --no-cpu-baseline > $O/bench_with_lidar.json 2>/dev/null; python -c "import json; d=json.load(open('$O/bench_with_lidar.json')); print('with-lidar', d['value'], d['ms_per_step'])"
