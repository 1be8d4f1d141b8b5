#!/bin/bash
# one-launch stream compaction with 16-byte loads, eight in flight (the 4-byte form took as long as the three launches it replaced)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5v
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_gpu_knn.py tests/test_gpu_serving.py tests/test_gpu_graph.py -x -q -m gpu \
  -k "compaction or layout or native_search or serving or graph or simple_test or pipelined" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 360 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -36 $O/dense_stage_kernels.txt | grep -E "dense stage|flag|reduce|rows_to_h2|argmax"
timeout 200 python tools/kbench.py search 2>&1 | grep -v amdgpu | tail -12 | tee $O/kbench_search.txt
