#!/bin/bash
# final defaults (z-column upsample-add on, mix off): pipelined loop / graphs == eager, A/B tests, decoder goldens; dense stage; bench
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
timeout 200 python -m pytest tests/test_gpu_serving.py tests/test_gpu_conv.py tests/test_gpu_modules.py tests/test_gpu_graph.py -x -q -m gpu \
  -k "pipelined_test_loop or three_graphs or eight or interp_column or upsample or occhead or decoder or graph_replay" > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
timeout 60 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -1 $O/dense_stage_kernels.txt
timeout 100 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
