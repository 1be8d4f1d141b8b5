#!/bin/bash
# the z-column interpolation kernels inside the product: serving graphs, boundary, decoder / head modules, the full-size r50 scene,
# the compaction forms on the final dispatch; then the dense stage kernel by kernel and the default bench line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5y
mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_gpu_serving.py tests/test_gpu_graph.py tests/test_gpu_boundary.py tests/test_gpu_conv.py tests/test_gpu_knn.py tests/test_gpu_modules.py tests/test_gpu_parity_full.py -x -q -m gpu \
  -k "serving or graph or simple_test or pipelined or boundary or train_render or interp_column or upsample or occhead or compaction or decoder or head or hot_path_vs or r50_hot_path or (seed_sweep and 5-1.0) or stress200" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 300 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -3 $O/dense_stage_kernels.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-330 $O/bench_default.json
