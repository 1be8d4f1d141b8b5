#!/bin/bash
# one-launch stream compaction, batched split-K second pass, one H2 conversion for both G1 gather GEMMs:
# direct tests + the suites that run through them, the dense stage kernel by kernel, the default bench line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5u
mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_gpu_knn.py tests/test_gpu_modules.py tests/test_gpu_h2_engine.py tests/test_gpu_conv.py -x -q -m gpu \
  -k "compaction or layout or g1_shared or bifuser or splitk or conv3d_bn or con_enc0 or native_search or hot_path" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 420 python -m pytest tests/test_gpu_serving.py tests/test_gpu_graph.py tests/test_gpu_lidar.py -x -q -m gpu > $O/pytest_b.txt 2>&1; tail -3 $O/pytest_b.txt
timeout 360 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -36 $O/dense_stage_kernels.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
