#!/bin/bash
# FPS bucket kernel (grids above 640 buckets): grouped refresh (four dirty buckets per pass through per-wave LDS lists) against one
# bucket per wave; parity tests for both; the split-con_enc.0 regression test with the fill kernel and with the memset node
ulimit -c 0
mkdir -p gpurun_out/r5o
for g in 1 0; do
  COOCC_FPS_GROUPED=$g timeout 900 python -m pytest tests/test_gpu_knn.py -x -q -m gpu > gpurun_out/r5o/pytest_g$g.txt 2>&1; tail -1 gpurun_out/r5o/pytest_g$g.txt
  COOCC_FPS_GROUPED=$g timeout 300 python tools/kbench.py fps > gpurun_out/r5o/kbench_fps_g$g.txt 2>&1; grep "^fps" gpurun_out/r5o/kbench_fps_g$g.txt
done
COOCC_FPS_GROUPED=1 COOCC_FPSDBG_GRID=200,200,16 timeout 300 python tools/kbench.py fpsdbg > gpurun_out/r5o/fpsdbg_g1.txt 2>&1; head -18 gpurun_out/r5o/fpsdbg_g1.txt | cut -c1-90
timeout 600 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k "scatter_form" > gpurun_out/r5o/pytest_split.txt 2>&1; tail -2 gpurun_out/r5o/pytest_split.txt
COOCC_MAP_MEMSET=1 timeout 600 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k "scatter_form" > gpurun_out/r5o/pytest_split_memset.txt 2>&1; tail -2 gpurun_out/r5o/pytest_split_memset.txt
