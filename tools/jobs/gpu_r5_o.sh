#!/bin/bash
# FPS bucket kernel (grids above 640 buckets): grouped (four buckets per pass) against one bucket per wave, both with the
# row-reduced winner and 24-bit multiplies; parity tests for both
ulimit -c 0
mkdir -p gpurun_out/r5o
for g in 1 0; do
  COOCC_FPS_GROUPED=$g timeout 900 python -m pytest tests/test_gpu_knn.py -x -q -m gpu > gpurun_out/r5o/pytest_g$g.txt 2>&1; tail -1 gpurun_out/r5o/pytest_g$g.txt
  COOCC_FPS_GROUPED=$g timeout 300 python tools/kbench.py fps > gpurun_out/r5o/kbench_fps_g$g.txt 2>&1; cat gpurun_out/r5o/kbench_fps_g$g.txt
done
COOCC_FPS_GROUPED=0 COOCC_FPSDBG_GRID=200,200,16 timeout 300 python tools/kbench.py fpsdbg > gpurun_out/r5o/fpsdbg_g0.txt 2>&1; head -18 gpurun_out/r5o/fpsdbg_g0.txt | cut -c1-90
