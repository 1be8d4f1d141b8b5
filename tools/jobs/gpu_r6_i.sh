# round 6, call i: ServingPipeline outside bench.py, knob by knob, against tools/dense_concurrency.py on the same box
mkdir -p gpurun_out/r6i
O=gpurun_out/r6i
( timeout 300 python tools/dense_concurrency.py 3:3 6:3 6:3:ev 12:3 2:2 4:4 2>&1 | grep -v amdgpu
  for rep in 1 2; do
  timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  COOCC_SERVING_DIAG_SKIP=7 timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  COOCC_SERVING_DIAG_SKIP=7 COOCC_SERVING_PROBE=nowait timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  COOCC_SERVING_DIAG_SKIP=7 COOCC_SERVING_PROBE=nowait,noev timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  done
  COOCC_SERVING_DIAG_SKIP=7 timeout 200 python tools/serving_probe.py 12 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 12 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 9 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 6 2 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 8 2 60 2>&1 | grep "serving alone"
) | tee $O/serving_probe.txt
