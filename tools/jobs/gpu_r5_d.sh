#!/bin/bash
# round 5: which switch makes test_pipelined_test_loop_equals_per_sample_calls differ (eager simple_test vs the N-slot loop)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5d
mkdir -p $O
cd $R
T=tests/test_gpu_serving.py::test_pipelined_test_loop_equals_per_sample_calls
for env in "A=1" "COOCC_MERGED_PRED_Q=0" "COOCC_FINE2_H2=0" "A=2"; do
  echo "== $env" >> $O/summary.txt
  env $env timeout 600 python -m pytest $T -x -q -m gpu 2>&1 | grep -E "passed|failed|differs" | tail -n 3 >> $O/summary.txt
done
cat $O/summary.txt
