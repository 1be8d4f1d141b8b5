#!/bin/bash
# which of the two z-column kernels (if any) makes the pipelined loop differ from eager calls: COOCC_INTERP_COLUMN = 0 / 1 / 2, three runs each
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5z
mkdir -p $O
cd $R
for mask in 2 1 0; do
  for rep in 1 2 3; do
    COOCC_INTERP_COLUMN=$mask timeout 120 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k "pipelined_test_loop or three_graphs" > $O/p_${mask}_${rep}.txt 2>&1
    echo "mask $mask rep $rep: $(tail -1 $O/p_${mask}_${rep}.txt) $(grep -o 'sample [0-9]*: [a-z_]* differs' $O/p_${mask}_${rep}.txt | head -1)" | tee -a $O/summary.txt
  done
done
