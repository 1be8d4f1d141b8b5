#!/bin/bash
# soak of the shipped defaults (z-column FPN add on): the pipelined loop and the three-graphs test, five more runs
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5s
mkdir -p $O
cd $R
for rep in 1 2 3 4 5; do
  timeout 60 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k "pipelined_test_loop or three_graphs" > $O/p_${rep}.txt 2>&1
  echo "default rep $rep: $(tail -1 $O/p_${rep}.txt) $(grep -o 'sample [0-9]*: [a-z_]* differs' $O/p_${rep}.txt | head -1)" | tee -a $O/summary.txt
done
