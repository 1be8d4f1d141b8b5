#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5j
mkdir -p $O
cd $R
COOCC_PRINT_ERR=1 timeout 1500 python -m pytest tests/test_gpu_parity_full.py -x -q -m gpu -s > $O/pytest_parity_full.txt 2>&1
tail -n 5 $O/pytest_parity_full.txt
grep -E "\[err\].*(fine|pred_f)" $O/pytest_parity_full.txt | head -20
COOCC_FINE2_H2=0 timeout 600 python -m pytest tests/test_gpu_serving.py -x -q -m gpu -k calibrate -s > $O/calibrate_fine2_off.txt 2>&1
grep -E "con_enc tiles|passed|failed" $O/calibrate_fine2_off.txt
