#!/bin/bash
# z-column forms of the FPN upsample-add and the OccHead mix (half columns): bit equality, then their times in the dense stage
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5x
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "interp_column or upsample_add or occhead_mix" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
timeout 300 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -40 $O/dense_stage_kernels.txt | grep -E "dense stage|mix|upsample_add"
