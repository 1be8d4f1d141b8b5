#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
timeout 900 python tools/debug/fine2_corunner.py > $O/corunner.txt 2>&1
tail -n 12 $O/corunner.txt
