#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
timeout 900 python tools/debug/fine2_concurrent.py full full+inv persistent+inv > $O/concurrent.txt 2>&1
tail -n 12 $O/concurrent.txt
