#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5e
mkdir -p $O
cd $R
timeout 900 python tools/debug/fine2_pipeline_race.py > $O/race.txt 2>&1
tail -n 40 $O/race.txt
