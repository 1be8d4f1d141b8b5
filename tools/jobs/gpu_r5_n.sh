#!/bin/bash
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5n
mkdir -p $O
cd $R
rep() { name=$1; shift; ok=0; bad=0; oob=0; vals=""; for i in 1 2 3 4 5; do "$@" > $O/$name.json 2> $O/$name.$i.err; if [ $? -eq 0 ]; then ok=$((ok+1)); vals="$vals $(grep -o '"value": [0-9.]*' $O/$name.json | head -1 | cut -d' ' -f2)"; else bad=$((bad+1)); fi; oob=$((oob + $(grep -c 'k_sparse_tap_sum: voxel' $O/$name.json))); done; echo "$name ok=$ok crashed=$bad oob_lines=$oob values:$vals"; }
rep openocc_fill env COOCC_SERVING_AHEAD_LARGE=3 timeout 200 python bench.py --config openocc --steps 10 --warmup 2 --no-cpu-baseline --windows 5
rep stress200_fill env COOCC_SERVING_AHEAD_LARGE=3 timeout 250 python bench.py --config stress200 --steps 5 --warmup 2 --no-cpu-baseline --windows 6
rep openocc_memset env COOCC_MAP_MEMSET=1 COOCC_SERVING_AHEAD_LARGE=3 timeout 200 python bench.py --config openocc --steps 10 --warmup 2 --no-cpu-baseline --windows 5
