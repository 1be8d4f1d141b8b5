#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
timeout 300 python tools/serving_trace.py r50 36 6 3 0 > $O/trace_6_3_0.txt 2>&1
head -n 45 $O/trace_6_3_0.txt
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench1.json 2>> $O/bench.err
python -c "
import json
d = json.load(open('$O/bench1.json')); print('bench1', d['value'], d['ms_per_step'])"
