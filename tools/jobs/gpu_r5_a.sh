#!/bin/bash
# round 5, first call: the serving / ADVICE changes + the new bench modes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_serving.py tests/test_gpu_graph.py tests/test_gpu_bench.py -x -q -m gpu -s > $O/pytest_serving.txt 2>&1
tail -15 $O/pytest_serving.txt
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --api pipelined_test > $O/bench_pipelined.json 2> $O/bench_pipelined.err
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --api simple_test > $O/bench_simple.json 2> $O/bench_simple.err
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('window_ms_per_step'), d.get('graph'))
except Exception as e: print('$f'.split('/')[-1], 'FAILED', e)"; done
tail -n 5 $O/*.err
