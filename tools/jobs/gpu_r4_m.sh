#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4m
mkdir -p $O
cd $R
timeout 600 python bench.py --with-lidar --steps 40 --warmup 3 > $O/bench_with_lidar.json 2> $O/bench_with_lidar.err
tail -n 3 $O/bench_with_lidar.err
python -c "
import json
d = json.load(open('$O/bench_with_lidar.json')); print('with-lidar', d['value'], d['ms_per_step'], d['config']['workload'], d.get('graph', {}).get('eager_fallbacks'))"
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/bench.json 2>/dev/null
python -c "
import json
d = json.load(open('$O/bench.json')); print('default', d['value'], d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_serving.py tests/test_gpu_bench.py -q 2>&1 | tail -n 3
