#!/bin/bash
# the per-config bench lines on the final tree (one run each)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fin
mkdir -p $O
cd $R
timeout 400 python bench.py --config r101 --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_bench_r101.json 2>/dev/null
timeout 400 python bench.py --config openocc --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_bench_openocc_f32.json 2>/dev/null
timeout 400 python bench.py --config openocc --dtype f16 --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_bench_openocc_f16.json 2>/dev/null
timeout 400 python bench.py --config stress200 --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_bench_stress200.json 2>/dev/null
timeout 400 python bench.py --config stress200_r101 --steps 20 --warmup 3 --no-cpu-baseline > $O/r4_bench_stress200_r101.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing --api simple_test > $O/r4_bench_api_simple_test.json 2>/dev/null
timeout 400 python bench.py --with-lidar --steps 40 --warmup 3 > $O/r4_bench_with_lidar.json 2>/dev/null
timeout 300 python bench.py --train --steps 10 --warmup 2 > $O/r4_bench_train.json 2>/dev/null
for f in $O/r4_bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])
except Exception as e: print('$f'.split('/')[-1], 'FAILED', e)"; done
