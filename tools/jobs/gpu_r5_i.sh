#!/bin/bash
# round 5: state of the tree -- default bench, dense stage listing, per-config lines
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5i
mkdir -p $O
cd $R
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['window_ms_per_step'], d.get('roofline_pool'))"
COOCC_FINE2_H2=1 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_fine2.json 2> $O/bench_fine2.err
python -c "
import json; d=json.load(open('$O/bench_fine2.json')); print('fine2 on', d['value'], d['ms_per_step'], d['window_ms_per_step'])"
COOCC_POOL_SEG=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_poolseg0.json 2> $O/bench_poolseg0.err
python -c "
import json; d=json.load(open('$O/bench_poolseg0.json')); print('pool seg 0', d['value'], d['ms_per_step'], d['window_ms_per_step'], d.get('roofline_pool'))"
timeout 600 python bench.py --config r101 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_r101.json 2> $O/bench_r101.err
python -c "
import json; d=json.load(open('$O/bench_r101.json')); print('r101', d['value'], d['ms_per_step'], d['window_ms_per_step'], d.get('roofline_pool'))"
bash tools/dense_stage_kernels.sh $O/dense_stage.txt
head -n 1 $O/dense_stage.txt
