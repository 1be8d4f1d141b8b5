#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4k
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_h2_engine.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 8
timeout 600 python bench.py --train --steps 10 --warmup 2 > $O/train.json 2> $O/train.err
COOCC_TRAIN_H2_DGRAD=1 COOCC_TRAIN_H2_DGRAD_SCALE=4096 timeout 600 python bench.py --train --steps 10 --warmup 2 > $O/train_dgrad_h2.json 2> $O/train_dgrad.err
COOCC_TRAIN_H2=0 timeout 600 python bench.py --train --steps 10 --warmup 2 > $O/train_f32.json 2> $O/train_f32.err
python -c "
import json
for f in ('train', 'train_dgrad_h2', 'train_f32'):
    d = json.load(open('$O/%s.json' % f)); print(f, d['value'], d['ms_per_step'], d.get('kernel_groups_ms_per_step'))"
