#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4k
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_h2_engine.py tests/test_gpu_backward.py tests/test_abi.py -q -x > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 12
timeout 600 python bench.py --train --steps 10 --warmup 2 > $O/train.json 2> $O/train.err
timeout 600 python bench.py --train --steps 10 --warmup 2 --train-prefetch 0 > $O/train_noprefetch.json 2> $O/train_np.err
COOCC_TRAIN_H2_DGRAD=0 timeout 600 python bench.py --train --steps 10 --warmup 2 > $O/train_dgrad_f32.json 2> $O/train_dgrad.err
COOCC_TRAIN_H2=0 timeout 600 python bench.py --train --steps 10 --warmup 2 > $O/train_f32.json 2> $O/train_f32.err
python -c "
import json
for f in ('train', 'train_noprefetch', 'train_dgrad_f32', 'train_f32'):
    try:
        d = json.load(open('$O/%s.json' % f)); print(f, d['value'], d['ms_per_step'], d.get('kernel_groups_ms_per_step'))
    except Exception as e: print(f, 'FAILED', e)"
tail -n 5 $O/train.err
