#!/bin/bash
# round 5: pointwise GEMM kernel (k_gemm_h2p): parity tests that cover the 1x1x1 layers, dense stage listing, A/B bench
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5l
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_h2_engine.py tests/test_gpu_modules.py tests/test_gpu_graph.py tests/test_gpu_serving.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -n 6 $O/pytest.txt
bash tools/dense_stage_kernels.sh $O/dense_stage.txt
head -n 12 $O/dense_stage.txt
COOCC_H2_POINTWISE=0 bash tools/dense_stage_kernels.sh $O/dense_stage_off.txt
head -n 1 $O/dense_stage_off.txt
for e in 1 0 1 0; do
COOCC_H2_POINTWISE=$e timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err
python -c "
import json; d=json.load(open('$O/bench_$e.json')); print('pointwise=$e', d['value'], d['ms_per_step'], d['window_ms_per_step'])"
done
