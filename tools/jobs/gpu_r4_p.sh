#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_h2_engine.py tests/test_gpu_graph.py -q > $O/pytest.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.txt | tail -n 5
for abl in 0 32; do
  export COOCC_H2_ABLATE=$abl
  [ "$abl" = "0" ] && unset COOCC_H2_ABLATE
  timeout 300 python tools/dense_concurrency.py 2>&1 | grep -v amdgpu.ids | sed "s/^/abl $abl: /" | head -n 2
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('abl $abl bench', d['value'], d['ms_per_step'], 'gemm frac', r['frac'], 'alone', r.get('frac_alone'), 'avg_ms_alone', r.get('avg_launch_ms_alone'))"
done
