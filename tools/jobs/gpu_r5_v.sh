#!/bin/bash
# full GPU suite on the tree with the one-launch compaction / batched split-K second pass / shared G1 conversion,
# then the dense stage kernel by kernel and the default bench line
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5w
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 > $O/pytest_gpu_full.txt 2>&1; tail -3 $O/pytest_gpu_full.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $O/smoke.txt
timeout 300 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -3 $O/dense_stage_kernels.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-300 $O/bench_default.json
