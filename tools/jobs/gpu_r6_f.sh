# round 6, call f: forked branches inside the dense stage (core.Fork: render block, image branch, pyramid levels 1-3) + H2_MIN_FLOPS 5e7:
# parity / bit-equality tests, dense stage alone (wall per replay), dense graphs in flight, bench; the pipeline gap by left-out sub-stage
mkdir -p gpurun_out/r6f
O=gpurun_out/r6f
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_serving.py tests/test_gpu_corunner.py tests/test_gpu_modules.py tests/test_gpu_bench.py -x -q > $O/pytest_branches.txt 2>&1
tail -6 $O/pytest_branches.txt
bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt
head -1 $O/dense_stage_kernels.txt
COOCC_BRANCHES=0 bash tools/dense_stage_kernels.sh $O/dense_stage_kernels_branches0.txt
head -1 $O/dense_stage_kernels_branches0.txt
timeout 300 python tools/dense_concurrency.py 2>&1 | grep -v amdgpu > $O/dense_concurrency.txt; cat $O/dense_concurrency.txt
COOCC_BRANCHES=0 timeout 300 python tools/dense_concurrency.py 2>&1 | grep -v amdgpu > $O/dense_concurrency_branches0.txt; cat $O/dense_concurrency_branches0.txt
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/pipeline_gap.txt
}
run "branches 1, slots 6 streams 3" --slots 6 --streams 3
COOCC_BRANCHES=0 run "branches 0, slots 6 streams 3" --slots 6 --streams 3
run "branches 1, slots 6 streams 2" --slots 6 --streams 2
run "branches 1, slots 5 streams 2" --slots 5 --streams 2
run "branches 1, slots 8 streams 4" --slots 8 --streams 4
for k in 1 2 4 6 7; do
  COOCC_SERVING_DIAG_SKIP=$k run "branches 1, slots 6 streams 3, DIAG_SKIP=$k (1 copies, 2 pooling, 4 index search)" --slots 6 --streams 3
done
timeout 300 python bench.py --api simple_test --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('simple_test api', d['value'], d['ms_per_step'])" | tee -a $O/pipeline_gap.txt
