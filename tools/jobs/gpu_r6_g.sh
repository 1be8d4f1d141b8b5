# round 6, call g: the pipeline gap by left-out sub-stage (COOCC_SERVING_DIAG_SKIP) on the shipped configuration (branches off), the
# new bench tests (--also, 8 ranks on one GPU), and the default bench line
mkdir -p gpurun_out/r6g
O=gpurun_out/r6g
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/pipeline_gap.txt
}
run "full pipeline (slots 6, dense streams 3)" --slots 6 --streams 3
for k in 1 2 4 6 7; do
  COOCC_SERVING_DIAG_SKIP=$k run "DIAG_SKIP=$k (1 input copies, 2 pooling, 4 index search left out)" --slots 6 --streams 3
done
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7, dense streams 4 slots 6" --slots 6 --streams 4
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7, dense streams 4 slots 8" --slots 8 --streams 4
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7, dense streams 2 slots 6" --slots 6 --streams 2
timeout 300 python tools/dense_concurrency.py 2>&1 | grep -v amdgpu | tee $O/dense_concurrency.txt
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q -k "also or eight or default_line" > $O/pytest_bench_new.txt 2>&1; tail -4 $O/pytest_bench_new.txt
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d.get('window_ms_per_step'), d.get('also'))"
