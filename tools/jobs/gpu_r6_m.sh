# round 6, call m: paired-FPS searches without the side-queue wait; the gap table again (host-side slot wait); slots x streams
mkdir -p gpurun_out/r6m
O=gpurun_out/r6m
( for k in 0 1 2 4 6 7; do
    COOCC_SERVING_DIAG_SKIP=$k timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  done
  for sp in "8 4" "7 4" "6 4" "8 3" "10 4" "10 5" "12 4" "9 3"; do
    timeout 200 python tools/serving_probe.py $sp 60 2>&1 | grep "serving alone"
  done
) | tee $O/serving_probe_gap.txt
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_serving.py tests/test_gpu_knn.py tests/test_gpu_modules.py -x -q > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d.get('window_ms_per_step'), d['also']['stress200_r101'].get('value'))"
COOCC_BENCH_TIME_DENSE=0 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TIME_DENSE=0', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TIME_DENSE=1', d['value'], d['ms_per_step'])"
