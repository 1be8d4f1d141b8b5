# round 6, call j: device-scope events (streams.DeviceEvent, hipEventDisableSystemFence) in the serving loop / native search
mkdir -p gpurun_out/r6j
O=gpurun_out/r6j
( timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  COOCC_LIGHT_EVENTS=0 COOCC_SEARCH_COUNT_FENCE=1 timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  COOCC_SEARCH_COUNT_FENCE=1 timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 6 2 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 8 4 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 6 4 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 5 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 4 2 60 2>&1 | grep "serving alone"
) | tee $O/serving_probe_light_events.txt
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_serving.py tests/test_gpu_bench.py tests/test_gpu_knn.py tests/test_gpu_boundary.py -x -q > $O/pytest_light_events.txt 2>&1
tail -4 $O/pytest_light_events.txt
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d.get('window_ms_per_step'), d['also']['stress200_r101'].get('value'))"
timeout 300 python bench.py --api pipelined_test --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined_test api', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --api simple_test --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('simple_test api', d['value'], d['ms_per_step'])"
