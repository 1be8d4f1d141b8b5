# round 6, call l: the search's wait for its slot's last replay on the HOST (sleeping event wait in the helper thread) instead of on its stream
mkdir -p gpurun_out/r6l
O=gpurun_out/r6l
( COOCC_SERVING_DIAG_SKIP=7 timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  COOCC_SLOT_WAIT=device timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 6 2 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 6 4 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 8 4 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 8 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 5 3 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 5 2 60 2>&1 | grep "serving alone"
  timeout 200 python tools/serving_probe.py 10 4 60 2>&1 | grep "serving alone"
) | tee $O/serving_probe_hostwait.txt
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d.get('window_ms_per_step'), d['also']['stress200_r101'].get('value'))"
