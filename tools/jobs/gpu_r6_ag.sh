#!/bin/bash
# NOTE: COOCC_IMAGE_BRANCH_AHEAD existed for this call only (profiles/r6_image_branch_ahead.txt).
# round 6, call ag: the frame-only part of the fine branch made in the search stage (COOCC_IMAGE_BRANCH_AHEAD, default 1): tests, then A/B
O=gpurun_out/r6ag
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_serving.py tests/test_gpu_bench.py -x -q 2>&1 | tail -5 > $O/tests.txt
cat $O/tests.txt
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'), 'dense', d['graph'].get('dense_stage_ms'))")
  echo "$l: $v" | tee -a $O/ab.txt
}
for rep in 1 2 3; do
  COOCC_IMAGE_BRANCH_AHEAD=0 run "image branch inside the dense stage (round-6 first half)"
  run "image branch in the search stage (default)"
done
for i in 1 2; do COOCC_IMAGE_BRANCH_AHEAD=0 python bench.py --steps 20 --warmup 5 --also none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5, inside:', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; python bench.py --steps 20 --warmup 5 --also none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5, ahead:', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done
