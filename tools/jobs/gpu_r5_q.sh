#!/bin/bash
# split-K target of the small-grid layers inside the serving pipeline (three dense graphs share the chip)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5q
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
for rep in 1 2 3; do
for t in 512 384 320 256 192; do
  v=$(COOCC_SPLITK_TARGET=$t timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "target $t: $v" | tee -a $O/splitk_target2.txt
done
done
for t in 512 256; do
  COOCC_SPLITK_TARGET=$t timeout 300 python tools/graph_probe.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/target $t: /" | tee -a $O/splitk_target2.txt
  COOCC_SPLITK_TARGET=$t timeout 300 python bench.py --config r101 --steps 30 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r101 target $t:', d['value'], d['ms_per_step'])" | tee -a $O/splitk_target2.txt
  COOCC_SPLITK_TARGET=$t timeout 300 python bench.py --config stress200 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stress200 target $t:', d['value'], d['ms_per_step'])" | tee -a $O/splitk_target2.txt
done
