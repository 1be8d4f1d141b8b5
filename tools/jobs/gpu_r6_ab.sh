#!/bin/bash
# NOTE: COOCC_FPS_LDS_KB was an experiment of this call only; the knob was removed afterwards (profiles/r6_fps_lds_hog.txt).
# round 6, call ab: the FPS chain's workgroup asks for LDS it does not use, so that no GEMM tile shares its compute unit (COOCC_FPS_LDS_KB)
O=gpurun_out/r6ab
mkdir -p $O
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'), 'dense', d['graph'].get('dense_stage_ms'))")
  echo "$l: $v" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run "default"
  COOCC_FPS_LDS_KB=100 run "FPS workgroup + 100 KB LDS"
  COOCC_FPS_LDS_KB=130 run "FPS workgroup + 130 KB LDS"
  COOCC_FPS_LDS_KB=150 run "FPS workgroup + 150 KB LDS"
done
python tools/kbench.py fps 2>&1 | grep -v amdgpu | head -4 >> $O/ab.txt
COOCC_FPS_LDS_KB=130 python tools/kbench.py fps 2>&1 | grep -v amdgpu | head -4 >> $O/ab.txt
for i in 1 2; do python bench.py --steps 20 --warmup 5 --also none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5:', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done
for i in 1 2; do COOCC_FPS_LDS_KB=130 python bench.py --steps 20 --warmup 5 --also none --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5, 130 KB:', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt; done
