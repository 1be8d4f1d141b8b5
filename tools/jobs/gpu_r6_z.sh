#!/bin/bash
# round 6, call z: the two parked kernels rebuilt with COOCC_SCALAR_FP32 (no packed-fp32 op_sel forms) -- bit stability next to the
# strongest triggers, then what un-parking them is worth (A/B on one box)
O=gpurun_out/r6z
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libmfma_co.so tools/proto/mfma_corunner.hip > $O/build.log 2>&1
python tools/isa_lint.py > $O/isa_lint.txt 2>&1
for k in 0 3 4; do python tools/debug/mix_trigger2.py $k 256 35076 2>&1 | grep synthetic >> $O/stability.txt; done
for p in h2p gemm chain; do python tools/debug/mix_trigger.py $p 2>&1 | grep co-runner >> $O/stability.txt; done
COOCC_FINE2_IMG_INSIDE=1 timeout 300 python tools/debug/fine2_corunner.py 2>&1 | grep -v Warning | tail -12 >> $O/stability.txt
timeout 600 python -m pytest tests/test_gpu_corunner.py -q -rxX 2>&1 | tail -8 >> $O/stability.txt
cat $O/stability.txt
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'), 'dense', d['graph'].get('dense_stage_ms'))")
  echo "$l: $v" | tee -a $O/ab.txt
}
for rep in 1 2; do
  run "default (both parked)"
  COOCC_INTERP_COLUMN=3 run "mix half-column on"
  COOCC_FINE2_IMG_INSIDE=1 run "one-launch fine branch on"
  COOCC_INTERP_COLUMN=3 COOCC_FINE2_IMG_INSIDE=1 run "both on"
done
