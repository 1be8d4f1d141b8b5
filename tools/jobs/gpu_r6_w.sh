#!/bin/bash
# Which piece of a split-f16 layer triggers the parked half-column mix's zero load beat?  (tools/debug/mix_trigger.py)
out=gpurun_out/r6w
mkdir -p $out
run() { echo "== $*" >> $out/mix_trigger.txt; env "$@" 2>&1 | grep -v Warning | tail -3 >> $out/mix_trigger.txt; }
P="python tools/debug/mix_trigger.py"
run X=1 $P none
run X=1 $P h2p
run X=1 $P chain
run X=1 $P f32
run X=1 $P wino_in
run X=1 $P wino_out
run X=1 $P to_h2
run X=1 $P gemm
for a in 1 2 4 8 16 7 15; do run COOCC_H2_ABLATE=$a $P gemm; done
cat $out/mix_trigger.txt
