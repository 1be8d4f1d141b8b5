#!/bin/bash
# Synthetic matrix-core co-runners next to the parked half-column mix (tools/debug/mix_trigger2.py)
out=gpurun_out/r6x
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libmfma_co.so tools/proto/mfma_corunner.hip > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
P="python tools/debug/mix_trigger2.py"
for kind in 0 1 2 3 4 5; do
  for nv in 256 208 128; do
    $P $kind $nv 35076 2>&1 | grep synthetic >> $out/mix_trigger2.txt
  done
done
$P 0 256 0 2>&1 | grep synthetic >> $out/mix_trigger2.txt
$P 0 256 65536 2>&1 | grep synthetic >> $out/mix_trigger2.txt
$P 0 256 35076 512 512 2>&1 | grep synthetic >> $out/mix_trigger2.txt
$P 0 0 0 2>&1 | grep synthetic >> $out/mix_trigger2.txt
cat $out/mix_trigger2.txt
