#!/bin/bash
# packed-fp32 op_sel forms next to gfx950's 128-bit-operand MFMAs (tools/proto/pk_opsel_probe.hip)
out=gpurun_out/r6y
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_opsel_probe tools/proto/pk_opsel_probe.hip > $out/build.log 2>&1 || { tail -5 $out/build.log; exit 1; }
timeout 300 /tmp/pk_opsel_probe > $out/pk_opsel_probe.txt 2>&1
cat $out/pk_opsel_probe.txt
