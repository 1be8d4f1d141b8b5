# round 6, call b: the co-runner defect's discriminators (VERDICT r5 "Next round" 5a), <= 10 GPU-minutes
#   1. the SGPR write-after-write probe (tools/proto/sgpr_waw_probe.hip): VALU carry-out -> s[N:N+1], then s_and_saveexec on the same pair
#   2. what the differing rows of the half-column mix hold (NaN prefill = skipped store | abs/max = overwritten data), VAR 0 and VAR 2
mkdir -p gpurun_out/r6b
/opt/rocm/bin/rocminfo | grep -i -m4 "xnack\|gfx950" > gpurun_out/r6b/rocminfo.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/sgpr_waw_probe tools/proto/sgpr_waw_probe.hip > gpurun_out/r6b/probe_build.txt 2>&1
timeout 300 /tmp/sgpr_waw_probe > gpurun_out/r6b/sgpr_waw_probe.txt 2>&1
timeout 300 python tools/debug/mix_values.py 0 2 > gpurun_out/r6b/mix_values.txt 2>&1
cat gpurun_out/r6b/rocminfo.txt gpurun_out/r6b/sgpr_waw_probe.txt gpurun_out/r6b/mix_values.txt
