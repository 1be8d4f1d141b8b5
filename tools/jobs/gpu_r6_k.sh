# round 6, call k: which part of the per-replay event costs the loop 20 %: the record, or a second stream waiting for it?
mkdir -p gpurun_out/r6k
O=gpurun_out/r6k
( for p in "" nowait nowait,recnowait nowait,dummywait nowait,noev noev recnowait; do
    COOCC_SERVING_DIAG_SKIP=7 COOCC_SERVING_PROBE=$p timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
  done
  COOCC_LIGHT_EVENTS=0 COOCC_SERVING_DIAG_SKIP=7 COOCC_SERVING_PROBE=nowait,recnowait timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone"
) | tee $O/serving_probe_event_parts.txt
