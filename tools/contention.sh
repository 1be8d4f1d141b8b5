#!/bin/bash
# bench.py (the driver's command) under host-CPU contention: N busy-loop processes beside it (the boxes are shared; load averages
# of 40-60 are common).  Arms: one sample in flight, two, and the default (chosen from untimed bursts).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
one() { python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["samples_in_flight"], d["value"], d.get("stream_probe", {}).get("samples_per_s", ""))'; }
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
for hogs in 0 16 24; do
  pids=""
  for i in $(seq 1 $hogs); do python -c 'while True: pass' & pids="$pids $!"; done
  sleep 1
  echo "== $hogs busy processes, loadavg $(cut -d" " -f1-3 /proc/loadavg)"
  for rep in 1 2 3; do for args in "--streams 1" "--streams 2" ""; do echo -n "${args:-auto}: "; one $args; done; done
  for p in $pids; do kill $p; done
  wait 2>/dev/null
done
