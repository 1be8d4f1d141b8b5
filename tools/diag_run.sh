#!/bin/bash
# host-issue diagnostics of the S = 1 / S = 2 pipelines + kernel stats of the OpenOccupancy configuration
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
out=$R/gpurun_out/diag.txt
: > $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2 3; do
  for S in 1 2; do
    echo "== rep $rep streams $S" >> $out
    python $R/bench.py --steps 100 --warmup 10 --streams $S --no-cpu-baseline --diag 2>> $out | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], "samples/s")' >> $out
  done
done
rm -rf /tmp/p_oo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_oo -o b -- python $R/bench.py --config openocc --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp /tmp/p_oo/b_kernel_stats.csv $R/gpurun_out/openocc_kernel_stats.csv
python $R/tools/kstats.py $R/gpurun_out/openocc_kernel_stats.csv 30 >> $out
cat $out
