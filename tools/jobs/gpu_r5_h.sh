#!/bin/bash
# round 5: where does k_pool_sum_seg's time go?  (timing experiments; results of the ablated runs are wrong by construction)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
export TMPDIR=/tmp
for which in r50 r101; do
for env in "A=0" "COOCC_POOL_ABLATE=1" "COOCC_POOL_ABLATE=2" "COOCC_POOL_LONG_BLOCKS=8" "COOCC_POOL_G16=0" "COOCC_POOL_G16=0 COOCC_POOL_ABLATE=1"; do
( cd /tmp && rm -rf /tmp/pp && env $env COOCC_POOLPROF=$which timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/tools/kbench.py poolprof > /tmp/pp.log 2>&1 )
python - <<PY >> $O/ablate.txt 2>&1
import csv
rows = list(csv.DictReader(open("/tmp/pp/p_kernel_stats.csv")))
for r in rows:
    if "k_pool_sum_seg" in r["Name"]:
        print("%-5s %-40s k_pool_sum_seg avg %8.1f us" % ("$which", "$env", float(r["AverageNs"]) / 1e3))
PY
done
done
cat $O/ablate.txt
