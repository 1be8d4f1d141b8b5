#!/bin/bash
# pooling sums: XCD-sector voxel order (COOCC_POOL_XCD=1) against the plain order -- time, parity, HBM fetch at r101 and r50
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  COOCC_POOL_XCD=$x timeout 300 python $R/tools/kbench.py pool 2>&1 | grep "^pool" > $O/kbench_pool_xcd$x.txt; cat $O/kbench_pool_xcd$x.txt | cut -c1-150
  for w in r50 r101; do
    rm -rf /tmp/pp
    COOCC_POOL_XCD=$x COOCC_POOLPROF=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o s -- python $R/tools/kbench.py poolprof > /dev/null 2>&1
    echo "== xcd=$x $w" >> $O/stats.txt
    python - >> $O/stats.txt <<PY
import csv
for r in list(csv.DictReader(open("/tmp/pp/s_kernel_stats.csv")))[:6]:
    print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pq
      COOCC_POOL_XCD=$x COOCC_POOLPROF=$w timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pq -o b -- python $R/tools/kbench.py poolprof > /dev/null 2>&1
      echo "== xcd=$x $w $c" >> $O/pmc.txt
      python $R/tools/pmc_summary.py --only k_pool_sum_seg,k_seg_hist,k_csr_fill $c=/tmp/pq/b_counter_collection.csv >> $O/pmc.txt 2>&1 < /dev/null
    done
  done
done
cat $O/stats.txt | grep -E "==|k_pool_sum_seg"
grep -E "==|k_pool_sum_seg" $O/pmc.txt
cd $R
COOCC_POOL_XCD=1 timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pool or lift or splat" > $O/pytest_xcd1.txt 2>&1; tail -2 $O/pytest_xcd1.txt
