#!/bin/bash
# round 5: the ray-segment form of the fused lift (x) splat: parity tests, times alone (r50 / r101), kernel trace + HBM counters at r101
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -m gpu -k "pool or lift or splat" > $O/pytest_pool.txt 2>&1
tail -n 15 $O/pytest_pool.txt

timeout 900 python -m pytest tests/test_gpu_lidar.py -x -q -m gpu > $O/pytest_lidar.txt 2>&1
tail -n 25 $O/pytest_lidar.txt

for seg in 1 0; do
  echo "== COOCC_POOL_SEG=$seg" >> $O/kbench_pool.txt
  COOCC_POOL_SEG=$seg timeout 300 python tools/kbench.py pool 2>&1 | grep -v amdgpu.ids >> $O/kbench_pool.txt
done
cat $O/kbench_pool.txt
for which in r50 r101; do
( cd /tmp && rm -rf /tmp/pp && COOCC_POOLPROF=$which timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $R/tools/kbench.py poolprof > /tmp/pp.log 2>&1 )
echo "== $which" >> $O/poolprof_kernel_stats.txt
python - <<PY >> $O/poolprof_kernel_stats.txt 2>&1
import csv
rows = list(csv.DictReader(open("/tmp/pp/p_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_seg", "k_scan", "k_csr", "k_pool", "k_keys", "ncdhw", "camera_mats")):
        print("%-60s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
cat $O/poolprof_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && COOCC_POOLPROF=r101 timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/q_$c -o b -- python $R/tools/kbench.py poolprof > /dev/null 2>&1 )
done
python $R/tools/pmc_summary.py --only k_pool_sum_seg,k_seg_hist,k_csr_fill,k_scan_local FETCH_SIZE=/tmp/q_FETCH_SIZE/b_counter_collection.csv WRITE_SIZE=/tmp/q_WRITE_SIZE/b_counter_collection.csv > $O/poolprof_pmc_hbm.txt 2>&1 < /dev/null
cat $O/poolprof_pmc_hbm.txt
