#!/bin/bash
# PMC look at the sparse encoder's row-table GEMMs (separate counter passes, no trace domains)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm
  timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pm -o p -- python $R/tools/kbench.py lidar > /dev/null 2>&1
  python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
try:
    rows = list(csv.DictReader(open("/tmp/pm/p_counter_collection.csv")))
except Exception as e:
    print("$tag: no counter file", e); rows = []
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    if "k_gemm_h2w" not in k and "k_conv<" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
for k, c in agg.items():
    print("$tag | %-40s calls %3d | " % (k, calls[k]) + "  ".join("%s %.4g" % (n, v / max(calls[k], 1)) for n, v in sorted(c.items())))
PY
done 2>&1 | tee $O/lidar_pmc.txt
