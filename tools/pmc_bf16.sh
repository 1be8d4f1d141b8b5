#!/bin/bash
# SQ counters of the bf16 convolution kernels at the OpenOccupancy shapes (tools/kbench.py convbf16)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_bf
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_bf -o b -- python $R/tools/kbench.py convbf16 > /dev/null 2>&1
python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open("/tmp/p_bf/b_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
print("%-42s %6s %12s %10s %10s %14s %12s" % ("kernel", "calls", "MFMA_busy/CU", "parked", "issue_stall", "bank_conflict", "active_inst"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:8]:
    busy = max(c["SQ_BUSY_CU_CYCLES"], 1.0); wave = max(c["SQ_WAVE_CYCLES"], 1.0)
    print("%-42s %6d %12.3f %10.3f %10.3f %14.3f %12.3f" % (k, calls[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, c["SQ_WAIT_ANY"] / wave, c["SQ_WAIT_INST_ANY"] / wave, c["SQ_LDS_BANK_CONFLICT"] / busy, c["SQ_ACTIVE_INST_ANY"] / wave))
PY
