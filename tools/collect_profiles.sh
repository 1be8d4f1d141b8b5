#!/bin/bash
# Round profile collection on the GPU box (writes under gpurun_out/prof_r1/).  Every rocprofv3 run is bounded by
# `timeout`; counter passes are separate from --stats / trace passes (MI355X guide).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --streams 1 --no-cpu-baseline"
timeout 300 python $R/bench.py --steps 20 --warmup 3 > $O/bench_streams1.json 2> $O/bench_streams1.err
timeout 300 python $R/bench.py --steps 30 --warmup 4 --streams 3 --no-cpu-baseline > $O/bench_streams3.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o b -- $B --steps 10 --warmup 3 --no-kernel-timing > /dev/null 2>&1
cp /tmp/p_stats/b_kernel_stats.csv $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o b -- $B --steps 3 --warmup 1 --no-kernel-timing > /dev/null 2>&1
done
python $R/tools/pmc_summary.py FETCH_SIZE=/tmp/p_FETCH_SIZE/b_counter_collection.csv WRITE_SIZE=/tmp/p_WRITE_SIZE/b_counter_collection.csv > $O/bench_pmc_hbm.txt 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_sq -o b -- $B --steps 3 --warmup 1 --no-kernel-timing > /dev/null 2>&1
python - > $O/bench_pmc_sq.txt 2>&1 < /dev/null <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open("/tmp/p_sq/b_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
print("%-46s %6s %10s %10s %10s %12s" % ("kernel", "calls", "MFMA_util", "wait_any", "wait_lds", "bank_conflict"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:16]:
    busy = max(c["SQ_BUSY_CU_CYCLES"], 1.0); wave = max(c["SQ_WAVE_CYCLES"], 1.0)
    print("%-46s %6d %10.3f %10.3f %10.3f %12.0f" % (k, calls[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy / 4.0 if False else c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy,
          c["SQ_WAIT_INST_ANY"] / wave, c["SQ_WAIT_INST_LDS"] / wave, c["SQ_LDS_BANK_CONFLICT"]))
PY
timeout 500 python $R/tools/kbench.py fps knn conv render > $O/kbench.txt 2>&1
python $R/tools/kstats.py $O/bench_kernel_stats.csv 14 < /dev/null
cat $O/bench_streams1.json | cut -c1-1500
head -12 $O/bench_pmc_hbm.txt; head -8 $O/bench_pmc_sq.txt
