#!/bin/bash
# Round-2 profile collection on the GPU box (writes under gpurun_out/prof_r2/; copy into profiles/ afterwards).  Every
# rocprofv3 run is bounded by `timeout`; counter passes are separate from --stats / trace passes (MI355X guide).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1      # clocks / caches warm before the recorded runs
timeout 400 python $R/bench.py > $O/r2_bench_default.json 2> $O/r2_bench_default.err                     # the driver's command (S = 2)
timeout 300 python $R/bench.py --steps 50 --warmup 3 --streams 1 --no-cpu-baseline > $O/r2_bench_streams1.json 2>/dev/null
timeout 300 python $R/bench.py --steps 50 --warmup 3 --streams 1 --no-cpu-baseline --kernel-table > $O/r2_bench_kernel_table.json 2> $O/r2_bench_kernel_table.txt
for S in 1 2; do
  rm -rf /tmp/p_stats$S
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats$S -o b -- python $R/bench.py --steps 10 --warmup 3 --streams $S --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  cp /tmp/p_stats$S/b_kernel_stats.csv $O/r2_bench_streams${S}_kernel_stats.csv
  cp /tmp/p_stats$S/b_kernel_trace.csv /tmp/trace_s$S.csv 2>/dev/null
done
B="python $R/bench.py --streams 1 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o b -- $B > /dev/null 2>&1
done
python $R/tools/pmc_summary.py FETCH_SIZE=/tmp/p_FETCH_SIZE/b_counter_collection.csv WRITE_SIZE=/tmp/p_WRITE_SIZE/b_counter_collection.csv > $O/r2_bench_pmc_hbm.txt 2>&1 < /dev/null
python $R/tools/make_traffic.py /tmp/p_FETCH_SIZE/b_counter_collection.csv /tmp/p_WRITE_SIZE/b_counter_collection.csv r50 > $O/r2_traffic.json 2> $O/r2_traffic.err
rm -rf /tmp/p_sq
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_sq -o b -- $B > /dev/null 2>&1
python - > $O/r2_bench_pmc_sq.txt 2>&1 < /dev/null <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open("/tmp/p_sq/b_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
print("%-46s %6s %10s %10s %10s %12s" % ("kernel", "calls", "MFMA_busy/CU_busy", "wait_any", "wait_lds", "bank_conflict"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:18]:
    busy = max(c["SQ_BUSY_CU_CYCLES"], 1.0); wave = max(c["SQ_WAVE_CYCLES"], 1.0)
    print("%-46s %6d %10.3f %10.3f %10.3f %12.0f" % (k, calls[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, c["SQ_WAIT_INST_ANY"] / wave, c["SQ_WAIT_INST_LDS"] / wave, c["SQ_LDS_BANK_CONFLICT"]))
PY
# other configurations + the reduced-precision path: median of three runs each (shared boxes)
bash $R/tools/record_configs.sh
timeout 600 python $R/tools/kbench.py fps knn conv render pool > $O/r2_kbench.txt 2>&1
timeout 300 python $R/tools/kbench.py convbf16 2>&1 | grep -v amdgpu.ids > $O/r2_kbench_bf16.txt
timeout 300 bash $R/tools/pmc_bf16.sh > $O/r2_pmc_bf16.txt 2>&1
python $R/tools/kstats.py $O/r2_bench_streams2_kernel_stats.csv 16 < /dev/null
cut -c1-900 $O/r2_bench_default.json
head -14 $O/r2_bench_pmc_hbm.txt; head -8 $O/r2_bench_pmc_sq.txt
