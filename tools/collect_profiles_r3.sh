#!/bin/bash
# Round-3 profile collection on the GPU box (writes under gpurun_out/prof_r3/; copy into profiles/ afterwards).  Every
# rocprofv3 run is bounded by `timeout`; counter passes are separate from the --kernel-trace --stats passes (MI355X guide).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1      # clocks / caches warm before the recorded runs
timeout 400 python $R/bench.py > $O/r3_bench_default.json 2> $O/r3_bench_default.err        # the driver's command
timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --kernel-table > $O/r3_bench_kernel_table.json 2> $O/r3_bench_kernel_table.txt
timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --graph 0 > $O/r3_bench_eager.json 2>/dev/null
COOCC_CONV_ENGINE=f32 timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/r3_bench_engine_f32.json 2>/dev/null
# the default command under the kernel trace (graph launches are traced kernel by kernel)
rm -rf /tmp/p_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp /tmp/p_stats/b_kernel_stats.csv $O/r3_bench_kernel_stats.csv
# one dense-stage graph replay, kernel by kernel (nothing else on the GPU)
rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $R/tools/graph_probe.py > $O/r3_graph_probe.txt 2>&1
python $R/tools/graph_trace.py /tmp/gp/gp_kernel_trace.csv > $O/r3_dense_stage_kernels.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o b -- $B > /dev/null 2>&1
done
python $R/tools/pmc_summary.py FETCH_SIZE=/tmp/p_FETCH_SIZE/b_counter_collection.csv WRITE_SIZE=/tmp/p_WRITE_SIZE/b_counter_collection.csv > $O/r3_bench_pmc_hbm.txt 2>&1 < /dev/null
python $R/tools/make_traffic.py /tmp/p_FETCH_SIZE/b_counter_collection.csv /tmp/p_WRITE_SIZE/b_counter_collection.csv r50 > $O/r3_traffic.json 2> $O/r3_traffic.err
rm -rf /tmp/p_sq
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_sq -o b -- $B > /dev/null 2>&1
python - > $O/r3_bench_pmc_sq.txt 2>&1 < /dev/null <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open("/tmp/p_sq/b_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
print("%-46s %6s %10s %10s %10s %10s %12s" % ("kernel", "calls", "MFMA_busy/CU_busy", "wait_any", "wait_inst", "wait_lds", "bank_conflict"))
for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:18]:
    busy = max(c["SQ_BUSY_CU_CYCLES"], 1.0); wave = max(c["SQ_WAVE_CYCLES"], 1.0)
    print("%-46s %6d %10.3f %10.3f %10.3f %10.3f %12.0f" % (k, calls[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, c["SQ_WAIT_ANY"] / wave, c["SQ_WAIT_INST_ANY"] / wave, c["SQ_WAIT_INST_LDS"] / wave, c["SQ_LDS_BANK_CONFLICT"]))
PY
timeout 300 python $R/tools/h2_check.py acc time direct general 2>&1 | grep -v amdgpu.ids > $O/r3_h2_check.txt
# the two stages alone: dense-stage ceiling at 1..4 graphs in flight, the prefetch stage (pooling + search) kernel by kernel
timeout 300 python $R/tools/dense_concurrency.py 2>&1 | grep -v amdgpu.ids > $O/r3_dense_concurrency.txt
rm -rf /tmp/sp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $R/tools/search_probe.py > $O/r3_search_stage.txt 2>&1
python - >> $O/r3_search_stage.txt 2>&1 < /dev/null <<PY
import csv
rows = list(csv.DictReader(open("/tmp/sp/s_kernel_stats.csv")))
tot = 0.0
print("per sample (23 calls traced), kernel trace of tools/search_probe.py:")
for r in rows[:32]:
    per = float(r["TotalDurationNs"]) / 23 / 1e3
    tot += per
    print("%-62s x%5.1f  %8.1f us/sample  avg %7.1f us" % (r["Name"][:62], int(r["Calls"]) / 23, per, float(r["AverageNs"]) / 1e3))
print("sum %.1f us per sample" % tot)
PY
timeout 300 python $R/tools/kbench.py fps fpsdbg pool 2>&1 | grep -v amdgpu.ids > $O/r3_kbench_search.txt
cut -c1-1200 $O/r3_bench_default.json
head -30 $O/r3_dense_stage_kernels.txt; head -12 $O/r3_bench_pmc_hbm.txt; head -8 $O/r3_bench_pmc_sq.txt
