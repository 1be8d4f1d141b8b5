#!/bin/bash
# Round-6 profile collection on the GPU box (writes under gpurun_out/prof_r6/; copy into profiles/ afterwards).  Every rocprofv3
# run is bounded by `timeout`; counter passes are separate from the --kernel-trace --stats passes (MI355X guide).
set -u
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r6
mkdir -p $O
cd $R
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1      # clocks / caches warm before the recorded runs
timeout 400 python $R/bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err        # the driver's command
timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --kernel-table > $O/r6_bench_kernel_table.json 2> $O/r6_bench_kernel_table.txt
timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing --api simple_test > $O/r6_bench_api_simple_test.json 2>/dev/null
timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-kernel-timing --api pipelined_test > $O/r6_bench_api_pipelined_test.json 2>/dev/null
timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --graph 0 > $O/r6_bench_eager.json 2>/dev/null
COOCC_CONV_ENGINE=f32 timeout 300 python $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/r6_bench_engine_f32.json 2>/dev/null
timeout 300 python $R/bench.py --train --steps 10 --warmup 2 > $O/r6_bench_train.json 2> $O/r6_bench_train.err
COOCC_TRAIN_H2=0 timeout 300 python $R/bench.py --train --steps 10 --warmup 2 > $O/r6_bench_train_f32.json 2>/dev/null
timeout 300 python $R/bench.py --train --steps 10 --warmup 2 --train-prefetch 0 > $O/r6_bench_train_noprefetch.json 2>/dev/null
COOCC_TRAIN_H2_WGRAD=0 timeout 300 python $R/bench.py --train --steps 10 --warmup 2 > $O/r6_bench_train_wgrad_f32.json 2>/dev/null
COOCC_TRAIN_H2_DGRAD=0 timeout 300 python $R/bench.py --train --steps 10 --warmup 2 > $O/r6_bench_train_dgrad_f32.json 2>/dev/null
timeout 300 python $R/bench.py --train --steps 20 --warmup 3 --no-kernel-timing > $O/r6_bench_train_notimers.json 2>/dev/null
timeout 400 python $R/bench.py --with-lidar --steps 40 --warmup 3 > $O/r6_bench_with_lidar.json 2>/dev/null
rm -rf /tmp/tt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o s -- python $R/bench.py --train --steps 10 --warmup 2 > /tmp/tt_train.json 2>/dev/null
python - > $O/r6_train_kernels.txt 2>&1 < /dev/null <<PY
import csv, json
rows = list(csv.DictReader(open("/tmp/tt/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
d = json.load(open("/tmp/tt_train.json"))
print("bench.py --train --steps 10 --warmup 2 under rocprofv3 --kernel-trace --stats: %.2f ms/step wall (traced); kernel time %.1f ms over 12 steps + setup" % (d["ms_per_step"], tot / 1e6))
for r in rows[:60]:
    print("  %-78s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
( echo "tools/kbench.py lidar (280 k points -> 120 k voxels -> SparseLiDAREnc8x), MI355X, round 6"; echo; echo "== COOCC_LIDAR_H2=1 (default: rule-book GEMMs with Cin % 32 == 0 on the split-f16 engine)"; COOCC_LIDAR_H2=1 timeout 300 python $R/tools/kbench.py lidar 2>&1 | grep -v amdgpu.ids; echo; echo "== COOCC_LIDAR_H2=0 (fp32-MFMA row-table kernels, rounds 1-3)"; COOCC_LIDAR_H2=0 timeout 300 python $R/tools/kbench.py lidar 2>&1 | grep -v amdgpu.ids ) > $O/r6_kbench_lidar.txt
timeout 300 python $R/tools/serving_trace.py r50 36 6 3 0 2>&1 | grep -v amdgpu.ids > $O/r6_serving_trace.txt
for cfg in r101 stress200 stress200_r101; do
  timeout 400 python $R/bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline > $O/r6_bench_$cfg.json 2> $O/r6_bench_$cfg.err
done
timeout 400 python $R/bench.py --config openocc --steps 20 --warmup 3 --no-cpu-baseline > $O/r6_bench_openocc_f32.json 2>/dev/null
timeout 400 python $R/bench.py --config openocc --dtype f16 --steps 20 --warmup 3 --no-cpu-baseline > $O/r6_bench_openocc_f16.json 2>/dev/null
# slots x dense streams x searches ahead
for sp in "6 3 0" "6 2 0" "6 4 0" "8 3 0"; do
  set -- $sp
  v=$(timeout 200 python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing --slots $1 --streams $2 --ahead $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "slots $1 dense-streams $2 ahead $3: $v" >> $O/r6_pipeline_sweep.txt
done
# the default command under the kernel trace (graph launches are traced kernel by kernel)
rm -rf /tmp/p_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o b -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
cp /tmp/p_stats/b_kernel_stats.csv $O/r6_bench_kernel_stats.csv
# one dense-stage graph replay, kernel by kernel (nothing else on the GPU)
rm -rf /tmp/gp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $R/tools/graph_probe.py > $O/r6_graph_probe.txt 2>&1
python $R/tools/graph_trace.py /tmp/gp/gp_kernel_trace.csv --seq > $O/r6_dense_stage_kernels.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o b -- $B > /dev/null 2>&1
done
python $R/tools/pmc_summary.py FETCH_SIZE=/tmp/p_FETCH_SIZE/b_counter_collection.csv WRITE_SIZE=/tmp/p_WRITE_SIZE/b_counter_collection.csv > $O/r6_bench_pmc_hbm.txt 2>&1 < /dev/null
python $R/tools/make_traffic.py /tmp/p_FETCH_SIZE/b_counter_collection.csv /tmp/p_WRITE_SIZE/b_counter_collection.csv r50 > $O/r6_traffic_r50.json 2> $O/r6_traffic.err
# the r101 render pair's traffic (roofline_render_r101 is measured at that size)
BR="python $R/bench.py --config r101 --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/q_$c
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/q_$c -o b -- $BR > /dev/null 2>&1
done
python $R/tools/make_traffic.py /tmp/q_FETCH_SIZE/b_counter_collection.csv /tmp/q_WRITE_SIZE/b_counter_collection.csv r101 > $O/r6_traffic_r101.json 2>> $O/r6_traffic.err
python - > $O/r6_traffic.json <<PY
import json
a = json.load(open("$O/r6_traffic_r50.json")); b = json.load(open("$O/r6_traffic_r101.json"))
for k, v in b.items():
    if isinstance(v, dict) and "r101" in v:
        a.setdefault(k, {}).update(r101=v["r101"])
json.dump(a, open("/dev/stdout", "w"), indent=1, sort_keys=True)
PY
python $R/tools/pmc_summary.py --only k_render,k_upsample_maps,k_pool_sum_seg,k_seg_hist,k_csr_fill FETCH_SIZE=/tmp/q_FETCH_SIZE/b_counter_collection.csv WRITE_SIZE=/tmp/q_WRITE_SIZE/b_counter_collection.csv > $O/r6_bench_r101_pmc_render_pool.txt 2>&1 < /dev/null
rm -rf /tmp/p_sq
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_sq -o b -- $B > /dev/null 2>&1
python - > $O/r6_bench_pmc_sq.txt 2>&1 < /dev/null <<PY
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
timeout 300 python $R/tools/dense_concurrency.py 2>&1 | grep -v amdgpu.ids > $O/r6_dense_concurrency.txt
rm -rf /tmp/sp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $R/tools/search_probe.py > $O/r6_search_stage.txt 2>&1
python - >> $O/r6_search_stage.txt 2>&1 < /dev/null <<PY
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
timeout 300 python $R/tools/kbench.py fps pool 2>&1 | grep -v amdgpu.ids > $O/r6_kbench_search.txt
cp $O/r6_kbench_search.txt $O/r6_kbench_fps.txt
timeout 300 python $R/tools/kbench.py trainfull 2>&1 | grep -v amdgpu.ids > $O/r6_kbench_train.txt
cut -c1-1500 $O/r6_bench_default.json
for f in $O/r6_bench_*.json; do python - <<PY
import json
try:
    d = json.load(open("$f"))
    rr = d.get("roofline_render_r101") or {}
    print("$f".split("/")[-1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline_pool") or {}).get("avg_ms_per_step"), rr.get("frac"), (rr.get("geometry_in_kernel") or {}).get("frac"))
except Exception as e:
    print("$f".split("/")[-1], "FAILED", e)
PY
done
cat $O/r6_pipeline_sweep.txt
head -n 30 $O/r6_dense_stage_kernels.txt; head -n 12 $O/r6_bench_pmc_hbm.txt; head -n 8 $O/r6_bench_pmc_sq.txt; cat $O/r6_kbench_search.txt | tail -n 4; head -n 16 $O/r6_kbench_train.txt
