# round 6, call h: the search stage costs nothing (call g: 270.5 with it, 273.0 without); what then separates the serving loop (273)
# from the dense stage alone on 3 streams (316)?  Number of graph instances, the per-replay events, the timing events, the host loop
mkdir -p gpurun_out/r6h
O=gpurun_out/r6h
timeout 600 python tools/dense_concurrency.py 3:3 6:3 6:3:ev 4:4 8:4 6:2 4:2 2:2 12:3 2>&1 | grep -v amdgpu | tee $O/dense_concurrency_slots.txt
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/pipeline_gap2.txt
}
COOCC_SERVING_DIAG_SKIP=7 COOCC_BENCH_TIME_DENSE=0 run "DIAG_SKIP=7 TIME_DENSE=0 slots 6 streams 3" --slots 6 --streams 3
COOCC_BENCH_TIME_DENSE=0 run "TIME_DENSE=0 slots 6 streams 3" --slots 6 --streams 3
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7 slots 4 streams 3" --slots 4 --streams 3
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7 slots 3 streams 3" --slots 3 --streams 3
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7 slots 4 streams 4" --slots 4 --streams 4
COOCC_SERVING_DIAG_SKIP=7 run "DIAG_SKIP=7 slots 5 streams 4" --slots 5 --streams 4
run "slots 4 streams 3" --slots 4 --streams 3
run "slots 5 streams 3" --slots 5 --streams 3
run "slots 5 streams 4" --slots 5 --streams 4
run "slots 6 streams 4" --slots 6 --streams 4
run "slots 7 streams 4" --slots 7 --streams 4
