# round 6, call d: where the pipeline loses against the dense stage alone -- slots x dense streams x searches ahead, with and without
# the FPS chains on reserved CUs (ServingPipeline(reserve_cus=), new this round)
mkdir -p gpurun_out/r6d
O=gpurun_out/r6d
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {   # label, args...
  l=$1; shift
  v=$(timeout 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/pipeline_sweep.txt
}
run "slots 6 streams 3 ahead 0 reserve 0" --slots 6 --streams 3
run "slots 6 streams 3 ahead 0 reserve 2" --slots 6 --streams 3 --reserve-cus 2
run "slots 6 streams 3 ahead 0 reserve 4" --slots 6 --streams 3 --reserve-cus 4
run "slots 6 streams 3 ahead 0 reserve 8" --slots 6 --streams 3 --reserve-cus 8
run "slots 8 streams 4 ahead 0 reserve 0" --slots 8 --streams 4
run "slots 8 streams 4 ahead 0 reserve 4" --slots 8 --streams 4 --reserve-cus 4
run "slots 7 streams 4 ahead 0 reserve 4" --slots 7 --streams 4 --reserve-cus 4
run "slots 6 streams 4 ahead 0 reserve 4" --slots 6 --streams 4 --reserve-cus 4
run "slots 10 streams 4 ahead 0 reserve 4" --slots 10 --streams 4 --reserve-cus 4
run "slots 5 streams 2 ahead 0 reserve 4" --slots 5 --streams 2 --reserve-cus 4
run "slots 4 streams 2 ahead 0 reserve 4" --slots 4 --streams 2 --reserve-cus 4
run "slots 6 streams 2 ahead 0 reserve 4" --slots 6 --streams 2 --reserve-cus 4
run "slots 8 streams 4 ahead 2 reserve 4" --slots 8 --streams 4 --ahead 2 --reserve-cus 4
run "slots 10 streams 5 ahead 0 reserve 4" --slots 10 --streams 5 --reserve-cus 4
timeout 300 python tools/serving_trace.py r50 36 6 3 0 2>&1 | grep -v amdgpu.ids > $O/serving_trace_6_3.txt
head -12 $O/serving_trace_6_3.txt
