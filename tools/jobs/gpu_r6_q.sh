# round 6, call q: slots x dense streams x searches ahead on the driver command's loop after the cross-queue-wait fixes; other configs
mkdir -p gpurun_out/r6q
O=gpurun_out/r6q
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/pipeline_sweep.txt
}
run "slots 6 streams 3" --slots 6 --streams 3
run "slots 8 streams 4" --slots 8 --streams 4
run "slots 7 streams 4" --slots 7 --streams 4
run "slots 7 streams 3" --slots 7 --streams 3
run "slots 8 streams 4 ahead 3" --slots 8 --streams 4 --ahead 3
run "slots 9 streams 4" --slots 9 --streams 4
run "slots 10 streams 5" --slots 10 --streams 5
run "slots 5 streams 3" --slots 5 --streams 3
run "r101 slots 6 streams 3" --config r101
run "r101 slots 8 streams 4" --config r101 --slots 8 --streams 4
run "api pipelined_test" --api pipelined_test
run "api simple_test" --api simple_test
run "with-lidar" --with-lidar
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing"
run "openocc f32" --config openocc
run "openocc f16" --config openocc --dtype f16
run "stress200_r101" --config stress200_r101
run "stress200_r101 slots 8 streams 4" --config stress200_r101 --slots 8 --streams 4
