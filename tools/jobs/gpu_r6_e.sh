# round 6, call e: (1) reserved CUs again with 4 masked queues instead of 18, (2) COOCC_H2_MIN_FLOPS sweep on the dense stage,
# (3) the swizzled k_fine2_h2, parity of both under the golden / fp64-anchored tests
mkdir -p gpurun_out/r6e
O=gpurun_out/r6e
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 200 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/pipeline_sweep2.txt
}
run "slots 6 streams 3 reserve 0" --slots 6 --streams 3
run "slots 6 streams 3 reserve 4 (3 masked dense + 1 masked fps)" --slots 6 --streams 3 --reserve-cus 4
run "slots 6 streams 3 reserve 8" --slots 6 --streams 3 --reserve-cus 8
run "slots 5 streams 3 reserve 8" --slots 5 --streams 3 --reserve-cus 8
run "slots 8 streams 4 reserve 8" --slots 8 --streams 4 --reserve-cus 8
for f in 1e9 3e8 1e8 5e7 0; do
  COOCC_H2_MIN_FLOPS=$f bash tools/dense_stage_kernels.sh $O/dense_stage_kernels_minflops_$f.txt
  echo "COOCC_H2_MIN_FLOPS=$f: $(head -1 $O/dense_stage_kernels_minflops_$f.txt)" | tee -a $O/minflops.txt
done
COOCC_H2_MIN_FLOPS=5e7 run "H2_MIN_FLOPS=5e7 slots 6 streams 3" --slots 6 --streams 3
COOCC_H2_MIN_FLOPS=5e7 timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_parity_full.py tests/test_gpu_graph.py -x -q > $O/pytest_minflops_5e7.txt 2>&1
tail -5 $O/pytest_minflops_5e7.txt
