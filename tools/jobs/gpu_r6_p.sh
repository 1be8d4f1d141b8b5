# round 6, call p: 16-byte paired H2 stores (store_h2_pair) in the Winograd transforms and the resampling twins
mkdir -p gpurun_out/r6p
O=gpurun_out/r6p
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_h2_engine.py tests/test_gpu_modules.py tests/test_gpu_graph.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt; head -12 $O/dense_stage_kernels.txt
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['window_ms_per_step'])"
