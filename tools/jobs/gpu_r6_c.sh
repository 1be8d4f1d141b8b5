# round 6, call c: the per-stage co-runner guard (tests/test_gpu_corunner.py), the new ADVICE tests, and a same-box BASELINE of this
# round's starting tree: default bench line, dense stage kernel by kernel, dense graphs in flight 1..4
mkdir -p gpurun_out/r6c
O=gpurun_out/r6c
timeout 600 python -m pytest tests/test_gpu_corunner.py tests/test_gpu_serving.py::test_simple_test_falls_back_to_eager_when_the_library_refuses_the_captured_form tests/test_gpu_conv.py::test_sparse_tap_sum_raises_the_sticky_fault_word_on_a_corrupted_map tests/test_gpu_lidar.py -x -q -rxX > $O/pytest_corunner.txt 2>&1
tail -15 $O/pytest_corunner.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "import json; d=json.load(open('$O/bench_default.json')); print('bench', d['value'], d['ms_per_step'], d.get('windows'))"
bash tools/dense_stage_kernels.sh $O/dense_stage_kernels.txt
head -3 $O/dense_stage_kernels.txt
timeout 300 python tools/dense_concurrency.py > $O/dense_concurrency.txt 2>&1
cat $O/dense_concurrency.txt | tail -5
