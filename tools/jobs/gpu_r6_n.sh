# round 6, call n: the multi-rank bench tests with dist.GatherThread; what of the index search costs the loop (COOCC_SEARCH_DIAG_SKIP)
mkdir -p gpurun_out/r6n
O=gpurun_out/r6n
timeout 1500 python -m pytest tests/test_gpu_bench.py -x -q --durations=8 > $O/pytest_bench.txt 2>&1; tail -14 $O/pytest_bench.txt
( for k in 0 1 2 3 4 7; do
    COOCC_SEARCH_DIAG_SKIP=$k timeout 200 python tools/serving_probe.py 6 3 60 2>&1 | grep "serving alone" | sed "s/^/SEARCH_DIAG_SKIP=$k  /"
  done ) | tee $O/serving_probe_search_parts.txt
