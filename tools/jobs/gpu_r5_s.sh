#!/bin/bash
# k_gemm_h2n<4> on the G1 gather GEMMs of the dense stage: A/B inside the pipeline (alternating runs) and the dense stage kernel by kernel
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5v
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
for rep in 1 2 3 4 5; do
for g in 1 0; do
  COOCC_H2_NARROW=$g timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default narrow $g:', d['value'], d['ms_per_step'], d['window_ms_per_step'])" | tee -a $O/bench2.txt
done
done
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  rm -rf /tmp/gp
  COOCC_H2_NARROW=$g timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o gp -- python $R/tools/graph_probe.py > /dev/null 2>&1
  python $R/tools/graph_trace.py /tmp/gp/gp_kernel_trace.csv --seq > $O/dense_stage_narrow$g.txt 2>&1
  head -1 $O/dense_stage_narrow$g.txt; grep -E "k_gemm_h2n|k_gemm_h2w<true" $O/dense_stage_narrow$g.txt | head -4
done
