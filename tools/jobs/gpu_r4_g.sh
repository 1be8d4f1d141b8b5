#!/bin/bash
# occupancy of the pipelined step (tools/pipeline_gaps.py over a kernel trace of the default bench line)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -o s -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench_traced.err
python $R/tools/pipeline_gaps.py /tmp/pg/s_kernel_trace.csv 60 $O/timeline_10ms.txt | tee $O/pipeline_gaps.txt
head -n 2 /tmp/pg/s_kernel_trace.csv > $O/trace_head.csv
