#!/bin/bash
# One dense-stage graph replay, kernel by kernel, with nothing else on the GPU:  tools/dense_stage_kernels.sh OUT.txt [config]
# (rocprofv3 --kernel-trace over tools/graph_probe.py, summarised by tools/graph_trace.py --seq)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; CFG=${2:-r50}
T=$(mktemp -d /tmp/gp.XXXX)
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $T -o gp -- python $R/tools/graph_probe.py $CFG > $T/probe.txt 2>&1 )
python $R/tools/graph_trace.py $T/gp_kernel_trace.csv --seq > $OUT 2>&1
cat $T/probe.txt | tail -n 4 >> $OUT
rm -rf $T
