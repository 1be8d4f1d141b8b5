#!/bin/bash
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in openocc stress200; do
timeout 200 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
echo "$cfg rc=$? $(grep -o '"value": [0-9.]*' /tmp/b.json | head -1)"
done
