#!/bin/bash
# package power and shader clock while bench.py keeps the GPU busy (is the dense stage running into the power limit?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --steps 1500 --warmup 10 --no-cpu-baseline --no-kernel-timing > /tmp/pb.json 2>/dev/null &
sleep 14
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Power (W)\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
wait
cut -c78-110 /tmp/pb.json
