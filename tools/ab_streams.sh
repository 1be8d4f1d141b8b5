#!/bin/bash
# Spread of the S = 1 / S = 2 pipelines over many processes on one box.  Output: gpurun_out/ab_streams.txt
mkdir -p gpurun_out
out=gpurun_out/ab_streams.txt
: > $out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1   # page the image in
for rep in 1 2 3 4; do
  for S in 1 2 2 2; do
    line=$(python bench.py --steps 100 --warmup 10 --streams $S --no-cpu-baseline 2>/dev/null | tail -1)
    echo "rep $rep streams $S: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "samples/s", d["ms_per_step"], "ms/step")')" >> $out
  done
done
cat $out
