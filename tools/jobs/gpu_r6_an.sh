R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r6; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --train --steps 10 --warmup 2 > $O/r6_bench_train.json 2>/dev/null
timeout 300 python $R/bench.py --train --steps 20 --warmup 3 --no-kernel-timing > $O/r6_bench_train_notimers.json 2>/dev/null
rm -rf /tmp/tt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tt -o s -- python $R/bench.py --train --steps 10 --warmup 2 > /tmp/tt_train.json 2>/dev/null
python - > $O/r6_train_kernels.txt 2>&1 < /dev/null <<PY
import csv, json
rows = list(csv.DictReader(open("/tmp/tt/s_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
d = json.load(open("/tmp/tt_train.json"))
print("bench.py --train --steps 10 --warmup 2 under rocprofv3 --kernel-trace --stats: %.2f ms/step wall (traced); kernel time %.1f ms over 12 steps + setup" % (d["ms_per_step"], tot / 1e6))
for r in rows[:60]:
    print("  %-78s calls %5s total %8.2f ms avg %8.1f us" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
head -1 $O/r6_train_kernels.txt; python -c "
import json
for f in ('r6_bench_train','r6_bench_train_notimers'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])"
