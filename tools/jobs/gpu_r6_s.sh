# round 6, call s: searches ahead (helper threads) 1 / 2 / 3 at N = 1 (the multi-rank default is 1); COOCC_SPLIT_C0; cgroup quota of the box
mkdir -p gpurun_out/r6s
O=gpurun_out/r6s
cat /sys/fs/cgroup/cpu.max > $O/cgroup.txt 2>&1; nproc >> $O/cgroup.txt
B="python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-kernel-timing"
run() {
  l=$1; shift
  v=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('window_ms_per_step'))")
  echo "$l: $v" | tee -a $O/ahead.txt
}
run "ahead 3 (default)"
run "ahead 1" --ahead 1
run "ahead 2" --ahead 2
run "ahead 1 slots 4" --ahead 1 --slots 4
COOCC_SPLIT_C0=1 run "COOCC_SPLIT_C0=1"
cat $O/cgroup.txt
