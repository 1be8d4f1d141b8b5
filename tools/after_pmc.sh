#!/bin/bash
# does a rocprofv3 counter pass leave the box in a state that slows the next bench run?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
one() { python $R/bench.py --config openocc --steps 10 --warmup 2 --no-cpu-baseline --diag 2>/tmp/err.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; grep diag /tmp/err.txt | cut -c1-330; }
echo "== before"; one
echo "== uptime"; cat /proc/loadavg
rm -rf /tmp/p_x; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_x -o b -- python $R/bench.py --streams 1 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 > /dev/null 2>&1
echo "== right after a pmc pass"; cat /proc/loadavg; one
ps aux --sort=-%cpu | head -5 | cut -c1-150
sleep 20
echo "== 20 s later"; cat /proc/loadavg; one
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
