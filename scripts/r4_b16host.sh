#!/bin/bash
# B = 16: is the step host-bound?  host enqueue time vs step time, host cProfile, kernel trace (sum of kernel durations per step)
set -o pipefail
O=gpurun_out/r4g; mkdir -p $O
MMFM_SIDE_DW=0 timeout -k 10 200 python bench.py --batch 16 --steps 200 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-profile --host-profile $O/host_profile_b16.txt 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms; host enqueue', d['host_enqueue_ms_per_step'])" || exit 1
cd /tmp && export TMPDIR=/tmp
MMFM_SIDE_DW=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b16 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-profile > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4g/prof/**/b16_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('sum of kernel durations over the run (110 steps + setup): %.1f ms'%(tot/1e6))
for r in rows[:16]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
tail -2 $O/prof.log
