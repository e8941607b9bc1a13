#!/bin/bash
# B = 16 (the reference's batch): which plan is fastest when the step is dispatch-bound
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
run() { echo "== $1"; env $1 timeout -k 10 200 python bench.py --batch 16 --steps 60 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-profile 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'])"; }
run "MMFM_X=0"
run "MMFM_ATTN_KEEPBITS=0"
run "MMFM_FUSED=15"
run "MMFM_FUSED=15 MMFM_ATTN_KEEPBITS=0"
run "MMFM_FUSED=8"
run "MMFM_FUSED=3"
python scripts/step_launches.py 16 > $O/launches_b16.txt 2>&1; tail -5 $O/launches_b16.txt
