#!/bin/bash
# B = 16 (the reference's batch): which plan is fastest when the step is bound by the fixed cost of its launches
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
run() { echo "== $1"; env $1 timeout -k 10 200 python bench.py --batch 16 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-profile 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'])" || exit 1; }
run "MMFM_X=0"
run "MMFM_FUSED=4"
run "MMFM_FUSED=12"
run "MMFM_FUSED=6"
run "MMFM_FUSED=15"
run "MMFM_FUSED=4 MMFM_MLP_BWD_SPLIT=0"
run "MMFM_X=0"
