#!/bin/bash
# round 4: column-block split of the async-ring row-owner linears, fused mask 11 below 12,288 rows by default: whole suite, small-batch plans
set -o pipefail
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
run() { echo "== B=$2 $1"; env $1 timeout -k 10 200 python bench.py --batch $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-profile 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'])" || exit 1; }
run "MMFM_X=0" 16
run "MMFM_FUSED=0" 16
run "MMFM_FUSED=11 MMFM_ROWGEMM_NSPLIT=0" 16
run "MMFM_X=0" 8
run "MMFM_FUSED=0" 8
run "MMFM_X=0" 32
run "MMFM_FUSED=0" 32
run "MMFM_X=0" 48
run "MMFM_FUSED=0" 48
