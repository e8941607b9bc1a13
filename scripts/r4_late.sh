#!/bin/bash
# round 4: small-batch plans - LayerNorm-fed linears' slab reductions join the segment's batched reduction: parity (model + DDP), then B = 16 / 32
set -o pipefail
O=gpurun_out/r4q; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py tests/test_entry_script_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
run() { echo "== B=$2 $1"; env $1 timeout -k 10 200 python bench.py --batch $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --no-kernel-profile 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms', d['value'], d['final_loss'])" || exit 1; }
run "MMFM_X=0" 16
run "MMFM_X=0" 32
run "MMFM_X=0" 8
