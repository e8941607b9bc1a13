#!/bin/bash
# round 4: row-owner forward-type linears on the asynchronous LDS-DMA ring (MMFM_ROWGEMM_RING=1) vs the register-staged ring: parity, step A/B
set -o pipefail
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rowchain_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for v in 1 3 1 3; do
  MMFM_ROWGEMM_RING=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>$O/bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RING=$v', d['ms_per_step'], d['kernel_breakdown_ms'])" || exit 1
done
