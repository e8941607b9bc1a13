#!/bin/bash
# round 4: whole GPU suite, then the bench step with / without the keep-bit attention path on the same box
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for kb in 1 0 1 0; do
  MMFM_ATTN_KEEPBITS=$kb timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>$O/bench_kb$kb.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KEEPBITS=$kb', d['ms_per_step'], d['kernel_breakdown_ms'])"
done
