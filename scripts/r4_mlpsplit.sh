#!/bin/bash
# round 4: MLP-backward front-half kernel (dx == NULL) + rowgemm(ln_bwd): parity, then the step with / without it on the same box
set -o pipefail
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rowchain_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "mlp or long_exact" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for sp in 0 1 0 1; do
  MMFM_MLP_BWD_SPLIT=$sp timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>$O/bench_sp$sp.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SPLIT=$sp', d['ms_per_step'], d['kernel_breakdown_ms'])" || exit 1
done
