#!/bin/bash
set -o pipefail
O=gpurun_out/r4h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention or dropout" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for w in 512 0 512 0 256 1024; do
  MMFM_ATTN_WARM=$w MMFM_ATTN_WARM_BWD=$w timeout -k 10 120 python scripts/attn_bench.py 1024 0.4 20 bf16 1 2>&1 | grep attn_ | sed "s/^/warm=$w /"
done
