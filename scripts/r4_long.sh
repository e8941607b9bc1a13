#!/bin/bash
set -o pipefail
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -k "long_keepbit or attention_bf16_fwd_bwd or attention_fast" > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED)|passed|failed|^E  .*Error" $O/pytest.log | head -20
timeout -k 10 400 python scripts/config5_step.py 256 > $O/config5.log 2>&1; tail -25 $O/config5.log
