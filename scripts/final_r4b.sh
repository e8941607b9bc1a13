#!/bin/bash
# round 4, final sources: the whole GPU suite, then the measurement set of scripts/final_r4.sh on the same box
set -o pipefail
O=gpurun_out/final_r4; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
bash scripts/final_r4.sh
