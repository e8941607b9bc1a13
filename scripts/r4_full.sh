#!/bin/bash
# round 4: whole GPU suite + the bench line (default flags) on one box
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 500 python bench.py > $O/bench_bf16_B1024.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench_bf16_B1024.json; echo
