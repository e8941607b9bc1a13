#!/bin/bash
# round 4: whole GPU suite + the bench line (default flags) on one box
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_short.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_short.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_breakdown_ms'])"
