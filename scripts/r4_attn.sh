#!/bin/bash
# round 4: attention kernel parity + timing on one box
set -o pipefail
mkdir -p gpurun_out/r4a
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention or dropout" > gpurun_out/r4a/pytest_attn.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4a/pytest_attn.log
grep -E "^(FAILED|PASSED)|passed|failed" gpurun_out/r4a/pytest_attn.log | tail -30
rm -f gpurun_out/r4a/bench.txt
for kb in 1 0; do
  timeout -k 10 120 python scripts/attn_bench.py 1024 0.4 20 bf16 $kb >> gpurun_out/r4a/bench.txt 2>&1
done
MMFM_ATTN_BWD_FLAGS=0 timeout -k 10 120 python scripts/attn_bench.py 1024 0.4 20 bf16 1 >> gpurun_out/r4a/bench.txt 2>&1
timeout -k 10 120 python scripts/attn_bench.py 1024 0.0 20 bf16 1 >> gpurun_out/r4a/bench.txt 2>&1
grep attn_ gpurun_out/r4a/bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof -o attn --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/attn_bench.py 1024 0.4 10 bf16 1 > $GRAFT_REPO_ROOT/gpurun_out/r4a/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4a/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | head -8'
