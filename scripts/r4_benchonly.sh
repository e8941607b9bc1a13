#!/bin/bash
# the default bench line with the PMC traffic of the same sources attached (profiles/pmc_traffic.json must carry this tree's src_sha)
set -o pipefail
O=gpurun_out/final_r4; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_bf16_B1024_traffic.json 2> $O/bench2.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_bf16_B1024_traffic.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline'])"
