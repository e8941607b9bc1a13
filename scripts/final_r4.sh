#!/bin/bash
# round 4 measurement set on ONE box: bench line (default flags), rocprofv3 kernel stats of the same command, per-launch table, PMC traffic
set -o pipefail
O=gpurun_out/final_r4; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench_bf16_B1024.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_bf16_B1024.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['configs'])"
bash scripts/prof_bench.sh > $O/prof.log 2>&1; cp gpurun_out/prof/r04_kernel_stats.csv $O/ 2>/dev/null; ls gpurun_out/prof | head
timeout -k 10 300 python scripts/step_launches.py 1024 > $O/step_launches_B1024.txt 2>&1; tail -3 $O/step_launches_B1024.txt
bash scripts/pmc_traffic.sh > $O/pmc.log 2>&1; cp gpurun_out/pmc/pmc_traffic.json gpurun_out/pmc/pmc_traffic_detail.json $O/ 2>/dev/null; tail -3 $O/pmc.log
timeout -k 10 300 python bench.py --batch 256 --no-cpu-baseline --no-extra-legs > $O/bench_bf16_B256.json 2> $O/b256.err
timeout -k 10 300 python bench.py --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-extra-legs > $O/bench_bf16_B16.json 2> $O/b16.err
timeout -k 10 300 python bench.py --batch 256 --dtype fp32 --no-cpu-baseline --no-extra-legs > $O/bench_fp32_B256.json 2> $O/f256.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["ms_per_step"], "ms", d["value"])
    except Exception as e: print(f, "FAILED", e)
PY
