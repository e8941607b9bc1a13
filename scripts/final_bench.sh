# the round's measurement set: GPU tests, smoke, bench at the three batch sizes SURVEY.md §8d asks for, fp32 parity mode
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench_bf16_B1024.json 2> $O/bench_bf16_B1024.err; tail -c 300 $O/bench_bf16_B1024.json; echo
timeout -k 10 300 python bench.py --batch 256 --no-cpu-baseline > $O/bench_bf16_B256.json 2> $O/b256.err
timeout -k 10 300 python bench.py --batch 16 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_bf16_B16.json 2> $O/b16.err
timeout -k 10 300 python bench.py --batch 256 --dtype fp32 --no-cpu-baseline > $O/bench_fp32_B256.json 2> $O/f256.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "ms", d["value"], d["unit"], "roofline", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
