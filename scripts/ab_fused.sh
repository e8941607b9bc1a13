# same-box comparison of the fused-group mask (MMFM_FUSED) on the whole bench step
for cfg in 10 14 15 11 10; do
  MMFM_FUSED=$cfg timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSED=$cfg', d['ms_per_step'], d['kernel_breakdown_ms'])"
done
