# same-box comparison of the fused-group mask (MMFM_FUSED) and the MLP kernel variant on the whole bench step
for cfg in "10 1" "14 1" "14 0" "15 1" "10 1"; do
  set -- $cfg
  MMFM_FUSED=$1 MMFM_MLP_V1=$2 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSED=$1 MLP_V1=$2', d['ms_per_step'], d['kernel_breakdown_ms'])"
done
