# same-box comparison of the GEMM epilogue's store policy on the whole bench step (MMFM_GEMM_NT: bit 0 = C, bit 1 = pre_out)
for nt in 3 11 3 11; do
  MMFM_GEMM_NT=$nt timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NT=$nt', d['ms_per_step'], d['kernel_breakdown_ms'])"
done
