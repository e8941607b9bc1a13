# Build the committed library for the A side first:
#   mkdir -p /tmp/prev scripts/ab && git archive HEAD multi_modal_foundation_model_amd/csrc include | tar -x -C /tmp/prev \
#     && make -C /tmp/prev/multi_modal_foundation_model_amd/csrc -j8 && cp /tmp/prev/multi_modal_foundation_model_amd/libmmfm_hip.so scripts/ab/libmmfm_hip_prev.so
# same-box A/B: committed build (scripts/ab/libmmfm_hip_prev.so) vs working tree, microbench then whole step
echo "== prev"; MMFM_LIB=$PWD/scripts/ab/libmmfm_hip_prev.so timeout -k 10 120 python scripts/gemm_bench.py 1024 2>/dev/null | grep -E "qkv|proj|down|head"
echo "== new";  timeout -k 10 120 python scripts/gemm_bench.py 1024 2>/dev/null | grep -E "qkv|proj|down|head"
echo "== prev bench"; MMFM_LIB=$PWD/scripts/ab/libmmfm_hip_prev.so timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_breakdown_ms'], d['gemm_layouts'])"
echo "== new bench"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_breakdown_ms'], d['gemm_layouts'])"
