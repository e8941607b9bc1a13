# diagnostic library with s_memtime phase stamps in the fast attention kernels: libmmfm_astamp.so (use with MMFM_LIB=.../libmmfm_astamp.so)
cd "$(dirname "$0")/../../multi_modal_foundation_model_amd/csrc" || exit 1
mkdir -p build_stamp
for f in api gemm gemm_bf16 layernorm attention attention_bf16 stitch loss optim metrics rowgemm mlp_fused; do cp build/$f.o build_stamp/ 2>/dev/null; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMMFM_ATTN_STAMP -c attention_fast.hip -o build_stamp/attention_fast.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmfm_astamp.so build_stamp/*.o
