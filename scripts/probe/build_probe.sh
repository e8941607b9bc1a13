# diagnostic library with parts of the fused MLP backward loop compiled out: libmmfm_probe<N>.so, N = MMFM_PROBE bit mask
# (1 = no g / du stores, 2 = no GELU algebra); results are wrong, timing only.  Use with MMFM_LIB=.../libmmfm_probe<N>.so
N=${1:-1}
cd "$(dirname "$0")/../../multi_modal_foundation_model_amd/csrc" || exit 1
mkdir -p build_probe$N
for f in api gemm gemm_bf16 layernorm attention attention_bf16 stitch loss optim metrics rowgemm; do cp build/$f.o build_probe$N/ 2>/dev/null; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMMFM_PROBE=$N -c mlp_fused.hip -o build_probe$N/mlp_fused.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmfm_probe$N.so build_probe$N/*.o
