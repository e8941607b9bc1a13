# diagnostic library with parts of the fused MLP backward loop compiled out: libmmfm_probe<N>.so, N = MMFM_PROBE bit mask
# (1 = no g / du stores, 2 = no GELU algebra); results are wrong, timing only.  Use with MMFM_LIB=.../libmmfm_probe<N>.so
N=${1:-1}
cd "$(dirname "$0")/../../multi_modal_foundation_model_amd/csrc" || exit 1
mkdir -p build_probe$N
cp build/*.o build_probe$N/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMMFM_PROBE=$N -c mlp_fused.hip -o build_probe$N/mlp_fused.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmfm_probe$N.so build_probe$N/*.o
