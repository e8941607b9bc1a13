# diagnostic library with s_memtime phase stamps in the MLP kernels: libmmfm_stamp.so (use with MMFM_LIB=.../libmmfm_stamp.so)
cd "$(dirname "$0")/../../multi_modal_foundation_model_amd/csrc" || exit 1
mkdir -p build_stamp
cp build/*.o build_stamp/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DMMFM_STAMP -c mlp_fused.hip -o build_stamp/mlp_fused.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmfm_stamp.so build_stamp/*.o
