# diagnostic library whose streaming dW kernel keeps a smaller LDS ring (so that another kernel's workgroups fit beside it on a CU):
# libmmfm_dw<KB>.so.  Use with MMFM_LIB=.../libmmfm_dw<KB>.so (scripts/attn_gemm_overlap.py)
KB=${1:-72}
cd "$(dirname "$0")/../../multi_modal_foundation_model_amd/csrc" || exit 1
mkdir -p build_dw$KB
cp build/*.o build_dw$KB/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DMMFM_DW_LDS_KB=$KB -c gemm_dw.hip -o build_dw$KB/gemm_dw.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmmfm_dw$KB.so build_dw$KB/*.o
