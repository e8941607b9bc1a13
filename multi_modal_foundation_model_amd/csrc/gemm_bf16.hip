// bf16 MFMA GEMM (throughput mode) — see mmfm_gemm in include/mmfm.h.
#include "common.h"

int mmfm_gemm_bf16_launch(const mmfm_gemm_desc* d, hipStream_t st) {
    (void)d; (void)st;
    return mmfm_set_error(-1, "mmfm_gemm: bf16 path not built yet");
}
