// Masked Poisson-NLL / MSE loss (mm.py:79-82,217-239), forward reduction and backward.
// One wavefront per (b,t) row; unmasked rows are skipped without touching their data, so the
// algorithmic bytes are (rows masked) * N * (sizeof(T) + 4).  fp32 accumulation, fixed-order
// two-stage reduction (bitwise reproducible); n_examples is an exact int64 from mmfm_mask_prep.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float loss_elem(int kind, float p, float t) {
    if (kind == 0) return __expf(p) - t * p;     // PoissonNLLLoss(log_input=True, full=False)
    const float d = p - t;
    return d * d;                                  // MSELoss
}
__device__ __forceinline__ float loss_grad(int kind, float p, float t) {
    return kind == 0 ? __expf(p) - t : 2.f * (p - t);
}

template <typename T>
__global__ __launch_bounds__(256) void loss_fwd_kernel(int kind, const T* __restrict__ pred, const float* __restrict__ target,
                                                       const uint8_t* __restrict__ rowmask, int mask_ld, int Tn, int64_t R, int N,
                                                       float* __restrict__ part) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < R; row += (int64_t)gridDim.x * 4) {
        if (!rowmask[(row / Tn) * mask_ld + (row % Tn)]) continue;
        for (int c = lane; c < N; c += 64) s += loss_elem(kind, io<T>::ld(pred + (size_t)row * N + c), target[(size_t)row * N + c]);
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void loss_sum_kernel(const float* __restrict__ part, int n, float* out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += part[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ void loss_finalize_kernel(const float* loss_sum, const int64_t* count, int M, float* loss, float* inv_n) {
    float tot = 0.f;
    int64_t n = 0;
    for (int m = 0; m < M; ++m) { tot += loss_sum[m]; n += count[m]; }
    loss[0] = tot / (float)n;          // 0/0 -> NaN exactly like the reference (mm.py:237)
    inv_n[0] = 1.f / (float)n;
}

template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(int kind, const T* __restrict__ pred, const float* __restrict__ target,
                                                       const uint8_t* __restrict__ rowmask, int mask_ld, int Tn, int64_t R, int N,
                                                       const float* __restrict__ grad_out, const float* __restrict__ inv_n,
                                                       T* __restrict__ dpred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float g = grad_out[0] * inv_n[0];
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < R; row += (int64_t)gridDim.x * 4) {
        const bool on = rowmask[(row / Tn) * mask_ld + (row % Tn)] != 0;
        for (int c = lane; c < N; c += 64) {
            const size_t o = (size_t)row * N + c;
            // unmasked rows: mask * g with mask = 0 - i.e. 0 normally, NaN when nothing at all is masked (g = grad / 0 = inf), which is
            // what upstream's (loss * mask).sum() / mask.sum() hands to autograd (mm.py:217-239): every gradient of that step is NaN there
            io<T>::st(dpred + o, on ? g * loss_grad(kind, io<T>::ld(pred + o), target[o]) : g * 0.f);
        }
    }
}

int loss_blocks(int64_t R) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, (R + 3) / 4)); }

}  // namespace

extern "C" int64_t mmfm_masked_loss_workspace(int64_t R, int N) { (void)N; return (int64_t)loss_blocks(R) * sizeof(float); }

extern "C" int mmfm_masked_loss_fwd(int dtype, int kind, const void* pred, const float* target, const uint8_t* rowmask, int mask_ld,
                                    int T, int64_t R, int N, float* loss_sum, void* workspace, int64_t workspace_bytes, mmfm_stream stream) {
    MMFM_REQUIRE(pred && target && rowmask && loss_sum, "mmfm_masked_loss_fwd: null pointer");
    MMFM_REQUIRE((kind == 0 || kind == 1) && R > 0 && N > 0 && T > 0 && R % T == 0 && mask_ld >= T, "mmfm_masked_loss_fwd: bad arguments");
    MMFM_REQUIRE(workspace && workspace_bytes >= mmfm_masked_loss_workspace(R, N), "mmfm_masked_loss_fwd: workspace too small");
    const int nb = loss_blocks(R);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(loss_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, kind, (const float*)pred, target, rowmask, mask_ld, T, R, N, (float*)workspace);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(loss_fwd_kernel<uint16_t>, dim3(nb), dim3(256), 0, st, kind, (const uint16_t*)pred, target, rowmask, mask_ld, T, R, N, (float*)workspace);
    else
        return mmfm_set_error(-1, "mmfm_masked_loss_fwd: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_masked_loss_fwd");
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, nb, loss_sum);
    MMFM_LAUNCH_CHECK("mmfm_masked_loss_fwd(sum)");
    return 0;
}

extern "C" int mmfm_loss_finalize(const float* loss_sum, const int64_t* count, int M, float* loss, float* inv_n, mmfm_stream stream) {
    MMFM_REQUIRE(loss_sum && count && loss && inv_n && M > 0, "mmfm_loss_finalize: bad arguments");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss_sum, count, M, loss, inv_n);
    MMFM_LAUNCH_CHECK("mmfm_loss_finalize");
    return 0;
}

extern "C" int mmfm_masked_loss_bwd(int dtype, int kind, const void* pred, const float* target, const uint8_t* rowmask, int mask_ld,
                                    int T, int64_t R, int N, const float* grad_out, const float* inv_n, void* dpred, mmfm_stream stream) {
    MMFM_REQUIRE(pred && target && rowmask && grad_out && inv_n && dpred, "mmfm_masked_loss_bwd: null pointer");
    MMFM_REQUIRE((kind == 0 || kind == 1) && R > 0 && N > 0 && T > 0 && R % T == 0 && mask_ld >= T, "mmfm_masked_loss_bwd: bad arguments");
    const int nb = loss_blocks(R);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(loss_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, kind, (const float*)pred, target, rowmask, mask_ld, T, R, N, grad_out, inv_n, (float*)dpred);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(loss_bwd_kernel<uint16_t>, dim3(nb), dim3(256), 0, st, kind, (const uint16_t*)pred, target, rowmask, mask_ld, T, R, N, grad_out, inv_n, (uint16_t*)dpred);
    else
        return mmfm_set_error(-1, "mmfm_masked_loss_bwd: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_masked_loss_bwd");
    return 0;
}
