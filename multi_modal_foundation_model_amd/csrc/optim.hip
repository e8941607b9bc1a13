// Fused AdamW over a flat fp32 range (+ bf16 weight refresh), elementwise dropout, casts.
// HBM-bound: AdamW moves 28 B/param (read p,g,m,v; write p,m,v) + 2 B with the bf16 copy.
#include "common.h"
#include <algorithm>

namespace {

// hyper (device): [0] 1-lr*wd  [1] 1-beta1  [2] beta2  [3] 1-beta2  [4] lr/bias_corr1
//                 [5] sqrt(bias_corr2)  [6] eps  [7] grad_scale      (host computes them in double)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, uint16_t* __restrict__ pb, int64_t n,
                                                    const float* __restrict__ hyper) {
    const float decay = hyper[0], omb1 = hyper[1], b2 = hyper[2], omb2 = hyper[3], step = hyper[4], bc2s = hyper[5], eps = hyper[6],
                gs = hyper[7];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gs;
        float pi = p[i] * decay;
        const float mi = m[i] + omb1 * (gi - m[i]);           // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + omb2 * gi * gi;          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vi) / bc2s + eps;
        pi -= step * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (pb) pb[i] = f2bf(pi);
    }
}

template <typename T>
__global__ void dropout_apply_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t n, mmfm_dropout da) {
    const Drop dr = drop_init(da);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        io<T>::st(dst + i, dr.apply(io<T>::ld(src + i), (uint64_t)i));
}

// bf16, n % 8 == 0, 16-B aligned: 8 elements per thread (the scalar kernel above moves 2 B per lane: 3.3 TB/s)
__global__ __launch_bounds__(256) void dropout_apply8_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int64_t n8, mmfm_dropout da) {
    const Drop dr = drop_init(da);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 g = *reinterpret_cast<const uint4*>(src + 8 * i);
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
        uint32_t ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v0 = dr.apply(__uint_as_float(gw[j] << 16), (uint64_t)(8 * i + 2 * j));
            const float v1 = dr.apply(__uint_as_float(gw[j] & 0xffff0000u), (uint64_t)(8 * i + 2 * j + 1));
            ow[j] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
        }
        *reinterpret_cast<uint4*>(dst + 8 * i) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}

int ew_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(4096, (n + 255) / 256)); }

}  // namespace

extern "C" int mmfm_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, const float* hyper, mmfm_stream stream) {
    MMFM_REQUIRE(p && g && m && v && hyper && n > 0, "mmfm_adamw_step: bad arguments");
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (uint16_t*)p_bf16, n, hyper);
    MMFM_LAUNCH_CHECK("mmfm_adamw_step");
    return 0;
}

extern "C" int mmfm_dropout_apply(int dtype, const void* src, void* dst, int64_t R, int N, mmfm_dropout drop, mmfm_stream stream) {
    MMFM_REQUIRE(src && dst && R > 0 && N > 0, "mmfm_dropout_apply: bad arguments");
    const int64_t n = R * N;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (const float*)src, (float*)dst, n, drop);
    else if (dtype == MMFM_BF16 && n % 8 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0)
        hipLaunchKernelGGL(dropout_apply8_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, n / 8, drop);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(dropout_apply_kernel<uint16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (const uint16_t*)src, (uint16_t*)dst, n, drop);
    else
        return mmfm_set_error(-1, "mmfm_dropout_apply: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_dropout_apply");
    return 0;
}

extern "C" int mmfm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, mmfm_stream stream) {
    MMFM_REQUIRE(src && dst && n > 0, "mmfm_cast_f32_to_bf16: bad arguments");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, n);
    MMFM_LAUNCH_CHECK("mmfm_cast_f32_to_bf16");
    return 0;
}
