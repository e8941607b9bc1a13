// LayerNorm forward/backward (mmfm_layernorm_fwd / _bwd in include/mmfm.h).
// One 64-lane wavefront per row; H/64 elements per lane held in registers (H <= 1024, H % 4 == 0);
// two-pass statistics in fp32 (mean, then centred variance) like torch's CPU/GPU kernels.
// HBM-bound: algorithmic bytes fwd = 2*R*H*sizeof(T) (+8R stats), bwd = 4*R*H*sizeof(T).
#include "common.h"
#include <algorithm>

namespace {

constexpr int MAXV = 4;  // float4 chunks per lane -> H <= 64*4*4 = 1024

__device__ __forceinline__ int64_t destitch_row(int64_t r, int L, int T, int64_t Btot) {
    if (T <= 0) return r;
    const int64_t b = r / L;
    const int l = (int)(r % L);
    return (int64_t)(l / T) * (Btot * T) + b * T + (l % T);
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t R, int H,
                                                     float eps, int dsL, int dsT) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = H / 256 + ((H % 256) ? 1 : 0);
    const int64_t Btot = dsT > 0 ? R / dsL : 0;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < R; row += (int64_t)gridDim.x * 4) {
        float4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nv && c < H) {
                v[i] = io<T>::ld4(x + (size_t)row * H + c);
                s += v[i].x + v[i].y + v[i].z + v[i].w;
            }
        }
        const float mu = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < H) {
                const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
                q += a * a + b * b + cc * cc + dd * dd;
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        const int64_t orow = destitch_row(row, dsL, dsT, Btot);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < H) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + c);
                const float4 bt = *reinterpret_cast<const float4*>(beta + c);
                float4 o;
                o.x = (v[i].x - mu) * rs * g.x + bt.x;
                o.y = (v[i].y - mu) * rs * g.y + bt.y;
                o.z = (v[i].z - mu) * rs * g.z + bt.z;
                o.w = (v[i].w - mu) * rs * g.w + bt.w;
                io<T>::st4(y + (size_t)orow * H + c, o);
            }
        }
    }
}

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  per-block partial dgamma/dbeta
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* dres, T* dx,
                                                     float* __restrict__ part, int64_t R, int H, int dsL, int dsT) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = H / 256 + ((H % 256) ? 1 : 0);
    const int64_t Btot = dsT > 0 ? R / dsL : 0;
    float4 ag[MAXV], ab[MAXV], g[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        ag[i] = ab[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = i * 256 + lane * 4;
        if (i < nv && c < H) g[i] = *reinterpret_cast<const float4*>(gamma + c);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < R; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const int64_t yrow = destitch_row(row, dsL, dsT, Btot);
        float4 d[MAXV], xh[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            d[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nv && c < H) {
                d[i] = io<T>::ld4(dy + (size_t)yrow * H + c);
                const float4 xv = io<T>::ld4(x + (size_t)row * H + c);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
                ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y;
                ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
                d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;
                s1 += d[i].x + d[i].y + d[i].z + d[i].w;
                s2 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
            }
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = i * 256 + lane * 4;
            if (i < nv && c < H) {
                float4 o;
                o.x = rs * (d[i].x - m1 - xh[i].x * m2);
                o.y = rs * (d[i].y - m1 - xh[i].y * m2);
                o.z = rs * (d[i].z - m1 - xh[i].z * m2);
                o.w = rs * (d[i].w - m1 - xh[i].w * m2);
                if (dres) {
                    const float4 rr = io<T>::ld4(dres + (size_t)row * H + c);
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                }
                io<T>::st4(dx + (size_t)row * H + c, o);
            }
        }
    }
    // cross-wave reduction of the column partials, fixed order (deterministic)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = i * 256 + lane * 4;
        if (i < nv && c < H) {
            *reinterpret_cast<float4*>(&red[(wave * 2 + 0) * H + c]) = ag[i];
            *reinterpret_cast<float4*>(&red[(wave * 2 + 1) * H + c]) = ab[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * H; c += 256) {
        const int which = c / H, col = c % H;
        const float s = red[(0 * 2 + which) * H + col] + red[(1 * 2 + which) * H + col] +
                        red[(2 * 2 + which) * H + col] + red[(3 * 2 + which) * H + col];
        part[(size_t)blockIdx.x * 2 * H + c] = s;
    }
}

// sum the per-block partials: 64 columns x 4 partial-lanes per block, coalesced along the columns
__global__ __launch_bounds__(256) void ln_bwd_finalize(const float* __restrict__ part, int nblk, int H, float* dgamma, float* dbeta,
                                                       int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < 2 * H)
        for (int b = pl; b < nblk; b += 4) s += part[(size_t)b * 2 * H + c];
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < 2 * H) {
        s = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
        float* dst = c < H ? dgamma + c : dbeta + (c - H);
        *dst = accumulate ? *dst + s : s;
    }
}

// enough workgroups to fill every wave slot of the chip (256 CUs x 8 blocks of 4 waves): streaming kernels
// hide HBM latency with waves in flight; the backward keeps fewer blocks (its per-block partials are reduced after)
int ln_blocks(int64_t R) { return (int)std::max<int64_t>(1, std::min<int64_t>(2048, (R + 3) / 4)); }
int ln_bwd_blocks(int64_t R) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, (R + 3) / 4)); }

}  // namespace

extern "C" int mmfm_reduce_slabs(float*, const float*, int64_t, int, int64_t, int, mmfm_stream);

extern "C" int mmfm_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                                  float* mean, float* rstd, int64_t R, int H, float eps, int dsL, int dsT,
                                  mmfm_stream stream) {
    MMFM_REQUIRE(x && gamma && beta && y && mean && rstd, "mmfm_layernorm_fwd: null pointer");
    MMFM_REQUIRE(R > 0 && H > 0 && H % 4 == 0 && H <= 1024, "mmfm_layernorm_fwd: H=%d must be a multiple of 4, <= 1024", H);
    MMFM_REQUIRE(dsT == 0 || (dsL > 0 && dsL % dsT == 0 && R % dsL == 0), "mmfm_layernorm_fwd: bad destitch L=%d T=%d", dsL, dsT);
    dim3 grid(ln_blocks(R)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, st, (const float*)x, gamma, beta, (float*)y, mean, rstd, R, H, eps, dsL, dsT);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<uint16_t>, grid, block, 0, st, (const uint16_t*)x, gamma, beta, (uint16_t*)y, mean, rstd, R, H, eps, dsL, dsT);
    else
        return mmfm_set_error(-1, "mmfm_layernorm_fwd: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_layernorm_fwd");
    return 0;
}

extern "C" int64_t mmfm_layernorm_bwd_workspace(int64_t R, int H) { return (int64_t)ln_bwd_blocks(R) * 2 * H * sizeof(float); }

extern "C" int mmfm_layernorm_bwd(int dtype, const void* dy, const void* x, const float* mean, const float* rstd,
                                  const float* gamma, const void* dres, void* dx, float* dgamma, float* dbeta,
                                  int accumulate, int64_t R, int H, int dsL, int dsT, void* workspace,
                                  int64_t workspace_bytes, mmfm_stream stream) {
    MMFM_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "mmfm_layernorm_bwd: null pointer");
    MMFM_REQUIRE(R > 0 && H > 0 && H % 4 == 0 && H <= 1024, "mmfm_layernorm_bwd: H=%d must be a multiple of 4, <= 1024", H);
    MMFM_REQUIRE(dsT == 0 || (dsL > 0 && dsL % dsT == 0 && R % dsL == 0), "mmfm_layernorm_bwd: bad destitch");
    MMFM_REQUIRE(workspace && workspace_bytes >= mmfm_layernorm_bwd_workspace(R, H), "mmfm_layernorm_bwd: workspace too small");
    const int nblk = ln_bwd_blocks(R);
    const size_t lds = (size_t)4 * 2 * H * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nblk), dim3(256), lds, st, (const float*)dy, (const float*)x, mean, rstd, gamma,
                           (const float*)dres, (float*)dx, (float*)workspace, R, H, dsL, dsT);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<uint16_t>, dim3(nblk), dim3(256), lds, st, (const uint16_t*)dy, (const uint16_t*)x, mean, rstd,
                           gamma, (const uint16_t*)dres, (uint16_t*)dx, (float*)workspace, R, H, dsL, dsT);
    else
        return mmfm_set_error(-1, "mmfm_layernorm_bwd: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_layernorm_bwd");
    // per-block partials [nblk][2][H] -> dgamma, dbeta (two tall-skinny slab reductions)
    if (dbeta == dgamma + H)     // adjacent in the flat gradient buffer (the engine's layout): one reduction over [2H]
        return mmfm_reduce_slabs(dgamma, (const float*)workspace, 2 * (int64_t)H, nblk, 2 * (int64_t)H, accumulate, stream);
    if (int rc = mmfm_reduce_slabs(dgamma, (const float*)workspace, H, nblk, 2 * (int64_t)H, accumulate, stream)) return rc;
    return mmfm_reduce_slabs(dbeta, (const float*)workspace + H, H, nblk, 2 * (int64_t)H, accumulate, stream);
}
