// LayerNorm forward/backward (mmfm_layernorm_fwd / _bwd in include/mmfm.h).
// One 64-lane wavefront per row; H/64 elements per lane held in registers (H <= 1024, H % 4 == 0);
// two-pass statistics in fp32 (mean, then centred variance) like torch's CPU/GPU kernels.
// HBM-bound: algorithmic bytes fwd = 2*R*H*sizeof(T) (+8R stats), bwd = 4*R*H*sizeof(T).
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {


__device__ __forceinline__ int64_t destitch_row(int64_t r, int L, int T, int64_t Btot) {
    if (T <= 0) return r;
    const int64_t b = r / L;
    const int l = (int)(r % L);
    return (int64_t)(l / T) * (Btot * T) + b * T + (l % T);
}

// NV = 256-column chunks per lane (H <= 256*NV); UNR = rows a wave keeps in flight per iteration.  Measured at [204800, 256]
// bf16 (scripts/ln_bench.py): forward 41 us (5.1 TB/s) at UNR = 1 or 2, 56 us at 4, 63 us at 8 (registers cost more
// waves than the extra rows in flight buy); backward 135 / 113 / 103 us at UNR = 1 / 2 / 4.  The kernels are
// latency-bound at one row per wave (16 waves/CU x 1.5 KB in flight = 3 TB/s by Little's law, measured 137 us for
// the [204800, 256] bf16 backward = 3.07 TB/s): all loads of UNR consecutive rows are issued before the first use.
template <typename T, int NV, int UNR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int64_t R, int H,
                                                     float eps, int dsL, int dsT) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Btot = dsT > 0 ? R / dsL : 0;
    float4 g[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        g[i] = bt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < H) { g[i] = *reinterpret_cast<const float4*>(gamma + c); bt[i] = *reinterpret_cast<const float4*>(beta + c); }
    }
    const float invH = 1.f / (float)H;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * UNR; row0 < R; row0 += (int64_t)gridDim.x * 4 * UNR) {
        float4 v[UNR][NV];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                v[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < H && row0 + u < R) v[u][i] = io<T>::ld4(x + (size_t)(row0 + u) * H + c);
            }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t row = row0 + u;
            if (row >= R) break;                   // wave-uniform
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s += v[u][i].x + v[u][i].y + v[u][i].z + v[u][i].w;   // zero beyond H
            const float mu = wave_sum(s) * invH;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                if (c < H) {
                    const float a = v[u][i].x - mu, b = v[u][i].y - mu, cc = v[u][i].z - mu, dd = v[u][i].w - mu;
                    q += a * a + b * b + cc * cc + dd * dd;
                }
            }
            const float rs = rsqrtf(wave_sum(q) * invH + eps);
            if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
            const int64_t orow = destitch_row(row, dsL, dsT, Btot);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                if (c < H) {
                    float4 o;
                    o.x = (v[u][i].x - mu) * rs * g[i].x + bt[i].x;
                    o.y = (v[u][i].y - mu) * rs * g[i].y + bt[i].y;
                    o.z = (v[u][i].z - mu) * rs * g[i].z + bt[i].z;
                    o.w = (v[u][i].w - mu) * rs * g[i].w + bt[i].w;
                    io<T>::st4(y + (size_t)orow * H + c, o);
                }
            }
        }
    }
}

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  per-block partial dgamma/dbeta
template <typename T, int NV, int UNR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* dres, T* dx,
                                                     float* __restrict__ part, int64_t R, int H, int dsL, int dsT) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [4][2][H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Btot = dsT > 0 ? R / dsL : 0;
    float4 ag[NV], ab[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = ab[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = i * 256 + lane * 4;
        if (c < H) g[i] = *reinterpret_cast<const float4*>(gamma + c);
    }
    const float invH = 1.f / (float)H;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * UNR; row0 < R; row0 += (int64_t)gridDim.x * 4 * UNR) {
        float4 d[UNR][NV], xv[UNR][NV], rr[UNR][NV];
        float mu[UNR], rs[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t row = row0 + u;
            const bool ok = row < R;
            mu[u] = ok ? mean[row] : 0.f;
            rs[u] = ok ? rstd[row] : 0.f;
            const int64_t yrow = ok ? destitch_row(row, dsL, dsT, Btot) : 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                d[u][i] = xv[u][i] = rr[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < H && ok) {
                    d[u][i] = io<T>::ld4(dy + (size_t)yrow * H + c);
                    xv[u][i] = io<T>::ld4(x + (size_t)row * H + c);
                    if (dres) rr[u][i] = io<T>::ld4(dres + (size_t)row * H + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t row = row0 + u;
            if (row >= R) break;                   // wave-uniform
            float4 xh[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c < H) {
                    float4& dd = d[u][i];
                    xh[i] = make_float4((xv[u][i].x - mu[u]) * rs[u], (xv[u][i].y - mu[u]) * rs[u], (xv[u][i].z - mu[u]) * rs[u],
                                        (xv[u][i].w - mu[u]) * rs[u]);
                    ab[i].x += dd.x; ab[i].y += dd.y; ab[i].z += dd.z; ab[i].w += dd.w;
                    ag[i].x += dd.x * xh[i].x; ag[i].y += dd.y * xh[i].y; ag[i].z += dd.z * xh[i].z; ag[i].w += dd.w * xh[i].w;
                    dd.x *= g[i].x; dd.y *= g[i].y; dd.z *= g[i].z; dd.w *= g[i].w;
                    s1 += dd.x + dd.y + dd.z + dd.w;
                    s2 += dd.x * xh[i].x + dd.y * xh[i].y + dd.z * xh[i].z + dd.w * xh[i].w;
                }
            }
            const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = i * 256 + lane * 4;
                if (c < H) {
                    float4 o;
                    o.x = rs[u] * (d[u][i].x - m1 - xh[i].x * m2) + rr[u][i].x;
                    o.y = rs[u] * (d[u][i].y - m1 - xh[i].y * m2) + rr[u][i].y;
                    o.z = rs[u] * (d[u][i].z - m1 - xh[i].z * m2) + rr[u][i].z;
                    o.w = rs[u] * (d[u][i].w - m1 - xh[i].w * m2) + rr[u][i].w;
                    io<T>::st4(dx + (size_t)row * H + c, o);
                }
            }
        }
    }
    // cross-wave reduction of the column partials, fixed order (deterministic)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < H) {
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
int ln_blocks(int64_t R) {
    static const int cap = [] { const char* e = getenv("MMFM_LN_FWD_BLOCKS"); return e ? atoi(e) : 2048; }();
    return (int)std::max<int64_t>(1, std::min<int64_t>(cap, (R + 15) / 16));
}
int ln_bwd_blocks(int64_t R) {
    static const int cap = [] { const char* e = getenv("MMFM_LN_BWD_BLOCKS"); return e ? atoi(e) : 1024; }();
    return (int)std::max<int64_t>(1, std::min<int64_t>(cap, (R + 15) / 16));
}

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
    if (dtype != MMFM_F32 && dtype != MMFM_BF16) return mmfm_set_error(-1, "mmfm_layernorm_fwd: bad dtype %d", dtype);
#define LN_FWD(TT, NVV, UU) hipLaunchKernelGGL((ln_fwd_kernel<TT, NVV, UU>), grid, block, 0, st, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, R, H, eps, dsL, dsT)
    static const int fwd_unr = [] { const char* e = getenv("MMFM_LN_FWD_UNR"); return e ? atoi(e) : 2; }();
#define LN_FWD_T(TT) if (H <= 256) { if (fwd_unr == 8) LN_FWD(TT, 1, 8); else if (fwd_unr == 2) LN_FWD(TT, 1, 2); else if (fwd_unr == 1) LN_FWD(TT, 1, 1); else LN_FWD(TT, 1, 4); } else if (H <= 512) LN_FWD(TT, 2, 2); else LN_FWD(TT, 4, 1)
    if (dtype == MMFM_F32) { LN_FWD_T(float); } else { LN_FWD_T(uint16_t); }
#undef LN_FWD_T
#undef LN_FWD
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
    if (dtype != MMFM_F32 && dtype != MMFM_BF16) return mmfm_set_error(-1, "mmfm_layernorm_bwd: bad dtype %d", dtype);
#define LN_BWD(TT, NVV, UU) hipLaunchKernelGGL((ln_bwd_kernel<TT, NVV, UU>), dim3(nblk), dim3(256), lds, st, (const TT*)dy, (const TT*)x, mean, rstd, gamma, (const TT*)dres, (TT*)dx, (float*)workspace, R, H, dsL, dsT)
    static const int bwd_unr = [] { const char* e = getenv("MMFM_LN_BWD_UNR"); return e ? atoi(e) : 4; }();
#define LN_BWD_T(TT) if (H <= 256) { if (bwd_unr == 2) LN_BWD(TT, 1, 2); else if (bwd_unr == 1) LN_BWD(TT, 1, 1); else LN_BWD(TT, 1, 4); } else if (H <= 512) LN_BWD(TT, 2, 2); else LN_BWD(TT, 4, 1)
    if (dtype == MMFM_F32) { LN_BWD_T(float); } else { LN_BWD_T(uint16_t); }
#undef LN_BWD_T
#undef LN_BWD
    MMFM_LAUNCH_CHECK("mmfm_layernorm_bwd");
    // per-block partials [nblk][2][H] -> dgamma, dbeta (two tall-skinny slab reductions)
    if (dbeta == dgamma + H)     // adjacent in the flat gradient buffer (the engine's layout): one reduction over [2H]
        return mmfm_reduce_slabs(dgamma, (const float*)workspace, 2 * (int64_t)H, nblk, 2 * (int64_t)H, accumulate, stream);
    if (int rc = mmfm_reduce_slabs(dgamma, (const float*)workspace, H, nblk, 2 * (int64_t)H, accumulate, stream)) return rc;
    return mmfm_reduce_slabs(dbeta, (const float*)workspace + H, H, nblk, 2 * (int64_t)H, accumulate, stream);
}
