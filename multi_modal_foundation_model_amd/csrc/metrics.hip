// Evaluation reductions on the device, replacing host Python loops of the reference's eval path:
//   mmfm_r2_series       - utils/utils.py:107-115 (metrics_list "r2": torcheval R2Score per (neuron, trial) over the time
//                          bins, 50 x B host calls per session in trainer/base.py:252-262)
//   mmfm_bits_per_spike  - utils/eval_utils.py:1051-1119 (neg_log_likelihood, bits_per_spike; NLB co-smoothing metric)
// Both accumulate in fp64 (the reference computes them in float64 numpy / torcheval double sums) in a fixed order.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wavefront per series (g, c): r2 = 1 - sum_s (y - p)^2 / sum_s (y - mean_s y)^2 ; element (g, s, c) of a strided view
__global__ __launch_bounds__(256) void r2_series_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int64_t g0, int64_t g1,
                                                        int64_t g2, int64_t p0, int64_t p1, int64_t p2, int G, int S, int C,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t series = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (series >= (int64_t)G * C) return;
    const int g = (int)(series / C), c = (int)(series % C);
    const float* y = gt + g * g0 + c * g2;
    const float* p = pred + g * p0 + c * p2;
    double sy = 0.0;
    for (int s = lane; s < S; s += 64) sy += (double)y[s * g1];
    const double mean = wave_sum_d(sy) / (double)S;
    double res = 0.0, tot = 0.0;
    for (int s = lane; s < S; s += 64) {
        const double yv = (double)y[s * g1], d = yv - (double)p[s * p1], m = yv - mean;
        res += d * d;
        tot += m * m;
    }
    res = wave_sum_d(res);
    tot = wave_sum_d(tot);
    if (lane == 0) out[series] = (float)(1.0 - res / tot);       // tot == 0 -> -inf / nan, masked by the caller like np.ma.masked_invalid
}

// lgamma(n + 1) for spike counts: exact table for small integers, lgamma otherwise
__device__ __forceinline__ double log_factorial(float s) {
    return lgamma((double)s + 1.0);
}

// partial sums per block: [0] nll_model, [1] nll_null, [2] total spikes   (rates == 0 -> 1e-9, eval_utils.py:1083-1089)
__global__ __launch_bounds__(256) void bps_partial_kernel(const float* __restrict__ rates, const float* __restrict__ spikes,
                                                          const float* __restrict__ colsum, int64_t R, int N, double* __restrict__ part) {
    __shared__ double red[3][4];
    double a = 0.0, b = 0.0, c = 0.0;
    const int64_t total = R * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const double s = (double)spikes[i];
        double r = (double)rates[i];
        if (r == 0.0) r = 1e-9;
        double mu = (double)colsum[n] / (double)R;
        if (mu == 0.0) mu = 1e-9;
        const double lf = log_factorial(spikes[i]);
        a += r - s * log(r) + lf;
        b += mu - s * log(mu) + lf;
        c += s;
    }
    a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
    __syncthreads();
    if (threadIdx.x < 3) part[(size_t)blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ void bps_final_kernel(const double* __restrict__ part, int nblk, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = 0; i < nblk; ++i) { a += part[3 * i]; b += part[3 * i + 1]; c += part[3 * i + 2]; }
    out[0] = (float)((b - a) / c / 0.6931471805599453);
    out[1] = (float)a;
    out[2] = (float)b;
    out[3] = (float)c;
}

// per-neuron bits per spike (spiking_activity_recon_eval, utils/eval_utils.py:846-851: bits_per_spike on one neuron's
// [trials, bins, 1] slice, N host calls upstream).  The log n! terms of the model and the null likelihood cancel:
//   nll_null - nll_model = R*mu - log(mu) * S - A,   A = sum_t (r - s log r),  S = sum_t s,  mu = S / R  (0 -> 1e-9).
// Stage 1: each block owns 64 columns x a row chunk and writes fp64 partials [chunk][2][N]; stage 2 sums the chunks in
// order and finishes the formula.
__global__ __launch_bounds__(256) void bpsn_partial_kernel(const float* __restrict__ rates, const float* __restrict__ spikes, int64_t R, int N,
                                                           int rows_per_chunk, double* __restrict__ part) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    double a = 0.0, sm = 0.0;
    if (col < N)
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const double s = (double)spikes[r * N + col];
            double rt = (double)rates[r * N + col];
            if (rt == 0.0) rt = 1e-9;
            a += rt - s * log(rt);
            sm += s;
        }
    red[0][rl][cl] = a;
    red[1][rl][cl] = sm;
    __syncthreads();
    if (rl == 0 && col < N) {
        double* o = part + (size_t)blockIdx.y * 2 * N;
        o[col] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
        o[N + col] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    }
}

__global__ __launch_bounds__(256) void bpsn_final_kernel(const double* __restrict__ part, int nchunk, int64_t R, int N, float* __restrict__ out) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= N) return;
    double a = 0.0, sm = 0.0;
    for (int c = 0; c < nchunk; ++c) { a += part[(size_t)c * 2 * N + col]; sm += part[(size_t)c * 2 * N + N + col]; }
    double mu = sm / (double)R;
    if (mu == 0.0) mu = 1e-9;
    out[col] = (float)(((double)R * mu - log(mu) * sm - a) / sm / 0.6931471805599453);      // S = 0 -> inf / nan like upstream
}

int bpsn_chunks(int64_t R) { return (int)std::max<int64_t>(1, std::min<int64_t>(256, R / 64)); }

int bps_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 255) / 256)); }

}  // namespace

extern "C" int64_t mmfm_colsum_workspace(int64_t R, int N);
extern "C" int mmfm_colsum(int dtype, const void* x, int64_t R, int N, int ld, float* out, int accumulate, void* workspace, int64_t workspace_bytes,
                           mmfm_stream stream);

extern "C" int mmfm_r2_series(const float* gt, const int64_t* gt_strides, const float* pred, const int64_t* pred_strides, int G, int S, int C,
                              float* out, mmfm_stream stream) {
    MMFM_REQUIRE(gt && pred && gt_strides && pred_strides && out, "mmfm_r2_series: null pointer");
    MMFM_REQUIRE(G > 0 && S > 0 && C > 0, "mmfm_r2_series: bad shape G=%d S=%d C=%d", G, S, C);
    const int64_t series = (int64_t)G * C;
    hipLaunchKernelGGL(r2_series_kernel, dim3((unsigned)((series + 3) / 4)), dim3(256), 0, (hipStream_t)stream, gt, pred, gt_strides[0], gt_strides[1],
                       gt_strides[2], pred_strides[0], pred_strides[1], pred_strides[2], G, S, C, out);
    MMFM_LAUNCH_CHECK("mmfm_r2_series");
    return 0;
}

extern "C" int64_t mmfm_bits_per_spike_workspace(int64_t R, int N) {
    const int64_t cs = (mmfm_colsum_workspace(R, N) + 255) / 256 * 256;
    return cs + (((int64_t)N * 4 + 255) / 256 * 256) + (int64_t)bps_blocks(R * N) * 3 * (int64_t)sizeof(double);
}

extern "C" int mmfm_bits_per_spike(const float* rates, const float* spikes, int64_t R, int N, float* out, void* workspace, int64_t workspace_bytes,
                                   mmfm_stream stream) {
    MMFM_REQUIRE(rates && spikes && out && workspace, "mmfm_bits_per_spike: null pointer");
    MMFM_REQUIRE(R > 0 && N > 0, "mmfm_bits_per_spike: bad shape");
    MMFM_REQUIRE(workspace_bytes >= mmfm_bits_per_spike_workspace(R, N), "mmfm_bits_per_spike: workspace too small");
    const int64_t cs = (mmfm_colsum_workspace(R, N) + 255) / 256 * 256;
    char* ws = reinterpret_cast<char*>(workspace);
    float* colsum = reinterpret_cast<float*>(ws + cs);
    double* part = reinterpret_cast<double*>(ws + cs + (((int64_t)N * 4 + 255) / 256 * 256));
    if (int rc = mmfm_colsum(MMFM_F32, spikes, R, N, N, colsum, 0, ws, cs, stream)) return rc;      // per-neuron totals -> null-model rates
    const int nblk = bps_blocks(R * N);
    hipLaunchKernelGGL(bps_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, rates, spikes, colsum, R, N, part);
    hipLaunchKernelGGL(bps_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, nblk, out);
    MMFM_LAUNCH_CHECK("mmfm_bits_per_spike");
    return 0;
}

extern "C" int64_t mmfm_bits_per_spike_neurons_workspace(int64_t R, int N) { return (int64_t)bpsn_chunks(R) * 2 * N * (int64_t)sizeof(double); }

extern "C" int mmfm_bits_per_spike_neurons(const float* rates, const float* spikes, int64_t R, int N, float* out, void* workspace,
                                           int64_t workspace_bytes, mmfm_stream stream) {
    MMFM_REQUIRE(rates && spikes && out && workspace, "mmfm_bits_per_spike_neurons: null pointer");
    MMFM_REQUIRE(R > 0 && N > 0, "mmfm_bits_per_spike_neurons: bad shape");
    MMFM_REQUIRE(workspace_bytes >= mmfm_bits_per_spike_neurons_workspace(R, N), "mmfm_bits_per_spike_neurons: workspace too small");
    const int nch = bpsn_chunks(R);
    const int rpc = (int)((R + nch - 1) / nch);
    hipLaunchKernelGGL(bpsn_partial_kernel, dim3((N + 63) / 64, nch), dim3(256), 0, (hipStream_t)stream, rates, spikes, R, N, rpc, (double*)workspace);
    hipLaunchKernelGGL(bpsn_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nch, R, N, out);
    MMFM_LAUNCH_CHECK("mmfm_bits_per_spike_neurons");
    return 0;
}
