// Mask/index preparation and the per-modality "stitch" (tokens + modality/position embeddings
// scattered into the concatenated [B, M*T, H] sequence), forward and backward.
// mm.py:90-110,141-175,245-275; encoder_embeddings.py:56-61; decoder_embeddings.py:56-61.
// HBM-bound elementwise/gather work; index semantics are bit-exact (sample-0 quirk included).
#include "common.h"
#include <algorithm>

namespace {

constexpr int MAXM = 8;
struct MaskSrc {
    const int64_t* p[MAXM];
    int64_t stride[MAXM];
    int64_t channels[MAXM];
};

__global__ void mask_prep_kernel(MaskSrc src, const int64_t* __restrict__ attn, int B, int T, int M, uint8_t* __restrict__ tokmask,
                                 uint8_t* __restrict__ keypad, uint8_t* __restrict__ keep0, uint8_t* __restrict__ mod_id,
                                 unsigned long long* __restrict__ count) {
    const int L = M * T;
    unsigned long long local[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) local[m] = 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)B * L; idx += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / L), l = (int)(idx % L), m = l / T, t = l % T;
        const int64_t a = attn[(size_t)b * T + t];
        const int64_t v = src.p[m][((size_t)b * T + t) * src.stride[m]] & a;      // mm.py:270
        tokmask[idx] = (uint8_t)(v != 0);
        keypad[idx] = (uint8_t)(a != 0);
        if (b == 0) {
            keep0[l] = (uint8_t)(v != 1);                                            // mm.py:145: argwhere(mask[0] == 1)
            mod_id[l] = (uint8_t)m;
        }
#pragma unroll
        for (int mm = 0; mm < MAXM; ++mm)
            if (mm == m) local[mm] += (unsigned long long)v * (unsigned long long)src.channels[mm];
    }
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
            unsigned long long s = local[m];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(&count[m], s);              // integer: order-independent, exact
        }
    }
}

// (a captured hipMemsetAsync node replayed with a stale fill pattern on this stack, so the counters are
// zeroed by a kernel of our own: plain stream order, identical eager and under hipGraph replay)
__global__ void zero_u64_kernel(unsigned long long* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0ull;
}

template <typename T>
__global__ __launch_bounds__(256) void stitch_fwd_kernel(const T* __restrict__ tok, const float* __restrict__ mod_row,
                                                         const float* __restrict__ pos, const int64_t* __restrict__ ts,
                                                         const uint8_t* __restrict__ keep0, T* __restrict__ x, T* __restrict__ emb,
                                                         int B, int Tn, int L, int m, int H, int max_F) {
    const int C4 = H / 4;
    const int64_t total = (int64_t)B * Tn * C4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / C4;
        const int c = (int)(idx % C4) * 4;
        const int b = (int)(r / Tn), t = (int)(r % Tn);
        const int l = m * Tn + t;
        const float4 mr = *reinterpret_cast<const float4*>(mod_row + c);
        const int tsv = (int)min((int64_t)(max_F - 1), max((int64_t)0, ts[r]));     // memory-safe even for bad stamps
        const float4 pr = *reinterpret_cast<const float4*>(pos + (size_t)tsv * H + c);
        float4 e = make_float4(mr.x + pr.x, mr.y + pr.y, mr.z + pr.z, mr.w + pr.w);
        const size_t o = ((size_t)b * L + l) * H + c;
        if (emb) io<T>::st4(emb + o, e);
        if (keep0[l]) {
            const float4 tv = io<T>::ld4(tok + (size_t)r * H + c);
            e.x += tv.x; e.y += tv.y; e.z += tv.z; e.w += tv.w;
        }
        io<T>::st4(x + o, e);
    }
}

// block = CW threads (one column each); grid = (H/CW, nchunks); LDS table [max_F][CW]
template <typename T>
__global__ void stitch_bwd_kernel(const T* __restrict__ dx, const T* __restrict__ dextra, const int64_t* __restrict__ ts,
                                  const uint8_t* __restrict__ keep0, mmfm_dropout dropa, T* __restrict__ d_tok,
                                  float* __restrict__ part, int B, int Tn, int L, int m, int H, int max_F, int bper) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    const int CW = blockDim.x, j = threadIdx.x, col = blockIdx.x * CW + j;
    const Drop dr = drop_init(dropa);
    for (int f = 0; f < max_F; ++f) tab[f * CW + j] = 0.f;
    float macc = 0.f;
    const int b0 = blockIdx.y * bper, b1 = min(B, b0 + bper);
    for (int b = b0; b < b1; ++b) {
        for (int t = 0; t < Tn; ++t) {
            const int l = m * Tn + t;
            const size_t o = ((size_t)b * L + l) * H + col;
            const float g = io<T>::ld(dx + o);
            float e = g;
            if (dextra) e += io<T>::ld(dextra + o);
            const int64_t r = (int64_t)b * Tn + t;
            const int tsv = (int)min((int64_t)(max_F - 1), max((int64_t)0, ts[r]));
            tab[tsv * CW + j] += e;                   // a thread owns its column: no race, fixed order
            macc += e;
            if (d_tok) io<T>::st(d_tok + (size_t)r * H + col, keep0[l] ? dr.apply(g, (uint64_t)r * H + col) : 0.f);
        }
    }
    float* out = part + (size_t)blockIdx.y * (max_F + 1) * H;
    for (int f = 0; f < max_F; ++f) out[(size_t)f * H + col] = tab[f * CW + j];
    out[(size_t)max_F * H + col] = macc;
}

// ---- bf16 throughput path of mmfm_stitch_bwd: the scatter d_pos[ts[b][t]] += e[b][t] and the sum d_mod += e ARE a matrix
// product with a one-hot matrix:  [d_pos; d_mod] = OH^T E,  OH[row][f] = (f == ts[row]), OH[row][max_F] = 1 for the rows of
// modality m (zero rows elsewhere), E = dx viewed as [B*L][H].  It runs as a split-K dW-style launch of the bf16 MFMA GEMM
// (fp32 accumulation of exact 0/1 products = the plain fp32 sum, fixed order), 1 + 1 launches for dx and dextra; measured
// 429 us -> ~90 us per call at B = 1024 against the per-column LDS scatter kernel above (2-byte accesses, 4 waves/CU).
__global__ __launch_bounds__(256) void onehot_kernel(const int64_t* __restrict__ ts, uint16_t* __restrict__ oh, int B, int Tn, int L, int m,
                                                     int max_F, int ohc) {
    const int cpr = ohc / 8;
    const int64_t total = (int64_t)B * L * cpr;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / cpr;
        const int c0 = (int)(idx % cpr) * 8;
        const int b = (int)(row / L), l = (int)(row % L), t = l - m * Tn;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (t >= 0 && t < Tn) {
            const int tsv = (int)min((int64_t)(max_F - 1), max((int64_t)0, ts[(int64_t)b * Tn + t]));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j == tsv || c0 + j == max_F) w[j >> 1] |= 0x3F80u << (16 * (j & 1));      // bf16 1.0
        }
        *reinterpret_cast<uint4*>(oh + row * ohc + c0) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// d_tok[b*T+t][:] = keep0[m*T+t] ? dropout'(dx[b][m*T+t][:]) : 0      (bf16, 16-B accesses; H % 8 == 0)
__global__ __launch_bounds__(256) void stitch_dtok_kernel(const uint16_t* __restrict__ dx, const uint8_t* __restrict__ keep0, mmfm_dropout dropa,
                                                          uint16_t* __restrict__ d_tok, int B, int Tn, int L, int m, int H) {
    const Drop dr = drop_init(dropa);
    const int C8 = H / 8;
    const int64_t total = (int64_t)B * Tn * C8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / C8;
        const int c = (int)(idx % C8) * 8;
        const int b = (int)(r / Tn), l = m * Tn + (int)(r % Tn);
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (keep0[l]) {
            const uint4 g = *reinterpret_cast<const uint4*>(dx + ((size_t)b * L + l) * H + c);
            if (dr.on()) {
                const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
                uint32_t ow[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v0 = dr.apply(__uint_as_float(gw[j] << 16), (uint64_t)r * H + c + 2 * j);
                    const float v1 = dr.apply(__uint_as_float(gw[j] & 0xffff0000u), (uint64_t)r * H + c + 2 * j + 1);
                    ow[j] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                }
                o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else {
                o = g;
            }
        }
        *reinterpret_cast<uint4*>(d_tok + (size_t)r * H + c) = o;
    }
}

// Loader collate (loader/base.py:304-450, utils/dataset_utils.py:38-43): CSR (uint8 counts) ->
// dense [B][max_T][max_N] fp32, truncated / right-padded with pad_value, plus the two attention masks.
// One wavefront per (trial, time bin): the lanes fill the row (coalesced), then scatter the row's non-zeros with
// float atomic adds (duplicate (row, col) entries of a CSR add up, like scipy's toarray()).
struct CollateArgs {
    const uint8_t* data;
    const int32_t* indices;
    const int64_t* indptr;        // concatenated per-trial indptr arrays (T_b + 1 entries each)
    const int64_t* indptr_off;    // [B] offset of trial b's indptr
    const int64_t* nnz_off;       // [B] offset of trial b's data/indices
    const int32_t* T_b;
    const int32_t* N_b;
};
__global__ __launch_bounds__(256) void collate_csr_kernel(CollateArgs a, int B, int max_T, int max_N, float pad, float* __restrict__ out,
                                                          int64_t* __restrict__ tmask, int64_t* __restrict__ smask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < (int64_t)B * max_T; row += (int64_t)gridDim.x * 4) {
        const int b = (int)(row / max_T), t = (int)(row % max_T);
        const int Tb = a.T_b[b], Nb = a.N_b[b];
        const bool live = t < Tb;                       // rows past the trial's length are padding
        const int nvalid = min(Nb, max_N);
        float* o = out + row * max_N;
        for (int n = lane; n < max_N; n += 64) o[n] = (live && n < nvalid) ? 0.f : pad;
        if (lane == 0) tmask[row] = live ? 1 : 0;
        if (t == 0)
            for (int n = lane; n < max_N; n += 64) smask[(size_t)b * max_N + n] = n < nvalid ? 1 : 0;
        if (!live) continue;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        const int64_t* ip = a.indptr + a.indptr_off[b];
        const int64_t z0 = ip[t], z1 = ip[t + 1], base = a.nnz_off[b];
        for (int64_t z = z0 + lane; z < z1; z += 64) {
            const int col = a.indices[base + z];
            if (col >= 0 && col < nvalid) atomicAdd(o + col, (float)a.data[base + z]);
        }
    }
}

int pick_cw(int H, int max_F) {
    for (int cw : {256, 128, 64, 32, 16, 8, 4})
        if (H % cw == 0 && (size_t)max_F * cw * 4 <= 60 * 1024) return cw;
    return 0;
}
int stitch_chunks(int B, int H, int cw) { return std::max(1, std::min(B, 512 / std::max(1, H / cw))); }
// one-hot GEMM path (bf16, H % 8 == 0): split-K geometry over K = B*L rows and the workspace carve-up
struct OhGeo { int ohc, S, kchunk; int64_t oh_bytes, slab; };
OhGeo oh_geo(int B, int L, int H, int max_F) {
    OhGeo g;
    g.ohc = (max_F + 1 + 7) & ~7;
    const int64_t K = (int64_t)B * L;
    const int tiles = ((max_F + 1 + 127) / 128) * ((H + 127) / 128);
    int S = (int)std::max<int64_t>(1, std::min<int64_t>(K / 512, (512 + tiles - 1) / tiles));
    g.kchunk = (int)(((K + S - 1) / S + 63) / 64 * 64);
    g.S = (int)((K + g.kchunk - 1) / g.kchunk);
    g.oh_bytes = ((int64_t)K * g.ohc * 2 + 255) / 256 * 256;
    g.slab = (int64_t)(max_F + 1) * H;
    return g;
}
bool oh_path(int dtype, int H) { return dtype == MMFM_BF16 && H % 8 == 0; }

}  // namespace

extern "C" int mmfm_reduce_slabs(float*, const float*, int64_t, int, int64_t, int, mmfm_stream);
int mmfm_gemm_bf16_launch(const mmfm_gemm_desc* dp, hipStream_t st);       // gemm_bf16.hip

extern "C" int mmfm_mask_prep(int B, int T, int M, const int64_t* const* mask_src, const int64_t* mask_stride,
                              const int64_t* attn, const int64_t* channels, uint8_t* tokmask, uint8_t* keypad,
                              uint8_t* keep0, uint8_t* mod_id, int64_t* count, mmfm_stream stream) {
    MMFM_REQUIRE(B > 0 && T > 0 && M > 0 && M <= MAXM, "mmfm_mask_prep: bad shape B=%d T=%d M=%d (M <= %d)", B, T, M, MAXM);
    MMFM_REQUIRE(mask_src && mask_stride && attn && channels && tokmask && keypad && keep0 && mod_id && count, "mmfm_mask_prep: null pointer");
    MMFM_REQUIRE(M * T <= 65535, "mmfm_mask_prep: sequence too long");
    MaskSrc src;
    for (int m = 0; m < MAXM; ++m) {
        src.p[m] = m < M ? mask_src[m] : nullptr;
        src.stride[m] = m < M ? mask_stride[m] : 0;
        src.channels[m] = m < M ? channels[m] : 0;
        MMFM_REQUIRE(m >= M || (src.p[m] && src.stride[m] > 0), "mmfm_mask_prep: modality %d has no mask source", m);
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_u64_kernel, dim3(1), dim3(64), 0, st, (unsigned long long*)count, M);
    MMFM_LAUNCH_CHECK("mmfm_mask_prep(zero)");
    const int64_t n = (int64_t)B * M * T;
    hipLaunchKernelGGL(mask_prep_kernel, dim3((int)std::min<int64_t>(256, (n + 255) / 256)), dim3(256), 0, st, src, attn, B, T, M,
                       tokmask, keypad, keep0, mod_id, (unsigned long long*)count);
    MMFM_LAUNCH_CHECK("mmfm_mask_prep");
    return 0;
}

extern "C" int mmfm_collate_csr(int B, int max_T, int max_N, float pad_value, const uint8_t* data, const int32_t* indices,
                                const int64_t* indptr, const int64_t* indptr_off, const int64_t* nnz_off, const int32_t* T_b,
                                const int32_t* N_b, float* out, int64_t* time_mask, int64_t* space_mask, mmfm_stream stream) {
    MMFM_REQUIRE(B > 0 && max_T > 0 && max_N > 0, "mmfm_collate_csr: bad shape");
    MMFM_REQUIRE(indptr && indptr_off && nnz_off && T_b && N_b && out && time_mask && space_mask, "mmfm_collate_csr: null pointer");
    CollateArgs a{data, indices, indptr, indptr_off, nnz_off, T_b, N_b};
    const int64_t rows = (int64_t)B * max_T;
    hipLaunchKernelGGL(collate_csr_kernel, dim3((int)std::min<int64_t>(4096, (rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, B, max_T,
                       max_N, pad_value, out, time_mask, space_mask);
    MMFM_LAUNCH_CHECK("mmfm_collate_csr");
    return 0;
}

extern "C" int mmfm_stitch_fwd(int dtype, const void* tok, const float* mod_emb_row, const float* pos_emb, const int64_t* ts,
                               const uint8_t* keep0, void* x, void* emb, int B, int T, int L, int m, int H, int max_F, mmfm_stream stream) {
    MMFM_REQUIRE(tok && mod_emb_row && pos_emb && ts && keep0 && x, "mmfm_stitch_fwd: null pointer");
    MMFM_REQUIRE(B > 0 && T > 0 && H > 0 && H % 4 == 0 && m >= 0 && (m + 1) * T <= L && max_F > 0, "mmfm_stitch_fwd: bad shape");
    const int64_t n = (int64_t)B * T * (H / 4);
    dim3 grid((int)std::min<int64_t>(4096, (n + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(stitch_fwd_kernel<float>, grid, block, 0, st, (const float*)tok, mod_emb_row, pos_emb, ts, keep0, (float*)x, (float*)emb, B, T, L, m, H, max_F);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(stitch_fwd_kernel<uint16_t>, grid, block, 0, st, (const uint16_t*)tok, mod_emb_row, pos_emb, ts, keep0, (uint16_t*)x, (uint16_t*)emb, B, T, L, m, H, max_F);
    else
        return mmfm_set_error(-1, "mmfm_stitch_fwd: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_stitch_fwd");
    return 0;
}

extern "C" int64_t mmfm_stitch_bwd_workspace(int dtype, int B, int T, int L, int H, int max_F) {
    (void)T;
    if (oh_path(dtype, H)) {
        const OhGeo g = oh_geo(B, L, H, max_F);
        return g.oh_bytes + 2 * (int64_t)g.S * g.slab * (int64_t)sizeof(float);
    }
    const int cw = pick_cw(H, max_F);
    if (!cw) return -1;
    return (int64_t)stitch_chunks(B, H, cw) * (max_F + 1) * H * sizeof(float);
}

extern "C" int mmfm_stitch_bwd(int dtype, const void* dx, const void* dextra, const int64_t* ts, const uint8_t* keep0,
                               mmfm_dropout drop, void* d_tok, float* d_mod_row, float* d_pos, int acc_mod, int acc_pos, int B, int T,
                               int L, int m, int H, int max_F, void* workspace, int64_t workspace_bytes, mmfm_stream stream) {
    MMFM_REQUIRE(dx && ts && keep0 && d_mod_row && d_pos, "mmfm_stitch_bwd: null pointer");
    MMFM_REQUIRE(B > 0 && T > 0 && H > 0 && max_F > 0 && m >= 0 && (m + 1) * T <= L, "mmfm_stitch_bwd: bad shape");
    if (oh_path(dtype, H)) {
        const OhGeo g = oh_geo(B, L, H, max_F);
        MMFM_REQUIRE(workspace && workspace_bytes >= mmfm_stitch_bwd_workspace(dtype, B, T, L, H, max_F), "mmfm_stitch_bwd: workspace too small");
        MMFM_REQUIRE((uintptr_t)dx % 16 == 0 && (!dextra || (uintptr_t)dextra % 16 == 0) && (!d_tok || (uintptr_t)d_tok % 16 == 0) &&
                     (uintptr_t)workspace % 16 == 0, "mmfm_stitch_bwd: bf16 tensors must be 16-byte aligned");
        hipStream_t st = (hipStream_t)stream;
        uint16_t* oh = reinterpret_cast<uint16_t*>(workspace);
        float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + g.oh_bytes);
        const int64_t nch = (int64_t)B * L * (g.ohc / 8);
        hipLaunchKernelGGL(onehot_kernel, dim3((int)std::min<int64_t>(4096, (nch + 255) / 256)), dim3(256), 0, st, ts, oh, B, T, L, m, max_F, g.ohc);
        if (d_tok) {
            const int64_t n = (int64_t)B * T * (H / 8);
            hipLaunchKernelGGL(stitch_dtok_kernel, dim3((int)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, st, (const uint16_t*)dx, keep0,
                               drop, (uint16_t*)d_tok, B, T, L, m, H);
        }
        MMFM_LAUNCH_CHECK("mmfm_stitch_bwd(one-hot)");
        mmfm_gemm_desc gd = {};
        gd.dtype = MMFM_BF16; gd.c_f32 = 1;
        gd.A = oh; gd.lda = g.ohc; gd.a_kcontig = 0;
        gd.ldb = H; gd.b_kcontig = 0;
        gd.M = max_F + 1; gd.N = H; gd.K = B * L; gd.ldc = H;
        gd.splits = g.S; gd.kchunk = g.kchunk; gd.slab_stride = g.slab;
        gd.act_scale = 1.f;
        int nsl = 0;
        for (const void* src : {dx, dextra}) {
            if (!src) continue;
            gd.B = src;
            gd.C = slabs + (int64_t)nsl * g.slab;
            if (int rc = mmfm_gemm_bf16_launch(&gd, st)) return rc;
            nsl += g.S;
        }
        if (int rc = mmfm_reduce_slabs(d_pos, slabs, (int64_t)max_F * H, nsl, g.slab, acc_pos, stream)) return rc;
        return mmfm_reduce_slabs(d_mod_row, slabs + (int64_t)max_F * H, H, nsl, g.slab, acc_mod, stream);
    }
    const int cw = pick_cw(H, max_F);
    MMFM_REQUIRE(cw > 0, "mmfm_stitch_bwd: no column slab fits LDS for H=%d max_F=%d", H, max_F);
    const int nch = stitch_chunks(B, H, cw);
    MMFM_REQUIRE(workspace && workspace_bytes >= mmfm_stitch_bwd_workspace(dtype, B, T, L, H, max_F), "mmfm_stitch_bwd: workspace too small");
    const int bper = (B + nch - 1) / nch;
    dim3 grid(H / cw, nch), block(cw);
    const size_t lds = (size_t)max_F * cw * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MMFM_F32)
        hipLaunchKernelGGL(stitch_bwd_kernel<float>, grid, block, lds, st, (const float*)dx, (const float*)dextra, ts, keep0, drop, (float*)d_tok,
                           (float*)workspace, B, T, L, m, H, max_F, bper);
    else if (dtype == MMFM_BF16)
        hipLaunchKernelGGL(stitch_bwd_kernel<uint16_t>, grid, block, lds, st, (const uint16_t*)dx, (const uint16_t*)dextra, ts, keep0, drop,
                           (uint16_t*)d_tok, (float*)workspace, B, T, L, m, H, max_F, bper);
    else
        return mmfm_set_error(-1, "mmfm_stitch_bwd: bad dtype %d", dtype);
    MMFM_LAUNCH_CHECK("mmfm_stitch_bwd");
    const int64_t stride = (int64_t)(max_F + 1) * H;
    if (int rc = mmfm_reduce_slabs(d_pos, (const float*)workspace, (int64_t)max_F * H, nch, stride, acc_pos, stream)) return rc;
    return mmfm_reduce_slabs(d_mod_row, (const float*)workspace + (size_t)max_F * H, H, nch, stride, acc_mod, stream);
}
