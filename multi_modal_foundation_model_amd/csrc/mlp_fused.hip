// mmfm_mlp_fwd / mmfm_mlp_bwd / mmfm_ln_linear_grad: the transformer MLP block as ONE launch each way (bf16, width 256 -> 512 -> 256).
// Reference: MLP.forward (mm_utils.py:50-52) behind ln2 with the residual add (encoder_embeddings.py:114,
// decoder_embeddings.py:145) and its autograd.  Structure: rowchain.h (a wavefront owns 32 token rows end to end).
//
// forward, per 32-row tile of a wave:
//   x (16 operands) -> LayerNorm in registers -> for each of the 16 intermediate tiles t:
//        U_t  = Wp_up[32t..][:] . x_hat            16 MFMAs, one weight chunk [32][256]
//        g_t  = gelu(U_t + b)  -> two bf16 operands (accumulator tile = next product's operand)
//        Y   += W_down[:, 32t..32t+32] . g_t       16 MFMAs into the 8 output tiles, one chunk [256][32]
//   y = x + dropout(Y + b_down): HBM traffic 2 x R x 256 x 2 B (+ the x_hat side output for the backward) instead of the
//   un-fused LN (2) + up (5) + down (4) = 11 x R x 256 x 2 B.
// backward recomputes U_t / g_t from the saved x_hat (48 MFMAs per t instead of 32) and does the LayerNorm backward on the
// accumulated d(x_hat) row in registers.
#include "rowchain.h"
#include <stdlib.h>
#include <algorithm>

using namespace rowchain;

namespace {

constexpr int NT = 256, NW = 4;

__global__ __launch_bounds__(NT) void mlp_fwd_kernel(const mmfm_mlp_desc d) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + NW * STG_BYTES + 768 * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, h = lane >> 5;
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const uint16_t* Wup = reinterpret_cast<const uint16_t*>(d.w_up);
    const uint16_t* Wdn = reinterpret_cast<const uint16_t*>(d.w_down);
    const int rot = d.rotate ? (int)(blockIdx.x & 15) : 0;
    auto src = [=](int g) {
        const int idx = g & 31, tt = ((idx >> 1) + rot) & 15;
        WChunk c;
        if (idx & 1) { c.base = Wdn + 32 * tt; c.ld = 512; c.kind = 2; }
        else { c.base = Wup + (size_t)(32 * tt) * 256; c.ld = 256; c.kind = 0; }
        return c;
    };
    char* stg = smem + LDS_BYTES + wave * STG_BYTES;
    float* lb_up = reinterpret_cast<float*>(smem + LDS_BYTES + NW * STG_BYTES);
    float* lb_dn = lb_up + 512;
    stage_vec(lb_up, d.b_up, 512, t, NT);
    stage_vec(lb_dn, d.b_down, 256, t, NT);
    const Drop dr = drop_init(d.drop);
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2), XH = gbuf(d.xhat, d.R * 512), RS = gbuf(d.rstd, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2;
    RING_DECL(NT);
    RING_START(smem, my_passes * 32, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        const uint32_t row = wrow0 + m;
        opnd x[16];
        load_rows_lines<4>(stg, x, X, wrow0, ldxb, lane, m, h);
        const float rs = ln_rows(x, d.eps);
        store_rows_lines<4, false>(stg, XH, wrow0, 512u, lane, m, h, x);
        st4f(RS, h == 0 ? row * 4u : 0xfffffff0u, rs);
        f32x16 Y8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) Y8[i] = zero16();
        for (int ti = 0; ti < 16; ++ti) {
            const int tt = (ti + rot) & 15;
            const char* slot;
            RING_STEP(src, slot);
            f32x16 U = mma16(slot, x, zero16(), m, h);
            add_vec(U, lb_up, tt, h);
#pragma unroll
            for (int i = 0; i < 16; ++i) U[i] = gelu_fast(U[i]);
            opnd g0, g1;
            acc_to_opnd(U, g0, g1);
            RING_STEP(src, slot);
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) {
                Y8[t2] = mfma(wfragB(slot, t2, 0, m, h), g0, Y8[t2]);
                Y8[t2] = mfma(wfragB(slot, t2, 1, m, h), g1, Y8[t2]);
            }
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const Lines xl = fetch_lines(X, wrow0, ldxb, 128u * tp, lane);
            stage_lines(stg, xl, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t2 = 2 * tp + j;
                add_vec(Y8[t2], lb_dn, t2, h);
                if (dr.on()) {
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        Y8[t2][i] = dr.keep((uint64_t)row * 256u + (uint64_t)feat(t2, i, h)) ? Y8[t2][i] * dr.scale : 0.f;
                }
                const f32x16 r = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) Y8[t2][i] += r[i];
            }
            stage_tile(stg, 0, m, h, Y8[2 * tp]);
            stage_tile(stg, 1, m, h, Y8[2 * tp + 1]);
            flush_lines<false>(stg, Y, wrow0, ldyb, 128u * tp, lane);
        }
    }
}

__global__ __launch_bounds__(NT) void mlp_bwd_kernel(const mmfm_mlp_desc d) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + 3 * NW * STG_BYTES + 512 * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, h = lane >> 5;
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const uint16_t* Wup = reinterpret_cast<const uint16_t*>(d.w_up);
    const uint16_t* WdnT = reinterpret_cast<const uint16_t*>(d.w_down_t);
    const uint16_t* WupT = reinterpret_cast<const uint16_t*>(d.w_up_t);
    const int rot = d.rotate ? (int)(blockIdx.x & 7) * 2 : 0;            // even: tile pairs of g / du complete together
    auto src = [=](int g) {
        const int idx = g % 48, ti = idx / 3, k = idx - 3 * ti, tt = (ti + rot) & 15;
        WChunk c;
        if (k == 0) { c.base = Wup + (size_t)(32 * tt) * 256; c.ld = 256; c.kind = 0; }
        else if (k == 1) { c.base = WdnT + (size_t)(32 * tt) * 256; c.ld = 256; c.kind = 0; }
        else { c.base = WupT + 32 * tt; c.ld = 512; c.kind = 2; }
        return c;
    };
    char* stg = smem + LDS_BYTES + wave * STG_BYTES;
    char* stg_g = smem + LDS_BYTES + (NW + wave) * STG_BYTES;
    char* stg_du = smem + LDS_BYTES + (2 * NW + wave) * STG_BYTES;
    float* lb_up = reinterpret_cast<float*>(smem + LDS_BYTES + 3 * NW * STG_BYTES);
    stage_vec(lb_up, d.b_up, 512, t, NT);
    const Drop dr = drop_init(d.drop);
    const GBuf XH = gbuf(d.xhat, d.R * 512), RS = gbuf(d.rstd, d.R * 4), DY = gbuf(d.dy, d.R * d.lddy * 2), T1 = gbuf(d.t1, d.R * 512);
    const GBuf G = gbuf(d.g, d.R * 1024), DU = gbuf(d.du, d.R * 1024), DX = gbuf(d.dx, d.R * d.lddx * 2);
    const uint32_t lddyb = d.lddy * 2, lddxb = d.lddx * 2;
    RING_DECL(NT);
    RING_START(smem, my_passes * 48, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        const uint32_t row = wrow0 + m;
        opnd x[16], t1[16];
        load_rows_lines<4>(stg, x, XH, wrow0, 512u, lane, m, h);
        load_rows_lines<4>(stg, t1, DY, wrow0, lddyb, lane, m, h);
        const float rs = ld4f(RS, row * 4u);
        if (dr.on()) {                                      // dropout'(dy): counter row*256 + k, k = 16s + 8h + j
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float f[8]; unpack8f(t1[s], f);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    f[j] = dr.keep((uint64_t)row * 256u + (uint64_t)(16 * s + 8 * h + j)) ? f[j] * dr.scale : 0.f;
                t1[s] = pack8o(f);
            }
        }
        store_rows_lines<4, true>(stg, T1, wrow0, 512u, lane, m, h, t1);
        f32x16 DH[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) DH[i] = zero16();
        for (int ti = 0; ti < 16; ++ti) {
            const int tt = (ti + rot) & 15;
            const char* slot;
            RING_STEP(src, slot);
            f32x16 U = mma16(slot, x, zero16(), m, h);
            add_vec(U, lb_up, tt, h);
            RING_STEP(src, slot);
            f32x16 DG = mma16(slot, t1, zero16(), m, h);
            f32x16 Gt;
#pragma unroll
            for (int i = 0; i < 16; ++i) { Gt[i] = gelu_fast(U[i]); DG[i] *= gelu_grad_fast(U[i]); }
            opnd d0, d1;
            acc_to_opnd(DG, d0, d1);
            RING_SYNC_WRITE(src);
            stage_tile(stg_g, ti & 1, m, h, Gt);
            stage_tile(stg_du, ti & 1, m, h, DG);
            if (ti & 1) {                                     // uniform: the pair (tt-1, tt) is complete -> whole 128-B lines
                flush_lines<true>(stg_g, G, wrow0, 1024u, 64u * (tt - 1), lane);
                flush_lines<true>(stg_du, DU, wrow0, 1024u, 64u * (tt - 1), lane);
            }
            RING_FETCH(src, slot);
#pragma unroll
            for (int t2 = 0; t2 < 8; ++t2) {
                DH[t2] = mfma(wfragB(slot, t2, 0, m, h), d0, DH[t2]);
                DH[t2] = mfma(wfragB(slot, t2, 1, m, h), d1, DH[t2]);
            }
        }
        // LayerNorm backward on the row: dx = dy + rstd * (dh - mean(dh) - x_hat * mean(dh * x_hat))
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const Lines xl = fetch_lines(XH, wrow0, 512u, 128u * tp, lane);
            stage_lines(stg, xl, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += DH[2 * tp + j][i]; s2 = fmaf(DH[2 * tp + j][i], xt[i], s2); }
            }
        }
        s1 = xhalf(s1) * (1.f / 256.f);
        s2 = xhalf(s2) * (1.f / 256.f);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const Lines xl = fetch_lines(XH, wrow0, 512u, 128u * tp, lane);
            const Lines rl = fetch_lines(DY, wrow0, lddyb, 128u * tp, lane);
            stage_lines(stg, xl, lane);
            f32x16 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[j][i] = rs * (DH[2 * tp + j][i] - s1 - xt[i] * s2);
            }
            stage_lines(stg, rl, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 r = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[j][i] += r[i];
            }
            stage_tile(stg, 0, m, h, o[0]);
            stage_tile(stg, 1, m, h, o[1]);
            flush_lines<false>(stg, DX, wrow0, lddxb, 128u * tp, lane);
        }
    }
}

// dW = gamma * G + db x beta; dgamma = colsum(W * G); dbeta = W^T db.  One block per 32 columns k, 8 row groups.
__global__ __launch_bounds__(256) void ln_linear_grad_kernel(const float* __restrict__ Gdb, const float* __restrict__ W,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int N, int K,
                                                            float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate) {
    __shared__ float red[2][8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, k = blockIdx.x * 32 + tx;
    const float* db = Gdb + (size_t)N * K;
    float ag = 0.f, ab = 0.f;
    if (k < K) {
        const float g = gamma[k], b = beta[k];
        for (int n = ty; n < N; n += 8) {
            const float Gv = Gdb[(size_t)n * K + k], w = W[(size_t)n * K + k], dbn = db[n];
            dW[(size_t)n * K + k] = fmaf(g, Gv, dbn * b);
            ag = fmaf(w, Gv, ag);
            ab = fmaf(w, dbn, ab);
        }
    }
    red[0][ty][tx] = ag; red[1][ty][tx] = ab;
    __syncthreads();
    if (ty == 0 && k < K) {
        float sg = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sg += red[0][i][tx]; sb += red[1][i][tx]; }
        dgamma[k] = accumulate ? dgamma[k] + sg : sg;
        dbeta[k] = accumulate ? dbeta[k] + sb : sb;
    }
    if (blockIdx.x == 0) for (int n = threadIdx.x; n < N; n += 256) dbias[n] = db[n];
}

int grid_for(int64_t R, int per_cu) {
    const int64_t npass = (R + 32 * NW - 1) / (32 * NW);
    return (int)std::max<int64_t>(1, std::min<int64_t>(npass, 256 * per_cu));
}

int check(const mmfm_mlp_desc& d, bool bwd) {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    MMFM_REQUIRE(d.R > 0 && d.w_up && d.b_up, "mmfm_mlp: null operand / empty problem");
    MMFM_REQUIRE((d.R + 128) * (int64_t)std::max(std::max(d.ldx, d.ldy), std::max(std::max(d.lddy, d.lddx), 512)) * 2 < (int64_t)1 << 31,
                 "mmfm_mlp: tensors beyond 2 GiB are not addressable by the 32-bit buffer offsets");
    MMFM_REQUIRE(al16(d.x) && al16(d.w_up) && al16(d.w_down) && al16(d.y) && al16(d.xhat) && al16(d.dy) && al16(d.w_down_t) && al16(d.w_up_t) &&
                 al16(d.t1) && al16(d.g) && al16(d.du) && al16(d.dx) && al16(d.b_up) && al16(d.b_down), "mmfm_mlp: operands must be 16-byte aligned");
    if (!bwd) MMFM_REQUIRE(d.x && d.w_down && d.b_down && d.y && d.ldx % 8 == 0 && d.ldy % 8 == 0 && d.ldx >= 256 && d.ldy >= 256, "mmfm_mlp_fwd: bad arguments");
    else MMFM_REQUIRE(d.xhat && d.rstd && d.dy && d.w_down_t && d.w_up_t && d.g && d.du && d.dx && d.lddy % 8 == 0 && d.lddx % 8 == 0 && d.lddy >= 256 &&
                      d.lddx >= 256, "mmfm_mlp_bwd: bad arguments");
    return 0;
}

}  // namespace

extern "C" int mmfm_mlp_fwd(const mmfm_mlp_desc* dp, mmfm_stream stream) {
    const mmfm_mlp_desc d = *dp;
    if (int rc = check(d, false)) return rc;
    static const int per_cu = [] { const char* e = getenv("MMFM_MLP_WG_PER_CU"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    hipLaunchKernelGGL(mlp_fwd_kernel, dim3(grid_for(d.R, per_cu)), dim3(NT), 0, (hipStream_t)stream, d);
    MMFM_LAUNCH_CHECK("mmfm_mlp_fwd");
    return 0;
}

extern "C" int mmfm_mlp_bwd(const mmfm_mlp_desc* dp, mmfm_stream stream) {
    const mmfm_mlp_desc d = *dp;
    if (int rc = check(d, true)) return rc;
    static const int per_cu = [] { const char* e = getenv("MMFM_MLP_WG_PER_CU"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(grid_for(d.R, per_cu)), dim3(NT), 0, (hipStream_t)stream, d);
    MMFM_LAUNCH_CHECK("mmfm_mlp_bwd");
    return 0;
}

extern "C" int mmfm_ln_linear_grad(const float* Gdb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW,
                                   float* dbias, float* dgamma, float* dbeta, int accumulate_ln, mmfm_stream stream) {
    MMFM_REQUIRE(Gdb && W && gamma && beta && dW && dbias && dgamma && dbeta && N > 0 && K > 0, "mmfm_ln_linear_grad: null argument");
    hipLaunchKernelGGL(ln_linear_grad_kernel, dim3(cdiv(K, 32)), dim3(256), 0, (hipStream_t)stream, Gdb, W, gamma, beta, N, K, dW, dbias, dgamma,
                       dbeta, accumulate_ln);
    MMFM_LAUNCH_CHECK("mmfm_ln_linear_grad");
    return 0;
}
