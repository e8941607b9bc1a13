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
// accumulated d(x_hat) row in registers - or, with dx == NULL (the engine's default since round 4), stops after t1 / g / du in an
// eight-wave kernel and leaves d(x_hat) + the LayerNorm backward to mmfm_rowgemm(ln_bwd, K = 512).
#ifdef MMFM_STAMP
#define RING_BARRIER_STAMP STAMP(7)
#endif
#include "rowchain.h"
#include <stdlib.h>
#include <algorithm>

using namespace rowchain;

#ifdef MMFM_STAMP
// diagnostic build only (scripts/probe/build_stamp.sh): per-phase shader-cycle totals of wave 0 of every workgroup
__device__ unsigned long long mmfm_probe_acc[16];
#define STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_a[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_a[i] += n_ - st_t; st_t = n_; } while (0)
#define STAMP_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&mmfm_probe_acc[i_], st_a[i_]); } while (0)
extern "C" int mmfm_probe_read(unsigned long long* host16, int reset) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(mmfm_probe_acc), 128);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mmfm_probe_acc), z, 128); }
    return 0;
}
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH
#endif

namespace {

constexpr int NT = 256, NW = 4;

// MMFM_PROBE (diagnostic builds only, scripts/probe/build_probe.sh): bit 0 drops the g / du stores of the loop, bit 1 the GELU algebra
#ifndef MMFM_PROBE
#define MMFM_PROBE 0
#endif
__global__ __launch_bounds__(NT) void mlp_bwd_kernel(const mmfm_mlp_desc d) {
    // LDS: three 16 KB ring slots (asynchronous ring, rowchain.h) | per-wave staging for g and du (the g area doubles as the
    // prologue / epilogue staging) | b_up | a 16 KB per-wave stash of the pass's x_hat rows (four [32 rows][128 B] staging images):
    // the LayerNorm-backward epilogue reads x_hat twice more, and fetching it from memory again put two more dependent round trips
    // (and 210 MB) into every pass
    constexpr int RING_B = RINGA_SLOTS * CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];        // RING_B + 2 * NW * STG_BYTES + 512 * 4 + NW * 4 * STG_BYTES
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    char* xstash = smem + RING_B + 2 * NW * STG_BYTES + 512 * 4 + wave * 4 * STG_BYTES;
    const __amdgpu_buffer_rsrc_t rs_up = wbuf(d.w_up), rs_dnt = wbuf(d.w_down_t), rs_upt = wbuf(d.w_up_t);
    const int rot = d.rotate ? (int)(blockIdx.x & 7) * 2 : 0;            // even: tile pairs of g / du complete together
    // chunk sequence of a tile ti: Wp_up rows (recompute u), W_down^T rows (dg), Wp_up^T columns (d x_hat; unit-permuted copy)
    auto src = [=](int g) {
        const int idx = g % 48;                             // up(0) dg(0) | up(ti) dg(ti) dh(ti-1), ti = 1..15 | dh(15)
        int ti, k;
        if (idx < 2) { ti = 0; k = idx; }
        else if (idx == 47) { ti = 15; k = 2; }
        else { const int j = idx - 2; ti = j / 3 + 1; k = j - 3 * (ti - 1); if (k == 2) --ti; }
        const int tt = (ti + rot) & 15;
        AChunk c;
        if (k == 0) { c.rs = rs_up; c.off = (uint32_t)(32 * tt) * 512u; c.ldb = 512u; c.kind = 0; }
        else if (k == 1) { c.rs = rs_dnt; c.off = (uint32_t)(32 * tt) * 512u; c.ldb = 512u; c.kind = 0; }
        else { c.rs = rs_upt; c.off = (uint32_t)(32 * tt) * 2u; c.ldb = 1024u; c.kind = 2; }
        return c;
    };
    char* stg_g = smem + RING_B + wave * STG_BYTES;
    char* stg_du = smem + RING_B + (NW + wave) * STG_BYTES;
    char* stg = stg_g;                                                   // prologue / epilogue staging: the loop's g area is idle then
    float* lb_up = reinterpret_cast<float*>(smem + RING_B + 2 * NW * STG_BYTES);
    stage_vec(lb_up, d.b_up, 512, t, NT);
    const Drop dr = drop_init(d.drop);
    const GBuf XH = gbuf(d.xhat, d.R * 512), RS = gbuf(d.rstd, d.R * 4), DY = gbuf(d.dy, d.R * d.lddy * 2), T1 = gbuf(d.t1, d.R * 512);
    const GBuf G = gbuf(d.g, d.R * 1024), DU = gbuf(d.du, d.R * 1024), DX = gbuf(d.dx, d.R * d.lddx * 2);
    const uint32_t lddyb = d.lddy * 2, lddxb = d.lddx * 2;
    const ALane<NT> ring_al = alane_init<NT>(t, 512u, 1024u);
    const AFrag fr = afrag_init(m, h);
    RINGA_DECL(NT);
    RINGA_START((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, my_passes * 48, src);
    STAMP_DECL;
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        const uint32_t row = wrow0 + m;
        opnd x[16], t1[16];
        {   // x_hat: fetched as whole lines into the stash (which doubles as the transposition area for the operand reads) and kept;
            // the dy rows are requested in the same breath (one round trip for both row blocks)
            Lines L[4], Ld[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) L[q] = fetch_lines(XH, wrow0, 512u, 128u * q, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) Ld[q] = fetch_lines(DY, wrow0, lddyb, 128u * q, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                stage_lines(xstash + q * STG_BYTES, L[q], lane);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x[4 * q + s4] = unstage_opnd(xstash + q * STG_BYTES, s4, m, h);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                stage_lines(stg, Ld[q], lane);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t1[4 * q + s4] = unstage_opnd(stg, s4, m, h);
            }
        }
        const float rs = ld4f(RS, row * 4u);
        if (dr.on()) {                                      // dropout'(dy): the row's keys, then features k = 16s + 8h + j (rowchain.h RowDrop)
            const RowDrop rd = rowdrop_init(dr, row);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float f[8]; unpack8f(t1[s], f);
                drop8(rd, f, 16 * s + 8 * h);
                t1[s] = pack8o(f);
            }
        }
        store_rows_lines<4, true>(stg, T1, wrow0, 512u, lane, m, h, t1);
        f32x16 DH[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) DH[i] = zero16();
        STAMP(0);
        // Software pipeline over the 16 intermediate tiles: the d(x_hat) MFMAs of tile ti-1 are issued BETWEEN the GELU pairs of tile ti
        // (hipcc would cluster them in front of the vector work: one wave per SIMD, nothing else to fill the MFMA pipe's shadow), so
        // the ring carries  up(0) dg(0) | up(1) dg(1) dh(0) | ... | up(15) dg(15) dh(14) | dh(15).
        // ODD tiles complete a pair of g / du tiles: their 16 line stores leave at the end of the iteration, and the next two ring
        // steps may therefore leave eight more operations in flight (the EXTRA argument of RINGA_SYNC).
        opnd d0, d1;                                         // du operands of the previous tile
#define DH_READS(SET, Q) do { ALDS_READ_B(wv[SET][0], slot, fr, 2 * (Q), 0); ALDS_READ_B(wv[SET][1], slot, fr, 2 * (Q), 1);          \
                              ALDS_READ_B(wv[SET][2], slot, fr, 2 * (Q) + 1, 0); ALDS_READ_B(wv[SET][3], slot, fr, 2 * (Q) + 1, 1); } while (0)
#define DH_MMAS(SET, Q) do { DH[2 * (Q)] = mfma_u4(wv[SET][0], d0, DH[2 * (Q)]); DH[2 * (Q) + 1] = mfma_u4(wv[SET][2], d0, DH[2 * (Q) + 1]);  \
                             DH[2 * (Q)] = mfma_u4(wv[SET][1], d1, DH[2 * (Q)]); DH[2 * (Q) + 1] = mfma_u4(wv[SET][3], d1, DH[2 * (Q) + 1]);  \
                             __builtin_amdgcn_sched_barrier(0); } while (0)
#define GELU_PAIRS(I0) do { gelu_fb_pair(U, Gt, DG, I0); gelu_fb_pair(U, Gt, DG, (I0) + 2); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MLP_BWD_TILE(TI, ODD, FIRST, EXTRA_AB)                                                   \
        {                                                                                        \
            const int ti = (TI), tt = (ti + rot) & 15;                                           \
            uint32_t slot;                                                                       \
            f32x16 U, DG;                                                                        \
            {                                                                                    \
                RINGA_SYNC(src, slot, EXTRA_AB);                                                 \
                STAMP(1);                                                                        \
                U = mma16a<4>(slot, fr, x, zero16(), [&](int g_) { RINGA_PIECE(g_); });          \
            }                                                                                    \
            add_vec(U, lb_up, tt, h);                                                            \
            STAMP(2);                                                                            \
            {                                                                                    \
                RINGA_SYNC(src, slot, EXTRA_AB);                                                 \
                STAMP(1);                                                                        \
                DG = mma16a<4>(slot, fr, t1, zero16(), [&](int g_) { RINGA_PIECE(g_); });        \
            }                                                                                    \
            STAMP(2);                                                                            \
            f32x16 Gt;                                                                           \
            if (FIRST) {                                                                         \
                gelu_fwd_bwd16(U, Gt, DG);                                                       \
            } else {                                                                             \
                RINGA_SYNC(src, slot, 0);                                                        \
                STAMP(1);                                                                        \
                uint4 wv[2][4];                                                                  \
                DH_READS(0, 0); DH_READS(1, 1);                                                  \
                ALDS_WAITN(4); DH_MMAS(0, 0); RINGA_PIECE(0); DH_READS(0, 2); GELU_PAIRS(0);     \
                ALDS_WAITN(4); DH_MMAS(1, 1); RINGA_PIECE(1); DH_READS(1, 3); GELU_PAIRS(4);     \
                ALDS_WAITN(4); DH_MMAS(0, 2); RINGA_PIECE(2); GELU_PAIRS(8);                     \
                ALDS_WAITN(0); DH_MMAS(1, 3); RINGA_PIECE(3); GELU_PAIRS(12);                    \
            }                                                                                    \
            acc_to_opnd(DG, d0, d1);                                                             \
            STAMP(3);                                                                            \
            stage_tile(stg_g, ODD, m, h, Gt);                                                    \
            stage_tile(stg_du, ODD, m, h, DG);                                                   \
            if (ODD) {                          /* the pair (tt-1, tt) is complete -> whole 128-B lines */ \
                flush_lines<true>(stg_g, G, wrow0, 1024u, 64u * (tt - 1), lane);                 \
                flush_lines<true>(stg_du, DU, wrow0, 1024u, 64u * (tt - 1), lane);               \
            }                                                                                    \
            STAMP(4);                                                                            \
        }
        MLP_BWD_TILE(0, 0, 1, 0)
        MLP_BWD_TILE(1, 1, 0, 0)
        for (int tp = 1; tp < 8; ++tp) {
            MLP_BWD_TILE(2 * tp, 0, 0, 8)
            MLP_BWD_TILE(2 * tp + 1, 1, 0, 0)
        }
        {   // d(x_hat) of the last tile
            uint32_t slot;
            RINGA_SYNC(src, slot, 8);
            STAMP(1);
            uint4 wv[2][4];
            DH_READS(0, 0); DH_READS(1, 1);
            ALDS_WAITN(4); DH_MMAS(0, 0); RINGA_PIECE(0); DH_READS(0, 2);
            ALDS_WAITN(4); DH_MMAS(1, 1); RINGA_PIECE(1); DH_READS(1, 3);
            ALDS_WAITN(4); DH_MMAS(0, 2); RINGA_PIECE(2);
            ALDS_WAITN(0); DH_MMAS(1, 3); RINGA_PIECE(3);
            STAMP(5);
        }
#undef MLP_BWD_TILE
#undef DH_READS
#undef DH_MMAS
#undef GELU_PAIRS
        // LayerNorm backward on the row: dx = dy + rstd * (dh - mean(dh) - x_hat * mean(dh * x_hat)); x_hat out of the stash, the four
        // dy line groups requested up front (the operand registers of the loop are dead here)
        Lines rl4[4];
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) rl4[tp] = fetch_lines(DY, wrow0, lddyb, 128u * tp, lane);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(xstash + tp * STG_BYTES, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += DH[2 * tp + j][i]; s2 = fmaf(DH[2 * tp + j][i], xt[i], s2); }
            }
        }
        s1 = xhalf(s1) * (1.f / 256.f);
        s2 = xhalf(s2) * (1.f / 256.f);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            f32x16 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(xstash + tp * STG_BYTES, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[j][i] = rs * (DH[2 * tp + j][i] - s1 - xt[i] * s2);
            }
            stage_lines(stg, rl4[tp], lane);
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
        STAMP(6);
    }
    STAMP_FLUSH;
}

// ------------------------------------------------------------------------------------------------ backward, front half only
// mmfm_mlp_bwd with dx == NULL: t1 = dropout'(dy), g = gelu(u), du = dg * gelu'(u) for every row and NOTHING else - no d(x_hat)
// accumulators (128 registers of the kernel above), no x_hat stash, no LayerNorm-backward epilogue.  The caller finishes with
// mmfm_rowgemm(ln_bwd): dx = dy + LN'(du . Wp_up) (the K = 512 dX + LayerNorm-backward kernel the key/value linear already uses).
// What the split buys: this half fits 256 registers, so EIGHT waves share a workgroup's weight ring (two per SIMD - the fused kernel
// runs one wave per SIMD at 448 registers and pays every LDS / MFMA latency once) and a pass streams 2 x 256 KB of weights for 256
// rows instead of 3 x 256 KB for 128; what it costs: du is read back once (210 MB at R = 204,800).
constexpr int NTD = 512, NWD = 8;
__global__ __launch_bounds__(NTD, 2) void mlp_bwd_du_kernel(const mmfm_mlp_desc d) {
    constexpr int RING_B = RINGA_SLOTS * CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];        // RING_B + 2 * NWD * STG_BYTES + 512 * 4
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    const int64_t npass = (d.R + 32 * NWD - 1) / (32 * NWD);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const __amdgpu_buffer_rsrc_t rs_up = wbuf(d.w_up), rs_dnt = wbuf(d.w_down_t);
    const int rot = d.rotate ? (int)(blockIdx.x & 7) * 2 : 0;            // even: tile pairs of g / du complete together
    auto src = [=](int g) {                                               // up(ti) dg(ti), ti = 0..15
        const int idx = g & 31, ti = idx >> 1, tt = (ti + rot) & 15;
        AChunk c;
        if ((idx & 1) == 0) { c.rs = rs_up; c.off = (uint32_t)(32 * tt) * 512u; c.ldb = 512u; c.kind = 0; }
        else { c.rs = rs_dnt; c.off = (uint32_t)(32 * tt) * 512u; c.ldb = 512u; c.kind = 0; }
        return c;
    };
    char* stg_g = smem + RING_B + wave * STG_BYTES;
    char* stg_du = smem + RING_B + (NWD + wave) * STG_BYTES;
    char* stg = stg_g;                                                   // prologue staging: the loop's g area is idle then
    float* lb_up = reinterpret_cast<float*>(smem + RING_B + 2 * NWD * STG_BYTES);
    stage_vec(lb_up, d.b_up, 512, t, NTD);
    const Drop dr = drop_init(d.drop);
    const GBuf XH = gbuf(d.xhat, d.R * 512), DY = gbuf(d.dy, d.R * d.lddy * 2), T1 = gbuf(d.t1, d.R * 512);
    const GBuf G = gbuf(d.g, d.R * 1024), DU = gbuf(d.du, d.R * 1024);
    const uint32_t lddyb = d.lddy * 2;
    const ALane<NTD> ring_al = alane_init<NTD>(t, 512u, 1024u);
    const AFrag fr = afrag_init(m, h);
    RINGA_DECL(NTD);
    RINGA_START((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, my_passes * 32, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NWD + wave) * 32);
        const uint32_t row = wrow0 + m;
        opnd x[16], t1[16];
        {   // x_hat and dy rows requested in the same breath (one round trip for both row blocks)
            Lines L[4], Ld[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) L[q] = fetch_lines(XH, wrow0, 512u, 128u * q, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) Ld[q] = fetch_lines(DY, wrow0, lddyb, 128u * q, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                stage_lines(stg, L[q], lane);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x[4 * q + s4] = unstage_opnd(stg, s4, m, h);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                stage_lines(stg, Ld[q], lane);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t1[4 * q + s4] = unstage_opnd(stg, s4, m, h);
            }
        }
        if (dr.on()) {                                      // dropout'(dy) (rowchain.h RowDrop: the forward's decisions)
            const RowDrop rd = rowdrop_init(dr, row);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float f[8]; unpack8f(t1[s], f);
                drop8(rd, f, 16 * s + 8 * h);
                t1[s] = pack8o(f);
            }
        }
        store_rows_lines<4, true>(stg, T1, wrow0, 512u, lane, m, h, t1);
        // ODD tiles complete a pair of g / du tiles: their 16 line stores leave at the end of the iteration, and the next two ring steps
        // may therefore leave eight more operations in flight (the EXTRA argument of RINGA_SYNC; same accounting as the fused kernel)
#define MLP_DU_TILE(TI, ODD, EXTRA_AB)                                                           \
        {                                                                                        \
            const int ti = (TI), tt = (ti + rot) & 15;                                           \
            uint32_t slot;                                                                       \
            f32x16 U, DG;                                                                        \
            {                                                                                    \
                RINGA_SYNC(src, slot, EXTRA_AB);                                                 \
                U = mma16a<4>(slot, fr, x, zero16(), [&](int g_) { if (g_ < 1024 / NTD) RINGA_PIECE(g_); }); \
            }                                                                                    \
            add_vec(U, lb_up, tt, h);                                                            \
            {                                                                                    \
                RINGA_SYNC(src, slot, EXTRA_AB);                                                 \
                DG = mma16a<4>(slot, fr, t1, zero16(), [&](int g_) { if (g_ < 1024 / NTD) RINGA_PIECE(g_); }); \
            }                                                                                    \
            f32x16 Gt;                                                                           \
            gelu_fwd_bwd16(U, Gt, DG);                                                           \
            stage_tile(stg_g, ODD, m, h, Gt);                                                    \
            stage_tile(stg_du, ODD, m, h, DG);                                                   \
            if (ODD) {                          /* the pair (tt-1, tt) is complete -> whole 128-B lines */ \
                flush_lines<true>(stg_g, G, wrow0, 1024u, 64u * (tt - 1), lane);                 \
                flush_lines<true>(stg_du, DU, wrow0, 1024u, 64u * (tt - 1), lane);               \
            }                                                                                    \
        }
        MLP_DU_TILE(0, 0, 0)
        MLP_DU_TILE(1, 1, 0)
        for (int tp = 1; tp < 8; ++tp) {
            MLP_DU_TILE(2 * tp, 0, 8)
            MLP_DU_TILE(2 * tp + 1, 1, 0)
        }
#undef MLP_DU_TILE
    }
}

// ------------------------------------------------------------------------------------------------ wave-pair versions
// Forward: waves w and w+4 own the SAME 32 rows and split the work, so each fits 256 registers and every SIMD runs two waves
// (VALU of one beside MFMAs of the other): both hold x_hat; wave A takes intermediate tiles 2u, wave B 2u+1 (up-projection + GELU),
// they swap the bf16 operands of g through LDS, and each accumulates HALF of the 256 output columns over all of g.  Chunks are
// 32 KB (two 16 KB sub-blocks, one per intermediate tile of the pair).  (Measured and no longer compiled - DESIGN.md section 3b: a
// one-wave-per-tile forward, 300 us against 257 us, and a wave-pair backward, 539 us against 478 us for the one-wave kernel above.)
constexpr int NT8 = 512;

// The weights stream through the asynchronous ring (rowchain.h): three 32 KB slots filled by LDS-DMA, chunk cc+2 requested - one
// request behind each MFMA group - while chunk cc is multiplied (register-staged ring, one chunk ahead: 251 us; this: 213 us).
// d.w_down must be the unit-permuted copy (mmfm_prep_entry.WpP): the DMA cannot permute on the way in.
// Measured and not kept (round 3): a producer / consumer split of the pair (one wave LayerNorm + up-projection + GELU, the other the
// down-projection one ring step behind): 222 / 199 us with and without dropout against 216 / 206 - the producer's GELU (1,300 cycles
// per tile, as long as 40 MFMAs) becomes the critical path and the consumer idles at the barrier.
__global__ __launch_bounds__(NT8) void mlp_fwd8_kernel(const mmfm_mlp_desc d) {
    constexpr int NPAIR = 4;
    constexpr int RING_B = RINGA_SLOTS * CHUNK2;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // RING_B + 8 * STG_BYTES + NPAIR * 2 * 2048 + 768 * 4
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    const int role = wave >> 2, pw = wave & 3;
    const int64_t npass = (d.R + 32 * NPAIR - 1) / (32 * NPAIR);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const __amdgpu_buffer_rsrc_t rs_up = wbuf(d.w_up), rs_dn = wbuf(d.w_down);
    const int rot = d.rotate ? (int)(blockIdx.x & 7) : 0;
    auto src = [=](int g) {
        const int idx = g & 15, u = ((idx >> 1) + rot) & 7;
        AChunk2 c;
        if (idx & 1) {
            c.s[0].rs = rs_dn; c.s[0].off = (uint32_t)(64 * u) * 2u; c.s[0].ldb = 1024u; c.s[0].kind = 2;
            c.s[1].rs = rs_dn; c.s[1].off = (uint32_t)(64 * u + 32) * 2u; c.s[1].ldb = 1024u; c.s[1].kind = 2;
        } else {
            c.s[0].rs = rs_up; c.s[0].off = (uint32_t)(64 * u) * 512u; c.s[0].ldb = 512u; c.s[0].kind = 0;
            c.s[1].rs = rs_up; c.s[1].off = (uint32_t)(64 * u + 32) * 512u; c.s[1].ldb = 512u; c.s[1].kind = 0;
        }
        return c;
    };
    char* stg = smem + RING_B + wave * STG_BYTES;
    char* exch = smem + RING_B + 8 * STG_BYTES + pw * 4096;            // [role][2 operands][64 lanes][16 B]
    float* lb_up = reinterpret_cast<float*>(smem + RING_B + 8 * STG_BYTES + NPAIR * 4096);
    float* lb_dn = lb_up + 512;
    stage_vec(lb_up, d.b_up, 512, t, NT8);
    stage_vec(lb_dn, d.b_down, 256, t, NT8);
    const Drop dr = drop_init(d.drop);
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2);
    const GBuf XH = gbuf(role == 0 ? d.xhat : nullptr, d.R * 512), RS = gbuf(role == 0 ? d.rstd : nullptr, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2;
    const ALane<NT8> ring_al = alane_init<NT8>(t, 512u, 1024u);
    const AFrag fr = afrag_init(m, h);
    RINGA_DECL(NT8);
    RINGA2_START((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, my_passes * 16, src);
    STAMP_DECL;
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NPAIR + pw) * 32);
        const uint32_t row = wrow0 + m;
        opnd x[16];
        load_rows_lines<4>(stg, x, X, wrow0, ldxb, lane, m, h);
        const float rs = ln_rows(x, d.eps);
        store_rows_lines<4, true>(stg, XH, wrow0, 512u, lane, m, h, x);       // role 1: zero-sized buffer, dropped
        st4f(RS, h == 0 ? row * 4u : 0xfffffff0u, rs);
        f32x16 Yh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) Yh[i] = zero16();
        STAMP(0);
        for (int ui = 0; ui < 8; ++ui) {
            const int u = (ui + rot) & 7;
            uint32_t slot;
            f32x16 U;
            {
                RINGA2_SYNC(src, slot);
                STAMP(1);
                U = mma16a<4>(slot + (uint32_t)role * CHUNK, fr, x, zero16(), [&](int g_) { RINGA2_PIECE(g_); });
            }
            STAMP(2);
            add_vec(U, lb_up, 2 * u + role, h);
            gelu16(U);
            opnd g0, g1;
            acc_to_opnd(U, g0, g1);
            *reinterpret_cast<uint4*>(exch + role * 2048 + lane * 16) = as_u4(g0);
            *reinterpret_cast<uint4*>(exch + role * 2048 + 1024 + lane * 16) = as_u4(g1);
            STAMP(3);
            RINGA2_SYNC(src, slot);                                             // its barrier publishes the pair's operands
            STAMP(4);
            const opnd p0 = as_opnd(*reinterpret_cast<const uint4*>(exch + (role ^ 1) * 2048 + lane * 16));
            const opnd p1 = as_opnd(*reinterpret_cast<const uint4*>(exch + (role ^ 1) * 2048 + 1024 + lane * 16));
            const opnd e0 = role == 0 ? g0 : p0, e1 = role == 0 ? g1 : p1;      // tile 2u   (even)
            const opnd o0 = role == 0 ? p0 : g0, o1 = role == 0 ? p1 : g1;      // tile 2u+1 (odd)
            // the immediate selects the output tile, the slot address the role's tile quartet; one group = the four operands of an output
            // tile (both sub-blocks, both k-steps), two groups of reads in flight, counted waits
            const uint32_t sq = slot + (uint32_t)role * (4u * 2048u);
            uint4 wv[2][4];
#define F8_READ(SET, J) do { ALDS_READ_B(wv[SET][0], sq, fr, J, 0); ALDS_READ_B(wv[SET][1], sq, fr, J, 1);                       \
                             ALDS_READ_B(wv[SET][2], sq + CHUNK, fr, J, 0); ALDS_READ_B(wv[SET][3], sq + CHUNK, fr, J, 1); } while (0)
            F8_READ(0, 0);
            F8_READ(1, 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < 3) ALDS_WAITN(4); else ALDS_WAITN(0);
                Yh[j] = mfma_u4(wv[j & 1][0], e0, Yh[j]);
                Yh[j] = mfma_u4(wv[j & 1][1], e1, Yh[j]);
                Yh[j] = mfma_u4(wv[j & 1][2], o0, Yh[j]);
                Yh[j] = mfma_u4(wv[j & 1][3], o1, Yh[j]);
                __builtin_amdgcn_sched_barrier(0);
                RINGA2_PIECE(j);
                if (j == 0) F8_READ(0, 2);
                if (j == 1) F8_READ(1, 3);
            }
#undef F8_READ
            STAMP(5);
        }
        const RowDrop rd = rowdrop_init(dr, row);                               // (unused values when dropout is off)
        Lines xl2[2];                                                           // both residual line pairs requested together
#pragma unroll
        for (int q = 0; q < 2; ++q) xl2[q] = fetch_lines(X, wrow0, ldxb, 128u * (2 * role + q), lane);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int lp = 2 * role + q;                                        // line pair = output tiles 2*lp, 2*lp+1
            stage_lines(stg, xl2[q], lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int t2 = 2 * lp + j;
                add_vec(Yh[2 * q + j], lb_dn, t2, h);
                if (dr.on()) drop16(rd, Yh[2 * q + j], t2, h);
                const f32x16 r = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) Yh[2 * q + j][i] += r[i];
            }
            stage_tile(stg, 0, m, h, Yh[2 * q]);
            stage_tile(stg, 1, m, h, Yh[2 * q + 1]);
            flush_lines<false>(stg, Y, wrow0, ldyb, 128u * lp, lane);
        }
        STAMP(6);
    }
    STAMP_FLUSH;
}

// dW = gamma * G + db x beta; dgamma = colsum(W * G); dbeta = W^T db.  Grid (K/32 column groups) x (NSPLIT row groups): every
// block writes its dW rows and a partial (dgamma, dbeta) row; the LAST block of a column group (agent-scope ticket) sums the
// NSPLIT partials in fixed order -> deterministic, one launch.
constexpr int LG_SPLIT = 16;
__global__ __launch_bounds__(256) void ln_linear_grad_kernel(const float* __restrict__ Gdb, const float* __restrict__ W,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int N, int K,
                                                            float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate, float* __restrict__ part,
                                                            unsigned int* __restrict__ ticket) {
    __shared__ float red[2][8][32];
    __shared__ int is_last;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, k = blockIdx.x * 32 + tx, sp = blockIdx.y;
    const float* db = Gdb + (size_t)N * K;
    const int rows_per = (N + LG_SPLIT - 1) / LG_SPLIT, n0 = sp * rows_per, n1 = min(N, n0 + rows_per);
    float ag = 0.f, ab = 0.f;
    if (k < K) {
        const float g = gamma[k], b = beta[k];
        for (int n = n0 + ty; n < n1; n += 8) {
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
        part[((size_t)sp * 2 + 0) * K + k] = sg;
        part[((size_t)sp * 2 + 1) * K + k] = sb;
    }
    if (blockIdx.x == 0) for (int n = n0 + threadIdx.x; n < n1; n += 256) dbias[n] = db[n];
    // publish the partial row, take a ticket; the last arriver of this column group reduces (cdna_hip_programming.md Guideline 16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int prev = __hip_atomic_fetch_add(&ticket[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == (unsigned int)(LG_SPLIT - 1));
        if (is_last) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); ticket[blockIdx.x] = 0u; }      // re-armed for the next launch
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (is_last && ty == 0 && k < K) {
        float sg = 0.f, sb = 0.f;
        for (int i = 0; i < LG_SPLIT; ++i) {
            sg += __builtin_nontemporal_load(&part[((size_t)i * 2 + 0) * K + k]);
            sb += __builtin_nontemporal_load(&part[((size_t)i * 2 + 1) * K + k]);
        }
        dgamma[k] = accumulate ? dgamma[k] + sg : sg;
        dbeta[k] = accumulate ? dbeta[k] + sb : sb;
    }
}

int grid_for(int64_t R, int per_cu, int nw = NW) {
    const int64_t npass = (R + 32 * nw - 1) / (32 * nw);
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
    else MMFM_REQUIRE(d.xhat && d.dy && d.w_down_t && d.g && d.du && d.t1 && d.lddy % 8 == 0 && d.lddy >= 256 &&
                      (d.dx == nullptr || (d.rstd && d.w_up_t && d.lddx % 8 == 0 && d.lddx >= 256)), "mmfm_mlp_bwd: bad arguments");
    return 0;
}

}  // namespace

extern "C" int mmfm_mlp_fwd(const mmfm_mlp_desc* dp, mmfm_stream stream) {
    const mmfm_mlp_desc d = *dp;
    if (int rc = check(d, false)) return rc;
    static const int per_cu = [] { const char* e = getenv("MMFM_MLP_WG_PER_CU"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    {
        constexpr int LDS_A = RINGA_SLOTS * CHUNK2 + 8 * STG_BYTES + 4 * 4096 + 768 * 4;
        if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(mlp_fwd8_kernel), LDS_A, "mmfm_mlp_fwd")) return rc;
        hipLaunchKernelGGL(mlp_fwd8_kernel, dim3(grid_for(d.R, per_cu, 4)), dim3(NT8), LDS_A, (hipStream_t)stream, d);
    }
    MMFM_LAUNCH_CHECK("mmfm_mlp_fwd");
    return 0;
}

extern "C" int mmfm_mlp_bwd(const mmfm_mlp_desc* dp, mmfm_stream stream) {
    const mmfm_mlp_desc d = *dp;
    if (int rc = check(d, true)) return rc;
    static const int per_cu = [] { const char* e = getenv("MMFM_MLP_WG_PER_CU"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
    if (d.dx == nullptr) {            // front half only (t1, g, du): the caller finishes with mmfm_rowgemm(ln_bwd)
        constexpr int LDS_D = RINGA_SLOTS * CHUNK + 2 * NWD * STG_BYTES + 512 * 4;
        if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(mlp_bwd_du_kernel), LDS_D, "mmfm_mlp_bwd(front half)")) return rc;
        hipLaunchKernelGGL(mlp_bwd_du_kernel, dim3(grid_for(d.R, per_cu, NWD)), dim3(NTD), LDS_D, (hipStream_t)stream, d);
        MMFM_LAUNCH_CHECK("mmfm_mlp_bwd(front half)");
        return 0;
    }
    constexpr int LDS_B = RINGA_SLOTS * CHUNK + 2 * NW * STG_BYTES + 512 * 4 + NW * 4 * STG_BYTES;
    if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(mlp_bwd_kernel), LDS_B, "mmfm_mlp_bwd")) return rc;
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(grid_for(d.R, per_cu)), dim3(NT), LDS_B, (hipStream_t)stream, d);
    MMFM_LAUNCH_CHECK("mmfm_mlp_bwd");
    return 0;
}

extern "C" int64_t mmfm_ln_linear_grad_workspace(int K) { return ((int64_t)LG_SPLIT * 2 * K + cdiv(K, 32) + 64) * 4; }

extern "C" int mmfm_ln_linear_grad(const float* Gdb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW,
                                   float* dbias, float* dgamma, float* dbeta, int accumulate_ln, void* workspace, int64_t workspace_bytes,
                                   mmfm_stream stream) {
    MMFM_REQUIRE(Gdb && W && gamma && beta && dW && dbias && dgamma && dbeta && N > 0 && K > 0, "mmfm_ln_linear_grad: null argument");
    MMFM_REQUIRE(workspace && workspace_bytes >= mmfm_ln_linear_grad_workspace(K), "mmfm_ln_linear_grad: workspace too small (must be ZEROED once before first use)");
    float* part = reinterpret_cast<float*>(workspace);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(part + (size_t)LG_SPLIT * 2 * K);
    hipLaunchKernelGGL(ln_linear_grad_kernel, dim3(cdiv(K, 32), LG_SPLIT), dim3(256), 0, (hipStream_t)stream, Gdb, W, gamma, beta, N, K, dW, dbias, dgamma,
                       dbeta, accumulate_ln, part, ticket);
    MMFM_LAUNCH_CHECK("mmfm_ln_linear_grad");
    return 0;
}
