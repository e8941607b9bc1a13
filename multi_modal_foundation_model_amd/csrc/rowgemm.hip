// mmfm_rowgemm / mmfm_prep_weights: row-owner GEMMs for the width-256 block linears (bf16 throughput mode).
//   y[R][N] = epi( pro(x)[R][K] . W[N][K]^T ),  K = 256, 512 or 768, token rows owned by wavefronts (rowchain.h).
// prologue : LayerNorm statistics + normalisation of the row in registers (K = 256; the affine part is folded into the
//            prepared weights), with x_hat / rstd side outputs for the backward
//            -> replaces nn.LayerNorm + the nn.Linear it feeds: ln1+qkv, query_norm+query, context_norm+key/value,
//               encoder_norm+decoder_proj_context (encoder_embeddings.py:110-112, decoder_embeddings.py:139-143, mm.py:290-292)
// epilogue : + bias, + residual, bf16 store  (attention out_proj + residual add, mm_utils.py:114 / encoder_embeddings.py:112)
//        or  LayerNorm BACKWARD on the full output row (N = 256) + residual gradient: the dX product of a linear that
//            was fed by a LayerNorm never writes d(x_hat) to memory (autograd of the sites above).
// activation rows are read once by these kernels: non-temporal line loads keep them from displacing the weights in L2
// (measured on the B = 1024 step: 7.83 -> 7.72 ms over the 72 launches; the MLP kernels re-read their rows and lose with it)
#define MMFM_ACT_LOAD_AUX 2
#include "rowchain.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

using namespace rowchain;

namespace {

constexpr int NT = 256, NW = 4;

// ------------------------------------------------------------------------------------------------ forward-type kernel
constexpr int BIAS_MAX = 1024;                      // floats of bias kept in LDS behind the ring and the staging areas

// NWV waves per workgroup (8 = two per SIMD wherever the registers allow: the barrier / LDS-write overhead of a chunk is
// paid once per 256 rows instead of 128 and a SIMD's MFMA pipe is fed by two waves), 32 rows each
template <int KP, bool LN, bool NTS, int NWV>
__global__ __launch_bounds__(NWV * 64, (KP == 1 ? 2 : 1)) void rowgemm_kernel(const mmfm_rowgemm_desc d) {
    constexpr int NT = NWV * 64, NW = NWV;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + NW * STG_BYTES + BIAS_MAX * 4];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    const int npair = d.N >> 6, cpp = 2 * npair * KP;                   // N % 64 == 0
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const uint16_t* W = reinterpret_cast<const uint16_t*>(d.w);
    const int ldw = d.ldw;
    // every workgroup walks the same weight tiles; each starts at a different tile pair
    const int rot = d.rotate ? (int)(blockIdx.x % npair) : 0;
    auto src = [=](int g) {
        const int idx = g % cpp, ti = idx / KP, p = idx - ti * KP;
        int pr = (ti >> 1) + rot; pr = pr >= npair ? pr - npair : pr;
        WChunk c;
        c.base = W + (size_t)(32 * (2 * pr + (ti & 1))) * ldw + 256 * p;
        c.ld = ldw; c.kind = 0;
        return c;
    };
    char* stg = smem + LDS_BYTES + wave * STG_BYTES;
    float* lbias = reinterpret_cast<float*>(smem + LDS_BYTES + NW * STG_BYTES);
    stage_vec(lbias, d.bias, d.N, t, NT);           // visible after the first chunk's barrier
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2), RES = gbuf(d.residual, d.R * d.ldr * 2);
    const GBuf XH = gbuf(d.xhat, d.R * 512), RS = gbuf(d.rstd, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2, ldrb = d.ldr * 2;
    RING_DECL(NT);
    RING_START(smem, my_passes * cpp, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        const bool live = wrow0 < (uint32_t)d.R;          // wave-uniform: a wave without rows only keeps the ring's barriers
        opnd x[16 * KP];
        if (live) load_rows_lines<4 * KP>(stg, x, X, wrow0, ldxb, lane, m, h);
        if constexpr (LN) if (live) {
            const float rs = ln_rows(x, d.eps);
            store_rows_lines<4, true>(stg, XH, wrow0, 512u, lane, m, h, x);
            st4f(RS, h == 0 ? (wrow0 + m) * 4u : 0xfffffff0u, rs);
        }
        for (int tp = 0; tp < npair; ++tp) {
            int pr = tp + rot; pr = pr >= npair ? pr - npair : pr;
            f32x16 acc[2];
            Lines res;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[j] = zero16();
#pragma unroll
                for (int p = 0; p < KP; ++p) {
                    RING_SYNC_WRITE(src);
                    if (j == 0 && p == 0) res = fetch_lines(RES, wrow0, ldrb, 128u * pr, lane);   // older than the chunk fetch
                    const char* slot;
                    RING_FETCH(src, slot);
                    if (live) acc[j] = mma16(slot, x + 16 * p, acc[j], m, h);
                }
                add_vec(acc[j], lbias, 2 * pr + j, h);
            }
            if (!live) continue;
            stage_lines(stg, res, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 r = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][i] += r[i];
            }
            stage_tile(stg, 0, m, h, acc[0]);
            stage_tile(stg, 1, m, h, acc[1]);
            flush_lines<NTS>(stg, Y, wrow0, ldyb, 128u * pr, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward-type kernel, asynchronous ring
// K = 256 (every forward-type launch of the step).  Same row ownership and epilogue as rowgemm_kernel; the weights come through the
// three-slot LDS-DMA ring of rowchain.h (chunk cc+2 requested - one request behind each MFMA group - while chunk cc is multiplied; no
// staging registers, no ds_write pass; the move that took the MLP forward from 251 to 213 us).  A vector-memory LOAD inside the ring
// loop would be waited for with a compiler-counted vmcnt that does not know the DMA requests, i.e. it would drain the prefetch: the
// residual lines of the whole pass are therefore requested in front of the loop (RES4: N = 256, the only residual site), and the only
// other traffic of the loop, the four line stores behind each tile pair, is accounted for in the ring's wait (EXTRA, below).
// NPV = N / 64 when known at compile time (4: unrolled, residual allowed), 0 = runtime.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <bool LN, bool NTS, int NPV>
__global__ __launch_bounds__(256, 2) void rowgemm_a_kernel(const mmfm_rowgemm_desc d) {
    constexpr int NT = 256, NW = 4, RING_B = RINGA_SLOTS * CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // RING_B + NW * STG_BYTES + BIAS_MAX * 4
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    // Column blocks (gridDim.y > 1: launches with fewer row passes than CUs - the reference's batch of 16 is 25 passes - where one workgroup
    // walking all of N is a chain of 2 N / 64 dependent ring steps; split, every workgroup re-reads its 128 rows and takes `npair` tile pairs
    // starting at `pbase`).  The LayerNorm side outputs are written by column block 0 only.
    const int np_all = d.N >> 6;                                                    // N % 64 == 0
    const int npb = NPV ? NPV : (np_all + (int)gridDim.y - 1) / (int)gridDim.y, pbase = (int)blockIdx.y * npb;
    const int npair = NPV ? NPV : min(npb, np_all - pbase), cpp = 2 * npair;
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0 || npair <= 0) return;
    const __amdgpu_buffer_rsrc_t rs_w = wbuf(d.w);
    const uint32_t ldwb = (uint32_t)d.ldw * 2u;
    const int rot = d.rotate ? (int)(blockIdx.x % npair) : 0;
    auto src = [=](int g) {
        const int ti = g % cpp;
        int pr = (ti >> 1) + rot; pr = pbase + (pr >= npair ? pr - npair : pr);
        AChunk c;
        c.rs = rs_w; c.off = (uint32_t)(32 * (2 * pr + (ti & 1))) * ldwb; c.ldb = ldwb; c.kind = 0;
        return c;
    };
    char* stg = smem + RING_B + wave * STG_BYTES;
    float* lbias = reinterpret_cast<float*>(smem + RING_B + NW * STG_BYTES);
    stage_vec(lbias, d.bias, d.N, t, NT);           // visible after the first chunk's barrier
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2), RES = gbuf(d.residual, d.R * d.ldr * 2);
    const GBuf XH = gbuf(blockIdx.y == 0 ? d.xhat : nullptr, d.R * 512), RS = gbuf(blockIdx.y == 0 ? d.rstd : nullptr, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2, ldrb = d.ldr * 2;
    const bool has_res = NPV != 0 && d.residual != nullptr;
    const ALane<NT> ring_al = alane_init<NT>(t, ldwb, 0u);
    const AFrag fr = afrag_init(m, h);
    RINGA_DECL(NT);
    RINGA_START((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, my_passes * cpp, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        const bool live = wrow0 < (uint32_t)d.R;          // wave-uniform: a wave without rows only keeps the ring turning
        opnd x[16];
        Lines res[NPV ? NPV : 1];
        if (live) {
            load_rows_lines<4>(stg, x, X, wrow0, ldxb, lane, m, h);
            if constexpr (LN) {
                const float rs = ln_rows(x, d.eps);
                store_rows_lines<4, true>(stg, XH, wrow0, 512u, lane, m, h, x);
                st4f(RS, h == 0 ? (wrow0 + m) * 4u : 0xfffffff0u, rs);
            }
            if constexpr (NPV != 0) if (has_res) {          // behind the LayerNorm: its fp32 row and these 64 registers do not fit together
#pragma unroll
                for (int tp = 0; tp < NPV; ++tp) {
                    int pr = tp + rot; pr = pbase + (pr >= npair ? pr - npair : pr);
                    res[tp] = fetch_lines(RES, wrow0, ldrb, 128u * pr, lane);
                }
            }
        }
        // One tile pair = two ring steps + four line stores.  The stores of pair tp-1 were issued behind the requests of chunks 2tp and
        // 2tp+1, so both steps of pair tp may leave them in flight (EXTRA = 4; a wave without rows has issued none: 0); the first pair
        // of a pass waits conservatively (its predecessors are the prologue's loads / stores, of which there may be none).
        auto pair = [&](int tp, auto first) {
            constexpr int EX = decltype(first)::value ? 0 : 4;
            int pr = tp + rot; pr = pbase + (pr >= npair ? pr - npair : pr);
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t slot;
                if (live) vm_wait_n<1024 / NT + EX>(); else vm_wait_n<1024 / NT>();
                RINGA_SYNC_NOWAIT(src, slot);
                if (live) {
                    acc[j] = mma16a<4>(slot, fr, x, zero16(), [&](int g_) { RINGA_PIECE(g_); });
                    add_vec(acc[j], lbias, 2 * pr + j, h);
                } else {
#pragma unroll
                    for (int q = 0; q < 1024 / NT; ++q) RINGA_PIECE(q);
                }
            }
            if (!live) return;
            if constexpr (NPV != 0) if (has_res) {
                stage_lines(stg, res[NPV ? tp : 0], lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x16 r = unstage_tile(stg, j, m, h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[j][i] += r[i];
                }
            }
            stage_tile(stg, 0, m, h, acc[0]);
            stage_tile(stg, 1, m, h, acc[1]);
            flush_lines<NTS>(stg, Y, wrow0, ldyb, 128u * pr, lane);
        };
        pair(0, std::true_type());
        if constexpr (NPV != 0) {
#pragma unroll
            for (int tp = 1; tp < NPV; ++tp) pair(tp, std::false_type());
        } else {
            for (int tp = 1; tp < npair; ++tp) pair(tp, std::false_type());
        }
    }
}
#pragma clang diagnostic pop

// ------------------------------------------------------------------------------------------------ dX + LayerNorm backward
// v = x . W^T (N = 256: the gradient wrt x_hat of the LayerNorm that fed the forward linear; W = prepared W'^T);
// y = dres + rstd * (v - mean(v) - x_hat * mean(v * x_hat))
template <int KP>
__global__ __launch_bounds__(NT) void rowgemm_lnbwd_kernel(const mmfm_rowgemm_desc d) {
    // + a 16 KB per-wave stash of the pass's x_hat rows (as in the MLP backward): the statistics loop fetches them once, the output
    // loop reads them from LDS instead of putting a second dependent round trip per line pair into every pass
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + NW * STG_BYTES + NW * 4 * STG_BYTES];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    constexpr int cpp = 8 * KP;
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const uint16_t* W = reinterpret_cast<const uint16_t*>(d.w);
    const int ldw = d.ldw;
    auto src = [=](int g) {
        const int idx = g % cpp, tt = idx / KP, p = idx - tt * KP;
        WChunk c;
        c.base = W + (size_t)(32 * tt) * ldw + 256 * p;
        c.ld = ldw; c.kind = 0;
        return c;
    };
    char* stg = smem + LDS_BYTES + wave * STG_BYTES;
    char* xstash = smem + LDS_BYTES + NW * STG_BYTES + wave * 4 * STG_BYTES;
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2), RES = gbuf(d.residual, d.R * d.ldr * 2);
    const GBuf XH = gbuf(d.bwd_xhat, d.R * 512), RS = gbuf(d.bwd_rstd, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2, ldrb = d.ldr * 2;
    RING_DECL(NT);
    RING_START(smem, my_passes * cpp, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        opnd x[16 * KP];
        load_rows_lines<4 * KP>(stg, x, X, wrow0, ldxb, lane, m, h);
        const float rs = ld4f(RS, (wrow0 + m) * 4u);
        f32x16 acc[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            Lines xl;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[2 * tp + j] = zero16();
#pragma unroll
                for (int p = 0; p < KP; ++p) {
                    RING_SYNC_WRITE(src);
                    if (j == 0 && p == 0) xl = fetch_lines(XH, wrow0, 512u, 128u * tp, lane);
                    const char* slot;
                    RING_FETCH(src, slot);
                    acc[2 * tp + j] = mma16(slot, x + 16 * p, acc[2 * tp + j], m, h);
                }
            }
            stage_lines(xstash + tp * STG_BYTES, xl, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(xstash + tp * STG_BYTES, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += acc[2 * tp + j][i]; s2 = fmaf(acc[2 * tp + j][i], xt[i], s2); }
            }
        }
        s1 = xhalf(s1) * (1.f / 256.f);
        s2 = xhalf(s2) * (1.f / 256.f);
        Lines rl4[4];                                   // the residual-gradient lines, requested together (the operand registers are dead here)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) rl4[tp] = fetch_lines(RES, wrow0, ldrb, 128u * tp, lane);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const Lines& rl = rl4[tp];
            f32x16 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(xstash + tp * STG_BYTES, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[j][i] = rs * (acc[2 * tp + j][i] - s1 - xt[i] * s2);
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
            flush_lines<false>(stg, Y, wrow0, ldyb, 128u * tp, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dX + LayerNorm backward, asynchronous ring
// The one-wave-per-tile kernel above with the weights on the three-slot LDS-DMA ring.  The x_hat lines of the pass are fetched and
// stashed in front of the ring loop (a load inside it would be waited for with a vmcnt that does not know the DMA requests and drain
// the prefetch), so the loop issues nothing but ring requests and MFMAs.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int KP>
__global__ __launch_bounds__(NT) void rowgemm_lnbwd_a_kernel(const mmfm_rowgemm_desc d) {
    constexpr int RING_B = RINGA_SLOTS * CHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // RING_B + NW * STG_BYTES + NW * 4 * STG_BYTES
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    constexpr int cpp = 8 * KP;
    const int64_t npass = (d.R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const __amdgpu_buffer_rsrc_t rs_w = wbuf(d.w);
    const uint32_t ldwb = (uint32_t)d.ldw * 2u;
    auto src = [=](int g) {
        const int idx = g % cpp, tt = idx / KP, p = idx - tt * KP;
        AChunk c;
        c.rs = rs_w; c.off = (uint32_t)(32 * tt) * ldwb + 512u * (uint32_t)p; c.ldb = ldwb; c.kind = 0;
        return c;
    };
    char* stg = smem + RING_B + wave * STG_BYTES;
    char* xstash = smem + RING_B + NW * STG_BYTES + wave * 4 * STG_BYTES;
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2), RES = gbuf(d.residual, d.R * d.ldr * 2);
    const GBuf XH = gbuf(d.bwd_xhat, d.R * 512), RS = gbuf(d.bwd_rstd, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2, ldrb = d.ldr * 2;
    const ALane<NT> ring_al = alane_init<NT>(t, ldwb, 0u);
    const AFrag fr = afrag_init(m, h);
    RINGA_DECL(NT);
    RINGA_START((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem, my_passes * cpp, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32);
        opnd x[16 * KP];
        {
            Lines xl[4];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) xl[tp] = fetch_lines(XH, wrow0, 512u, 128u * tp, lane);
            load_rows_lines<4 * KP>(stg, x, X, wrow0, ldxb, lane, m, h);
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) stage_lines(xstash + tp * STG_BYTES, xl[tp], lane);
        }
        const float rs = ld4f(RS, (wrow0 + m) * 4u);
        f32x16 acc[8];
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            acc[tt] = zero16();
#pragma unroll
            for (int p = 0; p < KP; ++p) {
                uint32_t slot;
                RINGA_SYNC(src, slot, 0);
                acc[tt] = mma16a<4>(slot, fr, x + 16 * p, acc[tt], [&](int g_) { RINGA_PIECE(g_); });
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(xstash + tp * STG_BYTES, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += acc[2 * tp + j][i]; s2 = fmaf(acc[2 * tp + j][i], xt[i], s2); }
            }
        }
        s1 = xhalf(s1) * (1.f / 256.f);
        s2 = xhalf(s2) * (1.f / 256.f);
        Lines rl4[4];                                   // the residual-gradient lines, requested together (the operand registers are dead here)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) rl4[tp] = fetch_lines(RES, wrow0, ldrb, 128u * tp, lane);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const Lines& rl = rl4[tp];
            f32x16 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 xt = unstage_tile(xstash + tp * STG_BYTES, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[j][i] = rs * (acc[2 * tp + j][i] - s1 - xt[i] * s2);
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
            flush_lines<false>(stg, Y, wrow0, ldyb, 128u * tp, lane);
        }
    }
}
#pragma clang diagnostic pop

// ------------------------------------------------------------------------------------------------ dX + LayerNorm backward, wave pairs
// Eight waves; waves w and w+4 own the SAME 32 rows and split the 256 output columns (tiles 0-3 / 4-7): 64 accumulator
// registers each instead of 128, the K = 256*KP input streamed one 256-wide piece at a time (64 registers + the next piece in
// flight) -> two waves per SIMD fit where the one-wave version needed 512 registers and spent its time copying operands out
// of the accumulator file.  A 32 KB chunk = the two groups' weight tiles of one (piece, tile) step; the row statistics of the
// LayerNorm backward are summed across the pair through LDS behind one extra barrier per pass.
template <int KP>
__global__ __launch_bounds__(512) void rowgemm_lnbwd8_kernel(const mmfm_rowgemm_desc d) {
    constexpr int NT = 512, NW = 8, NPAIR = 4;
    __shared__ __attribute__((aligned(16))) char smem[2 * CHUNK2 + NW * STG_BYTES + NW * 32 * 8];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), m = lane & 31, h = lane >> 5;
    const int half = wave >> 2, pw = wave & 3;
    constexpr int cpp = 4 * KP;
    const int64_t npass = (d.R + 32 * NPAIR - 1) / (32 * NPAIR);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    const uint16_t* W = reinterpret_cast<const uint16_t*>(d.w);
    const int ldw = d.ldw;
    auto src = [=](int g) {
        const int idx = g % cpp, p = idx >> 2, j = idx & 3;
        WChunk2 c;
        c.s[0].base = W + (size_t)(32 * j) * ldw + 256 * p; c.s[0].ld = ldw; c.s[0].kind = 0;
        c.s[1].base = W + (size_t)(32 * (4 + j)) * ldw + 256 * p; c.s[1].ld = ldw; c.s[1].kind = 0;
        return c;
    };
    char* stg = smem + 2 * CHUNK2 + wave * STG_BYTES;
    float2* exch = reinterpret_cast<float2*>(smem + 2 * CHUNK2 + NW * STG_BYTES);      // [wave][32 rows]
    const GBuf X = gbuf(d.x, d.R * d.ldx * 2), Y = gbuf(d.y, d.R * d.ldy * 2), RES = gbuf(d.residual, d.R * d.ldr * 2);
    const GBuf XH = gbuf(d.bwd_xhat, d.R * 512), RS = gbuf(d.bwd_rstd, d.R * 4);
    const uint32_t ldxb = d.ldx * 2, ldyb = d.ldy * 2, ldrb = d.ldr * 2;
    RING2_DECL(NT);
    RING2_START(smem, my_passes * cpp, src);
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t wrow0 = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NPAIR + pw) * 32);
        const bool live = wrow0 < (uint32_t)d.R;
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = zero16();
        Lines L[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) L[q] = fetch_lines(X, wrow0, ldxb, 128u * q, lane);
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            opnd x[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                stage_lines(stg, L[q], lane);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) x[4 * q + s4] = unstage_opnd(stg, s4, m, h);
                if (p + 1 < KP) L[q] = fetch_lines(X, wrow0, ldxb, 512u * (p + 1) + 128u * q, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* slot;
                RING2_STEP(src, slot);
                if (live) acc[j] = mma16<4>(slot + half * CHUNK, x, acc[j], m, h);
            }
        }
        // this wave's columns: tiles 4*half .. 4*half+3 = line pairs 2*half, 2*half+1
        f32x16 xt[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const Lines xl = fetch_lines(XH, wrow0, 512u, 128u * (2 * half + q), lane);
            stage_lines(stg, xl, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                xt[2 * q + j] = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += acc[2 * q + j][i]; s2 = fmaf(acc[2 * q + j][i], xt[2 * q + j][i], s2); }
            }
        }
        s1 = xhalf(s1); s2 = xhalf(s2);
        if (h == 0) exch[wave * 32 + m] = make_float2(s1, s2);
        __syncthreads();
        const float2 o2 = exch[(wave ^ 4) * 32 + m];
        s1 = (s1 + o2.x) * (1.f / 256.f);
        s2 = (s2 + o2.y) * (1.f / 256.f);
        const float rs = ld4f(RS, (wrow0 + m) * 4u);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const Lines rl = fetch_lines(RES, wrow0, ldrb, 128u * (2 * half + q), lane);
            stage_lines(stg, rl, lane);
            f32x16 o[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x16 r = unstage_tile(stg, j, m, h);
#pragma unroll
                for (int i = 0; i < 16; ++i) o[j][i] = r[i] + rs * (acc[2 * q + j][i] - s1 - xt[2 * q + j][i] * s2);
            }
            stage_tile(stg, 0, m, h, o[0]);
            stage_tile(stg, 1, m, h, o[1]);
            flush_lines<false>(stg, Y, wrow0, ldyb, 128u * (2 * half + q), lane);
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight preparation
// One block per (entry, 32-row tile of W): Wp = bf16(W * gamma[k]) [N][K], WpT = its transpose [K][N] (WpP / WpTP: the same, unit-permuted),
// bp[n] = bias[n] + sum_k W[n][k] * beta[k]  (the LayerNorm affine folded into the linear it feeds).
__global__ __launch_bounds__(256) void prep_weights_kernel(const mmfm_prep_entry* __restrict__ E, int ne) {
    __shared__ float tile[32][33];
    int e = 0;
    while (e + 1 < ne && (int)blockIdx.x >= E[e + 1].tile0) ++e;
    const mmfm_prep_entry en = E[e];
    const int n0 = ((int)blockIdx.x - en.tile0) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    uint16_t* Wp = reinterpret_cast<uint16_t*>(en.Wp);
    uint16_t* WpT = reinterpret_cast<uint16_t*>(en.WpT);
    uint16_t* WpP = reinterpret_cast<uint16_t*>(en.WpP);
    uint16_t* WpTP = reinterpret_cast<uint16_t*>(en.WpTP);
    // unit-permuted position of element i of a row: 4-element (8-byte) units 1 and 2 of every 16 swap places
    auto perm = [](int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); };
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < en.K; k0 += 32) {
        const int k = k0 + tx;
        const float g = (en.gamma && k < en.K) ? en.gamma[k] : 1.f;
        const float bt = (en.beta && k < en.K) ? en.beta[k] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + ty + 8 * j;
            const float v = (n < en.N && k < en.K) ? en.W[(size_t)n * en.K + k] : 0.f;
            const float w = v * g;
            dot[j] = fmaf(v, bt, dot[j]);
            if (Wp && n < en.N && k < en.K) Wp[(size_t)n * en.K + k] = f2bf(w);
            if (WpP && n < en.N && k < en.K && perm(k) < en.K) WpP[(size_t)n * en.K + perm(k)] = f2bf(w);    // K % 16 == 0 (mmfm.h)
            tile[ty + 8 * j][tx] = w;
        }
        __syncthreads();
        if (WpT || WpTP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k0 + ty + 8 * j, n = n0 + tx;
                if (kk < en.K && n < en.N) {
                    if (WpT) WpT[(size_t)kk * en.N + n] = f2bf(tile[tx][ty + 8 * j]);
                    if (WpTP && perm(n) < en.N) WpTP[(size_t)kk * en.N + perm(n)] = f2bf(tile[tx][ty + 8 * j]);   // N % 16 == 0 (mmfm.h)
                }
            }
        }
        __syncthreads();
    }
    if (en.bp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r = dot[j];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) r += __shfl_xor(r, o);
            const int n = n0 + ty + 8 * j;
            if (tx == 0 && n < en.N) en.bp[n] = (en.bias ? en.bias[n] : 0.f) + r;
        }
    }
}

int grid_for(int64_t R, int per_cu, int nw = NW) {
    const int64_t npass = (R + 32 * nw - 1) / (32 * nw);
    return (int)std::max<int64_t>(1, std::min<int64_t>(npass, 256 * per_cu));
}

}  // namespace

extern "C" int mmfm_rowgemm(const mmfm_rowgemm_desc* dp, mmfm_stream stream) {
    const mmfm_rowgemm_desc d = *dp;
    MMFM_REQUIRE(d.x && d.w && d.y && d.R > 0, "mmfm_rowgemm: null operand / empty problem");
    MMFM_REQUIRE(d.K == 256 || d.K == 512 || d.K == 768, "mmfm_rowgemm: K = %d (256, 512 or 768 only)", d.K);
    MMFM_REQUIRE(d.N > 0 && d.N % 64 == 0, "mmfm_rowgemm: N = %d must be a positive multiple of 64", d.N);
    MMFM_REQUIRE(d.ldx % 8 == 0 && d.ldw % 8 == 0 && d.ldy % 8 == 0 && d.ldx >= d.K && d.ldw >= d.K && d.ldy >= d.N,
                 "mmfm_rowgemm: leading dimensions (%d, %d, %d) must be multiples of 8 and cover the rows", d.ldx, d.ldw, d.ldy);
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    MMFM_REQUIRE(al16(d.x) && al16(d.w) && al16(d.y) && al16(d.residual) && al16(d.xhat) && al16(d.bwd_xhat) && al16(d.bias),
                 "mmfm_rowgemm: operands must be 16-byte aligned");
    MMFM_REQUIRE(!d.residual || (d.ldr % 8 == 0 && d.ldr >= d.N), "mmfm_rowgemm: ldr %d", d.ldr);
    MMFM_REQUIRE(!d.ln || d.K == 256, "mmfm_rowgemm: the LayerNorm prologue needs K = 256");
    const int64_t maxld = std::max<int64_t>(std::max(d.ldx, d.ldy), std::max(d.ldr, 256));
    MMFM_REQUIRE((d.R + 128) * maxld * 2 < (int64_t)1 << 31, "mmfm_rowgemm: tensors beyond 2 GiB are not addressable by the 32-bit buffer offsets");
    MMFM_REQUIRE(d.N <= BIAS_MAX, "mmfm_rowgemm: N = %d > %d", d.N, BIAS_MAX);
    // Launch shapes (measured on MI355X at R = 204,800 inside the bench step, profiles/r04_step_launches_B1024.txt; rejected variants - 8 skewed
    // waves, a dedicated weight-stager wave, one 8-wave workgroup per CU, wave pairs at K >= 512 - are described in DESIGN.md sections 3b / 3e):
    //   forward-type, K = 256 : 4 waves, 2 workgroups per CU, LDS-DMA ring (LN+qkv 119 us, LN+kv 94 us, LN+q 62 us, out_proj + residual 69 us,
    //                           plain 56 us); LayerNorm + residual in one launch, and K = 512 / 768: the register-staged ring
    //   LN-backward epilogue  : K = 256 -> wave pairs (8 waves, 101 us); K = 512 / 768 -> one wave per row tile on the LDS-DMA ring (137 / 174 us)
    static const int per_cu_env = [] { const char* e = getenv("MMFM_ROWGEMM_WG_PER_CU"); return e ? atoi(e) : 0; }();
    hipStream_t st = (hipStream_t)stream;
    if (d.ln_bwd) {
        MMFM_REQUIRE(d.N == 256 && d.bwd_xhat && d.bwd_rstd && !d.ln && !d.bias, "mmfm_rowgemm: ln_bwd needs N = 256, x_hat, rstd, no bias/ln");
        if (d.K == 256) {
            dim3 grid(grid_for(d.R, per_cu_env > 0 ? per_cu_env : 1, 4)), block(512);      // 4 row tiles (wave pairs) per pass
            hipLaunchKernelGGL(rowgemm_lnbwd8_kernel<1>, grid, block, 0, st, d);
        } else {
            dim3 grid(grid_for(d.R, per_cu_env > 0 ? per_cu_env : 1)), block(NT);
            static const int ring_env = [] { const char* e = getenv("MMFM_ROWGEMM_RING"); return (e ? atoi(e) : 3) & 2; }();   // bit 1 clear: register-staged ring
            constexpr int LDS_L = RINGA_SLOTS * CHUNK + NW * STG_BYTES + NW * 4 * STG_BYTES;
            if (ring_env && d.K == 512) {
                if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(rowgemm_lnbwd_a_kernel<2>), LDS_L, "mmfm_rowgemm")) return rc;
                hipLaunchKernelGGL(rowgemm_lnbwd_a_kernel<2>, grid, block, LDS_L, st, d);
            } else if (ring_env) {
                if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(rowgemm_lnbwd_a_kernel<3>), LDS_L, "mmfm_rowgemm")) return rc;
                hipLaunchKernelGGL(rowgemm_lnbwd_a_kernel<3>, grid, block, LDS_L, st, d);
            } else if (d.K == 512) hipLaunchKernelGGL(rowgemm_lnbwd_kernel<2>, grid, block, 0, st, d);
            else hipLaunchKernelGGL(rowgemm_lnbwd_kernel<3>, grid, block, 0, st, d);
        }
    } else {
#define RG_LAUNCH(KP, LN)                                                                                   \
    if (d.stream_out) hipLaunchKernelGGL((rowgemm_kernel<KP, LN, true, 4>), grid, block, 0, st, d);         \
    else hipLaunchKernelGGL((rowgemm_kernel<KP, LN, false, 4>), grid, block, 0, st, d);
        if (d.K != 256) {
            dim3 grid(grid_for(d.R, per_cu_env > 0 ? per_cu_env : 1, 4)), block(256);
            if (d.K == 512) { RG_LAUNCH(2, false) } else { RG_LAUNCH(3, false) }
        } else {
            dim3 grid(grid_for(d.R, per_cu_env > 0 ? per_cu_env : 2, 4)), block(256);
            static const int ring_env = [] { const char* e = getenv("MMFM_ROWGEMM_RING"); return (e ? atoi(e) : 3) & 1; }();   // bit 0 clear: register-staged ring
            if (ring_env && (!d.residual || (d.N == 256 && !d.ln))) {     // LayerNorm + residual (one launch per step) stays on the staged ring: 171 spills
                constexpr int LDS_A = RINGA_SLOTS * CHUNK + 4 * STG_BYTES + BIAS_MAX * 4;
#define RGA_LAUNCH(LN, NTS, NPV)                                                                                            \
                {                                                                                                          \
                    if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(rowgemm_a_kernel<LN, NTS, NPV>), LDS_A, "mmfm_rowgemm")) return rc; \
                    hipLaunchKernelGGL((rowgemm_a_kernel<LN, NTS, NPV>), grid, block, LDS_A, st, d);                        \
                }
#define RGA_LAUNCH2(LN, NPV) if (d.stream_out) RGA_LAUNCH(LN, true, NPV) else RGA_LAUNCH(LN, false, NPV)
                // fewer row passes than a quarter of the resident slots (2 workgroups per CU): split N into column blocks too (kernel comment)
                const int64_t npass = (d.R + 127) / 128;
                const int np_all = d.N >> 6;
                static const int nsplit_env = [] { const char* e = getenv("MMFM_ROWGEMM_NSPLIT"); return e ? atoi(e) : 1; }();
                const int ny = (nsplit_env && npass < 128) ? (int)std::max<int64_t>(1, std::min<int64_t>(np_all, 512 / npass)) : 1;
                if (d.residual) {                                  // N = 256: four pairs in one block, or one pair in each of four
                    if (ny >= 4) { grid.y = 4; RGA_LAUNCH2(false, 1) } else { RGA_LAUNCH2(false, 4) }
                } else {
                    grid.y = (unsigned)((np_all + (np_all + ny - 1) / ny - 1) / ((np_all + ny - 1) / ny));      // blocks of ceil(np_all / ny) pairs
                    if (d.ln) RGA_LAUNCH2(true, 0)
                    else RGA_LAUNCH2(false, 0)
                }
#undef RGA_LAUNCH2
#undef RGA_LAUNCH
            } else if (d.ln) { RG_LAUNCH(1, true) } else { RG_LAUNCH(1, false) }
        }
#undef RG_LAUNCH
    }
    MMFM_LAUNCH_CHECK("mmfm_rowgemm");
    return 0;
}

extern "C" int mmfm_prep_weights(const mmfm_prep_entry* entries_dev, int n_entries, int total_tiles, mmfm_stream stream) {
    MMFM_REQUIRE(entries_dev && n_entries > 0 && total_tiles > 0, "mmfm_prep_weights: empty table");
    hipLaunchKernelGGL(prep_weights_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, entries_dev, n_entries);
    MMFM_LAUNCH_CHECK("mmfm_prep_weights");
    return 0;
}
