// bf16 attention, dh = 64, any sequence length (BASELINE configs[4]: d_model 512, 8 heads, L = 600): the keep-bit kernels for heads
// whose K / V images do not fit the LDS.  Same contract as attention_bf16.hip (masks: key padding and DIAG; CAUSAL / SEP stay with the
// general kernels), reference: mm_utils.py:97-152.  They replace the round-1 tiled pair (attention_bf16.hip: 0.99 ms forward, 2.64 ms
// two-phase backward per launch at B = 256, L = 600 - 54 of the 98 ms config-5 step) whenever the caller provides the keep-bit
// workspace (mmfm_attn_desc.keepbits):
//   * dropout decisions are the generator kernel's bit tiles (attention_fast.hip): scalar lane masks where the lane is the query
//     (forward, dQ phase), the lane's own word where it is the key (dK / dV phase) - no hash in any of the three kernels;
//   * the forward keeps the first key tile's row maximum as the reference exponent (no running maximum, no rescaling; rows whose sum
//     overflows send the workgroup through an exact pass);
//   * eight waves per workgroup where the registers allow it (forward, dQ phase: 8 query tiles share every streamed K / V chunk), and
//     the next chunk's global loads are in flight while the current one is multiplied (one LDS buffer, register staging);
//   * the backward's per-query constants (delta, the output-dropout'd d_o) are computed ONCE by a small kernel instead of once per
//     workgroup and phase: d_o' lands in the dq buffer (each dQ workgroup reads its own rows before it overwrites them), delta in the
//     tail of the keep-bit workspace.
// The two-phase backward stays (a single pass needs every key tile of a head in one workgroup's registers: 19 tiles x 64 accumulator
// registers), so S and P are evaluated twice; what changed is what an evaluation costs.
#include "attn_common.h"
#include <algorithm>
#include <stdlib.h>

using namespace attn;

namespace {

constexpr int DH = 64, KS = DH / 16, DT = DH / 32;
constexpr int RS = DH * 2 + 16;          // 144-byte image rows: conflict-free 16-byte row fragments
constexpr int CH = 128;                  // streamed rows per chunk (four 32-row tiles)
constexpr int C8 = DH / 8;               // 16-byte pieces per row
constexpr float OVERFLOW_SUM = 1.2676506e30f;      // 2^100

// ---- chunk streaming: rows [c0, c0 + CH) of two [L][ld] bf16 tensors -> registers -> two LDS images (RS-byte rows); NPT pieces per thread
template <int NT>
struct Stager {
    static constexpr int NPT = CH * C8 / NT;
    uint4 x[NPT], y[NPT];
    __device__ __forceinline__ void load(const uint16_t* xg, int ldx, const uint16_t* yg, int ldy, int c0, int L, int t) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int idx = t + NT * j, row = c0 + idx / C8, col = idx % C8;
            x[j] = make_uint4(0u, 0u, 0u, 0u); y[j] = x[j];
            if (row < L) {
                x[j] = *reinterpret_cast<const uint4*>(xg + (size_t)row * ldx + 8 * col);
                y[j] = *reinterpret_cast<const uint4*>(yg + (size_t)row * ldy + 8 * col);
            }
        }
    }
    __device__ __forceinline__ void store(char* Xs, char* Ys, int t) const {
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int idx = t + NT * j, row = idx / C8, col = idx % C8;
            *reinterpret_cast<uint4*>(Xs + row * RS + col * 16) = x[j];
            *reinterpret_cast<uint4*>(Ys + row * RS + col * 16) = y[j];
        }
    }
};

__device__ __forceinline__ float tile_max16(const f32x16& st) {
    float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
    for (int r = 3; r < 16; ++r) mx = fmaxf(mx, st[r]);
    return xhalf_max(mx);
}

// key bias of a head (0 / -inf per key, -inf beyond Lk) into LDS + "some key of the head is padded" (wave-uniform)
template <int NW>
__device__ __forceinline__ int stage_kbias(float* kbias, int* wflag, const uint8_t* keypad, int b, int Lk, int LkP, int t, int wave, int lane) {
    int pad = 0;
    for (int i = t; i < LkP; i += NW * 64) {
        const bool ok = i < Lk && (keypad == nullptr || keypad[(size_t)b * Lk + i] != 0);
        kbias[i] = ok ? 0.f : -INFINITY;
        pad |= (i < Lk && !ok) ? 1 : 0;
    }
    const int wv = __any(pad) ? 1 : 0;
    if (lane == 0) wflag[wave] = wv;
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) any |= wflag[w];
    return __builtin_amdgcn_readfirstlane(any);
}

// S^T tile of key tile (rows kl*32.. of the K image) against the wave's Q fragments, starting from the key bias
__device__ __forceinline__ f32x16 score_tile(const char* Ks, const float* kb, int kl, const bf16x8v (&qf)[KS], bool dfix, int kh, int l31) {
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 kb4 = *reinterpret_cast<const float4*>(kb + 8 * g + 4 * kh);
        a[4 * g + 0] = kb4.x; a[4 * g + 1] = kb4.y; a[4 * g + 2] = kb4.z; a[4 * g + 3] = kb4.w;
    }
    if (dfix) {              // rare: padded keys and `eye |`: the diagonal keeps S[q][q] finite
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (mrow(r, kh) == l31) ? 0.f : a[r];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kl * 32 + l31) * RS + ks * 32 + kh * 16), qf[ks], a, 0, 0, 0);
    return a;
}

// ---------------------------------------------------------------------------------------------- forward
constexpr int F_NW = 8;
template <bool DROP>
__global__ __launch_bounds__(F_NW * 64, 4) void attn_fwd_long_kernel(const mmfm_attn_desc d, const float keep_scale) {
    constexpr int NW = F_NW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LkP = (Lk + 31) & ~31;
    const int nqt = (Lq + 31) >> 5, nkt = LkP >> 5, nch = (LkP + CH - 1) / CH;
    char* Ks = smem;                                   // [CH][RS]
    char* Vs = Ks + CH * RS;                           // [CH][RS]
    char* ost = Vs + CH * RS;                          // [NW][32][RS] output transpose tiles
    float* kbias = reinterpret_cast<float*>(ost + NW * 32 * RS);
    int* wflag = reinterpret_cast<int*>(kbias + LkP);
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;

    const int qt = blockIdx.y * NW + wave;
    const bool active = qt < nqt;                      // inactive waves still stage chunks and take part in the barriers
    const int q0 = qt * 32, q = q0 + l31;
    Stager<NT> stg;
    stg.load(kg, d.ldk, vg, d.ldv, 0, Lk, t);          // chunk 0 in flight under the prologue
    bf16x8v qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (active && q < Lq) v = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
        qf[ks] = __builtin_bit_cast(bf16x8v, v);
    }
    const int anypad = stage_kbias<NW>(kbias, wflag, d.keypad, b, Lk, LkP, t, wave, lane);
    const bool fixdiag = anypad && (d.flags & MMFM_ATTN_DIAG);
    const float c2 = d.scale * LOG2E;
    const masks_ptr mkp = reinterpret_cast<masks_ptr>(reinterpret_cast<uintptr_t>(d.keepbits)) + ((size_t)bh_ * nqt + (active ? qt : 0)) * nkt;

    float m_ref = 0.f, l_run = 0.f;
    f32x16 acc[DT];
    // one pass over the head's keys.  EXACT = false: reference exponent = first key tile's row maximum; true: running maximum
    auto pass = [&](auto exact_tag) {
        constexpr bool EXACT = decltype(exact_tag)::value;
        float m_run = -1e30f;
        l_run = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                           // readers of the previous chunk are done
            stg.store(Ks, Vs, t);
            // the next chunk's loads fly while this one is multiplied (after the last chunk: chunk 0 again, for a possible second pass)
            stg.load(kg, d.ldk, vg, d.ldv, (c + 1 < nch ? c + 1 : 0) * CH, Lk, t);
            __syncthreads();
            if (!active) continue;
            const int ntl = min(CH / 32, nkt - c * (CH / 32));
            for (int kl = 0; kl < ntl; ++kl) {
                const int kt = c * (CH / 32) + kl;
                const f32x16 st = score_tile(Ks, kbias + kt * 32, kl, qf, fixdiag && kt == qt, kh, l31);
                Masks16 mk;
                if (DROP) mk = ld_masks(mkp + kt);
                if (EXACT) {
                    const float m_new = v_max(m_run, tile_max16(st) * c2);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                    for (int i = 0; i < DT; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                    l_run *= alpha;
                    m_run = m_new;
                    m_ref = m_run;
                } else if (kt == 0) {
                    m_ref = v_max(tile_max16(st), -1e30f / c2) * c2;
                }
                uint32_t pk[8];
                float ps0 = 0.f, ps1 = 0.f;
                const float nm = -m_ref;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, nm));
                    float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r + 1], c2, nm));
                    ps0 = v_add(ps0, p0);
                    ps1 = v_add(ps1, p1);
                    if (DROP) {
                        p0 = v_keep(p0, mk.m[r]);
                        p1 = v_keep(p1, mk.m[r + 1]);
                    }
                    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(ps0), "+v"(ps1));     // see attention_fast.hip fwd_tile
                    pk[r >> 1] = pack2(p0, p1);
                }
                l_run = v_add(l_run, v_add(ps0, ps1));
                const bf16x8v lo = __builtin_bit_cast(bf16x8v, make_uint4(pk[0], pk[1], pk[2], pk[3]));
                const bf16x8v hi = __builtin_bit_cast(bf16x8v, make_uint4(pk[4], pk[5], pk[6], pk[7]));
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, RS, kl * 32, i * 32, lane), lo, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, RS, kl * 32 + 16, i * 32, lane), hi, acc[i], 0, 0, 0);
                }
            }
        }
    };
    pass(std::false_type{});
    float l_tot = xhalf_sum(l_run);
    // exact pass if ANY row of the workgroup overflowed its first-tile reference (the chunk barriers need every wave)
    const int bad = (active && __any(!(l_tot < OVERFLOW_SUM))) ? 1 : 0;
    __syncthreads();
    if (lane == 0) wflag[wave] = bad;
    __syncthreads();
    int redo = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) redo |= wflag[w];
    if (__builtin_amdgcn_readfirstlane(redo)) {
        pass(std::true_type{});
        l_tot = xhalf_sum(l_run);
    }
    if (!active) return;
    const Drop dout = drop_init(d.drop_o);
    const float inv = keep_scale / l_tot;
    if (kh == 0 && q < Lq) d.lse[(size_t)bh_ * Lq + q] = m_ref * LN2 + __logf(l_tot);
    // O^T (rows = d in registers, lane = query) -> bf16 rows [query][d] through the wave's staging tile, output dropout on the way
    char* tl = ost + wave * 32 * RS;
    uint16_t* og = reinterpret_cast<uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const uint64_t base = ((uint64_t)b * Lq + (uint64_t)q) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH);
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = i * 32 + 8 * g + 4 * kh;
            float v0 = acc[i][4 * g + 0] * inv, v1 = acc[i][4 * g + 1] * inv, v2 = acc[i][4 * g + 2] * inv, v3 = acc[i][4 * g + 3] * inv;
            if (dout.on()) {
                dout.apply2(v0, v1, base + d0);
                dout.apply2(v2, v3, base + d0 + 2);
            }
            *reinterpret_cast<uint2*>(tl + l31 * RS + d0 * 2) = make_uint2(pack2(v0, v1), pack2(v2, v3));
        }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 32 * C8 / 64; ++i) {
        const int idx = lane + 64 * i, row = idx / C8, c = idx % C8;
        if (q0 + row < Lq)
            *reinterpret_cast<uint4*>(og + (size_t)(q0 + row) * d.ldo + 8 * c) = *reinterpret_cast<const uint4*>(tl + row * RS + c * 16);
    }
}
size_t fwd_long_lds(int Lk) {
    const int LkP = (Lk + 31) & ~31;
    return (size_t)2 * CH * RS + (size_t)F_NW * 32 * RS + (size_t)LkP * 4 + 64;
}

// ---------------------------------------------------------------------------------------------- backward: per-query constants
// d_o' = dropout'(d_o) as bf16 -> the dq buffer (scratch until the dQ phase overwrites it with dq);  delta / dropout scale -> dl[bh][q]
__global__ __launch_bounds__(256) void attn_bwd_long_prep_kernel(const mmfm_attn_desc d, float* dl, const float inv_keep) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 16-byte piece per thread: (token row, head, piece)
    const int64_t total = (int64_t)d.B * d.Lq * d.heads * C8;
    const bool ok = idx < total;
    const int c = (int)(idx % C8), h = (int)((idx / C8) % d.heads);
    const int64_t row = ok ? idx / (C8 * d.heads) : 0;                     // b * Lq + q
    const Drop dout = drop_init(d.drop_o);
    uint4 g = make_uint4(0u, 0u, 0u, 0u), o = g;
    if (ok) {
        g = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(d.d_o) + row * d.lddo + h * DH + 8 * c);
        o = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(d.o) + row * d.ldo + h * DH + 8 * c);
    }
    const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {o.x, o.y, o.z, o.w};
    const uint64_t base = (uint64_t)row * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 8 * c);
    float part = 0.f, gd[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float g0 = __uint_as_float(gw[j] << 16), g1 = __uint_as_float(gw[j] & 0xffff0000u);
        const float o0 = __uint_as_float(ow[j] << 16), o1 = __uint_as_float(ow[j] & 0xffff0000u);
        part += g0 * o0 + g1 * o1;
        if (dout.on()) dout.apply2(g0, g1, base + 2 * j);
        gd[2 * j] = g0;
        gd[2 * j + 1] = g1;
    }
#pragma unroll
    for (int off = 1; off < C8; off <<= 1) part += __shfl_xor(part, off);       // the C8 pieces of a (row, head) sit in neighbouring lanes
    if (ok) {
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(d.dq) + row * d.lddq + h * DH + 8 * c) = __builtin_bit_cast(uint4, pack8(gd));
        if (c == 0) {
            const int64_t bq = row / d.Lq, qq = row % d.Lq;
            dl[(bq * d.heads + h) * d.Lq + qq] = part * inv_keep;
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward, dQ phase
// A workgroup owns eight query tiles (one per wave: Q, d_o', lse, delta in registers, the query on the lane) and streams K / V.
constexpr int Q_NW = 8;
template <bool DROP>
__global__ __launch_bounds__(Q_NW * 64, 4) void attn_bwd_long_dq_kernel(const mmfm_attn_desc d, const float* dl, const float keep_scale) {
    constexpr int NW = Q_NW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LkP = (Lk + 31) & ~31;
    const int nqt = (Lq + 31) >> 5, nkt = LkP >> 5, nch = (LkP + CH - 1) / CH;
    char* Ks = smem;
    char* Vs = Ks + CH * RS;
    char* sct = Vs + CH * RS;                          // [NW][32][RS] dQ transpose tiles
    float* kbias = reinterpret_cast<float*>(sct + NW * 32 * RS);
    int* wflag = reinterpret_cast<int*>(kbias + LkP);
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH;      // d_o' (prep kernel)

    const int qt = blockIdx.y * NW + wave;
    const bool active = qt < nqt;
    const int q = qt * 32 + l31;
    Stager<NT> stg;
    stg.load(kg, d.ldk, vg, d.ldv, 0, Lk, t);
    // Q fragments stay in registers; the d_o' fragments of the wave's rows live in its (until the end unused) transpose tile and are
    // re-read per key tile: with both resident next to two dQ tiles, S and dP the kernel spills 33 registers
    char* dot = sct + wave * 32 * RS;
    bf16x8v qfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 qv = make_uint4(0u, 0u, 0u, 0u), gv = qv;
        if (active && q < Lq) {
            qv = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
            gv = *reinterpret_cast<const uint4*>(dog + (size_t)q * d.lddq + ks * 16 + 8 * kh);
        }
        qfr[ks] = __builtin_bit_cast(bf16x8v, qv);
        *reinterpret_cast<uint4*>(dot + l31 * RS + ks * 32 + kh * 16) = gv;
    }
    float lq = 0.f, dq_ = 0.f;
    if (active && q < Lq) {
        lq = d.lse[(size_t)bh_ * Lq + q] * LOG2E;
        dq_ = dl[(size_t)bh_ * Lq + q];
    }
    const int anypad = stage_kbias<NW>(kbias, wflag, d.keypad, b, Lk, LkP, t, wave, lane);
    const bool fixdiag = anypad && (d.flags & MMFM_ATTN_DIAG);
    const float c2 = d.scale * LOG2E, osc = d.scale * keep_scale;
    const masks_ptr mkp = reinterpret_cast<masks_ptr>(reinterpret_cast<uintptr_t>(d.keepbits)) + ((size_t)bh_ * nqt + (active ? qt : 0)) * nkt;
    f32x16 dQt[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nch; ++c) {
        // (no register prefetch of the next chunk here: with Q, d_o' and two dQ tiles resident the 16 staging registers spill)
        if (c > 0) stg.load(kg, d.ldk, vg, d.ldv, c * CH, Lk, t);
        __syncthreads();
        stg.store(Ks, Vs, t);
        __syncthreads();
        if (!active) continue;
        const int ntl = min(CH / 32, nkt - c * (CH / 32));
        for (int kl = 0; kl < ntl; ++kl) {
            const int kt = c * (CH / 32) + kl;
            const f32x16 s = score_tile(Ks, kbias + kt * 32, kl, qfr, fixdiag && kt == qt, kh, l31);       // S^T[key][q] + key bias
            f32x16 dpv = zero;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)                                                                    // dP^T[key][q]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Vs, (kl * 32 + l31) * RS + ks * 32 + kh * 16),
                                                              rowfrag(dot, l31 * RS + ks * 32 + kh * 16), dpv, 0, 0, 0);
            Masks16 mk;
            if (DROP) mk = ld_masks(mkp + kt);
            uint32_t sk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float e[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r + j], c2, -lq));       // padded keys: exp2(-inf) = 0
                    const float pm = DROP ? v_keep(p, mk.m[r + j]) : p;
                    e[j] = __builtin_fmaf(pm, dpv[r + j], -v_mul(p, dq_));                             // dS / dropout scale
                }
                asm volatile("" : "+v"(e[0]), "+v"(e[1]));
                sk[r >> 1] = pack2(e[0], e[1]);
            }
            const bf16x8v sf[2] = {__builtin_bit_cast(bf16x8v, make_uint4(sk[0], sk[1], sk[2], sk[3])),
                                   __builtin_bit_cast(bf16x8v, make_uint4(sk[4], sk[5], sk[6], sk[7]))};
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    dQt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Ks, RS, kl * 32 + 16 * s2, i * 32, lane), sf[s2], dQt[i], 0, 0, 0);
        }
    }
    if (!active) return;
    store_tile_T<DH, DT>(sct + wave * 32 * RS, RS, dQt, reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH, d.lddq, qt * 32, Lq,
                         lane, osc);
}
size_t dq_long_lds(int Lk) {
    const int LkP = (Lk + 31) & ~31;
    return (size_t)2 * CH * RS + (size_t)Q_NW * 32 * RS + (size_t)LkP * 4 + 64;
}

// ---------------------------------------------------------------------------------------------- backward, dK / dV phase
// A workgroup owns four key tiles (one per wave: K, V operands and the 2 x 2 accumulator tiles in registers, the key on the lane) and
// streams Q / d_o' with their per-query constants.  128 accumulator + operand registers per wave: two waves per SIMD.
constexpr int K_NW = 4;
template <bool DROP>
__global__ __launch_bounds__(K_NW * 64, 2) void attn_bwd_long_dkv_kernel(const mmfm_attn_desc d, const float* dl, const float keep_scale) {
    constexpr int NW = K_NW, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    const int nqt = LqP >> 5, nkt = LkP >> 5, nch = (LqP + CH - 1) / CH;
    char* As = smem;                                   // Q chunk
    char* Bs = As + CH * RS;                           // d_o' chunk
    float* lse2 = reinterpret_cast<float*>(Bs + CH * RS);          // [CH] lse * log2 e of the chunk's queries
    float* dlc = lse2 + CH;                                         // [CH] delta / dropout scale
    char* sct = reinterpret_cast<char*>(dlc + CH);                  // [NW][32][RS] output transpose tiles
    int* wflag = reinterpret_cast<int*>(sct + NW * 32 * RS);
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH;      // d_o' (prep kernel)

    const int kt = blockIdx.y * NW + wave;
    const bool active = kt < nkt;
    const int key = kt * 32 + l31;
    Stager<NT> stg;
    stg.load(qg, d.ldq, dog, d.lddq, 0, Lq, t);
    float lse_n = 0.f, dl_n = 0.f;                     // the next chunk's per-query constants (threads 0 .. CH-1)
    auto load_consts = [&](int c0) {
        lse_n = 0.f; dl_n = 0.f;
        if (t < CH && c0 + t < Lq) { lse_n = d.lse[(size_t)bh_ * Lq + c0 + t] * LOG2E; dl_n = dl[(size_t)bh_ * Lq + c0 + t]; }
    };
    load_consts(0);
    bf16x8v kfr[KS], vfr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
        if (active && key < Lk) {
            kv = *reinterpret_cast<const uint4*>(kg + (size_t)key * d.ldk + ks * 16 + 8 * kh);
            vv = *reinterpret_cast<const uint4*>(vg + (size_t)key * d.ldv + ks * 16 + 8 * kh);
        }
        kfr[ks] = __builtin_bit_cast(bf16x8v, kv);
        vfr[ks] = __builtin_bit_cast(bf16x8v, vv);
    }
    // key bias of this lane's key; "a key of the head is padded" decides between the literal-zero and the bias-initialised S product
    int pad = 0;
    for (int i = t; i < Lk; i += NT) pad |= (d.keypad != nullptr && d.keypad[(size_t)b * Lk + i] == 0) ? 1 : 0;
    const int wv = __any(pad) ? 1 : 0;
    if (lane == 0) wflag[wave] = wv;
    const bool kok = key < Lk && (d.keypad == nullptr || d.keypad[(size_t)b * Lk + (key < Lk ? key : 0)] != 0);
    const float kbv = kok ? 0.f : -INFINITY;
    // keep words of (query tile i, this wave's key tile): key l31 = mrow(r, kh') sits in word 2 r + kh' of its tile
    const uint32_t* kbp = reinterpret_cast<const uint32_t*>(d.keepbits) + ((size_t)bh_ * nqt * nkt + min(kt, nkt - 1)) * 32 +
                          2 * ((l31 & 3) + 4 * (l31 >> 3)) + ((l31 >> 2) & 1);
    const int kw_stride = nkt * 32;
    uint32_t kw_next = 0u;
    if (DROP) kw_next = kbp[0];
    __syncthreads();
    int anypad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) anypad |= wflag[w];
    // (keys beyond Lk need no bias here: their K / V rows are zero, their accumulator lanes are never stored and nothing is exchanged)
    anypad = __builtin_amdgcn_readfirstlane(anypad);
    const float c2 = d.scale * LOG2E;
    f32x16 dKt[DT], dVt[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dKt[i][r] = 0.f; dVt[i][r] = 0.f; }
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        stg.store(As, Bs, t);
        if (t < CH) { lse2[t] = lse_n; dlc[t] = dl_n; }
        if (c + 1 < nch) { stg.load(qg, d.ldq, dog, d.lddq, (c + 1) * CH, Lq, t); load_consts((c + 1) * CH); }
        __syncthreads();
        if (!active) continue;
        const int ntl = min(CH / 32, nqt - c * (CH / 32));
        for (int ql = 0; ql < ntl; ++ql) {
            const int qt = c * (CH / 32) + ql;
            f32x16 s, dpv = zero;
            if (!anypad) {
                s = zero;
            } else {
                const bool dfix = (d.flags & MMFM_ATTN_DIAG) && qt == kt;
                int lv = l31;
                asm volatile("" : "+v"(lv));
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (dfix && mrow(r, kh) == lv) ? 0.f : kbv;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (ql * 32 + l31) * RS + ks * 32 + kh * 16;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), kfr[ks], s, 0, 0, 0);        // S[q][key]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), vfr[ks], dpv, 0, 0, 0);    // dP[q][key]
            }
            uint32_t wq = 0u;
            if (DROP) {
                wq = kw_next >> (4 * kh);              // bit (r & 3) + 8 (r >> 2) is now register r's query
                kw_next = kbp[(size_t)min(qt + 1, nqt - 1) * kw_stride];
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint32_t pkp[4], pks[4];
                const int qb = ql * 32 + 16 * s2 + 4 * kh;             // chunk-local queries qb + 0..3 and qb + 8 + 0..3
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const float4 l4 = *reinterpret_cast<const float4*>(lse2 + qb + 8 * e4), d4 = *reinterpret_cast<const float4*>(dlc + qb + 8 * e4);
                    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
                    float pm[4], ds[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 8 * s2 + 4 * e4 + i;
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lq[i]));
                        pm[i] = p;
                        if (DROP) {
                            int m = __builtin_amdgcn_sbfe((int)wq, (r & 3) + 8 * (r >> 2), 1);
                            asm volatile("" : "+v"(m));
                            pm[i] = __uint_as_float(__float_as_uint(p) & (uint32_t)m);
                        }
                        ds[i] = __builtin_fmaf(pm[i], dpv[r], -v_mul(p, dq_[i]));
                    }
                    pkp[2 * e4] = pack2(pm[0], pm[1]); pkp[2 * e4 + 1] = pack2(pm[2], pm[3]);
                    pks[2 * e4] = pack2(ds[0], ds[1]); pks[2 * e4 + 1] = pack2(ds[2], ds[3]);
                }
                const bf16x8v pf = __builtin_bit_cast(bf16x8v, make_uint4(pkp[0], pkp[1], pkp[2], pkp[3]));
                const bf16x8v sf = __builtin_bit_cast(bf16x8v, make_uint4(pks[0], pks[1], pks[2], pks[3]));
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Bs, RS, ql * 32 + 16 * s2, i * 32, lane), pf, dVt[i], 0, 0, 0);
                    dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, RS, ql * 32 + 16 * s2, i * 32, lane), sf, dKt[i], 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;
    char* tl = sct + wave * 32 * RS;
    store_tile_T<DH, DT>(tl, RS, dKt, reinterpret_cast<uint16_t*>(d.dk) + (size_t)b * Lk * d.lddk + h * DH, d.lddk, kt * 32, Lk, lane, d.scale * keep_scale);
    store_tile_T<DH, DT>(tl, RS, dVt, reinterpret_cast<uint16_t*>(d.dv) + (size_t)b * Lk * d.lddv + h * DH, d.lddv, kt * 32, Lk, lane, keep_scale);
}
size_t dkv_long_lds() { return (size_t)2 * CH * RS + (size_t)2 * CH * 4 + (size_t)K_NW * 32 * RS + 64; }

}  // namespace

int mmfm_attn_keepbits_launch(const mmfm_attn_desc& d, hipStream_t st);          // attention_fast.hip

// dh = 64 with the keep-bit workspace.  Returns -1000 when the general kernels must run.
int mmfm_attn_long_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st) {
    static const bool off = [] { const char* e = getenv("MMFM_ATTN_LONG"); return e && atoi(e) == 0; }();
    if (off || d.dh != DH || d.keepbits == nullptr || (d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP))) return -1000;
    if (d.Lq % 8 || d.Lk % 8) return -1000;
    const bool drop = d.drop_p.p > 0.f && d.drop_p.state != nullptr;
    if (drop && d.drop_p.p >= 1.f) return -1000;
    const bool al = d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 && (uintptr_t)d.q % 16 == 0 &&
                    (uintptr_t)d.k % 16 == 0 && (uintptr_t)d.v % 16 == 0 && (uintptr_t)d.o % 16 == 0 && (uintptr_t)d.keepbits % 128 == 0;
    if (!al) return -1000;
    const bool alb = !backward || (d.lddo % 8 == 0 && d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0 && (uintptr_t)d.d_o % 16 == 0 &&
                                   (uintptr_t)d.dq % 16 == 0 && (uintptr_t)d.dk % 16 == 0 && (uintptr_t)d.dv % 16 == 0);
    if (!alb) return mmfm_set_error(-1, "mmfm_attn_bwd(bf16, dh 64): gradient tensors must be 16-byte aligned with leading dims %% 8 == 0 on the "
                                        "keep-bit path (mmfm_attn_desc.keepbits)");
    const int nqt = (d.Lq + 31) / 32, nkt = (d.Lk + 31) / 32, bh = d.B * d.heads;
    const float keep_scale = drop ? 1.f / mmfm_attn_keep_prob(d.drop_p.p) : 1.f;
    // delta / dropout scale per (b, head, query): the tail of the keep-bit workspace, behind the bit tiles
    float* dl = reinterpret_cast<float*>(reinterpret_cast<char*>(d.keepbits) + (size_t)bh * nqt * nkt * 128);
#define LAUNCH(KERN, GRID, NTH, LDSB, WHAT, ...)                                                                    \
    {                                                                                                               \
        auto kern = KERN;                                                                                           \
        if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(kern), LDSB, WHAT)) return rc;                   \
        hipLaunchKernelGGL(kern, GRID, dim3(NTH), LDSB, st, __VA_ARGS__);                                           \
    }
    if (!backward) {
        if (drop) { if (int rc = mmfm_attn_keepbits_launch(d, st)) return rc; }
        const size_t lds = fwd_long_lds(d.Lk);
        const dim3 grid(bh, (nqt + F_NW - 1) / F_NW);
        if (drop) LAUNCH(attn_fwd_long_kernel<true>, grid, F_NW * 64, lds, "mmfm_attn_fwd(bf16, dh 64)", d, keep_scale)
        else LAUNCH(attn_fwd_long_kernel<false>, grid, F_NW * 64, lds, "mmfm_attn_fwd(bf16, dh 64)", d, keep_scale)
        MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16, dh 64)");
        return 0;
    }
    const int64_t pieces = (int64_t)d.B * d.Lq * d.heads * C8;
    hipLaunchKernelGGL(attn_bwd_long_prep_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, d, dl, 1.f / keep_scale);
    {
        const size_t lds = dkv_long_lds();
        const dim3 grid(bh, (nkt + K_NW - 1) / K_NW);
        if (drop) LAUNCH(attn_bwd_long_dkv_kernel<true>, grid, K_NW * 64, lds, "mmfm_attn_bwd(bf16, dh 64, dK dV)", d, dl, keep_scale)
        else LAUNCH(attn_bwd_long_dkv_kernel<false>, grid, K_NW * 64, lds, "mmfm_attn_bwd(bf16, dh 64, dK dV)", d, dl, keep_scale)
    }
    {
        const size_t lds = dq_long_lds(d.Lk);
        const dim3 grid(bh, (nqt + Q_NW - 1) / Q_NW);
        if (drop) LAUNCH(attn_bwd_long_dq_kernel<true>, grid, Q_NW * 64, lds, "mmfm_attn_bwd(bf16, dh 64, dQ)", d, dl, keep_scale)
        else LAUNCH(attn_bwd_long_dq_kernel<false>, grid, Q_NW * 64, lds, "mmfm_attn_bwd(bf16, dh 64, dQ)", d, dl, keep_scale)
    }
#undef LAUNCH
    MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16, dh 64)");
    return 0;
}
