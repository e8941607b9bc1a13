// bf16 MFMA attention (throughput mode), forward and backward: v_mfma_f32_32x32x16_bf16, fp32
// softmax statistics.  Same masks / dropout / LSE contract as attention.hip (the fp32 parity path).
//
// One workgroup per (batch, head); Q/K/V(/dO) of the head live in LDS as bf16 rows padded by 16 B
// (80-B rows at dh = 32: a ds_read_b128 of 8 consecutive d per lane is conflict-free).
// Orientation is chosen so that every product that follows a softmax sums over the accumulator's
// ROW index: the probabilities (and dS) never leave registers — registers 8s..8s+7 of the 32x32
// fp32 tile, packed to bf16, ARE the B operand of k-step s, whose k slots then mean rows
// 16s + 8(j>>2) + 4h + (j&3); the other operand is gathered in exactly that order with
// ds_read_b64_tr_b16 (hardware-transposed LDS read), so V, Q, dO and K are staged once, row-major.
//   forward : S^T = K Q^T  ->  online softmax (lane = query)  ->  O^T += V^T P^T
//   backward: phase A (wave owns 32 keys):  S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
//             phase B (wave owns 32 queries): S^T, dP^T, dQ^T += K^T dS^T        (no atomics)
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

__device__ __forceinline__ int mrow(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

__device__ __forceinline__ bf16x8v pack8(const float* p) {
    bf16x8v v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)p[j];
    return v;
}
__device__ __forceinline__ bf16x8v rowfrag(const char* S, int byte) {
    return __builtin_bit_cast(bf16x8v, *reinterpret_cast<const uint4*>(S + byte));
}
// Transposed operand from a row-major [row][col] bf16 image with RS-byte rows: lane (c = lane%32, h = lane/32)
// gets element j = image[rbase + 8*(j>>2) + 4*h + (j&3)][cbase + c]   (the k order of an accumulator-fed MFMA).
__device__ __forceinline__ bf16x8v trfrag(const char* S, int RS, int rbase, int cbase, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int row = rbase + 4 * (g >> 1) + q;
    const int col = cbase + 16 * (g & 1) + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + row * RS + col * 2));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + (row + 8) * RS + col * 2));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8v, v);
}

struct MaskCtx {
    const uint8_t* kpad;
    const uint8_t* modl;
    int flags;
    __device__ __forceinline__ bool allowed(int q, int k) const {
        bool a = (flags & MMFM_ATTN_CAUSAL) ? (k <= q) : (kpad[k] != 0);
        if ((flags & MMFM_ATTN_DIAG) && q == k) a = true;
        if ((flags & MMFM_ATTN_SEP) && modl[q] != modl[k]) a = true;
        return a;
    }
};

// Dropout on the probabilities, two decisions per hash: element (q, key) uses the low (even key) or high
// (odd key) 16 bits of hash(pair index), pair index = (bh*Lq + q) * ceil(Lk/2) + key/2.  Forward and both
// backward phases evaluate the same function, so no mask is stored.
struct Drop16 {
    uint32_t k0, k1, t16;
    float scale;
    bool on;
    __device__ __forceinline__ uint32_t hash(uint64_t pidx) const {
        return mix32(mix32((uint32_t)pidx ^ k0) + k1 + (uint32_t)(pidx >> 32) * 0x9E3779B9u);
    }
};
__device__ __forceinline__ Drop16 drop16_init(mmfm_dropout d) {
    const Drop b = drop_init(d);
    Drop16 r;
    r.k0 = b.k0; r.k1 = b.k1; r.t16 = b.thresh >> 16; r.scale = b.scale; r.on = b.on();
    return r;
}

// rows [0,L) x DH bf16 of one head -> LDS image with RS-byte rows and CPR 16-B chunks per row; the rest zero
template <int DH>
__device__ __forceinline__ void load_head16(char* __restrict__ dst, int RS, int CPR, const uint16_t* __restrict__ src, int ld, int L, int LP,
                                            int t, int nthreads) {
    constexpr int C8 = DH / 8;
    for (int idx = t; idx < LP * CPR; idx += nthreads) {
        const int row = idx / CPR, c = idx % CPR;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < L && c < C8) v = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + 8 * c);
        *reinterpret_cast<uint4*>(dst + row * RS + c * 16) = v;
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ============================================================================ forward
// element-wise part of one 32-key x 32-query tile: online softmax + dropout.  FULL = every (q, key) of the tile is
// valid and allowed (no padding, no causal/sep flags): the mask logic disappears.
template <bool FULL, bool DROP>
__device__ __forceinline__ bool fwd_tile(const f32x16& st, float c2, float& m_run, float& l_run, float& alpha, float (&pd)[16], int q,
                                         int Lq, int Lk, int kt, int kh, const MaskCtx& mk, const Drop16& dp, uint64_t pair_base) {
    uint32_t okm = 0xffffu;
    float mx = -INFINITY;
    if (FULL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
    } else {
        okm = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + mrow(r, kh);
            if ((key < Lk) && (q < Lq) && mk.allowed(q, key)) { okm |= 1u << r; mx = fmaxf(mx, st[r]); }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * c2);                      // c2 > 0
    if (__all(m_new == -INFINITY)) return false;
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_use);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -m_use));
        float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r + 1], c2, -m_use));
        if (!FULL) {
            p0 = (okm >> r) & 1 ? p0 : 0.f;
            p1 = (okm >> (r + 1)) & 1 ? p1 : 0.f;
        }
        ps += p0 + p1;
        if (DROP) {
            const uint32_t hsh = dp.hash(pair_base + (uint64_t)((kt * 32 + mrow(r, kh)) >> 1));
            p0 = (hsh & 0xffffu) >= dp.t16 ? p0 * dp.scale : 0.f;
            p1 = (hsh >> 16) >= dp.t16 ? p1 * dp.scale : 0.f;
        }
        pd[r] = p0;
        pd[r + 1] = p1;
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    return true;
}

template <int DH, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_bf16_kernel(const mmfm_attn_desc d) {
    constexpr int KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int KRS = DH * 2 + 16;          // K rows (row reads)
    constexpr int VRS = DT * 64;              // V rows (transposed reads only), zero padded to 32 columns
    constexpr int SLD = DT * 32 + 1;
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / d.heads, h = blockIdx.x % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LkP = (Lk + 31) & ~31, Lmx = max(Lq, Lk);
    char* Ks = smem;
    char* Vs = Ks + LkP * KRS;
    float* Sc = reinterpret_cast<float*>(Vs + LkP * VRS);
    int* wflag = reinterpret_cast<int*>(Sc + NW * 32 * SLD);          // [NW] per-wave "all keys valid" votes
    uint8_t* kpad = reinterpret_cast<uint8_t*>(wflag + NW);
    uint8_t* modl = kpad + LkP;
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    uint16_t* og = reinterpret_cast<uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;

    load_head16<DH>(Ks, KRS, KRS / 16, kg, d.ldk, Lk, LkP, t, NT);
    load_head16<DH>(Vs, VRS, VRS / 16, vg, d.ldv, Lk, LkP, t, NT);
    int allk = 1;
    for (int i = t; i < LkP; i += NT) {
        const uint8_t v = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
        kpad[i] = v;
        if (i < Lk) allk &= (v != 0);
    }
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += NT) modl[i] = d.mod_id[i];
    // block-wide AND through the dynamic LDS (no static __shared__: it would shift the 16-B aligned carve-up
    // and shrink the >64 KB opt-in limit)
    const int wave_vote = __all(allk) ? 1 : 0;      // all 64 lanes vote BEFORE any divergence
    if (lane == 0) wflag[wave] = wave_vote;
    __syncthreads();
    int vote = 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) vote &= wflag[w];
    const bool nomask = vote && !(d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP));

    const MaskCtx mk{kpad, modl, d.flags};
    const Drop16 dp = drop16_init(d.drop_p);
    const Drop dout = drop_init(d.drop_o);
    float* sc = Sc + wave * 32 * SLD;
    const int nqt = (Lq + 31) / 32, nkt = LkP / 32, LkH = (Lk + 1) >> 1;
    const float c2 = d.scale * LOG2E;

    for (int qt = wave; qt < nqt; qt += NW) {
        const int q0 = qt * 32, q = q0 + l31;
        const bool qfull = nomask && (q0 + 32 <= Lq);
        const uint64_t pair_base = ((uint64_t)blockIdx.x * Lq + (uint64_t)q) * (uint64_t)LkH;
        bf16x8v qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (q < Lq) v = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
            qf[ks] = __builtin_bit_cast(bf16x8v, v);
        }
        float m_run = -INFINITY, l_run = 0.f;
        f32x16 acc[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kt * 32 + l31) * KRS + ks * 32 + kh * 16), qf[ks], st, 0, 0, 0);
            float alpha, pd[16];
            bool live;
            if (qfull && kt * 32 + 32 <= Lk)
                live = dp.on ? fwd_tile<true, true>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, pair_base)
                             : fwd_tile<true, false>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, pair_base);
            else
                live = dp.on ? fwd_tile<false, true>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, pair_base)
                             : fwd_tile<false, false>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, pair_base);
            if (!live) continue;
            const bf16x8v pf0 = pack8(pd), pf1 = pack8(pd + 8);
            const bool rescale = !__all(alpha == 1.f);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, VRS, kt * 32, i * 32, lane), pf0, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, VRS, kt * 32 + 16, i * 32, lane), pf1, acc[i], 0, 0, 0);
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.f / l_tot;
        if (kh == 0 && q < Lq) d.lse[(size_t)blockIdx.x * Lq + q] = m_run * LN2 + __logf(l_tot);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[l31 * SLD + i * 32 + mrow(r, kh)] = acc[i][r] * inv;
        wave_lds_fence();
        constexpr int C4 = DH / 4;
        for (int idx = lane; idx < 32 * C4; idx += 64) {
            const int row = idx / C4, c = idx % C4;
            if (q0 + row < Lq) {
                const float* p = sc + row * SLD + 4 * c;
                const uint64_t base = ((uint64_t)b * Lq + (uint64_t)(q0 + row)) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 4 * c);
                float4 v;
                v.x = dout.apply(p[0], base + 0); v.y = dout.apply(p[1], base + 1);
                v.z = dout.apply(p[2], base + 2); v.w = dout.apply(p[3], base + 3);
                io<uint16_t>::st4(og + (size_t)(q0 + row) * d.ldo + 4 * c, v);
            }
        }
        wave_lds_fence();
    }
}

// ============================================================================ backward
// Phase A, one half-tile (accumulator rows 8*s2 .. 8*s2+7 = 8 queries, lane = key): P~ (dropped, scaled) and dS.
template <bool FULL, bool DROP>
__device__ __forceinline__ void bwdA_half(const f32x16& s, const f32x16& dpv, int s2, float (&pd)[8], float (&ds)[8], float c2, float scale,
                                          const float* lse2, const float* dlt, int qt, int key, int kh, int Lq, int Lk, const MaskCtx& mk,
                                          const Drop16& dp, uint64_t pbase, int LkH) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = 8 * s2 + e;
        const int q = qt * 32 + mrow(r, kh);
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lse2[q]));
        if (!FULL) p = ((q < Lq) && (key < Lk) && mk.allowed(q, key)) ? p : 0.f;
        float g = dpv[r];
        float pdrop = p;
        if (DROP) {
            const uint32_t hsh = dp.hash((pbase + (uint64_t)q) * (uint64_t)LkH + (uint64_t)(key >> 1));
            const bool keep = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= dp.t16;
            pdrop = keep ? p * dp.scale : 0.f;
            g = keep ? g * dp.scale : 0.f;
        }
        pd[e] = pdrop;
        ds[e] = p * (g - dlt[q]) * scale;
    }
}

// Phase B, one tile (lane = query, accumulator rows = keys): dS^T.
template <bool FULL, bool DROP>
__device__ __forceinline__ void bwdB_tile(const f32x16& s, const f32x16& dpv, float (&ds)[16], float c2, float scale, float lq, float dq_,
                                          int q, int kt, int kh, int Lq, int Lk, const MaskCtx& mk, const Drop16& dp, uint64_t pair_base) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const int key = kt * 32 + mrow(r, kh);
        float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lq));
        float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r + 1], c2, -lq));
        if (!FULL) {
            p0 = ((q < Lq) && (key < Lk) && mk.allowed(q, key)) ? p0 : 0.f;
            p1 = ((q < Lq) && (key + 1 < Lk) && mk.allowed(q, key + 1)) ? p1 : 0.f;
        }
        float g0 = dpv[r], g1 = dpv[r + 1];
        if (DROP) {
            const uint32_t hsh = dp.hash(pair_base + (uint64_t)(key >> 1));
            g0 = (hsh & 0xffffu) >= dp.t16 ? g0 * dp.scale : 0.f;
            g1 = (hsh >> 16) >= dp.t16 ? g1 * dp.scale : 0.f;
        }
        ds[r] = p0 * (g0 - dq_) * scale;
        ds[r + 1] = p1 * (g1 - dq_) * scale;
    }
}

// PHASE 0: dK, dV (waves own key tiles)   PHASE 1: dQ (waves own query tiles)   PHASE 2: both in one launch.
// Two single-phase launches keep each kernel under 256 registers at two waves per SIMD (no spills).
template <int DH, int NW, int PHASE>
__global__ __launch_bounds__(NW * 64) void attn_bwd_bf16_kernel(const mmfm_attn_desc d) {
    constexpr int KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int RS = DH * 2 + 16;
    constexpr int SLD = 33;
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / d.heads, h = blockIdx.x % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, Lmx = max(Lq, Lk);
    char* Qs = smem;
    char* dOs = Qs + LqP * RS;
    char* Ks = dOs + LqP * RS;
    char* Vs = Ks + LkP * RS;
    float* lse2 = reinterpret_cast<float*>(Vs + LkP * RS);    // lse * log2(e)
    float* dlt = lse2 + LqP;
    float* Sc = dlt + LqP;
    int* wflag = reinterpret_cast<int*>(Sc + NW * 32 * SLD);
    uint8_t* kpad = reinterpret_cast<uint8_t*>(wflag + NW);
    uint8_t* modl = kpad + LkP;
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * DH;
    const Drop16 dp = drop16_init(d.drop_p);
    const Drop dout = drop_init(d.drop_o);

    load_head16<DH>(Qs, RS, RS / 16, qg, d.ldq, Lq, LqP, t, NT);
    load_head16<DH>(Ks, RS, RS / 16, kg, d.ldk, Lk, LkP, t, NT);
    load_head16<DH>(Vs, RS, RS / 16, vg, d.ldv, Lk, LkP, t, NT);
    {   // dO = dropout'(d_o) as bf16;  delta = rowsum(d_o * o)
        constexpr int C8 = DH / 8;
        for (int idx = t; idx < LqP * C8; idx += NT) {
            const int row = idx / C8, c = idx % C8;
            uint4 g = make_uint4(0u, 0u, 0u, 0u), o = g;
            if (row < Lq) {
                g = *reinterpret_cast<const uint4*>(dog + (size_t)row * d.lddo + 8 * c);
                o = *reinterpret_cast<const uint4*>(og + (size_t)row * d.ldo + 8 * c);
            }
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {o.x, o.y, o.z, o.w};
            const uint64_t base = ((uint64_t)b * Lq + (uint64_t)row) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 8 * c);
            float part = 0.f, gd[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g0 = __uint_as_float(gw[j] << 16), g1 = __uint_as_float(gw[j] & 0xffff0000u);
                const float o0 = __uint_as_float(ow[j] << 16), o1 = __uint_as_float(ow[j] & 0xffff0000u);
                part += g0 * o0 + g1 * o1;
                gd[2 * j] = dout.apply(g0, base + 2 * j);
                gd[2 * j + 1] = dout.apply(g1, base + 2 * j + 1);
            }
#pragma unroll
            for (int off = 1; off < C8; off <<= 1) part += __shfl_xor(part, off);
            if (c == 0) dlt[row] = part;
            *reinterpret_cast<uint4*>(dOs + row * RS + c * 16) = __builtin_bit_cast(uint4, pack8(gd));
        }
        constexpr int PADC = RS / 16 - C8;           // zero the pad chunk(s) of dOs rows
        for (int idx = t; idx < LqP * PADC; idx += NT)
            *reinterpret_cast<uint4*>(dOs + (idx / PADC) * RS + (C8 + idx % PADC) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int i = t; i < LqP; i += NT) lse2[i] = (i < Lq) ? d.lse[(size_t)blockIdx.x * Lq + i] * LOG2E : 0.f;
    int allk = 1;
    for (int i = t; i < LkP; i += NT) {
        const uint8_t v = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
        kpad[i] = v;
        if (i < Lk) allk &= (v != 0);
    }
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += NT) modl[i] = d.mod_id[i];
    const int wave_vote = __all(allk) ? 1 : 0;      // all 64 lanes vote BEFORE any divergence
    if (lane == 0) wflag[wave] = wave_vote;
    __syncthreads();
    int vote = 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) vote &= wflag[w];
    const bool nomask = vote && !(d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP));

    const MaskCtx mk{kpad, modl, d.flags};
    float* sc = Sc + wave * 32 * SLD;
    const int nqt = LqP / 32, nkt = LkP / 32, LkH = (Lk + 1) >> 1;
    const uint64_t pbase = (uint64_t)blockIdx.x * Lq;
    const float c2 = d.scale * LOG2E;
    constexpr int CW = (DH < 32 ? DH : 32) / 4;

    // ---------------- phase A: wave owns key tile kt -> dK, dV
    if constexpr (PHASE != 1)
    for (int kt = wave; kt < nkt; kt += NW) {
        f32x16 dKt[DT], dVt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dKt[i][r] = 0.f; dVt[i][r] = 0.f; }
        const int key = kt * 32 + l31;
        bf16x8v kfr[KS], vfr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kfr[ks] = rowfrag(Ks, (kt * 32 + l31) * RS + ks * 32 + kh * 16);
            vfr[ks] = rowfrag(Vs, (kt * 32 + l31) * RS + ks * 32 + kh * 16);
        }
        for (int qt = 0; qt < nqt; ++qt) {
            f32x16 s, dpv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (qt * 32 + l31) * RS + ks * 32 + kh * 16;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Qs, off), kfr[ks], s, 0, 0, 0);        // S[q][key]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(dOs, off), vfr[ks], dpv, 0, 0, 0);   // dP[q][key]
            }
            const bool full = nomask && (kt * 32 + 32 <= Lk) && (qt * 32 + 32 <= Lq);
            // two half-tiles of 8 accumulator rows each: softmax/dropout algebra, pack to bf16, feed the MFMAs
            // (keeps only 16 fp32 temporaries live instead of 32)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pd[8], ds[8];
                if (full) {
                    if (dp.on) bwdA_half<true, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, pbase, LkH);
                    else bwdA_half<true, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, pbase, LkH);
                } else {
                    if (dp.on) bwdA_half<false, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, pbase, LkH);
                    else bwdA_half<false, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, pbase, LkH);
                }
                const bf16x8v pf = pack8(pd), sf = pack8(ds);
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(dOs, RS, qt * 32 + 16 * s2, i * 32, lane), pf, dVt[i], 0, 0, 0);
                    dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Qs, RS, qt * 32 + 16 * s2, i * 32, lane), sf, dKt[i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int ldo_ = which ? d.lddv : d.lddk;
            uint16_t* outg = reinterpret_cast<uint16_t*>(which ? d.dv : d.dk) + (size_t)b * Lk * ldo_ + h * DH;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[l31 * SLD + mrow(r, kh)] = which ? dVt[i][r] : dKt[i][r];
                wave_lds_fence();
                for (int idx = lane; idx < 32 * CW; idx += 64) {
                    const int row = idx / CW, c = idx % CW;
                    if (kt * 32 + row < Lk) {
                        const float* p = sc + row * SLD + 4 * c;
                        io<uint16_t>::st4(outg + (size_t)(kt * 32 + row) * ldo_ + i * 32 + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
                    }
                }
            }
        }
    }

    // ---------------- phase B: wave owns query tile qt -> dQ
    if constexpr (PHASE != 0)
    for (int qt = wave; qt < nqt; qt += NW) {
        f32x16 dQt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        const int q = qt * 32 + l31;
        const float lq = lse2[q], dq_ = dlt[q];
        bf16x8v qfr[KS], dofr[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qfr[ks] = rowfrag(Qs, (qt * 32 + l31) * RS + ks * 32 + kh * 16);
            dofr[ks] = rowfrag(dOs, (qt * 32 + l31) * RS + ks * 32 + kh * 16);
        }
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 s, dpv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (kt * 32 + l31) * RS + ks * 32 + kh * 16;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, off), qfr[ks], s, 0, 0, 0);        // S^T[key][q]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Vs, off), dofr[ks], dpv, 0, 0, 0);   // dP^T[key][q]
            }
            float ds[16];
            const bool full = nomask && (kt * 32 + 32 <= Lk) && (qt * 32 + 32 <= Lq);
            const uint64_t pair_base = (pbase + (uint64_t)q) * (uint64_t)LkH;
            if (full) {
                if (dp.on) bwdB_tile<true, true>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, pair_base);
                else bwdB_tile<true, false>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, pair_base);
            } else {
                if (dp.on) bwdB_tile<false, true>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, pair_base);
                else bwdB_tile<false, false>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, pair_base);
            }
            const bf16x8v sf[2] = {pack8(ds), pack8(ds + 8)};
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    dQt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Ks, RS, kt * 32 + 16 * s2, i * 32, lane), sf[s2], dQt[i], 0, 0, 0);
        }
        uint16_t* outg = reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[l31 * SLD + mrow(r, kh)] = dQt[i][r];
            wave_lds_fence();
            for (int idx = lane; idx < 32 * CW; idx += 64) {
                const int row = idx / CW, c = idx % CW;
                if (qt * 32 + row < Lq) {
                    const float* p = sc + row * SLD + 4 * c;
                    io<uint16_t>::st4(outg + (size_t)(qt * 32 + row) * d.lddq + i * 32 + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
                }
            }
        }
    }
}

// waves per workgroup of the backward kernel: 8 (two per SIMD, 256-register budget) or 4 (one per SIMD, 512)
int bwd_waves() {
    static int w = [] { const char* e = getenv("MMFM_ATTN_BWD_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
    return w;
}

constexpr int FWD_WAVES = 8;
size_t fwd_lds(int Lq, int Lk, int dh) {
    const int DT = (dh + 31) / 32, LkP = (Lk + 31) & ~31;
    return (size_t)LkP * (dh * 2 + 16) + (size_t)LkP * DT * 64 + (size_t)FWD_WAVES * 32 * (DT * 32 + 1) * 4 + LkP + std::max(Lq, Lk) + 64;
}
size_t bwd_lds(int Lq, int Lk, int dh, int nw) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    return (size_t)(2 * LqP + 2 * LkP) * (dh * 2 + 16) + (size_t)2 * LqP * 4 + (size_t)nw * 32 * 33 * 4 + LkP + std::max(Lq, Lk) + 64;
}

int opt_in_lds(const void* kern, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, bool> done;
    if (bytes <= 65536) return 0;
    std::lock_guard<std::mutex> g(mu);
    if (done.count(kern)) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mmfm_set_error((int)e, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    done[kern] = true;
    return 0;
}

}  // namespace

// returns MMFM_NOT_HANDLED (-1000) if this path does not take the shape (the caller falls back to the generic
// kernel), 0 on launch, otherwise an error code
int mmfm_attn_bf16_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st) {
    if (!(d.dh == 16 || d.dh == 32 || d.dh == 64)) return -1000;
    const bool al = d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 && (uintptr_t)d.q % 16 == 0 &&
                    (uintptr_t)d.k % 16 == 0 && (uintptr_t)d.v % 16 == 0 && (uintptr_t)d.o % 16 == 0;
    if (!al) return -1000;
    if (backward) {
        const bool alb = d.lddo % 8 == 0 && d.lddq % 4 == 0 && d.lddk % 4 == 0 && d.lddv % 4 == 0 && (uintptr_t)d.d_o % 16 == 0;
        if (!alb) return -1000;
        const int nw = bwd_waves();
        const size_t lds = bwd_lds(d.Lq, d.Lk, d.dh, nw);
        if (lds > 160 * 1024) return -1000;
#define BWD1(DHV, NWV, PH)                                                                                        \
        {                                                                                                         \
            auto kern = attn_bwd_bf16_kernel<DHV, NWV, PH>;                                                       \
            if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), lds)) return rc;                         \
            hipLaunchKernelGGL(kern, dim3(d.B * d.heads), dim3(NWV * 64), lds, st, d);                            \
        }
#define BWD(DHV)                                                                                                  \
        if (nw == 8) { BWD1(DHV, 8, 0) BWD1(DHV, 8, 1) } else { BWD1(DHV, 4, 2) }
        if (d.dh == 16) { BWD(16) } else if (d.dh == 32) { BWD(32) } else { BWD(64) }
#undef BWD
#undef BWD1
        MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16)");
        return 0;
    }
    const size_t lds = fwd_lds(d.Lq, d.Lk, d.dh);
    if (lds > 160 * 1024) return -1000;
#define FWD(DHV)                                                                                                  \
    {                                                                                                             \
        auto kern = attn_fwd_bf16_kernel<DHV, FWD_WAVES>;                                                         \
        if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), lds)) return rc;                             \
        hipLaunchKernelGGL(kern, dim3(d.B * d.heads), dim3(FWD_WAVES * 64), lds, st, d);                          \
    }
    if (d.dh == 16) FWD(16) else if (d.dh == 32) FWD(32) else FWD(64)
#undef FWD
    MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16)");
    return 0;
}
