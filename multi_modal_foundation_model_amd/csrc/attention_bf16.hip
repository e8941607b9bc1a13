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
#include "attn_common.h"
#include <algorithm>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

using namespace attn;

namespace {

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

// ============================================================================ forward
// element-wise part of one 32-key x 32-query tile: online softmax + dropout.  FULL = every (q, key) of the tile is
// valid and allowed (no padding, no causal/sep flags): the mask logic disappears.
// G = number of 8-key groups of the tile that hold any valid key (accumulator registers 4g..4g+3 are keys 8g + 4*kh + 0..3):
// the last key tile of a 200-token head has one (keys 192..199), so three quarters of its element-wise work is skipped.
template <bool FULL, bool DROP>
__device__ __forceinline__ bool fwd_tile(const f32x16& st, float c2, float& m_run, float& l_run, float& alpha, float (&pd)[16], int q,
                                         int Lq, int Lk, int kt, int kh, const MaskCtx& mk, const Drop16& dp, uint32_t ka, uint32_t kb, int G) {
    uint32_t okm = 0xffffu;
    float mx = -INFINITY;
    if (FULL) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < G) {
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) mx = fmaxf(mx, st[r]);
            }
    } else {
        okm = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < G) {
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) {
                    const int key = kt * 32 + mrow(r, kh);
                    if ((key < Lk) && (q < Lq) && mk.allowed(q, key)) { okm |= 1u << r; mx = fmaxf(mx, st[r]); }
                }
            }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * c2);                      // c2 > 0
    if (__all(m_new == -INFINITY)) return false;
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_use);
    float ps = 0.f;
    // pair index of register r: (kt*32 + mrow(r, kh)) >> 1 = (16 kt + 2 kh) | (mrow(r, 0) >> 1): disjoint bits, so the tile part is
    // XOR-ed into the row key once and every pair costs one XOR with a literal
    const uint32_t jbase = (uint32_t)(16 * kt + 2 * kh) ^ ka;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < G) {
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; r += 2) {
                float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -m_use));
                float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r + 1], c2, -m_use));
                if (!FULL) {
                    p0 = (okm >> r) & 1 ? p0 : 0.f;
                    p1 = (okm >> (r + 1)) & 1 ? p1 : 0.f;
                }
                ps += p0 + p1;
                if (DROP) {
                    const uint32_t hsh = dp.hash(jbase ^ (uint32_t)(mrow(r, 0) >> 1), kb);
                    p0 = (hsh & 0xffffu) >= dp.t16 ? p0 : 0.f;       // the 1/(1-p) factor rides on the final normalisation
                    p1 = (hsh >> 16) >= dp.t16 ? p1 : 0.f;
                }
                pd[r] = p0;
                pd[r + 1] = p1;
            }
        } else {
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) pd[r] = 0.f;
        }
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    return true;
}

// __launch_bounds__ second argument = waves per SIMD the register allocation must allow: 3 -> <= 168 VGPRs, so three
// 4-wave workgroups (49 KB of LDS each at L = 200, dh = 32) co-reside on a CU
template <int DH, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 2) void attn_fwd_bf16_kernel(const mmfm_attn_desc d) {
    constexpr int KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int KRS = DH * 2 + 16;          // K rows (row reads)
    constexpr int VRS = DT * 64;              // V rows (transposed reads only), zero padded to 32 columns
    constexpr int SLD = DT * 32 + 1;
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
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
    const int nqt = (Lq + 31) / 32, nkt = LkP / 32;
    const float c2 = d.scale * LOG2E;

    for (int qt = wave; qt < nqt; qt += NW) {
        const int q0 = qt * 32, q = q0 + l31;
        uint32_t ka, kb;                                   // dropout row keys of this lane's query
        dp.rowkeys((uint32_t)bh_ * (uint32_t)Lq + (uint32_t)q, ka, kb);
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

        // S^T of tile kt+1 is issued before the softmax of tile kt: the matrix pipe works under the VALU-heavy
        // softmax instead of stalling it at the top of every iteration
        auto score = [&](int kt) {
            f32x16 acc_s;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kt * 32 + l31) * KRS + ks * 32 + kh * 16), qf[ks], acc_s, 0, 0, 0);
            return acc_s;
        };
        // one key tile: softmax algebra on its scores, then O^T += V^T P^T
        auto tile = [&](int kt, const f32x16& st) {
            float alpha, pd[16];
            bool live;
            // valid keys of this tile in groups of 8; with no mask in play and whole groups the mask-free code runs on
            // the valid groups only.  Lanes of queries >= Lq then carry finite garbage that is never stored (their keys'
            // zero-padded rows give finite scores) - the softmax of a query is lane-local.
            const int nk = min(32, Lk - kt * 32), G = (nk + 7) >> 3;
            if (nomask && (nk & 7) == 0)
                live = dp.on ? fwd_tile<true, true>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G)
                             : fwd_tile<true, false>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G);
            else
                live = dp.on ? fwd_tile<false, true>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G)
                             : fwd_tile<false, false>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G);
            if (!live) return;
            const bf16x8v pf0 = pack8(pd), pf1 = pack8(pd + 8);
            const bool rescale = !__all(alpha == 1.f);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, VRS, kt * 32, i * 32, lane), pf0, acc[i], 0, 0, 0);
                if (G > 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, VRS, kt * 32 + 16, i * 32, lane), pf1, acc[i], 0, 0, 0);
            }
        };
        // two score tiles in flight, ping-pong (no accumulator copies: a rotating `st = st_next` costs 32 v_mov per tile in
        // this VALU-bound loop): the scores of tile kt+1 are issued before the element-wise work of tile kt
        f32x16 s0 = score(0), s1;
        for (int kt = 0; kt < nkt; kt += 2) {
            const bool has1 = kt + 1 < nkt;
            if (has1) s1 = score(kt + 1);
            tile(kt, s0);
            if (has1) {
                if (kt + 2 < nkt) s0 = score(kt + 2);
                tile(kt + 1, s1);
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = dp.scale / l_tot;
        if (kh == 0 && q < Lq) d.lse[(size_t)bh_ * Lq + q] = m_run * LN2 + __logf(l_tot);
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
// G = number of 8-query groups of the tile holding any valid query (registers 4g..4g+3 = queries 8g + 4*kh + 0..3); this half
// tile covers groups 2*s2 and 2*s2 + 1.  The caller skips a half tile with no valid group.
template <bool FULL, bool DROP>
__device__ __forceinline__ void bwdA_half(const f32x16& s, const f32x16& dpv, int s2, float (&pd)[8], float (&ds)[8], float c2, float scale,
                                          const float* lse2, const float* dlt, int qt, int key, int kh, int Lq, int Lk, const MaskCtx& mk,
                                          const Drop16& dp, const uint2* rkey, int G) {
    // One hash decides a PAIR of keys (even key: low 16 bits, odd key: high 16 bits).  Here the lane is the key, so the two
    // keys of a pair sit in neighbouring lanes and would both evaluate the same hash: instead the even lane hashes the
    // half-tile's first four queries, the odd lane its last four, and a quad-permute DPP move swaps them (12 -> 7.5 VALU
    // per decision; this kernel is VALU-issue bound).  With only the first group valid every lane hashes its own four.
    const bool both = G > 2 * s2 + 1;
    uint32_t hq[8];
    if (DROP) {
        const int par = key & 1;
        if (both) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qmine = qt * 32 + e + 8 * (2 * s2 + par) + 4 * kh;          // = mrow(8*s2 + e + 4*par, kh)
                const uint2 kk = rkey[qmine];                                           // row keys of query qmine (LDS table)
                const uint32_t mine = dp.hash((uint32_t)(key >> 1) ^ kk.x, kk.y);
                const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                hq[e] = par ? other : mine;          // query e      (hashed by the even lane)
                hq[e + 4] = par ? mine : other;      // query e + 4  (hashed by the odd lane)
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint2 kk = rkey[qt * 32 + e + 16 * s2 + 4 * kh];
                hq[e] = dp.hash((uint32_t)(key >> 1) ^ kk.x, kk.y);
                hq[e + 4] = 0u;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e >= 4 && !both) {
            pd[e] = 0.f;
            ds[e] = 0.f;
            continue;
        }
        const int r = 8 * s2 + e;
        const int q = qt * 32 + mrow(r, kh);
        float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lse2[q]));
        if (!FULL) p = ((q < Lq) && (key < Lk) && mk.allowed(q, key)) ? p : 0.f;
        float g = dpv[r];
        float pdrop = p;
        if (DROP) {
            const uint32_t hsh = hq[e];
            const bool keep = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= dp.t16;
            pdrop = keep ? p : 0.f;
            g = keep ? g : 0.f;
        }
        pd[e] = pdrop;
        ds[e] = p * (g - dlt[q]);
    }
}

// Phase B, one tile (lane = query, accumulator rows = keys): dS^T.
template <bool FULL, bool DROP>
__device__ __forceinline__ void bwdB_tile(const f32x16& s, const f32x16& dpv, float (&ds)[16], float c2, float scale, float lq, float dq_,
                                          int q, int kt, int kh, int Lq, int Lk, const MaskCtx& mk, const Drop16& dp, uint32_t ka, uint32_t kb) {
    const uint32_t jbase = (uint32_t)(16 * kt + 2 * kh) ^ ka;         // see fwd_tile
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
            const uint32_t hsh = dp.hash(jbase ^ (uint32_t)(mrow(r, 0) >> 1), kb);
            g0 = (hsh & 0xffffu) >= dp.t16 ? g0 : 0.f;
            g1 = (hsh >> 16) >= dp.t16 ? g1 : 0.f;
        }
        ds[r] = p0 * (g0 - dq_);
        ds[r + 1] = p1 * (g1 - dq_);
    }
}

// PHASE 0: dK, dV (waves own key tiles)   PHASE 1: dQ (waves own query tiles).  Two launches, each with only
// the operands it shares between waves in LDS:
//   phase 0: Q and dO of the head (row reads for S/dP, transposed reads for dV^T/dK^T) + lse, delta; the wave's own
//            K/V tile goes from global memory straight into registers;
//   phase 1: K and V (row reads for S^T/dP^T, transposed K for dQ^T); the wave's own Q/dO rows go straight into
//            registers, delta = rowsum(d_o * o) is a per-lane dot product.
// ~48 KB of LDS per 4-wave workgroup -> three workgroups per CU, which is what hides each one's load prologue.
template <int DH, int NW, int PHASE>
__global__ __launch_bounds__(NW * 64) void attn_bwd_bf16_kernel(const mmfm_attn_desc d) {
    constexpr int KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int RS = DH * 2 + 16;
    constexpr int NT = NW * 64;
    constexpr int C8 = DH / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, Lmx = max(Lq, Lk);
    const int LA = PHASE == 0 ? LqP : LkP;            // rows of the two shared images
    char* As = smem;                                  // phase 0: Q        phase 1: K
    char* Bs = As + LA * RS;                          // phase 0: dO       phase 1: V
    float* lse2 = reinterpret_cast<float*>(Bs + LA * RS);     // phase 0 only: lse * log2(e), [LqP]
    float* dlt = lse2 + (PHASE == 0 ? LqP : 0);               // phase 0 only: delta, [LqP]
    uint2* rkey = reinterpret_cast<uint2*>(dlt + (PHASE == 0 ? LqP : 0));      // phase 0 only: dropout row keys, [LqP]
    char* Sc = reinterpret_cast<char*>(rkey + (PHASE == 0 ? LqP : 0));   // [NW][32 rows x RS] bf16 transpose tiles
    int* wflag = reinterpret_cast<int*>(Sc + NW * 32 * RS);
    uint8_t* kpad = reinterpret_cast<uint8_t*>(wflag + NW);
    uint8_t* modl = kpad + LkP;
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * DH;
    const Drop16 dp = drop16_init(d.drop_p);
    // constant factors stay out of the per-element algebra: P~ and dP carry the keep mask only, delta is pre-divided by the
    // dropout scale, and dK / dQ (x softmax scale x dropout scale) and dV (x dropout scale) are scaled once, when stored
    const float inv_ds = 1.f / dp.scale, osc_dk = d.scale * dp.scale;
    const Drop dout = drop_init(d.drop_o);

    if constexpr (PHASE == 0) {
        load_head16<DH>(As, RS, RS / 16, qg, d.ldq, Lq, LqP, t, NT);
        // dO = dropout'(d_o) as bf16;  delta = rowsum(d_o * o)
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
            if (c == 0) dlt[row] = part * inv_ds;
            *reinterpret_cast<uint4*>(Bs + row * RS + c * 16) = __builtin_bit_cast(uint4, pack8(gd));
        }
        constexpr int PADC = RS / 16 - C8;
        for (int idx = t; idx < LqP * PADC; idx += NT)
            *reinterpret_cast<uint4*>(Bs + (idx / PADC) * RS + (C8 + idx % PADC) * 16) = make_uint4(0u, 0u, 0u, 0u);
        for (int i = t; i < LqP; i += NT) {
            lse2[i] = (i < Lq) ? d.lse[(size_t)bh_ * Lq + i] * LOG2E : 0.f;
            uint32_t ka, kb;
            dp.rowkeys((uint32_t)bh_ * (uint32_t)Lq + (uint32_t)i, ka, kb);
            rkey[i] = make_uint2(ka, kb);
        }
    } else {
        load_head16<DH>(As, RS, RS / 16, kg, d.ldk, Lk, LkP, t, NT);
        load_head16<DH>(Bs, RS, RS / 16, vg, d.ldv, Lk, LkP, t, NT);
    }
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
    const int nqt = LqP / 32, nkt = LkP / 32;
    const uint32_t pbase = (uint32_t)bh_ * (uint32_t)Lq;
    const float c2 = d.scale * LOG2E;
    char* sct = Sc + wave * 32 * RS;

    // ---------------- phase 0: wave owns key tile kt -> dK, dV          (As = Q image, Bs = dO image)
    if constexpr (PHASE == 0)
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
            uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
            if (key < Lk) {
                kv = *reinterpret_cast<const uint4*>(kg + (size_t)key * d.ldk + ks * 16 + 8 * kh);
                vv = *reinterpret_cast<const uint4*>(vg + (size_t)key * d.ldv + ks * 16 + 8 * kh);
            }
            kfr[ks] = __builtin_bit_cast(bf16x8v, kv);
            vfr[ks] = __builtin_bit_cast(bf16x8v, vv);
        }
        // S and dP of query tile qt+1 are issued before the element-wise work of tile qt (matrix pipe under VALU)
        f32x16 s_next, dp_next;
        auto scoresA = [&](int qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_next[r] = 0.f; dp_next[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (qt * 32 + l31) * RS + ks * 32 + kh * 16;
                s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), kfr[ks], s_next, 0, 0, 0);        // S[q][key]
                dp_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), vfr[ks], dp_next, 0, 0, 0);     // dP[q][key]
            }
        };
        scoresA(0);
        for (int qt = 0; qt < nqt; ++qt) {
            const f32x16 s = s_next, dpv = dp_next;
            if (qt + 1 < nqt) scoresA(qt + 1);
            const bool full = nomask && (kt * 32 + 32 <= Lk) && (qt * 32 + 32 <= Lq);
            constexpr int GRP = 4;
            // two half-tiles of 8 accumulator rows each: softmax/dropout algebra, pack to bf16, feed the MFMAs
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float pd[8], ds[8];
                if (full) {
                    if (dp.on) bwdA_half<true, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                    else bwdA_half<true, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                } else {
                    if (dp.on) bwdA_half<false, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                    else bwdA_half<false, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                }
                const bf16x8v pf = pack8(pd), sf = pack8(ds);
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Bs, RS, qt * 32 + 16 * s2, i * 32, lane), pf, dVt[i], 0, 0, 0);
                    dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, RS, qt * 32 + 16 * s2, i * 32, lane), sf, dKt[i], 0, 0, 0);
                }
            }
        }
        store_tile_T<DH, DT>(sct, RS, dKt, reinterpret_cast<uint16_t*>(d.dk) + (size_t)b * Lk * d.lddk + h * DH, d.lddk, kt * 32, Lk, lane, osc_dk);
        store_tile_T<DH, DT>(sct, RS, dVt, reinterpret_cast<uint16_t*>(d.dv) + (size_t)b * Lk * d.lddv + h * DH, d.lddv, kt * 32, Lk, lane, dp.scale);
    }

    // ---------------- phase 1: wave owns query tile qt -> dQ             (As = K image, Bs = V image)
    if constexpr (PHASE == 1)
    for (int qt = wave; qt < nqt; qt += NW) {
        f32x16 dQt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        const int q = qt * 32 + l31;
        bf16x8v qfr[KS], dofr[KS];
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 qv = make_uint4(0u, 0u, 0u, 0u), g = qv, o = qv;
            if (q < Lq) {
                qv = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
                g = *reinterpret_cast<const uint4*>(dog + (size_t)q * d.lddo + ks * 16 + 8 * kh);
                o = *reinterpret_cast<const uint4*>(og + (size_t)q * d.ldo + ks * 16 + 8 * kh);
            }
            qfr[ks] = __builtin_bit_cast(bf16x8v, qv);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {o.x, o.y, o.z, o.w};
            const uint64_t base = ((uint64_t)b * Lq + (uint64_t)q) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + ks * 16 + 8 * kh);
            float gd[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g0 = __uint_as_float(gw[j] << 16), g1 = __uint_as_float(gw[j] & 0xffff0000u);
                const float o0 = __uint_as_float(ow[j] << 16), o1 = __uint_as_float(ow[j] & 0xffff0000u);
                dpart += g0 * o0 + g1 * o1;
                gd[2 * j] = dout.apply(g0, base + 2 * j);
                gd[2 * j + 1] = dout.apply(g1, base + 2 * j + 1);
            }
            dofr[ks] = pack8(gd);
        }
        const float dq_ = (dpart + __shfl_xor(dpart, 32)) * inv_ds;      // delta[q] / dscale: the two half-waves hold the two halves of d
        uint32_t ka, kb;
        dp.rowkeys(pbase + (uint32_t)q, ka, kb);
        const float lq = (q < Lq) ? d.lse[(size_t)bh_ * Lq + q] * LOG2E : 0.f;
        f32x16 s_next, dp_next;
        auto scoresB = [&](int kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_next[r] = 0.f; dp_next[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (kt * 32 + l31) * RS + ks * 32 + kh * 16;
                s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), qfr[ks], s_next, 0, 0, 0);        // S^T[key][q]
                dp_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), dofr[ks], dp_next, 0, 0, 0);    // dP^T[key][q]
            }
        };
        scoresB(0);
        for (int kt = 0; kt < nkt; ++kt) {
            const f32x16 s = s_next, dpv = dp_next;
            if (kt + 1 < nkt) scoresB(kt + 1);
            float ds[16];
            const bool full = nomask && (kt * 32 + 32 <= Lk) && (qt * 32 + 32 <= Lq);
            if (full) {
                if (dp.on) bwdB_tile<true, true>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
                else bwdB_tile<true, false>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
            } else {
                if (dp.on) bwdB_tile<false, true>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
                else bwdB_tile<false, false>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
            }
            const bf16x8v sf[2] = {pack8(ds), pack8(ds + 8)};
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
                    dQt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, RS, kt * 32 + 16 * s2, i * 32, lane), sf[s2], dQt[i], 0, 0, 0);
        }
        store_tile_T<DH, DT>(sct, RS, dQt, reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH, d.lddq, qt * 32, Lq, lane, osc_dk);
    }
}

// ============================================================================ single-pass backward (Lk <= 224)
// The two-phase backward above evaluates the softmax / dropout / dS algebra twice (once per phase) and both phases are
// VALU-issue bound (SQ_ACTIVE_INST_ANY 70-85 % of the SIMD cycles, MFMA 16 %).  Here it is evaluated ONCE:
//   waves 0..6 each own one 32-key tile (a 200-token head has exactly 7) and walk the query tiles in lock step, exactly
//     like phase 0 (dK^T, dV^T in accumulators), and additionally drop their packed bf16 dS^T tile [32 keys][32 q] into
//     an LDS staging slot (two 8-B writes per half tile straight from the MFMA operand registers);
//   wave 7 - the wave a 7-tile head leaves idle - turns the staged tiles of the previous query tile into
//     dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] (K^T operands live in its registers for the whole kernel; the staged
//     tiles are read with the hardware-transposed ds_read_b64_tr_b16 in the same k order) and streams dQ out.
// One workgroup barrier per query tile hands a staging buffer over (double buffered).  Fixed summation order, no
// atomics: bitwise reproducible.  dK/dV leave through the workgroup's own, by then dead, Q/dO images.
constexpr int BW1_NW = 8, BW1_CW = 7, BW1_TS = 80, BW1_TILE = 32 * BW1_TS;

// PIPE: issue S/dP of query tile qt+1 before the element-wise work of tile qt (costs 32 VGPRs; at dh <= 32 the kernel runs
// four waves per SIMD under a 128-VGPR cap and lets the other waves cover the MFMA latency instead)
template <int DH, bool PIPE>
__global__ __launch_bounds__(BW1_NW * 64, DH <= 32 ? 4 : 2) void attn_bwd1_bf16_kernel(const mmfm_attn_desc d) {
    constexpr int NW = BW1_NW, CW = BW1_CW, TS = BW1_TS, TILE = BW1_TILE;
    constexpr int KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int RS = DH * 2 + 16;
    constexpr int NT = NW * 64;
    constexpr int C8 = DH / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    // the wave index as a SCALAR: the two roles below then sit behind a uniform branch and share the register file
    // (a branch on a per-lane value keeps the other side's live values allocated: 165 instead of ~110 VGPRs)
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, Lmx = max(Lq, Lk);
    char* As = smem;                                  // Q image
    char* Bs = As + LqP * RS;                         // dO image (output dropout applied)
    float* lse2 = reinterpret_cast<float*>(Bs + LqP * RS);
    float* dlt = lse2 + LqP;
    uint2* rkey = reinterpret_cast<uint2*>(dlt + LqP);        // dropout row keys per query
    char* stg = reinterpret_cast<char*>(rkey + LqP);  // [2][CW][32 keys x TS] dS^T tiles; first the K image (prologue only)
    char* sc7 = stg + 2 * CW * TILE;                  // [32 x RS] dQ transpose tile of wave 7
    int* wflag = reinterpret_cast<int*>(sc7 + 32 * RS);
    uint8_t* kpad = reinterpret_cast<uint8_t*>(wflag + NW);
    uint8_t* modl = kpad + LkP;
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * DH;
    const Drop16 dp = drop16_init(d.drop_p);
    // constant factors stay out of the per-element algebra: P~ and dP carry the keep mask only, delta is pre-divided by the
    // dropout scale, and dK / dQ (x softmax scale x dropout scale) and dV (x dropout scale) are scaled once, when stored
    const float inv_ds = 1.f / dp.scale, osc_dk = d.scale * dp.scale;
    const Drop dout = drop_init(d.drop_o);

    load_head16<DH>(As, RS, RS / 16, qg, d.ldq, Lq, LqP, t, NT);
    load_head16<DH>(stg, RS, RS / 16, kg, d.ldk, Lk, LkP, t, NT);          // K image, for wave 7's K^T operands only
    for (int idx = t; idx < LqP * C8; idx += NT) {                          // dO = dropout'(d_o) as bf16;  delta = rowsum(d_o * o)
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
        if (c == 0) dlt[row] = part * inv_ds;
        *reinterpret_cast<uint4*>(Bs + row * RS + c * 16) = __builtin_bit_cast(uint4, pack8(gd));
    }
    constexpr int PADC = RS / 16 - C8;
    for (int idx = t; idx < LqP * PADC; idx += NT)
        *reinterpret_cast<uint4*>(Bs + (idx / PADC) * RS + (C8 + idx % PADC) * 16) = make_uint4(0u, 0u, 0u, 0u);
    for (int i = t; i < LqP; i += NT) {
        lse2[i] = (i < Lq) ? d.lse[(size_t)bh_ * Lq + i] * LOG2E : 0.f;
        uint32_t ka, kb;
        dp.rowkeys((uint32_t)bh_ * (uint32_t)Lq + (uint32_t)i, ka, kb);
        rkey[i] = make_uint2(ka, kb);
    }
    int allk = 1;
    for (int i = t; i < LkP; i += NT) {
        const uint8_t v = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
        kpad[i] = v;
        if (i < Lk) allk &= (v != 0);
    }
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += NT) modl[i] = d.mod_id[i];
    const int wave_vote = __all(allk) ? 1 : 0;
    if (lane == 0) wflag[wave] = wave_vote;
    __syncthreads();
    int vote = 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) vote &= wflag[w];
    const bool nomask = vote && !(d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP));

    const MaskCtx mk{kpad, modl, d.flags};
    const int nqt = LqP / 32, nkt = LkP / 32;        // nkt <= CW (launcher)
    const float c2 = d.scale * LOG2E;

    if (wave < CW) {
        // ---------------- compute waves: one key tile each, all query tiles
        const int kt = wave;
        const bool active = kt < nkt;
        f32x16 dKt[DT], dVt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dKt[i][r] = 0.f; dVt[i][r] = 0.f; }
        const int key = kt * 32 + l31;
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
        f32x16 s_next, dp_next;
        auto scoresA = [&](int qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s_next[r] = 0.f; dp_next[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = (qt * 32 + l31) * RS + ks * 32 + kh * 16;
                s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), kfr[ks], s_next, 0, 0, 0);        // S[q][key]
                dp_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), vfr[ks], dp_next, 0, 0, 0);     // dP[q][key]
            }
        };
        __syncthreads();                               // wave 7 has its K^T operands: the K image is dead, the staging slots free
        if (PIPE && active) scoresA(0);
        for (int qt = 0; qt < nqt; ++qt) {
            if (active) {
                if (!PIPE) scoresA(qt);
                const f32x16 s = s_next, dpv = dp_next;
                if (PIPE && qt + 1 < nqt) scoresA(qt + 1);
                // valid queries of this tile in groups of 8 (accumulator registers 4g..4g+3): the last query tile of a
                // 200-token head has one group, so 3/4 of its element-wise work and half of its MFMAs are skipped.  With no
                // mask in play and whole groups the mask-free code runs: lanes of keys >= Lk then carry finite garbage that
                // only reaches dK/dV rows that are never stored, and dQ through K^T columns that are zero.
                const int nq = min(32, Lq - qt * 32), GRP = (nq + 7) >> 3;
                const bool full = nomask && (nq & 7) == 0;
                char* slot = stg + ((qt & 1) * CW + wave) * TILE + l31 * TS;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (2 * s2 >= GRP) continue;
                    float pd[8], ds[8];
                    if (full) {
                        if (dp.on) bwdA_half<true, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                        else bwdA_half<true, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                    } else {
                        if (dp.on) bwdA_half<false, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                        else bwdA_half<false, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2, dlt, qt, key, kh, Lq, Lk, mk, dp, rkey, GRP);
                    }
                    const bf16x8v pf = pack8(pd), sf = pack8(ds);
                    // dS^T[key = lane][q]: elements 0..3 are queries 16*s2 + 4*kh + 0..3, elements 4..7 the same + 8
                    const uint4 sw = __builtin_bit_cast(uint4, sf);
                    *reinterpret_cast<uint2*>(slot + (16 * s2 + 4 * kh) * 2) = make_uint2(sw.x, sw.y);
                    *reinterpret_cast<uint2*>(slot + (16 * s2 + 8 + 4 * kh) * 2) = make_uint2(sw.z, sw.w);
#pragma unroll
                    for (int i = 0; i < DT; ++i) {
                        dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Bs, RS, qt * 32 + 16 * s2, i * 32, lane), pf, dVt[i], 0, 0, 0);
                        dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, RS, qt * 32 + 16 * s2, i * 32, lane), sf, dKt[i], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                           // staging buffer (qt & 1) is complete; buffer ((qt+1) & 1) has been consumed
        }
        // Q / dO images are dead (every compute wave passed the last barrier): rows [32w, 32w+32) carry this wave's stores
        if (active) {
            store_tile_T<DH, DT>(As + 32 * wave * RS, RS, dKt, reinterpret_cast<uint16_t*>(d.dk) + (size_t)b * Lk * d.lddk + h * DH, d.lddk,
                                 kt * 32, Lk, lane, osc_dk);
            store_tile_T<DH, DT>(Bs + 32 * wave * RS, RS, dVt, reinterpret_cast<uint16_t*>(d.dv) + (size_t)b * Lk * d.lddv + h * DH, d.lddv,
                                 kt * 32, Lk, lane, dp.scale);
        }
    } else {
        // ---------------- wave 7: K^T operands of every key tile, hardware-transposed out of the K image and kept for the
        // whole kernel; then dQ of query tile qt from the staged dS^T tiles
        bf16x8v kT[CW][2][DT];
#pragma unroll
        for (int kt = 0; kt < CW; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < DT; ++i) kT[kt][s2][i] = trfrag(stg, RS, min(kt, nkt - 1) * 32 + 16 * s2, i * 32, lane);
        __syncthreads();
        uint16_t* dqg = reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH;
        for (int qt = 0; qt < nqt; ++qt) {
            __syncthreads();
            f32x16 dQt[DT];
#pragma unroll
            for (int i = 0; i < DT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
            const char* buf = stg + (qt & 1) * CW * TILE;
#pragma unroll
            for (int kt = 0; kt < CW; ++kt) {
                if (kt < nkt) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8v bfr = trfrag(buf + kt * TILE, TS, 16 * s2, 0, lane);
#pragma unroll
                        for (int i = 0; i < DT; ++i) dQt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[kt][s2][i], bfr, dQt[i], 0, 0, 0);
                    }
                }
            }
            store_tile_T<DH, DT>(sc7, RS, dQt, dqg, d.lddq, qt * 32, Lq, lane, osc_dk);
        }
    }
}

size_t bwd1_lds(int Lq, int Lk, int dh) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, RS = dh * 2 + 16;
    return (size_t)2 * LqP * RS + (size_t)4 * LqP * 4 + (size_t)2 * BW1_CW * BW1_TILE + (size_t)32 * RS + BW1_NW * 4 + LkP + std::max(Lq, Lk) + 64;
}
// shapes the single-pass backward takes: one key tile per compute wave, the K image fits the staging area, the dK/dV
// store tiles fit the Q/dO images
bool bwd1_ok(int Lq, int Lk, int dh) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, RS = dh * 2 + 16;
    static const bool off = [] { const char* e = getenv("MMFM_ATTN_BWD_SINGLE"); return e && atoi(e) == 0; }();
    return !off && LkP <= 32 * BW1_CW && LkP <= LqP && (size_t)LkP * RS <= (size_t)2 * BW1_CW * BW1_TILE && bwd1_lds(Lq, Lk, dh) <= 160 * 1024;
}

// ============================================================================ tiled bf16 kernels (long sequences)
// Heads whose K/V (forward) or Q/dO/K/V (backward) images do not fit the 160 KB of LDS (BASELINE config 5: L = 600,
// dh = 64).  grid = (B*heads, ceil(tiles/4)): a workgroup OWNS four 32-row tiles, one per wave, whose operands live in
// registers (straight from global memory), and STREAMS the other side through LDS in 128-row chunks, keeping the running
// softmax state / the gradient accumulators in registers across chunks.  Same element-wise code, masks, dropout hash and
// mask-free fast path as the untiled kernels; no atomics, fixed summation order.
constexpr int TCH16 = 128;       // streamed rows per chunk

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_tiled_kernel(const mmfm_attn_desc d) {
    constexpr int NW = 4, KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int KRS = DH * 2 + 16, VRS = DT * 64, SLD = DT * 32 + 1, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LkP = (Lk + 31) & ~31, Lmx = max(Lq, Lk);
    char* Ks = smem;                                   // [TCH16][KRS]
    char* Vs = Ks + TCH16 * KRS;                       // [TCH16][VRS]
    float* Sc = reinterpret_cast<float*>(Vs + TCH16 * VRS);
    int* wflag = reinterpret_cast<int*>(Sc + NW * 32 * SLD);
    uint8_t* kpad = reinterpret_cast<uint8_t*>(wflag + NW);
    uint8_t* modl = kpad + LkP;
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    uint16_t* og = reinterpret_cast<uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;

    int allk = 1;
    for (int i = t; i < LkP; i += NT) {
        const uint8_t v = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
        kpad[i] = v;
        if (i < Lk) allk &= (v != 0);
    }
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += NT) modl[i] = d.mod_id[i];
    const int wave_vote = __all(allk) ? 1 : 0;
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
    const int nqt = (Lq + 31) / 32;
    const float c2 = d.scale * LOG2E;
    const int qt = blockIdx.y * NW + wave;
    const bool active = qt < nqt;                      // inactive waves still take part in the chunk barriers
    const int q0 = qt * 32, q = q0 + l31;
    uint32_t ka, kb;                                       // dropout row keys of this lane's query
    dp.rowkeys((uint32_t)bh_ * (uint32_t)Lq + (uint32_t)q, ka, kb);
    bf16x8v qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (active && q < Lq) v = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
        qf[ks] = __builtin_bit_cast(bf16x8v, v);
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int c0 = 0; c0 < LkP; c0 += TCH16) {
        const int rows = min(TCH16, LkP - c0);
        __syncthreads();                               // readers of the previous chunk are done
        load_head16<DH>(Ks, KRS, KRS / 16, kg + (size_t)c0 * d.ldk, d.ldk, max(0, min(rows, Lk - c0)), rows, t, NT);
        load_head16<DH>(Vs, VRS, VRS / 16, vg + (size_t)c0 * d.ldv, d.ldv, max(0, min(rows, Lk - c0)), rows, t, NT);
        __syncthreads();
        if (!active) continue;
        auto score = [&](int kl) {
            f32x16 acc_s;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_s[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc_s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kl * 32 + l31) * KRS + ks * 32 + kh * 16), qf[ks], acc_s, 0, 0, 0);
            return acc_s;
        };
        auto tile = [&](int kl, const f32x16& st) {
            const int kt = c0 / 32 + kl;               // global key tile: key indices and the dropout counters
            float alpha, pd[16];
            bool live;
            const int nk = min(32, Lk - kt * 32), G = (nk + 7) >> 3;
            if (nomask && (nk & 7) == 0)
                live = dp.on ? fwd_tile<true, true>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G)
                             : fwd_tile<true, false>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G);
            else
                live = dp.on ? fwd_tile<false, true>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G)
                             : fwd_tile<false, false>(st, c2, m_run, l_run, alpha, pd, q, Lq, Lk, kt, kh, mk, dp, ka, kb, G);
            if (!live) return;
            const bf16x8v pf0 = pack8(pd), pf1 = pack8(pd + 8);
            const bool rescale = !__all(alpha == 1.f);
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, VRS, kl * 32, i * 32, lane), pf0, acc[i], 0, 0, 0);
                if (G > 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, VRS, kl * 32 + 16, i * 32, lane), pf1, acc[i], 0, 0, 0);
            }
        };
        const int nkl = rows / 32;
        f32x16 s0 = score(0), s1;
        for (int kl = 0; kl < nkl; kl += 2) {
            const bool has1 = kl + 1 < nkl;
            if (has1) s1 = score(kl + 1);
            tile(kl, s0);
            if (has1) {
                if (kl + 2 < nkl) s0 = score(kl + 2);
                tile(kl + 1, s1);
            }
        }
    }
    if (!active) return;                               // no barrier below
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = dp.scale / l_tot;
    if (kh == 0 && q < Lq) d.lse[(size_t)bh_ * Lq + q] = m_run * LN2 + __logf(l_tot);
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
}

// PHASE 0: the workgroup owns 4 key tiles (dK, dV; K/V operands in registers) and streams Q / dO chunks;
// PHASE 1: owns 4 query tiles (dQ; Q/dO operands and delta in registers) and streams K / V chunks.
template <int DH, int PHASE>
__global__ __launch_bounds__(256, 2) void attn_bwd_bf16_tiled_kernel(const mmfm_attn_desc d) {
    constexpr int NW = 4, KS = DH / 16, DT = (DH + 31) / 32;
    constexpr int RS = DH * 2 + 16, NT = NW * 64, C8 = DH / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, Lmx = max(Lq, Lk);
    char* As = smem;                                  // phase 0: Q chunk     phase 1: K chunk     [TCH16][RS]
    char* Bs = As + TCH16 * RS;                       // phase 0: dO chunk    phase 1: V chunk
    float* lse2 = reinterpret_cast<float*>(Bs + TCH16 * RS);   // phase 0: lse * log2(e) of the chunk  [TCH16]
    float* dlt = lse2 + TCH16;                                 // phase 0: delta of the chunk          [TCH16]
    uint2* rkey = reinterpret_cast<uint2*>(dlt + TCH16);       // phase 0: dropout row keys of the chunk     [TCH16]
    char* Sc = reinterpret_cast<char*>(rkey + TCH16);          // [NW][32 rows x RS] output transpose tiles
    int* wflag = reinterpret_cast<int*>(Sc + NW * 32 * RS);
    uint8_t* kpad = reinterpret_cast<uint8_t*>(wflag + NW);
    uint8_t* modl = kpad + LkP;
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * DH;
    const Drop16 dp = drop16_init(d.drop_p);
    // constant factors stay out of the per-element algebra: P~ and dP carry the keep mask only, delta is pre-divided by the
    // dropout scale, and dK / dQ (x softmax scale x dropout scale) and dV (x dropout scale) are scaled once, when stored
    const float inv_ds = 1.f / dp.scale, osc_dk = d.scale * dp.scale;
    const Drop dout = drop_init(d.drop_o);

    int allk = 1;
    for (int i = t; i < LkP; i += NT) {
        const uint8_t v = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
        kpad[i] = v;
        if (i < Lk) allk &= (v != 0);
    }
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += NT) modl[i] = d.mod_id[i];
    const int wave_vote = __all(allk) ? 1 : 0;
    if (lane == 0) wflag[wave] = wave_vote;
    __syncthreads();
    int vote = 1;
#pragma unroll
    for (int w = 0; w < NW; ++w) vote &= wflag[w];
    const bool nomask = vote && !(d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP));

    const MaskCtx mk{kpad, modl, d.flags};
    const uint32_t pbase = (uint32_t)bh_ * (uint32_t)Lq;
    const float c2 = d.scale * LOG2E;
    char* sct = Sc + wave * 32 * RS;

    if constexpr (PHASE == 0) {
        const int kt = blockIdx.y * NW + wave;
        const bool active = kt < LkP / 32;
        f32x16 dKt[DT], dVt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dKt[i][r] = 0.f; dVt[i][r] = 0.f; }
        const int key = kt * 32 + l31;
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
        for (int c0 = 0; c0 < LqP; c0 += TCH16) {
            const int rows = min(TCH16, LqP - c0);
            __syncthreads();
            load_head16<DH>(As, RS, RS / 16, qg + (size_t)c0 * d.ldq, d.ldq, max(0, min(rows, Lq - c0)), rows, t, NT);
            for (int idx = t; idx < rows * C8; idx += NT) {              // dO = dropout'(d_o) as bf16;  delta = rowsum(d_o * o)
                const int row = idx / C8, c = idx % C8, qrow = c0 + row;
                uint4 g = make_uint4(0u, 0u, 0u, 0u), o = g;
                if (qrow < Lq) {
                    g = *reinterpret_cast<const uint4*>(dog + (size_t)qrow * d.lddo + 8 * c);
                    o = *reinterpret_cast<const uint4*>(og + (size_t)qrow * d.ldo + 8 * c);
                }
                const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {o.x, o.y, o.z, o.w};
                const uint64_t base = ((uint64_t)b * Lq + (uint64_t)qrow) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 8 * c);
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
                if (c == 0) dlt[row] = part * inv_ds;
                *reinterpret_cast<uint4*>(Bs + row * RS + c * 16) = __builtin_bit_cast(uint4, pack8(gd));
            }
            constexpr int PADC = RS / 16 - C8;
            for (int idx = t; idx < rows * PADC; idx += NT)
                *reinterpret_cast<uint4*>(Bs + (idx / PADC) * RS + (C8 + idx % PADC) * 16) = make_uint4(0u, 0u, 0u, 0u);
            for (int i = t; i < rows; i += NT) {
                lse2[i] = (c0 + i < Lq) ? d.lse[(size_t)bh_ * Lq + c0 + i] * LOG2E : 0.f;
                uint32_t ka, kb;
                dp.rowkeys((uint32_t)bh_ * (uint32_t)Lq + (uint32_t)(c0 + i), ka, kb);
                rkey[i] = make_uint2(ka, kb);
            }
            __syncthreads();
            if (!active) continue;
            const float* lse2g = lse2 - c0;            // indexed by the GLOBAL query
            const float* dltg = dlt - c0;
            const uint2* rkeyg = rkey - c0;
            for (int ql = 0; ql < rows / 32; ++ql) {
                const int qt = c0 / 32 + ql;
                f32x16 s, dpv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int off = (ql * 32 + l31) * RS + ks * 32 + kh * 16;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), kfr[ks], s, 0, 0, 0);
                    dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), vfr[ks], dpv, 0, 0, 0);
                }
                const int nq = min(32, Lq - qt * 32), GRP = (nq + 7) >> 3;
                const bool full = nomask && (nq & 7) == 0;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    if (2 * s2 >= GRP) continue;
                    float pd[8], ds[8];
                    if (full) {
                        if (dp.on) bwdA_half<true, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2g, dltg, qt, key, kh, Lq, Lk, mk, dp, rkeyg, GRP);
                        else bwdA_half<true, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2g, dltg, qt, key, kh, Lq, Lk, mk, dp, rkeyg, GRP);
                    } else {
                        if (dp.on) bwdA_half<false, true>(s, dpv, s2, pd, ds, c2, d.scale, lse2g, dltg, qt, key, kh, Lq, Lk, mk, dp, rkeyg, GRP);
                        else bwdA_half<false, false>(s, dpv, s2, pd, ds, c2, d.scale, lse2g, dltg, qt, key, kh, Lq, Lk, mk, dp, rkeyg, GRP);
                    }
                    const bf16x8v pf = pack8(pd), sf = pack8(ds);
#pragma unroll
                    for (int i = 0; i < DT; ++i) {
                        dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Bs, RS, ql * 32 + 16 * s2, i * 32, lane), pf, dVt[i], 0, 0, 0);
                        dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, RS, ql * 32 + 16 * s2, i * 32, lane), sf, dKt[i], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) return;
        store_tile_T<DH, DT>(sct, RS, dKt, reinterpret_cast<uint16_t*>(d.dk) + (size_t)b * Lk * d.lddk + h * DH, d.lddk, kt * 32, Lk, lane, osc_dk);
        store_tile_T<DH, DT>(sct, RS, dVt, reinterpret_cast<uint16_t*>(d.dv) + (size_t)b * Lk * d.lddv + h * DH, d.lddv, kt * 32, Lk, lane, dp.scale);
    } else {
        const int qt = blockIdx.y * NW + wave;
        const bool active = qt < LqP / 32;
        f32x16 dQt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        const int q = qt * 32 + l31;
        bf16x8v qfr[KS], dofr[KS];
        float dpart = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint4 qv = make_uint4(0u, 0u, 0u, 0u), g = qv, o = qv;
            if (active && q < Lq) {
                qv = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
                g = *reinterpret_cast<const uint4*>(dog + (size_t)q * d.lddo + ks * 16 + 8 * kh);
                o = *reinterpret_cast<const uint4*>(og + (size_t)q * d.ldo + ks * 16 + 8 * kh);
            }
            qfr[ks] = __builtin_bit_cast(bf16x8v, qv);
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {o.x, o.y, o.z, o.w};
            const uint64_t base = ((uint64_t)b * Lq + (uint64_t)q) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + ks * 16 + 8 * kh);
            float gd[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g0 = __uint_as_float(gw[j] << 16), g1 = __uint_as_float(gw[j] & 0xffff0000u);
                const float o0 = __uint_as_float(ow[j] << 16), o1 = __uint_as_float(ow[j] & 0xffff0000u);
                dpart += g0 * o0 + g1 * o1;
                gd[2 * j] = dout.apply(g0, base + 2 * j);
                gd[2 * j + 1] = dout.apply(g1, base + 2 * j + 1);
            }
            dofr[ks] = pack8(gd);
        }
        const float dq_ = (dpart + __shfl_xor(dpart, 32)) * inv_ds;
        uint32_t ka, kb;
        dp.rowkeys(pbase + (uint32_t)q, ka, kb);
        const float lq = (active && q < Lq) ? d.lse[(size_t)bh_ * Lq + q] * LOG2E : 0.f;
        for (int c0 = 0; c0 < LkP; c0 += TCH16) {
            const int rows = min(TCH16, LkP - c0);
            __syncthreads();
            load_head16<DH>(As, RS, RS / 16, kg + (size_t)c0 * d.ldk, d.ldk, max(0, min(rows, Lk - c0)), rows, t, NT);
            load_head16<DH>(Bs, RS, RS / 16, vg + (size_t)c0 * d.ldv, d.ldv, max(0, min(rows, Lk - c0)), rows, t, NT);
            __syncthreads();
            if (!active) continue;
            for (int kl = 0; kl < rows / 32; ++kl) {
                const int kt = c0 / 32 + kl;
                f32x16 s, dpv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int off = (kl * 32 + l31) * RS + ks * 32 + kh * 16;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), qfr[ks], s, 0, 0, 0);          // S^T[key][q]
                    dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), dofr[ks], dpv, 0, 0, 0);     // dP^T[key][q]
                }
                float ds[16];
                const bool full = nomask && (kt * 32 + 32 <= Lk) && (qt * 32 + 32 <= Lq);
                if (full) {
                    if (dp.on) bwdB_tile<true, true>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
                    else bwdB_tile<true, false>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
                } else {
                    if (dp.on) bwdB_tile<false, true>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
                    else bwdB_tile<false, false>(s, dpv, ds, c2, d.scale, lq, dq_, q, kt, kh, Lq, Lk, mk, dp, ka, kb);
                }
                const bf16x8v sf[2] = {pack8(ds), pack8(ds + 8)};
#pragma unroll
                for (int i = 0; i < DT; ++i)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        dQt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, RS, kl * 32 + 16 * s2, i * 32, lane), sf[s2], dQt[i], 0, 0, 0);
            }
        }
        if (!active) return;
        store_tile_T<DH, DT>(sct, RS, dQt, reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH, d.lddq, qt * 32, Lq, lane, osc_dk);
    }
}

size_t fwd_tiled16_lds(int Lq, int Lk, int dh) {
    const int DT = (dh + 31) / 32, LkP = (Lk + 31) & ~31;
    return (size_t)TCH16 * (dh * 2 + 16) + (size_t)TCH16 * DT * 64 + (size_t)4 * 32 * (DT * 32 + 1) * 4 + 16 + LkP + std::max(Lq, Lk) + 64;
}
size_t bwd_tiled16_lds(int Lq, int Lk, int dh) {
    const int LkP = (Lk + 31) & ~31, RS = dh * 2 + 16;
    return (size_t)2 * TCH16 * RS + (size_t)4 * TCH16 * 4 + (size_t)4 * 32 * RS + 16 + LkP + std::max(Lq, Lk) + 64;
}

// waves per workgroup (MMFM_ATTN_FWD_WAVES / MMFM_ATTN_BWD_WAVES = 4 or 8).  Smaller workgroups let more of them
// co-reside on a CU, which is what hides each workgroup's dispatch + load prologue.
int env_waves(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : dflt;
    return (v == 4 || v == 8) ? v : dflt;
}
int fwd_waves() { static int w = env_waves("MMFM_ATTN_FWD_WAVES", 4); return w; }
// forward: 8 waves when the head has at least five query tiles and MMFM_ATTN_FWD_WAVES does not say otherwise (the forward needs
// 120 VGPRs since the dropout / scale algebra shrank: two 8-wave workgroups = four waves per SIMD against three 4-wave ones;
// measured in the B = 1024 step: 4.01 -> 3.77 ms over the 15 forward launches)
int fwd_waves_for(int Lq) {
    static const bool forced = getenv("MMFM_ATTN_FWD_WAVES") != nullptr;
    return forced ? fwd_waves() : (Lq >= 160 ? 8 : 4);
}
int bwd_waves() { static int w = env_waves("MMFM_ATTN_BWD_WAVES", 4); return w; }

size_t fwd_lds(int Lq, int Lk, int dh, int nw) {
    const int DT = (dh + 31) / 32, LkP = (Lk + 31) & ~31;
    return (size_t)LkP * (dh * 2 + 16) + (size_t)LkP * DT * 64 + (size_t)nw * 32 * (DT * 32 + 1) * 4 + LkP + std::max(Lq, Lk) + 64;
}
size_t bwd_lds(int Lq, int Lk, int dh, int nw, int phase) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31, RS = dh * 2 + 16;
    const size_t images = (size_t)2 * (phase == 0 ? LqP : LkP) * RS;
    return images + (phase == 0 ? (size_t)4 * LqP * 4 : 0) + (size_t)nw * 32 * RS + LkP + std::max(Lq, Lk) + 64;
}

// the attribute belongs to the (device, kernel) pair: a process that drives several GPUs must opt in on each
int opt_in_lds(const void* kern, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, bool> done;
    if (bytes <= 65536) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)kern ^ ((uint64_t)(dev + 1) << 56);
    std::lock_guard<std::mutex> g(mu);
    if (done.count(key)) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mmfm_set_error((int)e, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    done[key] = true;
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
    {   // shape decision shared by forward and backward (see use_tiled in attention.hip): the untiled kernels are used only
        // if all three fit; otherwise the tiled bf16 kernels take forward AND backward (one dropout hash layout per pair).
        // MMFM_ATTN_FORCE_TILED: 1 = tiled bf16 kernels for every shape, 2 = hand the call to the fp32-compute tiled kernels.
        static const int force_env = [] { const char* e = getenv("MMFM_ATTN_FORCE_TILED"); return e ? atoi(e) : 0; }();
        if (force_env == 2) return -1000;
        const int nwf = fwd_waves(), nwb = bwd_waves();
        const bool fits = fwd_lds(d.Lq, d.Lk, d.dh, nwf) <= 160 * 1024 && bwd_lds(d.Lq, d.Lk, d.dh, nwb, 0) <= 160 * 1024 &&
                          bwd_lds(d.Lq, d.Lk, d.dh, nwb, 1) <= 160 * 1024;
        if (!fits || force_env == 1) {
            const int gq = ((d.Lq + 31) / 32 + 3) / 4, gk = ((d.Lk + 31) / 32 + 3) / 4;
#define TILED16(KERN, GY, LDSB)                                                                                   \
            {                                                                                                     \
                auto kern = KERN;                                                                                 \
                if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), LDSB)) return rc;                    \
                hipLaunchKernelGGL(kern, dim3(d.B * d.heads, GY), dim3(256), LDSB, st, d);                        \
            }
            if (backward) {
                const bool alb = d.lddo % 8 == 0 && d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0 && (uintptr_t)d.d_o % 16 == 0 &&
                                 (uintptr_t)d.dq % 16 == 0 && (uintptr_t)d.dk % 16 == 0 && (uintptr_t)d.dv % 16 == 0;
                if (!alb) return mmfm_set_error(-1, "mmfm_attn_bwd(bf16, tiled): gradient tensors must be 16-byte aligned with leading dims % 8 == 0");
                const size_t lds = bwd_tiled16_lds(d.Lq, d.Lk, d.dh);
                if (lds > 160 * 1024) return -1000;
                if (d.dh == 16) { TILED16((attn_bwd_bf16_tiled_kernel<16, 0>), gk, lds) TILED16((attn_bwd_bf16_tiled_kernel<16, 1>), gq, lds) }
                else if (d.dh == 32) { TILED16((attn_bwd_bf16_tiled_kernel<32, 0>), gk, lds) TILED16((attn_bwd_bf16_tiled_kernel<32, 1>), gq, lds) }
                else { TILED16((attn_bwd_bf16_tiled_kernel<64, 0>), gk, lds) TILED16((attn_bwd_bf16_tiled_kernel<64, 1>), gq, lds) }
                MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16, tiled)");
            } else {
                const size_t lds = fwd_tiled16_lds(d.Lq, d.Lk, d.dh);
                if (lds > 160 * 1024 || bwd_tiled16_lds(d.Lq, d.Lk, d.dh) > 160 * 1024) return -1000;
                if (d.dh == 16) TILED16((attn_fwd_bf16_tiled_kernel<16>), gq, lds)
                else if (d.dh == 32) TILED16((attn_fwd_bf16_tiled_kernel<32>), gq, lds)
                else TILED16((attn_fwd_bf16_tiled_kernel<64>), gq, lds)
                MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16, tiled)");
            }
#undef TILED16
            return 0;
        }
    }
    if (backward) {
        const bool alb = d.lddo % 8 == 0 && d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0 && (uintptr_t)d.d_o % 16 == 0 &&
                         (uintptr_t)d.dq % 16 == 0 && (uintptr_t)d.dk % 16 == 0 && (uintptr_t)d.dv % 16 == 0;
        if (!alb) {
            // the forward of this shape ran the bf16-MFMA kernel; the fp32-compute fallback lays its dropout hash out differently,
            // so falling through would regenerate a DIFFERENT attention-dropout mask in the backward
            if (d.drop_p.p > 0.f && d.drop_p.state != nullptr)
                return mmfm_set_error(-1, "mmfm_attn_bwd(bf16): gradient tensors must be 16-byte aligned with leading dims %% 8 == 0 when "
                                          "attention dropout is on (the fallback kernel family draws a different mask than the forward did)");
            return -1000;
        }
        if (bwd1_ok(d.Lq, d.Lk, d.dh)) {
            const size_t lds = bwd1_lds(d.Lq, d.Lk, d.dh);
#define BWD1S(DHV)                                                                                                \
            {                                                                                                     \
                auto kern = attn_bwd1_bf16_kernel<DHV, (DHV > 32)>;                                                         \
                if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), lds)) return rc;                     \
                hipLaunchKernelGGL(kern, dim3(d.B * d.heads), dim3(BW1_NW * 64), lds, st, d);                     \
            }
            if (d.dh == 16) BWD1S(16) else if (d.dh == 32) BWD1S(32) else BWD1S(64)
#undef BWD1S
            MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16, single pass)");
            return 0;
        }
        const int nw = bwd_waves();
        const size_t lds0 = bwd_lds(d.Lq, d.Lk, d.dh, nw, 0), lds1 = bwd_lds(d.Lq, d.Lk, d.dh, nw, 1);
        if (lds0 > 160 * 1024 || lds1 > 160 * 1024) return -1000;
#define BWD1(DHV, NWV, PH)                                                                                        \
        {                                                                                                         \
            auto kern = attn_bwd_bf16_kernel<DHV, NWV, PH>;                                                       \
            const size_t lds = PH == 0 ? lds0 : lds1;                                                             \
            if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), lds)) return rc;                         \
            hipLaunchKernelGGL(kern, dim3(d.B * d.heads), dim3(NWV * 64), lds, st, d);                            \
        }
#define BWD(DHV)                                                                                                  \
        if (nw == 8) { BWD1(DHV, 8, 0) BWD1(DHV, 8, 1) } else { BWD1(DHV, 4, 0) BWD1(DHV, 4, 1) }
        if (d.dh == 16) { BWD(16) } else if (d.dh == 32) { BWD(32) } else { BWD(64) }
#undef BWD
#undef BWD1
        MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16)");
        return 0;
    }
    int nw = fwd_waves_for(d.Lq);
    if (nw == 8 && fwd_lds(d.Lq, d.Lk, d.dh, 8) > 160 * 1024) nw = 4;
    const size_t lds = fwd_lds(d.Lq, d.Lk, d.dh, nw);
    if (lds > 160 * 1024) return -1000;
#define FWD1(DHV, NWV)                                                                                            \
    {                                                                                                             \
        auto kern = attn_fwd_bf16_kernel<DHV, NWV>;                                                               \
        if (int rc = opt_in_lds(reinterpret_cast<const void*>(kern), lds)) return rc;                             \
        hipLaunchKernelGGL(kern, dim3(d.B * d.heads), dim3(NWV * 64), lds, st, d);                                \
    }
#define FWD(DHV) if (nw == 8) FWD1(DHV, 8) else FWD1(DHV, 4)
    if (d.dh == 16) { FWD(16) } else if (d.dh == 32) { FWD(32) } else { FWD(64) }
#undef FWD
#undef FWD1
    MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16)");
    return 0;
}
