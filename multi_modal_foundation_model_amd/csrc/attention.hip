// Masked scaled-dot-product attention, forward and backward (mmfm_attn_fwd / mmfm_attn_bwd).
//
// fp32 parity path on v_mfma_f32_32x32x2_f32.  One workgroup (4 wavefronts) per (batch, head);
// the head's K and V (L <= a few hundred tokens, dh <= 64) stay in LDS for the whole workgroup.
// The [B,h,L,L] mask of the reference is never materialised: keypad bytes + flags.
//
// Forward (flash style, online softmax): each wave owns 32-query tiles and walks 32-key tiles.
//   S^T = K Q^T is computed with the KEY on the MFMA row and the QUERY on the lane, so the
//   softmax statistics of a query are lane-local (one __shfl_xor(32) joins the two half-waves)
//   and the probabilities, still in accumulator registers, are directly the B operand of
//   O^T += V^T P^T (k-pairing (s&3)+8(s>>2) [+4 for the upper half]: no LDS round trip).
// Backward: recompute P from Q, K and the saved LSE.  Phase A: a wave owns a key tile and keeps
//   dK^T, dV^T in accumulators over all query tiles.  Phase B: a wave owns a query tile and keeps
//   dQ^T.  No atomics: results are bitwise reproducible.
#include "common.h"
#include <algorithm>
#include <mutex>
#include <unordered_map>

namespace {

__device__ __forceinline__ int mrow(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

struct MaskCtx {
    const uint8_t* kpad;   // LDS [LkP]
    const uint8_t* modq;   // LDS or null
    const uint8_t* modk;
    int flags;
    __device__ __forceinline__ bool allowed(int q, int k) const {
        bool a = (flags & MMFM_ATTN_CAUSAL) ? (k <= q) : (kpad[k] != 0);
        if ((flags & MMFM_ATTN_DIAG) && q == k) a = true;
        if ((flags & MMFM_ATTN_SEP) && modq[q] != modk[k]) a = true;
        return a;
    }
};

// cooperative load of rows [0,L) x DH floats of one head into LDS with row stride LD; rows [L,LP) zeroed
template <typename T, int DH>
__device__ __forceinline__ void load_head(float* __restrict__ dst, int LD, const T* __restrict__ src, int ld, int L, int LP,
                                          int t, int nthreads) {
    constexpr int C4 = DH / 4;
    for (int idx = t; idx < LP * C4; idx += nthreads) {
        const int row = idx / C4, c = idx % C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < L) v = io<T>::ld4(src + (size_t)row * ld + 4 * c);
        float* d = dst + row * LD + 4 * c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ============================================================================ forward
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const mmfm_attn_desc d) {
    constexpr int DT = (DH + 31) / 32;      // 32-row tiles of the head dim
    constexpr int LD = DH + 1;              // K / Q LDS row stride (odd: conflict-free column reads)
    constexpr int DVL = DT * 32;            // V LDS row stride (zero padded to the MFMA tile)
    constexpr int SLD = DVL + 1;            // per-wave scratch stride
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk;
    const int LkP = (Lk + 31) & ~31;
    const int Lmx = max(Lq, Lk);
    float* Ks = smem;                                  // [LkP][LD]
    float* Vs = Ks + LkP * LD;                         // [LkP][DVL]
    float* Sc = Vs + LkP * DVL;                        // [4][32][SLD]
    uint8_t* kpad = reinterpret_cast<uint8_t*>(Sc + 4 * 32 * SLD);   // [LkP]
    uint8_t* modl = kpad + LkP;                                      // [Lmx] (SEP only)

    const T* qg = reinterpret_cast<const T*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const T* kg = reinterpret_cast<const T*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const T* vg = reinterpret_cast<const T*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    T* og = reinterpret_cast<T*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;

    load_head<T, DH>(Ks, LD, kg, d.ldk, Lk, LkP, t, 256);
    {   // V: float4 rows, zero pad columns [DH, DVL) and rows [Lk, LkP)
        constexpr int C4 = DVL / 4;
        for (int idx = t; idx < LkP * C4; idx += 256) {
            const int row = idx / C4, c = idx % C4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < Lk && 4 * c < DH) v = io<T>::ld4(vg + (size_t)row * d.ldv + 4 * c);
            *reinterpret_cast<float4*>(Vs + row * DVL + 4 * c) = v;
        }
    }
    for (int i = t; i < LkP; i += 256) kpad[i] = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += 256) modl[i] = d.mod_id[i];
    __syncthreads();

    MaskCtx mk{kpad, modl, modl, d.flags};
    const Drop dp = drop_init(d.drop_p), dout = drop_init(d.drop_o);
    float* sc = Sc + wave * 32 * SLD;
    const int nqt = (Lq + 31) / 32, nkt = LkP / 32;

    for (int qt = wave; qt < nqt; qt += 4) {
        const int q0 = qt * 32;
        // ---- Q tile -> scratch (coalesced) -> operand registers, pre-scaled
        wave_lds_fence();
        {
            constexpr int C4 = DH / 4;
            for (int idx = lane; idx < 32 * C4; idx += 64) {
                const int row = idx / C4, c = idx % C4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q0 + row < Lq) v = io<T>::ld4(qg + (size_t)(q0 + row) * d.ldq + 4 * c);
                float* p = sc + row * SLD + 4 * c;
                p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
            }
        }
        wave_lds_fence();
        float qreg[DH / 2];
#pragma unroll
        for (int s = 0; s < DH / 2; ++s) qreg[s] = sc[l31 * SLD + 2 * s + kh] * d.scale;

        const int q = q0 + l31;
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
            const float* ka = Ks + (kt * 32 + l31) * LD + kh;
#pragma unroll
            for (int s = 0; s < DH / 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qreg[s], st, 0, 0, 0);
            float mx = -INFINITY;
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + mrow(r, kh);
                ok[r] = (key < Lk) && (q < Lq) && mk.allowed(q, key);
                if (ok[r]) mx = fmaxf(mx, st[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            if (__all(m_new == -INFINITY)) continue;       // nothing allowed yet for any query of this tile
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_use);
            float ps = 0.f, pd[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = ok[r] ? __expf(st[r] - m_use) : 0.f;
                ps += p;
                const int key = kt * 32 + mrow(r, kh);
                pd[r] = dp.apply(p, ((uint64_t)bh_ * Lq + (uint64_t)q) * Lk + (uint64_t)key);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float va = Vs[(kt * 32 + mrow(s, kh)) * DVL + i * 32 + l31];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, pd[s], acc[i], 0, 0, 0);
                }
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv = 1.f / l_tot;                    // 0 allowed keys -> 0 * inf = NaN, like SDPA
        if (kh == 0 && q < Lq) d.lse[((size_t)bh_) * Lq + q] = m_run + __logf(l_tot);
        // ---- O^T (rows d, lane = query) -> scratch [q][d] -> coalesced rows, output dropout fused
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[l31 * SLD + i * 32 + mrow(r, kh)] = acc[i][r] * inv;
        wave_lds_fence();
        {
            constexpr int C4 = DH / 4;
            for (int idx = lane; idx < 32 * C4; idx += 64) {
                const int row = idx / C4, c = idx % C4;
                if (q0 + row < Lq) {
                    const float* p = sc + row * SLD + 4 * c;
                    const uint64_t base = ((uint64_t)b * Lq + (uint64_t)(q0 + row)) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 4 * c);
                    float4 v;
                    v.x = dout.apply(p[0], base + 0); v.y = dout.apply(p[1], base + 1);
                    v.z = dout.apply(p[2], base + 2); v.w = dout.apply(p[3], base + 3);
                    io<T>::st4(og + (size_t)(q0 + row) * d.ldo + 4 * c, v);
                }
            }
        }
    }
}

// ============================================================================ backward
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const mmfm_attn_desc d) {
    constexpr int DT = (DH + 31) / 32;
    constexpr int LD = DH + 1;
    constexpr int SLD = 33;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk;
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    const int Lmx = max(Lq, Lk);
    float* Qs = smem;                       // [LqP][LD]
    float* dOs = Qs + LqP * LD;             // [LqP][LD]   dO = dropout'(d_o)
    float* Ks = dOs + LqP * LD;             // [LkP][LD]
    float* Vs = Ks + LkP * LD;              // [LkP][LD]
    float* lse = Vs + LkP * LD;             // [LqP]
    float* dlt = lse + LqP;                 // [LqP]  delta = rowsum(d_o * o)
    float* Sc = dlt + LqP;                  // [4][32][SLD]
    uint8_t* kpad = reinterpret_cast<uint8_t*>(Sc + 4 * 32 * SLD);
    uint8_t* modl = kpad + LkP;

    const T* qg = reinterpret_cast<const T*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const T* kg = reinterpret_cast<const T*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const T* vg = reinterpret_cast<const T*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const T* og = reinterpret_cast<const T*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const T* dog = reinterpret_cast<const T*>(d.d_o) + (size_t)b * Lq * d.lddo + h * DH;
    const Drop dp = drop_init(d.drop_p), dout = drop_init(d.drop_o);

    load_head<T, DH>(Qs, LD, qg, d.ldq, Lq, LqP, t, 256);
    load_head<T, DH>(Ks, LD, kg, d.ldk, Lk, LkP, t, 256);
    load_head<T, DH>(Vs, LD, vg, d.ldv, Lk, LkP, t, 256);
    {   // dO (output-dropout backward applied) and delta = sum_d d_o * o  (o = forward output AFTER drop_o)
        constexpr int C4 = DH / 4;
        for (int idx = t; idx < LqP * C4; idx += 256) {
            const int row = idx / C4, c = idx % C4;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f), o = g;
            if (row < Lq) {
                g = io<T>::ld4(dog + (size_t)row * d.lddo + 4 * c);
                o = io<T>::ld4(og + (size_t)row * d.ldo + 4 * c);
            }
            float part = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
#pragma unroll
            for (int off = 1; off < C4; off <<= 1) part += __shfl_xor(part, off);
            if (c == 0) dlt[row] = part;
            const uint64_t base = ((uint64_t)b * Lq + (uint64_t)row) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 4 * c);
            float* p = dOs + row * LD + 4 * c;
            p[0] = dout.apply(g.x, base + 0); p[1] = dout.apply(g.y, base + 1);
            p[2] = dout.apply(g.z, base + 2); p[3] = dout.apply(g.w, base + 3);
        }
    }
    for (int i = t; i < LqP; i += 256) lse[i] = (i < Lq) ? d.lse[(size_t)bh_ * Lq + i] : 0.f;
    for (int i = t; i < LkP; i += 256) kpad[i] = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += 256) modl[i] = d.mod_id[i];
    __syncthreads();

    MaskCtx mk{kpad, modl, modl, d.flags};
    float* sc = Sc + wave * 32 * SLD;
    const int nqt = LqP / 32, nkt = LkP / 32;
    const uint64_t pbase = (uint64_t)bh_ * Lq;

    // ---------------- phase A: wave owns key tile kt -> dK, dV
    for (int kt = wave; kt < nkt; kt += 4) {
        f32x16 dKt[DT], dVt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dKt[i][r] = 0.f; dVt[i][r] = 0.f; }
        const int key = kt * 32 + l31;
        const float* kb = Ks + (kt * 32 + l31) * LD + kh;
        const float* vb = Vs + (kt * 32 + l31) * LD + kh;
        for (int qt = 0; qt < nqt; ++qt) {
            f32x16 s, dpv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
            const float* qa = Qs + (qt * 32 + l31) * LD + kh;
            const float* da = dOs + (qt * 32 + l31) * LD + kh;
#pragma unroll
            for (int k2 = 0; k2 < DH / 2; ++k2) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * k2], kb[2 * k2], s, 0, 0, 0);       // S[q][key]
                dpv = __builtin_amdgcn_mfma_f32_32x32x2f32(da[2 * k2], vb[2 * k2], dpv, 0, 0, 0);   // dP[q][key]
            }
            float pd[16], ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + mrow(r, kh);
                const bool ok = (q < Lq) && (key < Lk) && mk.allowed(q, key);
                const float p = ok ? __expf(s[r] * d.scale - lse[q]) : 0.f;
                const uint64_t idx = (pbase + (uint64_t)q) * Lk + (uint64_t)key;
                const bool keep = !dp.on() || dp.keep(idx);
                pd[r] = keep ? p * dp.scale : 0.f;
                const float dpd = keep ? dpv[r] * dp.scale : 0.f;
                ds[r] = p * (dpd - dlt[q]) * d.scale;
            }
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const int dcol = i * 32 + l31;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    const int qrow = qt * 32 + mrow(k2, kh);
                    const float doT = (dcol < DH) ? dOs[qrow * LD + dcol] : 0.f;
                    const float qT = (dcol < DH) ? Qs[qrow * LD + dcol] : 0.f;
                    dVt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(doT, pd[k2], dVt[i], 0, 0, 0);   // dV^T[d][key]
                    dKt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qT, ds[k2], dKt[i], 0, 0, 0);    // dK^T[d][key]
                }
            }
        }
        // transpose through the wave's scratch, store coalesced rows
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            T* outg = reinterpret_cast<T*>(which ? d.dv : d.dk) + (size_t)b * Lk * (which ? d.lddv : d.lddk) + h * DH;
            const int ldo_ = which ? d.lddv : d.lddk;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[l31 * SLD + mrow(r, kh)] = which ? dVt[i][r] : dKt[i][r];
                wave_lds_fence();
                constexpr int CW = (DH < 32 ? DH : 32) / 4;
                for (int idx = lane; idx < 32 * CW; idx += 64) {
                    const int row = idx / CW, c = idx % CW;
                    if (kt * 32 + row < Lk) {
                        const float* p = sc + row * SLD + 4 * c;
                        io<T>::st4(outg + (size_t)(kt * 32 + row) * ldo_ + i * 32 + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
                    }
                }
            }
        }
    }

    // ---------------- phase B: wave owns query tile qt -> dQ
    for (int qt = wave; qt < nqt; qt += 4) {
        f32x16 dQt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        const int q = qt * 32 + l31;
        const float lq = lse[q], dq_ = dlt[q];
        const float* qb = Qs + (qt * 32 + l31) * LD + kh;
        const float* db = dOs + (qt * 32 + l31) * LD + kh;
        for (int kt = 0; kt < nkt; ++kt) {
            f32x16 s, dpv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
            const float* ka = Ks + (kt * 32 + l31) * LD + kh;
            const float* va = Vs + (kt * 32 + l31) * LD + kh;
#pragma unroll
            for (int k2 = 0; k2 < DH / 2; ++k2) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * k2], qb[2 * k2], s, 0, 0, 0);       // S^T[key][q]
                dpv = __builtin_amdgcn_mfma_f32_32x32x2f32(va[2 * k2], db[2 * k2], dpv, 0, 0, 0);   // dP^T[key][q]
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + mrow(r, kh);
                const bool ok = (q < Lq) && (key < Lk) && mk.allowed(q, key);
                const float p = ok ? __expf(s[r] * d.scale - lq) : 0.f;
                const uint64_t idx = (pbase + (uint64_t)q) * Lk + (uint64_t)key;
                const bool keep = !dp.on() || dp.keep(idx);
                const float dpd = keep ? dpv[r] * dp.scale : 0.f;
                ds[r] = p * (dpd - dq_) * d.scale;
            }
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                const int dcol = i * 32 + l31;
#pragma unroll
                for (int k2 = 0; k2 < 16; ++k2) {
                    const float kT = (dcol < DH) ? Ks[(kt * 32 + mrow(k2, kh)) * LD + dcol] : 0.f;
                    dQt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(kT, ds[k2], dQt[i], 0, 0, 0);    // dQ^T[d][q]
                }
            }
        }
        T* outg = reinterpret_cast<T*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[l31 * SLD + mrow(r, kh)] = dQt[i][r];
            wave_lds_fence();
            constexpr int CW = (DH < 32 ? DH : 32) / 4;
            for (int idx = lane; idx < 32 * CW; idx += 64) {
                const int row = idx / CW, c = idx % CW;
                if (qt * 32 + row < Lq) {
                    const float* p = sc + row * SLD + 4 * c;
                    io<T>::st4(outg + (size_t)(qt * 32 + row) * d.lddq + i * 32 + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
                }
            }
        }
    }
}

// ============================================================================ tiled variants (long sequences)
// Same arithmetic as the kernels above, for heads whose K/V (forward) or Q/dO/K/V (backward) do not fit the 160 KB of
// LDS (BASELINE config 5: L = 600, dh = 64).  grid = (B*heads, ceil(tiles/4)): a workgroup OWNS four 32-row tiles (one
// per wave) and STREAMS the other operand through LDS in 128-row chunks, keeping the flash-style running state in
// registers across chunks.  No atomics, no cross-workgroup reduction: results are bitwise reproducible, and the dropout
// decisions are the same function of (b, head, q, key) as in the untiled kernels.
constexpr int TCH = 128;       // streamed rows per chunk

template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_fwd_tiled_kernel(const mmfm_attn_desc d) {
    constexpr int DT = (DH + 31) / 32;
    constexpr int LD = DH + 1;
    constexpr int DVL = DT * 32;
    constexpr int SLD = DVL + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk;
    const int LkP = (Lk + 31) & ~31;
    const int Lmx = max(Lq, Lk);
    float* Ks = smem;                                  // [TCH][LD]
    float* Vs = Ks + TCH * LD;                         // [TCH][DVL]
    float* Sc = Vs + TCH * DVL;                        // [4][32][SLD]
    uint8_t* kpad = reinterpret_cast<uint8_t*>(Sc + 4 * 32 * SLD);   // [LkP]
    uint8_t* modl = kpad + LkP;                                      // [Lmx]

    const T* qg = reinterpret_cast<const T*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const T* kg = reinterpret_cast<const T*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const T* vg = reinterpret_cast<const T*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    T* og = reinterpret_cast<T*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;

    for (int i = t; i < LkP; i += 256) kpad[i] = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += 256) modl[i] = d.mod_id[i];

    MaskCtx mk{kpad, modl, modl, d.flags};
    const Drop dp = drop_init(d.drop_p), dout = drop_init(d.drop_o);
    float* sc = Sc + wave * 32 * SLD;
    const int nqt = (Lq + 31) / 32;
    const int qt = blockIdx.y * 4 + wave;
    const bool active = qt < nqt;                      // inactive waves still take part in the chunk barriers
    const int q0 = qt * 32;
    {
        constexpr int C4 = DH / 4;
        for (int idx = lane; idx < 32 * C4; idx += 64) {
            const int row = idx / C4, c = idx % C4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (active && q0 + row < Lq) v = io<T>::ld4(qg + (size_t)(q0 + row) * d.ldq + 4 * c);
            float* p = sc + row * SLD + 4 * c;
            p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        }
    }
    wave_lds_fence();
    float qreg[DH / 2];
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) qreg[s] = sc[l31 * SLD + 2 * s + kh] * d.scale;

    const int q = q0 + l31;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int c0 = 0; c0 < LkP; c0 += TCH) {
        const int rows = min(TCH, LkP - c0);
        __syncthreads();                               // readers of the previous chunk are done (and kpad/modl are visible)
        load_head<T, DH>(Ks, LD, kg + (size_t)c0 * d.ldk, d.ldk, max(0, min(rows, Lk - c0)), rows, t, 256);
        {
            constexpr int C4 = DVL / 4;
            for (int idx = t; idx < rows * C4; idx += 256) {
                const int row = idx / C4, c = idx % C4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + row < Lk && 4 * c < DH) v = io<T>::ld4(vg + (size_t)(c0 + row) * d.ldv + 4 * c);
                *reinterpret_cast<float4*>(Vs + row * DVL + 4 * c) = v;
            }
        }
        __syncthreads();
        if (!active) continue;
        for (int kt = 0; kt < rows / 32; ++kt) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const float* ka = Ks + (kt * 32 + l31) * LD + kh;
#pragma unroll
            for (int s = 0; s < DH / 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * s], qreg[s], st, 0, 0, 0);
            float mx = -INFINITY;
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = c0 + kt * 32 + mrow(r, kh);
                ok[r] = (key < Lk) && (q < Lq) && mk.allowed(q, key);
                if (ok[r]) mx = fmaxf(mx, st[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            if (__all(m_new == -INFINITY)) continue;
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_use);
            float ps = 0.f, pd[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = ok[r] ? __expf(st[r] - m_use) : 0.f;
                ps += p;
                const int key = c0 + kt * 32 + mrow(r, kh);
                pd[r] = dp.apply(p, ((uint64_t)bh_ * Lq + (uint64_t)q) * Lk + (uint64_t)key);
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float va = Vs[(kt * 32 + mrow(s, kh)) * DVL + i * 32 + l31];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, pd[s], acc[i], 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;                               // no barrier below this point
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (kh == 0 && q < Lq) d.lse[((size_t)bh_) * Lq + q] = m_run + __logf(l_tot);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[l31 * SLD + i * 32 + mrow(r, kh)] = acc[i][r] * inv;
    wave_lds_fence();
    {
        constexpr int C4 = DH / 4;
        for (int idx = lane; idx < 32 * C4; idx += 64) {
            const int row = idx / C4, c = idx % C4;
            if (q0 + row < Lq) {
                const float* p = sc + row * SLD + 4 * c;
                const uint64_t base = ((uint64_t)b * Lq + (uint64_t)(q0 + row)) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 4 * c);
                float4 v;
                v.x = dout.apply(p[0], base + 0); v.y = dout.apply(p[1], base + 1);
                v.z = dout.apply(p[2], base + 2); v.w = dout.apply(p[3], base + 3);
                io<T>::st4(og + (size_t)(q0 + row) * d.ldo + 4 * c, v);
            }
        }
    }
}

// PHASE 0: the workgroup owns 4 key tiles (dK, dV) and streams query chunks; PHASE 1: owns 4 query tiles (dQ), streams keys.
template <typename T, int DH, int PHASE>
__global__ __launch_bounds__(256) void attn_bwd_tiled_kernel(const mmfm_attn_desc d) {
    constexpr int DT = (DH + 31) / 32;
    constexpr int LD = DH + 1;
    constexpr int SLD = 33;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, kh = lane >> 5, l31 = lane & 31;
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk;
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    const int Lmx = max(Lq, Lk);
    float* Qs = smem;                       // [TCH][LD]   (rows relative to the query window qw0)
    float* dOs = Qs + TCH * LD;             // [TCH][LD]
    float* Ks = dOs + TCH * LD;             // [TCH][LD]   (rows relative to the key window kw0)
    float* Vs = Ks + TCH * LD;              // [TCH][LD]
    float* lse = Vs + TCH * LD;             // [TCH]
    float* dlt = lse + TCH;                 // [TCH]
    float* Sc = dlt + TCH;                  // [4][32][SLD]
    uint8_t* kpad = reinterpret_cast<uint8_t*>(Sc + 4 * 32 * SLD);
    uint8_t* modl = kpad + LkP;

    const T* qg = reinterpret_cast<const T*>(d.q) + (size_t)b * Lq * d.ldq + h * DH;
    const T* kg = reinterpret_cast<const T*>(d.k) + (size_t)b * Lk * d.ldk + h * DH;
    const T* vg = reinterpret_cast<const T*>(d.v) + (size_t)b * Lk * d.ldv + h * DH;
    const T* og = reinterpret_cast<const T*>(d.o) + (size_t)b * Lq * d.ldo + h * DH;
    const T* dog = reinterpret_cast<const T*>(d.d_o) + (size_t)b * Lq * d.lddo + h * DH;
    const Drop dp = drop_init(d.drop_p), dout = drop_init(d.drop_o);

    auto load_keys = [&](int kw0, int rows) {
        load_head<T, DH>(Ks, LD, kg + (size_t)kw0 * d.ldk, d.ldk, max(0, min(rows, Lk - kw0)), rows, t, 256);
        load_head<T, DH>(Vs, LD, vg + (size_t)kw0 * d.ldv, d.ldv, max(0, min(rows, Lk - kw0)), rows, t, 256);
    };
    auto load_queries = [&](int qw0, int rows) {
        load_head<T, DH>(Qs, LD, qg + (size_t)qw0 * d.ldq, d.ldq, max(0, min(rows, Lq - qw0)), rows, t, 256);
        constexpr int C4 = DH / 4;
        for (int idx = t; idx < rows * C4; idx += 256) {
            const int row = idx / C4, c = idx % C4, qrow = qw0 + row;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f), o = g;
            if (qrow < Lq) {
                g = io<T>::ld4(dog + (size_t)qrow * d.lddo + 4 * c);
                o = io<T>::ld4(og + (size_t)qrow * d.ldo + 4 * c);
            }
            float part = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
#pragma unroll
            for (int off = 1; off < C4; off <<= 1) part += __shfl_xor(part, off);
            if (c == 0) dlt[row] = part;
            const uint64_t base = ((uint64_t)b * Lq + (uint64_t)qrow) * (uint64_t)(d.heads * DH) + (uint64_t)(h * DH + 4 * c);
            float* p = dOs + row * LD + 4 * c;
            p[0] = dout.apply(g.x, base + 0); p[1] = dout.apply(g.y, base + 1);
            p[2] = dout.apply(g.z, base + 2); p[3] = dout.apply(g.w, base + 3);
        }
        for (int i = t; i < rows; i += 256) lse[i] = (qw0 + i < Lq) ? d.lse[(size_t)bh_ * Lq + qw0 + i] : 0.f;
    };

    for (int i = t; i < LkP; i += 256) kpad[i] = (i < Lk && d.keypad) ? d.keypad[(size_t)b * Lk + i] : 0;
    if (d.flags & MMFM_ATTN_SEP)
        for (int i = t; i < Lmx; i += 256) modl[i] = d.mod_id[i];
    MaskCtx mk{kpad, modl, modl, d.flags};
    float* sc = Sc + wave * 32 * SLD;
    const uint64_t pbase = (uint64_t)bh_ * Lq;
    const int own0 = blockIdx.y * TCH;                 // first owned row (key for phase 0, query for phase 1)
    const int tl = wave;                               // owned tile, local to the window

    if (PHASE == 0) {
        const int kt = blockIdx.y * 4 + wave;
        const bool active = kt < LkP / 32;
        load_keys(own0, min(TCH, LkP - own0));         // own0 < LkP by grid construction
        f32x16 dKt[DT], dVt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { dKt[i][r] = 0.f; dVt[i][r] = 0.f; }
        const int key = kt * 32 + l31;
        const float* kb = Ks + (tl * 32 + l31) * LD + kh;
        const float* vb = Vs + (tl * 32 + l31) * LD + kh;
        for (int c0 = 0; c0 < LqP; c0 += TCH) {
            const int rows = min(TCH, LqP - c0);
            __syncthreads();
            load_queries(c0, rows);
            __syncthreads();
            if (!active) continue;
            for (int qt = 0; qt < rows / 32; ++qt) {
                f32x16 s, dpv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
                const float* qa = Qs + (qt * 32 + l31) * LD + kh;
                const float* da = dOs + (qt * 32 + l31) * LD + kh;
#pragma unroll
                for (int k2 = 0; k2 < DH / 2; ++k2) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * k2], kb[2 * k2], s, 0, 0, 0);
                    dpv = __builtin_amdgcn_mfma_f32_32x32x2f32(da[2 * k2], vb[2 * k2], dpv, 0, 0, 0);
                }
                float pd[16], ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = qt * 32 + mrow(r, kh), q = c0 + ql;
                    const bool ok = (q < Lq) && (key < Lk) && mk.allowed(q, key);
                    const float p = ok ? __expf(s[r] * d.scale - lse[ql]) : 0.f;
                    const uint64_t idx = (pbase + (uint64_t)q) * Lk + (uint64_t)key;
                    const bool keep = !dp.on() || dp.keep(idx);
                    pd[r] = keep ? p * dp.scale : 0.f;
                    const float dpd = keep ? dpv[r] * dp.scale : 0.f;
                    ds[r] = p * (dpd - dlt[ql]) * d.scale;
                }
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const int dcol = i * 32 + l31;
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        const int qrow = qt * 32 + mrow(k2, kh);
                        const float doT = (dcol < DH) ? dOs[qrow * LD + dcol] : 0.f;
                        const float qT = (dcol < DH) ? Qs[qrow * LD + dcol] : 0.f;
                        dVt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(doT, pd[k2], dVt[i], 0, 0, 0);
                        dKt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qT, ds[k2], dKt[i], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) return;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            T* outg = reinterpret_cast<T*>(which ? d.dv : d.dk) + (size_t)b * Lk * (which ? d.lddv : d.lddk) + h * DH;
            const int ldo_ = which ? d.lddv : d.lddk;
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[l31 * SLD + mrow(r, kh)] = which ? dVt[i][r] : dKt[i][r];
                wave_lds_fence();
                constexpr int CW = (DH < 32 ? DH : 32) / 4;
                for (int idx = lane; idx < 32 * CW; idx += 64) {
                    const int row = idx / CW, c = idx % CW;
                    if (kt * 32 + row < Lk) {
                        const float* p = sc + row * SLD + 4 * c;
                        io<T>::st4(outg + (size_t)(kt * 32 + row) * ldo_ + i * 32 + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
                    }
                }
            }
        }
    } else {
        const int qt = blockIdx.y * 4 + wave;
        const bool active = qt < LqP / 32;
        load_queries(own0, min(TCH, LqP - own0));
        f32x16 dQt[DT];
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[i][r] = 0.f;
        const int q = qt * 32 + l31;
        const float* qb = Qs + (tl * 32 + l31) * LD + kh;
        const float* db = dOs + (tl * 32 + l31) * LD + kh;
        float lq = 0.f, dq_ = 0.f;
        for (int c0 = 0; c0 < LkP; c0 += TCH) {
            const int rows = min(TCH, LkP - c0);
            __syncthreads();
            load_keys(c0, rows);
            __syncthreads();
            if (!active) continue;
            lq = lse[tl * 32 + l31]; dq_ = dlt[tl * 32 + l31];
            for (int kt = 0; kt < rows / 32; ++kt) {
                f32x16 s, dpv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dpv[r] = 0.f; }
                const float* ka = Ks + (kt * 32 + l31) * LD + kh;
                const float* va = Vs + (kt * 32 + l31) * LD + kh;
#pragma unroll
                for (int k2 = 0; k2 < DH / 2; ++k2) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2 * k2], qb[2 * k2], s, 0, 0, 0);
                    dpv = __builtin_amdgcn_mfma_f32_32x32x2f32(va[2 * k2], db[2 * k2], dpv, 0, 0, 0);
                }
                float ds[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = c0 + kt * 32 + mrow(r, kh);
                    const bool ok = (q < Lq) && (key < Lk) && mk.allowed(q, key);
                    const float p = ok ? __expf(s[r] * d.scale - lq) : 0.f;
                    const uint64_t idx = (pbase + (uint64_t)q) * Lk + (uint64_t)key;
                    const bool keep = !dp.on() || dp.keep(idx);
                    const float dpd = keep ? dpv[r] * dp.scale : 0.f;
                    ds[r] = p * (dpd - dq_) * d.scale;
                }
#pragma unroll
                for (int i = 0; i < DT; ++i) {
                    const int dcol = i * 32 + l31;
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) {
                        const float kT = (dcol < DH) ? Ks[(kt * 32 + mrow(k2, kh)) * LD + dcol] : 0.f;
                        dQt[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(kT, ds[k2], dQt[i], 0, 0, 0);
                    }
                }
            }
        }
        if (!active) return;
        T* outg = reinterpret_cast<T*>(d.dq) + (size_t)b * Lq * d.lddq + h * DH;
#pragma unroll
        for (int i = 0; i < DT; ++i) {
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[l31 * SLD + mrow(r, kh)] = dQt[i][r];
            wave_lds_fence();
            constexpr int CW = (DH < 32 ? DH : 32) / 4;
            for (int idx = lane; idx < 32 * CW; idx += 64) {
                const int row = idx / CW, c = idx % CW;
                if (qt * 32 + row < Lq) {
                    const float* p = sc + row * SLD + 4 * c;
                    io<T>::st4(outg + (size_t)(qt * 32 + row) * d.lddq + i * 32 + 4 * c, make_float4(p[0], p[1], p[2], p[3]));
                }
            }
        }
    }
}

size_t fwd_tiled_lds_bytes(int Lq, int Lk, int dh) {
    const int DT = (dh + 31) / 32, LkP = (Lk + 31) & ~31, DVL = DT * 32;
    return (size_t)(TCH * (dh + 1) + TCH * DVL + 4 * 32 * (DVL + 1)) * 4 + LkP + std::max(Lq, Lk) + 16;
}
size_t bwd_tiled_lds_bytes(int Lq, int Lk, int dh) {
    const int LkP = (Lk + 31) & ~31;
    return (size_t)(4 * TCH * (dh + 1) + 2 * TCH + 4 * 32 * 33) * 4 + LkP + std::max(Lq, Lk) + 16;
}

size_t fwd_lds_bytes(int Lq, int Lk, int dh) {
    const int DT = (dh + 31) / 32, LkP = (Lk + 31) & ~31, DVL = DT * 32;
    return (size_t)(LkP * (dh + 1) + LkP * DVL + 4 * 32 * (DVL + 1)) * 4 + LkP + std::max(Lq, Lk) + 16;
}
size_t bwd_lds_bytes(int Lq, int Lk, int dh) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    return (size_t)(2 * LqP * (dh + 1) + 2 * LkP * (dh + 1) + 2 * LqP + 4 * 32 * 33) * 4 + LkP + std::max(Lq, Lk) + 16;
}

// Opt in to > 64 KiB of dynamic LDS once per kernel (not a stream operation; done on the first,
// un-captured call so later graph captures see no attribute change).
template <typename K>
int set_lds(K kern, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, size_t> done;             // per (device, kernel)
    if (bytes <= 65536) return 0;
    const void* fn = reinterpret_cast<const void*>(kern);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)fn ^ ((uint64_t)(dev + 1) << 56);
    std::lock_guard<std::mutex> g(mu);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mmfm_set_error((int)e, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    done[key] = 160 * 1024;
    return 0;
}

int check_common(const mmfm_attn_desc& d, const char* who) {
    MMFM_REQUIRE(d.dtype == MMFM_F32 || d.dtype == MMFM_BF16, "%s: bad dtype", who);
    MMFM_REQUIRE(d.B > 0 && d.heads > 0 && d.Lq > 0 && d.Lk > 0, "%s: bad shape", who);
    MMFM_REQUIRE(d.dh == 8 || d.dh == 16 || d.dh == 32 || d.dh == 64, "%s: head dim %d not in {8,16,32,64}", who, d.dh);
    MMFM_REQUIRE(d.q && d.k && d.v && d.o && d.lse, "%s: null tensor", who);
    const int hd = d.heads * d.dh, al = 4;
    MMFM_REQUIRE(d.ldq >= hd && d.ldk >= hd && d.ldv >= hd && d.ldo >= hd, "%s: leading dim < heads*dh", who);
    MMFM_REQUIRE(d.ldq % al == 0 && d.ldk % al == 0 && d.ldv % al == 0 && d.ldo % al == 0, "%s: leading dims must be multiples of 4", who);
    const size_t esz = d.dtype == MMFM_F32 ? 16 : 8;
    MMFM_REQUIRE((uintptr_t)d.q % esz == 0 && (uintptr_t)d.k % esz == 0 && (uintptr_t)d.v % esz == 0 && (uintptr_t)d.o % esz == 0,
                 "%s: q/k/v/o must be %zu-byte aligned", who, esz);
    MMFM_REQUIRE((d.flags & MMFM_ATTN_CAUSAL) || d.keypad, "%s: keypad required unless CAUSAL", who);
    MMFM_REQUIRE(!(d.flags & MMFM_ATTN_SEP) || d.mod_id, "%s: SEP needs mod_id", who);
    MMFM_REQUIRE(!(d.flags & (MMFM_ATTN_DIAG | MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP)) || d.Lq == d.Lk, "%s: DIAG/CAUSAL/SEP need Lq == Lk", who);
    return 0;
}

}  // namespace

#define ATTN_DISPATCH(KERN, TT, DHV)                                                             \
    {                                                                                            \
        auto kern = KERN<TT, DHV>;                                                               \
        if (int rc = set_lds(kern, lds)) return rc;                                              \
        hipLaunchKernelGGL(kern, dim3(d.B * d.heads), dim3(256), lds, (hipStream_t)stream, d);   \
    }

#define ATTN_DISPATCH_ALL(KERN)                                                  \
    if (d.dtype == MMFM_F32) {                                                   \
        switch (d.dh) {                                                          \
            case 8: ATTN_DISPATCH(KERN, float, 8) break;                         \
            case 16: ATTN_DISPATCH(KERN, float, 16) break;                       \
            case 32: ATTN_DISPATCH(KERN, float, 32) break;                       \
            default: ATTN_DISPATCH(KERN, float, 64) break;                       \
        }                                                                        \
    } else {                                                                     \
        switch (d.dh) {                                                          \
            case 8: ATTN_DISPATCH(KERN, uint16_t, 8) break;                      \
            case 16: ATTN_DISPATCH(KERN, uint16_t, 16) break;                    \
            case 32: ATTN_DISPATCH(KERN, uint16_t, 32) break;                    \
            default: ATTN_DISPATCH(KERN, uint16_t, 64) break;                    \
        }                                                                        \
    }

int mmfm_attn_bf16_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st);   // attention_bf16.hip (bf16 MFMA)
int mmfm_attn_fast_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st);   // attention_fast.hip (bf16, dh 32, L <= 224)
int mmfm_attn_long_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st);   // attention_long.hip (bf16, dh 64, keep-bit workspace)

#define ATTN_TILED(KERN, GY, ...)                                                                                  \
    {                                                                                                              \
        auto kern = KERN<__VA_ARGS__>;                                                                             \
        if (int rc = set_lds(kern, lds)) return rc;                                                                \
        hipLaunchKernelGGL(kern, dim3(d.B * d.heads, GY), dim3(256), lds, (hipStream_t)stream, d);                 \
    }
#define ATTN_TILED_ALL(KERN, GY, ...)                                                                              \
    if (d.dtype == MMFM_F32) {                                                                                     \
        switch (d.dh) {                                                                                            \
            case 8: ATTN_TILED(KERN, GY, float, 8 __VA_ARGS__) break;                                              \
            case 16: ATTN_TILED(KERN, GY, float, 16 __VA_ARGS__) break;                                            \
            case 32: ATTN_TILED(KERN, GY, float, 32 __VA_ARGS__) break;                                            \
            default: ATTN_TILED(KERN, GY, float, 64 __VA_ARGS__) break;                                            \
        }                                                                                                          \
    } else {                                                                                                       \
        switch (d.dh) {                                                                                            \
            case 8: ATTN_TILED(KERN, GY, uint16_t, 8 __VA_ARGS__) break;                                           \
            case 16: ATTN_TILED(KERN, GY, uint16_t, 16 __VA_ARGS__) break;                                         \
            case 32: ATTN_TILED(KERN, GY, uint16_t, 32 __VA_ARGS__) break;                                         \
            default: ATTN_TILED(KERN, GY, uint16_t, 64 __VA_ARGS__) break;                                         \
        }                                                                                                          \
    }

// One decision per SHAPE for forward and backward together: the bf16-MFMA kernels and the fp32-compute kernels draw their
// attention-dropout decisions from different hash layouts, so a forward/backward pair must never be split across them,
// and the untiled/tiled choice must not depend on which of the two passes happens to fit.
static bool use_tiled(const mmfm_attn_desc& d) {
    static const bool force = [] { const char* e = getenv("MMFM_ATTN_FORCE_TILED"); return e && atoi(e) != 0; }();
    return force || fwd_lds_bytes(d.Lq, d.Lk, d.dh) > 160 * 1024 || bwd_lds_bytes(d.Lq, d.Lk, d.dh) > 160 * 1024;
}

extern "C" int mmfm_attn_fwd(const mmfm_attn_desc* dp, mmfm_stream stream) {
    MMFM_REQUIRE(dp, "mmfm_attn_fwd: null descriptor");
    static const int no_remap = [] { const char* e = getenv("MMFM_ATTN_NO_REMAP"); return e ? atoi(e) : 0; }();
    mmfm_attn_desc d_ = *dp;
    d_.flags = (d_.flags & 0xff) | (no_remap ? 0x100 : 0);          // bit 8: plain workgroup order (common.h attn_xcd_remap)
    const mmfm_attn_desc d = d_;
    if (int rc = check_common(d, "mmfm_attn_fwd")) return rc;
    if (d.dtype == MMFM_BF16) {          // bf16 MFMA kernel; shapes it does not take fall through to fp32 compute on bf16 storage
        const int rf = mmfm_attn_fast_launch(d, false, (hipStream_t)stream);
        if (rf != -1000) return rf;
        const int rl = mmfm_attn_long_launch(d, false, (hipStream_t)stream);
        if (rl != -1000) return rl;
        const int rc = mmfm_attn_bf16_launch(d, false, (hipStream_t)stream);
        if (rc != -1000) return rc;
    }
    if (use_tiled(d)) {
        const size_t lds = fwd_tiled_lds_bytes(d.Lq, d.Lk, d.dh);
        MMFM_REQUIRE(lds <= 160 * 1024, "mmfm_attn_fwd: Lq=%d Lk=%d needs %zu B of LDS for the mask bytes alone", d.Lq, d.Lk, lds);
        const int gy = ((d.Lq + 31) / 32 + 3) / 4;
        ATTN_TILED_ALL(attn_fwd_tiled_kernel, gy)
        MMFM_LAUNCH_CHECK("mmfm_attn_fwd(tiled)");
        return 0;
    }
    const size_t lds = fwd_lds_bytes(d.Lq, d.Lk, d.dh);
    ATTN_DISPATCH_ALL(attn_fwd_kernel)
    MMFM_LAUNCH_CHECK("mmfm_attn_fwd");
    return 0;
}

extern "C" int mmfm_attn_bwd(const mmfm_attn_desc* dp, mmfm_stream stream) {
    MMFM_REQUIRE(dp, "mmfm_attn_bwd: null descriptor");
    static const int no_remap = [] { const char* e = getenv("MMFM_ATTN_NO_REMAP"); return e ? atoi(e) : 0; }();
    mmfm_attn_desc d_ = *dp;
    d_.flags = (d_.flags & 0xff) | (no_remap ? 0x100 : 0);
    const mmfm_attn_desc d = d_;
    if (int rc = check_common(d, "mmfm_attn_bwd")) return rc;
    MMFM_REQUIRE(d.d_o && d.dq && d.dk && d.dv, "mmfm_attn_bwd: null gradient tensor");
    const int hd = d.heads * d.dh;
    MMFM_REQUIRE(d.lddo >= hd && d.lddq >= hd && d.lddk >= hd && d.lddv >= hd, "mmfm_attn_bwd: gradient leading dim < heads*dh");
    MMFM_REQUIRE(d.lddo % 4 == 0 && d.lddq % 4 == 0 && d.lddk % 4 == 0 && d.lddv % 4 == 0, "mmfm_attn_bwd: gradient leading dims must be multiples of 4");
    if (d.dtype == MMFM_BF16) {
        const int rf = mmfm_attn_fast_launch(d, true, (hipStream_t)stream);
        if (rf != -1000) return rf;
        const int rl = mmfm_attn_long_launch(d, true, (hipStream_t)stream);
        if (rl != -1000) return rl;
        const int rc = mmfm_attn_bf16_launch(d, true, (hipStream_t)stream);
        if (rc != -1000) return rc;
    }
    if (use_tiled(d)) {
        const size_t lds = bwd_tiled_lds_bytes(d.Lq, d.Lk, d.dh);
        MMFM_REQUIRE(lds <= 160 * 1024, "mmfm_attn_bwd: Lq=%d Lk=%d needs %zu B of LDS for the mask bytes alone", d.Lq, d.Lk, lds);
        const int gk = ((d.Lk + 31) / 32 + 3) / 4, gq = ((d.Lq + 31) / 32 + 3) / 4;
#define COMMA_PHASE0 , 0
#define COMMA_PHASE1 , 1
        ATTN_TILED_ALL(attn_bwd_tiled_kernel, gk, COMMA_PHASE0)
        ATTN_TILED_ALL(attn_bwd_tiled_kernel, gq, COMMA_PHASE1)
#undef COMMA_PHASE0
#undef COMMA_PHASE1
        MMFM_LAUNCH_CHECK("mmfm_attn_bwd(tiled)");
        return 0;
    }
    const size_t lds = bwd_lds_bytes(d.Lq, d.Lk, d.dh);
    ATTN_DISPATCH_ALL(attn_bwd_kernel)
    MMFM_LAUNCH_CHECK("mmfm_attn_bwd");
    return 0;
}
