// bf16 attention, dh = 32, heads of up to 224 keys / 256 queries (the d_model-256 configurations: L = 200): the straight-line
// forward and single-pass backward that the step spends its attention time in.  Same contract as attention_bf16.hip
// (masks: key padding and DIAG; CAUSAL / SEP stay with the general kernels), reference: mm_utils.py:97-152.
//
// What changed against the general kernels, and why (round-2 counters: both were VALU-issue bound at 3,780 / 2,660 vector
// instructions per wave, MFMA pipe 7 % busy):
//   * attention-probability dropout comes from a precomputed KEEP-BIT MASK (mmfm_attn_desc.drop_mask), one bit per
//     (b, head, query, key), written by attn_dropmask_kernel in front of the forward and read back by the backward.
//     Layout [b*heads][query tile][key tile][32 words]: word w of a tile belongs to key mrow(w >> 1, w & 1), bit i to query i.
//       forward  (lane = query, accumulator register r = keys mrow(r, 0) | mrow(r, 1) in the two lane halves): words 2r, 2r+1
//                ARE the 64-bit lane mask of register r -> 128 contiguous bytes per tile arrive by scalar loads and each
//                decision is ONE v_cndmask with an SGPR-pair condition (was: 7-instruction hash per key pair + 2 compares +
//                2 selects with VCC hazards, ~6.5 instructions per element);
//       backward (lane = key): the lane's word, shifted by 4 * (lane half), holds the decisions of its 16 queries at fixed bit
//                positions: v_bfe_i32 + two ANDs per element (was ~9).
//     The generator is a pure-VALU kernel at full occupancy (11 instructions per two decisions, ~30 us per call).
//   * masks ride on the MFMA: the score accumulator starts at bias[key] (0 or -inf, an LDS table), so padded keys and the
//     head's ragged last tile need no compare / select at all, and the code has no per-group branches: the tile body is
//     instantiated per number of valid 8-key (forward) / 8-query (backward) groups and the loop calls the full one;
//   * lazy rescaling of the running maximum (threshold 2^6: the accumulator and the running sum are only rescaled when a
//     row's maximum grows by more than that; the LSE stays exact because m_run is only a reference point);
//   * the forward runs one query tile per wave with as many waves as the head has tiles (7 at L = 200: no idle eighth wave).
#include "attn_common.h"
#include <algorithm>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

using namespace attn;

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2v;
// The keep mask is read through the CONSTANT address space: a uniform load from it is always a scalar (s_load) instruction.
// (Through a plain global pointer hipcc's no-clobber analysis gave up behind the workgroup reduction of the prologue and
// fetched the masks with vector loads - the SGPR-pair v_cndmask then has no operand.)  The generator kernel has finished
// before this one starts and nothing in this kernel writes the buffer.
typedef const __attribute__((address_space(4))) u32x4v* cmask_ptr;

constexpr float LAZY_THR = 6.f;       // log2 units: probabilities stay below 2^6 between rescales

#ifdef MMFM_ATTN_STAMP
// diagnostic build only (scripts/probe/build_attn_stamp.sh): shader-cycle totals per phase, summed over all waves
__device__ unsigned long long mmfm_attn_probe_acc[16];
#define ASTAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_a[6] = {0, 0, 0, 0, 0, 0}
#define ASTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                       __builtin_amdgcn_sched_barrier(0); st_a[i] += n_ - st_t; st_t = n_; } while (0)
#define ASTAMP_FLUSH(base) do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&mmfm_attn_probe_acc[(base) + i_], st_a[i_]); } while (0)
#else
#define ASTAMP_DECL
#define ASTAMP(i)
#define ASTAMP_FLUSH(base)
#endif

// ---------------------------------------------------------------------------------------------- keep-mask generator
// One thread per 32-bit word (one key x 32 queries).  Per word two full-strength 32-bit keys; per PAIR of queries (j, j+16) a
// 7-instruction xorshift / 24-bit-multiply mix whose two 15-bit fields are compared with the threshold by one subtraction:
// ((f | 0x8000) - t) has bit 15 set iff f >= t, for both halves of the register at once.
__global__ __launch_bounds__(256) void attn_dropmask_kernel(uint32_t* __restrict__ out, uint32_t nwords, const mmfm_dropout dp) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= nwords) return;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(dp.state);
    const uint32_t k0 = mix32(s[0] + dp.site * 0x9E3779B9u), k1 = mix32(s[1] ^ (dp.site * 0x85EBCA6Bu + 0xC2B2AE35u));
    const uint32_t ja = mix32(gid ^ k0), kb = mix32(gid * 0x9E3779B9u + k1);
    double tt = (double)dp.p * 32768.0;
    const uint32_t t15 = tt >= 32768.0 ? 32768u : (uint32_t)tt;
    const uint32_t T = t15 * 0x10001u;
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) {
        uint32_t h = __umul24(ja ^ (j * 0x9E37u), 0x7FEB35u) + kb;
        h ^= h >> 13;
        h = __umul24(h, 0x46CA6Bu);
        h ^= h >> 16;
        const uint32_t y = ((h & 0x7FFF7FFFu) | 0x80008000u) - T;
        acc = (acc >> 1) | (y & 0x80008000u);
    }
    out[gid] = acc;
}

__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// p where the lane's bit of the 64-bit mask (an SGPR pair) is set, else 0: one VALU instruction
__device__ __forceinline__ float keep_sel(float p, uint32_t lo, uint32_t hi) {
    const uint64_t m = ((uint64_t)hi << 32) | lo;
    float o;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(o) : "v"(p), "s"(m));
    return o;
}

// ---------------------------------------------------------------------------------------------- forward
// One key tile (32 keys x 32 queries, lane = query): online softmax on the scores `st` (bias already inside), keep mask,
// O^T += V^T P^T.  G = 8-key groups of the tile that exist (registers 4g .. 4g+3).
template <int G, bool DROP>
__device__ __forceinline__ void fwd_tile(const f32x16& st, float c2, float& m_run, float& l_run, f32x16& acc, const u32x4v (&mk)[8],
                                         const char* Vs, int kt, int lane) {
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 4 * G; ++r) mx = fmaxf(mx, st[r]);
    const float mt = xhalf_max(mx) * c2;                       // c2 > 0; identical in both lane halves
    const bool grow = mt > m_run + LAZY_THR;                   // m_run = -inf: any finite score grows it
    if (__any(grow)) {
        const float m_new = grow ? mt : m_run;
        const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= alpha;
        l_run *= alpha;
        m_run = m_new;
    }
    const float m_use = (m_run == -INFINITY) ? 0.f : m_run;    // nothing allowed so far: every p below is exp2(-inf) = 0
    float pd[16];
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r < 4 * G) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -m_use));
            ps += p;
            pd[r] = DROP ? keep_sel(p, mk[r >> 1][2 * (r & 1)], mk[r >> 1][2 * (r & 1) + 1]) : p;
        } else {
            pd[r] = 0.f;
        }
    }
    l_run += ps;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, 64, kt * 32, 0, lane), pack8(pd), acc, 0, 0, 0);
    if (G > 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, 64, kt * 32 + 16, 0, lane), pack8(pd + 8), acc, 0, 0, 0);
}

constexpr int F_KRS = 80, F_VRS = 64, F_ORS = 80;

template <int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_fast_kernel(const mmfm_attn_desc d) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LkP = (Lk + 31) & ~31;
    const int nqt = (Lq + 31) >> 5, nkt = LkP >> 5;
    char* Ks = smem;
    char* Vs = Ks + LkP * F_KRS;
    float* kbias = reinterpret_cast<float*>(Vs + LkP * F_VRS);
    char* ost = reinterpret_cast<char*>(kbias + LkP);
    int* wflag = reinterpret_cast<int*>(ost + NW * 32 * F_ORS);          // per-wave "a key of this head is padded" votes (no static LDS:
                                                                         // it would shift the 16-B aligned carve-up, Guideline 17)
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * 32;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * 32;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * 32;
    uint16_t* og = reinterpret_cast<uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * 32;

    ASTAMP_DECL;
    load_head16<32>(Ks, F_KRS, F_KRS / 16, kg, d.ldk, Lk, LkP, t, NT);
    load_head16<32>(Vs, F_VRS, F_VRS / 16, vg, d.ldv, Lk, LkP, t, NT);
    int pad = 0;
    for (int i = t; i < LkP; i += NT) {
        const bool ok = i < Lk && (d.keypad == nullptr || d.keypad[(size_t)b * Lk + i] != 0);
        kbias[i] = ok ? 0.f : -INFINITY;
        pad |= (i < Lk && !ok);
    }
    ASTAMP(0);
    const int wv = __any(pad) ? 1 : 0;                  // all 64 lanes vote before any divergence
    if (lane == 0) wflag[wave] = wv;
    __syncthreads();                                    // also: the images are complete
    int anypad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) anypad |= wflag[w];
    const int qt = wave;
    if (qt >= nqt) return;
    ASTAMP(1);

    const int q0 = qt * 32, q = q0 + l31;
    bf16x8v qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (q < Lq) v = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
        qf[ks] = __builtin_bit_cast(bf16x8v, v);
    }
    const float c2 = d.scale * LOG2E;
    const cmask_ptr mp = (cmask_ptr)(uintptr_t)d.drop_mask + (size_t)(bh_ * nqt + qt) * nkt * 8;
    const bool fixdiag = anypad && (d.flags & MMFM_ATTN_DIAG);

    // S^T tile: rows = keys (registers), lane = query; the accumulator starts at the key bias
    auto score = [&](int kt) {
        f32x16 a;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 kb4 = *reinterpret_cast<const float4*>(kbias + kt * 32 + 8 * g + 4 * kh);
            a[4 * g + 0] = kb4.x; a[4 * g + 1] = kb4.y; a[4 * g + 2] = kb4.z; a[4 * g + 3] = kb4.w;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kt * 32 + l31) * F_KRS + ks * 32 + kh * 16), qf[ks], a, 0, 0, 0);
        if (fixdiag && kt == qt) {                     // rare: padded keys in the head and `eye |`: a query always sees itself
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kt * 32 + l31) * F_KRS + ks * 32 + kh * 16), qf[ks], z, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = (mrow(r, kh) == l31) ? z[r] : a[r];
        }
        return a;
    };
    auto masks = [&](u32x4v (&mk)[8], int kt) {
        if (DROP) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mk[i] = mp[kt * 8 + i];
        }
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int gtail = ((Lk - (nkt - 1) * 32) + 7) >> 3;             // 1..4 valid groups in the last key tile (Lk % 8 == 0: launcher)
    auto tile = [&](int kt, const f32x16& st, const u32x4v (&mk)[8]) {
        if (kt == nkt - 1 && gtail != 4) {
            if (gtail == 1) fwd_tile<1, DROP>(st, c2, m_run, l_run, acc, mk, Vs, kt, lane);
            else if (gtail == 2) fwd_tile<2, DROP>(st, c2, m_run, l_run, acc, mk, Vs, kt, lane);
            else fwd_tile<3, DROP>(st, c2, m_run, l_run, acc, mk, Vs, kt, lane);
        } else {
            fwd_tile<4, DROP>(st, c2, m_run, l_run, acc, mk, Vs, kt, lane);
        }
    };
    // two score tiles in flight, ping-pong: tile kt+1's MFMAs are issued before the element-wise work of tile kt.  ONE mask set
    // (32 SGPRs; two sets spill scalar registers): the scalar loads of tile kt+1 are issued right behind tile kt's last select
    // and land under its P.V products and the max / exp2 algebra of tile kt+1, whose selects come last.
    u32x4v mk[8];
    f32x16 s0 = score(0), s1;
    masks(mk, 0);
    ASTAMP(2);
    for (int kt = 0; kt < nkt; kt += 2) {
        const bool has1 = kt + 1 < nkt;
        if (has1) s1 = score(kt + 1);
        tile(kt, s0, mk);
        if (has1) {
            masks(mk, kt + 1);
            if (kt + 2 < nkt) s0 = score(kt + 2);
            tile(kt + 1, s1, mk);
            if (kt + 2 < nkt) masks(mk, kt + 2);
        }
    }
    ASTAMP(3);
    const Drop dout = drop_init(d.drop_o);
    const float dscale = DROP ? 1.f / (1.f - d.drop_p.p) : 1.f;
    const float l_tot = xhalf_sum(l_run);
    const float inv = dscale / l_tot;
    if (kh == 0 && q < Lq) d.lse[(size_t)bh_ * Lq + q] = m_run * LN2 + __logf(l_tot);
    // O^T (rows = d in registers, lane = query) -> bf16 rows [query][d] through the wave's staging tile, output dropout on the way
    char* tl = ost + wave * 32 * F_ORS;
    const uint64_t base = ((uint64_t)b * Lq + (uint64_t)q) * (uint64_t)(d.heads * 32) + (uint64_t)(h * 32);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 8 * g + 4 * kh;
        float v0 = acc[4 * g + 0] * inv, v1 = acc[4 * g + 1] * inv, v2 = acc[4 * g + 2] * inv, v3 = acc[4 * g + 3] * inv;
        if (dout.on()) {
            dout.apply2(v0, v1, base + d0);
            dout.apply2(v2, v3, base + d0 + 2);
        }
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
        bf16x4v pk;
        pk[0] = (__bf16)v0; pk[1] = (__bf16)v1; pk[2] = (__bf16)v2; pk[3] = (__bf16)v3;
        *reinterpret_cast<uint2*>(tl + l31 * F_ORS + d0 * 2) = __builtin_bit_cast(uint2, pk);
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = lane + 64 * i, row = idx >> 2, c = idx & 3;
        if (q0 + row < Lq)
            *reinterpret_cast<uint4*>(og + (size_t)(q0 + row) * d.ldo + 8 * c) = *reinterpret_cast<const uint4*>(tl + row * F_ORS + c * 16);
    }
    ASTAMP(4);
    ASTAMP_FLUSH(0);
}

size_t fwd_fast_lds(int Lk, int nw) {
    const int LkP = (Lk + 31) & ~31;
    return (size_t)LkP * (F_KRS + F_VRS + 4) + (size_t)nw * 32 * F_ORS + 64;
}

// ---------------------------------------------------------------------------------------------- backward (single pass)
// Structure of attn_bwd1_bf16_kernel (attention_bf16.hip): compute waves 0..6 own one 32-key tile each (lane = key; dK^T, dV^T
// in accumulators) and walk the query tiles in lock step, dropping their packed dS^T tile into an LDS staging slot; wave 7
// turns the staged tiles of the previous query tile into dQ.  One barrier per query tile; fixed order, no atomics.
constexpr int B_NW = 8, B_CW = 7, B_TS = 80, B_TILE = 32 * B_TS, B_RS = 80;

// one query tile of a compute wave.  GRP = 8-query groups of the tile that exist (registers 4g .. 4g+3 = queries 8g + 4 kh + 0..3)
template <int GRP, bool DROP>
__device__ __forceinline__ void bwd_tile(const f32x16& s, const f32x16& dpv, uint32_t w, float c2, const float* lse2, const float* dlt,
                                         const char* As, const char* Bs, char* slot, int qt, int kh, int lane, f32x16& dKt, f32x16& dVt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (2 * s2 >= GRP) continue;
        const bool both = GRP > 2 * s2 + 1;
        float pd[8], ds[8];
        const float4 la = *reinterpret_cast<const float4*>(lse2 + qt * 32 + 16 * s2 + 4 * kh);
        const float4 da = *reinterpret_cast<const float4*>(dlt + qt * 32 + 16 * s2 + 4 * kh);
        float4 lb = la, db = da;
        if (both) {
            lb = *reinterpret_cast<const float4*>(lse2 + qt * 32 + 16 * s2 + 8 + 4 * kh);
            db = *reinterpret_cast<const float4*>(dlt + qt * 32 + 16 * s2 + 8 + 4 * kh);
        }
        const float ls[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
        const float dl[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e >= 4 && !both) { pd[e] = 0.f; ds[e] = 0.f; continue; }
            const int r = 8 * s2 + e;
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -ls[e]));
            float g = dpv[r], pk = p;
            if (DROP) {
                const int pos = mrow(r, 0);                                           // bit of query mrow(r, kh) after the shift by 4 kh
                const uint32_t km = (uint32_t)(((int32_t)(w << (31 - pos))) >> 31);   // v_bfe_i32: all ones = keep
                g = __uint_as_float(__float_as_uint(g) & km);
                pk = __uint_as_float(__float_as_uint(p) & km);
            }
            pd[e] = pk;
            ds[e] = p * (g - dl[e]);
        }
        const bf16x8v pf = pack8(pd), sf = pack8(ds);
        // dS^T[key = lane][q]: elements 0..3 are queries 16*s2 + 4*kh + 0..3, elements 4..7 the same + 8
        const uint4 sw = __builtin_bit_cast(uint4, sf);
        *reinterpret_cast<uint2*>(slot + (16 * s2 + 4 * kh) * 2) = make_uint2(sw.x, sw.y);
        *reinterpret_cast<uint2*>(slot + (16 * s2 + 8 + 4 * kh) * 2) = make_uint2(sw.z, sw.w);
        dVt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Bs, B_RS, qt * 32 + 16 * s2, 0, lane), pf, dVt, 0, 0, 0);
        dKt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(As, B_RS, qt * 32 + 16 * s2, 0, lane), sf, dKt, 0, 0, 0);
    }
    if (GRP <= 2) {                                    // the dQ wave reads whole tiles: the second half must not be stale
        *reinterpret_cast<uint2*>(slot + (16 + 4 * kh) * 2) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(slot + (24 + 4 * kh) * 2) = make_uint2(0u, 0u);
    }
}

template <bool DROP>
__global__ __launch_bounds__(B_NW * 64, 4) void attn_bwd_fast_kernel(const mmfm_attn_desc d) {
    constexpr int NW = B_NW, CW = B_CW, TS = B_TS, TILE = B_TILE, RS = B_RS, NT = NW * 64, C8 = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    char* As = smem;                                  // Q image
    char* Bs = As + LqP * RS;                         // dO image (output dropout applied)
    float* lse2 = reinterpret_cast<float*>(Bs + LqP * RS);
    float* dlt = lse2 + LqP;
    float* kbias = dlt + LqP;
    char* stg = reinterpret_cast<char*>(kbias + LkP); // [2][CW][32 keys x TS] dS^T tiles; first the K image (prologue only)
    char* sc7 = stg + 2 * CW * TILE;                  // [32 x RS] dQ transpose tile of wave 7
    int* wflag = reinterpret_cast<int*>(sc7 + 32 * RS);
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * 32;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * 32;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * 32;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * 32;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * 32;
    // constant factors stay out of the per-element algebra: P~ and dP carry the keep mask only, delta is pre-divided by the
    // dropout scale, and dK / dQ (x softmax scale x dropout scale) and dV (x dropout scale) are scaled once, when stored
    const float dscale = DROP ? 1.f / (1.f - d.drop_p.p) : 1.f;
    const float inv_ds = 1.f / dscale, osc_dk = d.scale * dscale;
    const Drop dout = drop_init(d.drop_o);

    load_head16<32>(As, RS, RS / 16, qg, d.ldq, Lq, LqP, t, NT);
    load_head16<32>(stg, RS, RS / 16, kg, d.ldk, Lk, LkP, t, NT);          // K image, for wave 7's K^T operands only
    for (int idx = t; idx < LqP * C8; idx += NT) {                          // dO = dropout'(d_o) as bf16;  delta = rowsum(d_o * o)
        const int row = idx >> 2, c = idx & 3;
        uint4 g = make_uint4(0u, 0u, 0u, 0u), o = g;
        if (row < Lq) {
            g = *reinterpret_cast<const uint4*>(dog + (size_t)row * d.lddo + 8 * c);
            o = *reinterpret_cast<const uint4*>(og + (size_t)row * d.ldo + 8 * c);
        }
        const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, ow[4] = {o.x, o.y, o.z, o.w};
        const uint64_t base = ((uint64_t)b * Lq + (uint64_t)row) * (uint64_t)(d.heads * 32) + (uint64_t)(h * 32 + 8 * c);
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
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        if (c == 0) dlt[row] = part * inv_ds;
        *reinterpret_cast<uint4*>(Bs + row * RS + c * 16) = __builtin_bit_cast(uint4, pack8(gd));
    }
    for (int i = t; i < LqP; i += NT) {
        *reinterpret_cast<uint4*>(Bs + i * RS + 64) = make_uint4(0u, 0u, 0u, 0u);
        lse2[i] = (i < Lq) ? d.lse[(size_t)bh_ * Lq + i] * LOG2E : 0.f;
    }
    int pad = 0;
    for (int i = t; i < LkP; i += NT) {
        const bool ok = i < Lk && (d.keypad == nullptr || d.keypad[(size_t)b * Lk + i] != 0);
        kbias[i] = ok ? 0.f : -INFINITY;
        pad |= (i < Lk && !ok);
    }
    const int wv = __any(pad) ? 1 : 0;
    if (lane == 0) wflag[wave] = wv;
    __syncthreads();
    int anypad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) anypad |= wflag[w];
    const bool fixdiag = anypad && (d.flags & MMFM_ATTN_DIAG);
    const int nqt = LqP / 32, nkt = LkP / 32;        // nkt <= CW (launcher)
    const float c2 = d.scale * LOG2E;

    if (wave < CW) {
        // ---------------- compute waves: one key tile each, all query tiles
        const int kt = wave;
        const bool active = kt < nkt;
        f32x16 dKt, dVt;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dKt[r] = 0.f; dVt[r] = 0.f; }
        const int key = kt * 32 + l31;
        bf16x8v kfr[2], vfr[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 kv = make_uint4(0u, 0u, 0u, 0u), vv = kv;
            if (active && key < Lk) {
                kv = *reinterpret_cast<const uint4*>(kg + (size_t)key * d.ldk + ks * 16 + 8 * kh);
                vv = *reinterpret_cast<const uint4*>(vg + (size_t)key * d.ldv + ks * 16 + 8 * kh);
            }
            kfr[ks] = __builtin_bit_cast(bf16x8v, kv);
            vfr[ks] = __builtin_bit_cast(bf16x8v, vv);
        }
        const float kbv = active ? kbias[key] : 0.f;
        // the lane's mask word of query tile qt: word index of key l31 inside a tile (inverse of mrow), see the header comment
        const int widx = 2 * ((l31 & 3) + 4 * (l31 >> 3)) + ((l31 >> 2) & 1);
        const uint32_t* __restrict__ mw = reinterpret_cast<const uint32_t*>(d.drop_mask) + ((size_t)bh_ * nqt * nkt + (active ? kt : 0)) * 32 + widx;
        auto maskword = [&](int qt) -> uint32_t { return DROP ? (mw[(size_t)qt * nkt * 32] >> (4 * kh)) : 0u; };
        uint32_t w_cur = maskword(0);
        f32x16 s, dpv;
        auto scoresA = [&](int qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = kbv; dpv[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = (qt * 32 + l31) * RS + ks * 32 + kh * 16;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), kfr[ks], s, 0, 0, 0);        // S[q][key]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), vfr[ks], dpv, 0, 0, 0);    // dP[q][key]
            }
            if (fixdiag && qt == kt) {                 // rare: padded keys and `eye |`: S[q][q] is allowed even when key q is padded
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, (qt * 32 + l31) * RS + ks * 32 + kh * 16), kfr[ks], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (mrow(r, kh) == l31) ? z[r] : s[r];
            }
        };
        const int gtail = ((Lq - (nqt - 1) * 32) + 7) >> 3;            // valid 8-query groups of the last query tile (Lq % 8 == 0)
        __syncthreads();                               // wave 7 has its K^T operands: the K image is dead, the staging slots free
        for (int qt = 0; qt < nqt; ++qt) {
            if (active) {
                const uint32_t w = w_cur;
                if (qt + 1 < nqt) w_cur = maskword(qt + 1);
                scoresA(qt);
                char* slot = stg + ((qt & 1) * CW + wave) * TILE + l31 * TS;
                if (qt == nqt - 1 && gtail != 4) {
                    if (gtail == 1) bwd_tile<1, DROP>(s, dpv, w, c2, lse2, dlt, As, Bs, slot, qt, kh, lane, dKt, dVt);
                    else if (gtail == 2) bwd_tile<2, DROP>(s, dpv, w, c2, lse2, dlt, As, Bs, slot, qt, kh, lane, dKt, dVt);
                    else bwd_tile<3, DROP>(s, dpv, w, c2, lse2, dlt, As, Bs, slot, qt, kh, lane, dKt, dVt);
                } else {
                    bwd_tile<4, DROP>(s, dpv, w, c2, lse2, dlt, As, Bs, slot, qt, kh, lane, dKt, dVt);
                }
            }
            __syncthreads();                           // staging buffer (qt & 1) is complete; buffer ((qt+1) & 1) has been consumed
        }
        // Q / dO images are dead (every compute wave passed the last barrier): rows [32w, 32w+32) carry this wave's stores
        if (active) {
            const f32x16 dk1[1] = {dKt}, dv1[1] = {dVt};
            store_tile_T<32, 1>(As + 32 * wave * RS, RS, dk1, reinterpret_cast<uint16_t*>(d.dk) + (size_t)b * Lk * d.lddk + h * 32, d.lddk,
                                kt * 32, Lk, lane, osc_dk);
            store_tile_T<32, 1>(Bs + 32 * wave * RS, RS, dv1, reinterpret_cast<uint16_t*>(d.dv) + (size_t)b * Lk * d.lddv + h * 32, d.lddv,
                                kt * 32, Lk, lane, dscale);
        }
    } else {
        // ---------------- wave 7: K^T operands of every key tile, hardware-transposed out of the K image and kept for the
        // whole kernel; then dQ of query tile qt from the staged dS^T tiles
        bf16x8v kT[CW][2];
#pragma unroll
        for (int kt = 0; kt < CW; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) kT[kt][s2] = trfrag(stg, RS, min(kt, nkt - 1) * 32 + 16 * s2, 0, lane);
        __syncthreads();
        uint16_t* dqg = reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * 32;
        for (int qt = 0; qt < nqt; ++qt) {
            __syncthreads();
            f32x16 dQt[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[0][r] = 0.f;
            const char* buf = stg + (qt & 1) * CW * TILE;
#pragma unroll
            for (int kt = 0; kt < CW; ++kt) {
                if (kt < nkt) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        dQt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[kt][s2], trfrag(buf + kt * TILE, TS, 16 * s2, 0, lane), dQt[0], 0, 0, 0);
                }
            }
            store_tile_T<32, 1>(sc7, RS, dQt, dqg, d.lddq, qt * 32, Lq, lane, osc_dk);
        }
    }
}

size_t bwd_fast_lds(int Lq, int Lk) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    return (size_t)2 * LqP * B_RS + (size_t)2 * LqP * 4 + (size_t)LkP * 4 + (size_t)2 * B_CW * B_TILE + (size_t)32 * B_RS + 64;
}

int opt_in(const void* kern, size_t bytes) {
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

#ifdef MMFM_ATTN_STAMP
extern "C" int mmfm_attn_probe_read(unsigned long long* host16, int reset) {
    (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(mmfm_attn_probe_acc), 128);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(mmfm_attn_probe_acc), z, 128); }
    return 0;
}
#endif

extern "C" int64_t mmfm_attn_dropmask_bytes(int B, int heads, int Lq, int Lk) {
    return (int64_t)B * heads * ((Lq + 31) / 32) * ((Lk + 31) / 32) * 128;
}

// Shapes the fast pair takes (forward and backward decide alike: shape, flags and the presence of the mask buffer only).
// Returns -1000 when the general kernels must run.
int mmfm_attn_fast_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st) {
    static const bool off = [] { const char* e = getenv("MMFM_ATTN_FAST"); return e && atoi(e) == 0; }();
    if (off || d.dh != 32 || (d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP))) return -1000;
    const int nqt = (d.Lq + 31) / 32, nkt = (d.Lk + 31) / 32;
    if (d.Lq % 8 || d.Lk % 8 || nqt > 8 || nkt > B_CW || nkt > nqt) return -1000;
    const bool drop = d.drop_p.p > 0.f && d.drop_p.state != nullptr;
    if (drop && (d.drop_mask == nullptr || d.drop_p.p >= 1.f)) return -1000;
    const bool al = d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 && (uintptr_t)d.q % 16 == 0 &&
                    (uintptr_t)d.k % 16 == 0 && (uintptr_t)d.v % 16 == 0 && (uintptr_t)d.o % 16 == 0 && (uintptr_t)d.drop_mask % 16 == 0;
    // the backward's alignment is part of the SHARED decision: a forward that ran here must find its backward here too
    const bool alb = !backward || (d.lddo % 8 == 0 && d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0 && (uintptr_t)d.d_o % 16 == 0 &&
                                   (uintptr_t)d.dq % 16 == 0 && (uintptr_t)d.dk % 16 == 0 && (uintptr_t)d.dv % 16 == 0);
    if (!al) return -1000;
    if (!alb) {
        if (drop) return mmfm_set_error(-1, "mmfm_attn_bwd(bf16, mask): gradient tensors must be 16-byte aligned with leading dims %% 8 == 0 "
                                            "(the forward of this call drew its dropout from drop_mask; the general kernels would not)");
        return -1000;
    }
    if (drop) {
        const int64_t need = mmfm_attn_dropmask_bytes(d.B, d.heads, d.Lq, d.Lk);
        if (d.drop_mask_bytes < need) return mmfm_set_error(-1, "mmfm_attn: drop_mask holds %lld bytes, %lld needed", (long long)d.drop_mask_bytes, (long long)need);
    }
    const int grid = d.B * d.heads;
    if (!backward) {
        if (drop) {
            const uint32_t nwords = (uint32_t)(mmfm_attn_dropmask_bytes(d.B, d.heads, d.Lq, d.Lk) / 4);
            hipLaunchKernelGGL(attn_dropmask_kernel, dim3((nwords + 255) / 256), dim3(256), 0, st, reinterpret_cast<uint32_t*>(d.drop_mask), nwords, d.drop_p);
        }
        const int nw = nqt <= 4 ? 4 : (nqt == 7 ? 7 : (nqt <= 6 ? 6 : 8));
        const size_t lds = fwd_fast_lds(d.Lk, nw);
#define FWDF(NWV)                                                                                                   \
        {                                                                                                           \
            if (drop) { auto kern = attn_fwd_fast_kernel<NWV, true>; if (int rc = opt_in(reinterpret_cast<const void*>(kern), lds)) return rc; \
                        hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, d); }                         \
            else { auto kern = attn_fwd_fast_kernel<NWV, false>; if (int rc = opt_in(reinterpret_cast<const void*>(kern), lds)) return rc; \
                   hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, d); }                              \
        }
        if (nw == 4) FWDF(4) else if (nw == 6) FWDF(6) else if (nw == 7) FWDF(7) else FWDF(8)
#undef FWDF
        MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16, mask)");
        return 0;
    }
    const size_t lds = bwd_fast_lds(d.Lq, d.Lk);
    if (drop) { auto kern = attn_bwd_fast_kernel<true>; if (int rc = opt_in(reinterpret_cast<const void*>(kern), lds)) return rc;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(B_NW * 64), lds, st, d); }
    else { auto kern = attn_bwd_fast_kernel<false>; if (int rc = opt_in(reinterpret_cast<const void*>(kern), lds)) return rc;
           hipLaunchKernelGGL(kern, dim3(grid), dim3(B_NW * 64), lds, st, d); }
    MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16, mask)");
    return 0;
}
