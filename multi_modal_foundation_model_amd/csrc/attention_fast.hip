// bf16 attention, dh = 32, heads of up to 224 keys / 256 queries (the d_model-256 configurations: L = 200): the straight-line
// forward and single-pass backward that the step spends its attention time in.  Same contract as attention_bf16.hip
// (masks: key padding and DIAG; CAUSAL / SEP stay with the general kernels), reference: mm_utils.py:97-152.
//
// What changed against the general kernels, and why (round-2 counters: both were VALU-issue bound at 3,780 / 2,660 vector
// instructions per wave, MFMA pipe 7 % busy):
//   * attention-probability dropout keeps the counter hash of the general kernels (attn_common.h Drop16: one mix per PAIR of keys
//     under a per-(batch, head, query) row key, the same decisions bit for bit, so the general single-pass backward pairs with this
//     forward); both 16-bit fields are compared without extraction (word-select compares) and the selects run on fp32 values
//     before the PACKED bf16 conversion.  Measured and dropped on the way: a precomputed keep-bit mask
//     (generator kernel + scalar-loaded lane masks: the generator costs what the in-place hash costs - integer VALU issues at
//     4 cycles per wave instruction here - and the mask loads put a memory wait into every tile: 293 + 475 us against
//     264 + 435 us) and decisions taken from an int8 MFMA product of per-query / per-key random vectors (exact in both
//     orientations, 2 VALU instructions per element instead of 6, but 258 + 465 us: the two dependent MFMAs sit at the head
//     of every tile of the barrier-synchronised backward).
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

constexpr float LAZY_THR = 6.f;       // log2 units: probabilities stay below 2^6 between rescales

#ifdef MMFM_ATTN_STAMP
// diagnostic build only (scripts/probe/build_attn_stamp.sh): shader cycles per phase, one row of 6 per wave (plain stores: an
// atomic per wave on six shared words made the stamped kernel 25x slower and the shares meaningless), summed on the host
constexpr int APROBE_ROWS = 8192 * 8;
__device__ unsigned long long mmfm_attn_probe_acc[APROBE_ROWS * 6];
#define ASTAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_a[6] = {0, 0, 0, 0, 0, 0}
#define ASTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                       __builtin_amdgcn_sched_barrier(0); st_a[i] += n_ - st_t; st_t = n_; } while (0)
#define ASTAMP_FLUSH(base) do { if ((threadIdx.x & 63) == 0) { const int row_ = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % APROBE_ROWS; \
                                for (int i_ = 0; i_ < 6; ++i_) mmfm_attn_probe_acc[row_ * 6 + i_] = st_a[i_]; } } while (0)
#else
#define ASTAMP_DECL
#define ASTAMP(i)
#define ASTAMP_FLUSH(base)
#endif

__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// ---------------------------------------------------------------------------------------------- forward
// One key tile (32 keys x 32 queries, lane = query): online softmax on the scores `st` (bias already inside), dropout,
// O^T += V^T P^T.  G = 8-key groups of the tile that exist (registers 4g .. 4g+3).  jx = (pair index of the tile's first key of
// this lane half) ^ row key A, kb = row key B: register pair (r, r+1) = keys (kt*32 + mrow(r, kh), +1) = pair index
// (16 kt + 2 kh) | (mrow(r, 0) >> 1) - disjoint bits, so each pair costs one XOR with a literal (as in attention_bf16.hip).
template <int G, bool DROP>
__device__ __forceinline__ void fwd_tile(const f32x16& st, float c2, float& m_run, float& l_run, f32x16& acc, const Drop16& dp, uint32_t jx,
                                         uint32_t kb, const char* Vs, int kt, int lane) {
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
    for (int r = 0; r < 16; r += 2) {
        if (r < 4 * G) {
            float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -m_use));
            float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r + 1], c2, -m_use));
            ps += p0 + p1;
            if (DROP) {
                const uint32_t hsh = dp.hash(jx ^ (uint32_t)(mrow(r, 0) >> 1), kb);
                p0 = (uint16_t)hsh >= (uint16_t)dp.t16 ? p0 : 0.f;       // the 1/(1-p) factor rides on the final normalisation
                p1 = (hsh >> 16) >= dp.t16 ? p1 : 0.f;
                // opaque to the optimiser: otherwise hipcc converts every probability to bf16 on its own, selects on the 16-bit values
                // and permutes the halves together (16 cvt + 16 select + 8 perm per tile instead of 16 select + 8 packed cvt)
                asm volatile("" : "+v"(p0), "+v"(p1));
            }
            pd[r] = p0;
            pd[r + 1] = p1;
        } else {
            pd[r] = 0.f;
            pd[r + 1] = 0.f;
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
    // ONE memory round trip for the whole prologue: every global load of the workgroup (K and V chunks, the key-padding bytes, the
    // wave's own Q rows) is issued before the first wait.  (The generic loader - a load followed by its LDS store per loop trip - made
    // five to six DEPENDENT round trips of ~1.5 us each: half of a wave's 12 us life, see scripts/probe/attn_stamp.py.)
    // Chunk c = t + NT j (j = 0, 1; LkP * 4 <= 2 NT for every NW the launcher picks): row c >> 2, 16-B column c & 3; rows >= Lk are
    // zero-filled (a NaN bit pattern left in LDS would survive the -inf bias / the zero probability).
    const int qt = wave;
    const int q0 = qt * 32, q = q0 + l31;
    uint4 kc[2], vc[2], qv[2];
    uint8_t kpv = 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t + NT * j, row = c >> 2, col = c & 3;
        kc[j] = make_uint4(0u, 0u, 0u, 0u); vc[j] = kc[j];
        if (row < Lk) {
            kc[j] = *reinterpret_cast<const uint4*>(kg + (size_t)row * d.ldk + 8 * col);
            vc[j] = *reinterpret_cast<const uint4*>(vg + (size_t)row * d.ldv + 8 * col);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qv[ks] = make_uint4(0u, 0u, 0u, 0u);
        if (qt < nqt && q < Lq) qv[ks] = *reinterpret_cast<const uint4*>(qg + (size_t)q * d.ldq + ks * 16 + 8 * kh);
    }
    if (t < Lk && d.keypad != nullptr) kpv = d.keypad[(size_t)b * Lk + t];          // NT >= LkP for every launch shape
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t + NT * j, row = c >> 2, col = c & 3;
        if (row < LkP) {
            *reinterpret_cast<uint4*>(Ks + row * F_KRS + col * 16) = kc[j];
            *reinterpret_cast<uint4*>(Vs + row * F_VRS + col * 16) = vc[j];
        }
    }
    const bool kok = t < Lk && kpv != 0;
    if (t < LkP) kbias[t] = kok ? 0.f : -INFINITY;
    const int pad = (t < Lk && !kok) ? 1 : 0;
    ASTAMP(0);
    const int wv = __any(pad) ? 1 : 0;                  // all 64 lanes vote before any divergence
    if (lane == 0) wflag[wave] = wv;
    __syncthreads();                                    // also: the images are complete
    int anypad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) anypad |= wflag[w];
    if (qt >= nqt) return;
    ASTAMP(1);

    bf16x8v qf[2];
    qf[0] = __builtin_bit_cast(bf16x8v, qv[0]);
    qf[1] = __builtin_bit_cast(bf16x8v, qv[1]);
    const float c2 = d.scale * LOG2E;
    const Drop16 dp = drop16_init(d.drop_p);
    uint32_t ka = 0, kb = 0;                                    // dropout row keys of this lane's query
    if (DROP) dp.rowkeys((uint32_t)bh_ * (uint32_t)Lq + (uint32_t)q, ka, kb);
    const bool fixdiag = anypad && (d.flags & MMFM_ATTN_DIAG);

    // S^T tile: rows = keys (registers), lane = query; the accumulator starts at the key bias
    auto score = [&](int kt) {
        // rare: padded keys in the head and `eye |` (a query always sees itself): the diagonal tile starts from zero and takes the
        // key bias after the product, except on the diagonal
        const bool dfix = fixdiag && kt == qt;
        f32x16 a;
        if (!dfix) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 kb4 = *reinterpret_cast<const float4*>(kbias + kt * 32 + 8 * g + 4 * kh);
                a[4 * g + 0] = kb4.x; a[4 * g + 1] = kb4.y; a[4 * g + 2] = kb4.z; a[4 * g + 3] = kb4.w;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kt * 32 + l31) * F_KRS + ks * 32 + kh * 16), qf[ks], a, 0, 0, 0);
        if (dfix) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] += (mrow(r, kh) == l31) ? 0.f : kbias[kt * 32 + mrow(r, kh)];
        }
        return a;
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int gtail = ((Lk - (nkt - 1) * 32) + 7) >> 3;             // 1..4 valid groups in the last key tile (Lk % 8 == 0: launcher)
    auto tile = [&](int kt, const f32x16& st) {
        const uint32_t jx = (uint32_t)(16 * kt + 2 * kh) ^ ka;
        if (kt == nkt - 1 && gtail != 4) {
            if (gtail == 1) fwd_tile<1, DROP>(st, c2, m_run, l_run, acc, dp, jx, kb, Vs, kt, lane);
            else if (gtail == 2) fwd_tile<2, DROP>(st, c2, m_run, l_run, acc, dp, jx, kb, Vs, kt, lane);
            else fwd_tile<3, DROP>(st, c2, m_run, l_run, acc, dp, jx, kb, Vs, kt, lane);
        } else {
            fwd_tile<4, DROP>(st, c2, m_run, l_run, acc, dp, jx, kb, Vs, kt, lane);
        }
    };
    // two score tiles in flight, ping-pong: tile kt+1's MFMAs are issued before the element-wise work of tile kt
    f32x16 s0 = score(0), s1;
    ASTAMP(2);
    for (int kt = 0; kt < nkt; kt += 2) {
        const bool has1 = kt + 1 < nkt;
        if (has1) s1 = score(kt + 1);
        tile(kt, s0);
        if (has1) {
            if (kt + 2 < nkt) s0 = score(kt + 2);
            tile(kt + 1, s1);
        }
    }
    ASTAMP(3);
    const Drop dout = drop_init(d.drop_o);
    const float dscale = DROP ? dp.scale : 1.f;
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

// one query tile of a compute wave (lane = key, rows = queries), no attention dropout (with it the general single-pass kernel of
// attention_bf16.hip runs: see the header).  GRP = 8-query groups of the tile that exist (registers 4g .. 4g+3 = queries
// 8g + 4 kh + 0..3).
template <int GRP>
__device__ __forceinline__ void bwd_tile(const f32x16& s, const f32x16& dpv, float c2, const float2* ldl, const char* As, const char* Bs, char* slot,
                                         int qt, int kh, int lane, f32x16& dKt, f32x16& dVt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (2 * s2 >= GRP) continue;
        const bool both = GRP > 2 * s2 + 1;
        float pd[8], ds[8];
        const float2* lq = ldl + qt * 32 + 16 * s2 + 4 * kh;          // (lse * log2 e, delta) of the half tile's queries
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (e >= 4 && !both) { pd[e] = 0.f; ds[e] = 0.f; continue; }
            const int r = 8 * s2 + e;
            const float2 ld = lq[(e & 3) + 8 * (e >> 2)];
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -ld.x));
            pd[e] = p;
            ds[e] = p * (dpv[r] - ld.y);
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

__global__ __launch_bounds__(B_NW * 64, 4) void attn_bwd_fast_kernel(const mmfm_attn_desc d) {
    constexpr int NW = B_NW, CW = B_CW, TS = B_TS, TILE = B_TILE, RS = B_RS, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    char* As = smem;                                  // Q image
    char* Bs = As + LqP * RS;                         // dO image (output dropout applied)
    float2* ldl = reinterpret_cast<float2*>(Bs + LqP * RS);   // per query: (lse * log2 e, delta / dropout scale)
    float* kbias = reinterpret_cast<float*>(ldl + LqP);
    char* stg = reinterpret_cast<char*>(kbias + LkP); // [2][CW][32 keys x TS] dS^T tiles; first the K image (prologue only)
    char* sc7 = stg + 2 * CW * TILE;                  // [32 x RS] dQ transpose tile of wave 7
    int* wflag = reinterpret_cast<int*>(sc7 + 32 * RS);
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * 32;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * 32;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * 32;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * 32;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * 32;
    // the softmax scale stays out of the per-element algebra: dK / dQ are scaled once, when stored
    const float osc_dk = d.scale;
    const Drop dout = drop_init(d.drop_o);

    // ONE memory round trip for the prologue (see the forward): all of Q, K, d_o, o of the head, the LSE row, the key-padding bytes
    // and the compute waves' own K / V operand rows are requested before the first wait.  Chunk c = t + 512 j (j = 0, 1):
    // row c >> 2, 16-B column c & 3 (LkP <= LqP <= 256 -> LqP * 4 <= 1024).
    const int kt_own = wave < CW ? wave : 0, key_own = kt_own * 32 + l31;
    uint4 qc[2], kc[2], gc[2], oc[2], kfv[2], vfv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t + NT * j, row = c >> 2, col = c & 3;
        qc[j] = make_uint4(0u, 0u, 0u, 0u); kc[j] = qc[j]; gc[j] = qc[j]; oc[j] = qc[j];
        if (row < Lq) {
            qc[j] = *reinterpret_cast<const uint4*>(qg + (size_t)row * d.ldq + 8 * col);
            gc[j] = *reinterpret_cast<const uint4*>(dog + (size_t)row * d.lddo + 8 * col);
            oc[j] = *reinterpret_cast<const uint4*>(og + (size_t)row * d.ldo + 8 * col);
        }
        if (row < Lk) kc[j] = *reinterpret_cast<const uint4*>(kg + (size_t)row * d.ldk + 8 * col);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kfv[ks] = make_uint4(0u, 0u, 0u, 0u); vfv[ks] = kfv[ks];
        if (wave < CW && key_own < Lk) {
            kfv[ks] = *reinterpret_cast<const uint4*>(kg + (size_t)key_own * d.ldk + ks * 16 + 8 * kh);
            vfv[ks] = *reinterpret_cast<const uint4*>(vg + (size_t)key_own * d.ldv + ks * 16 + 8 * kh);
        }
    }
    float lsev = 0.f;
    uint8_t kpv = 1;
    if (t < Lq) lsev = d.lse[(size_t)bh_ * Lq + t];                               // NT = 512 >= LqP >= LkP
    if (t < Lk && d.keypad != nullptr) kpv = d.keypad[(size_t)b * Lk + t];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t + NT * j, row = c >> 2, col = c & 3;
        if (row < LqP) {
            *reinterpret_cast<uint4*>(As + row * RS + col * 16) = qc[j];
            // dO = dropout'(d_o) as bf16;  delta = rowsum(d_o * o) / dropout scale (the four 16-B columns of a row sit in neighbouring lanes)
            const uint32_t gw[4] = {gc[j].x, gc[j].y, gc[j].z, gc[j].w}, ow[4] = {oc[j].x, oc[j].y, oc[j].z, oc[j].w};
            const uint64_t base = ((uint64_t)b * Lq + (uint64_t)row) * (uint64_t)(d.heads * 32) + (uint64_t)(h * 32 + 8 * col);
            float part = 0.f, gd[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float g0 = __uint_as_float(gw[i] << 16), g1 = __uint_as_float(gw[i] & 0xffff0000u);
                const float o0 = __uint_as_float(ow[i] << 16), o1 = __uint_as_float(ow[i] & 0xffff0000u);
                part += g0 * o0 + g1 * o1;
                if (dout.on()) dout.apply2(g0, g1, base + 2 * i);
                gd[2 * i] = g0;
                gd[2 * i + 1] = g1;
            }
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            if (col == 0) ldl[row].y = part;
            *reinterpret_cast<uint4*>(Bs + row * RS + col * 16) = __builtin_bit_cast(uint4, pack8(gd));
        }
        if (row < LkP) *reinterpret_cast<uint4*>(stg + row * RS + col * 16) = kc[j];     // K image, for wave 7's K^T operands only
    }
    if (t < LqP) {
        *reinterpret_cast<uint4*>(As + t * RS + 64) = make_uint4(0u, 0u, 0u, 0u);        // the 16-B row pads (read by nobody, kept finite)
        *reinterpret_cast<uint4*>(Bs + t * RS + 64) = make_uint4(0u, 0u, 0u, 0u);
        ldl[t].x = lsev * LOG2E;
    }
    const bool kok = t < Lk && kpv != 0;
    if (t < LkP) kbias[t] = kok ? 0.f : -INFINITY;
    const int pad = (t < Lk && !kok) ? 1 : 0;
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
            kfr[ks] = __builtin_bit_cast(bf16x8v, kfv[ks]);
            vfr[ks] = __builtin_bit_cast(bf16x8v, vfv[ks]);
        }
        const float kbv = active ? kbias[key] : 0.f;
        f32x16 s, dpv;
        auto scoresA = [&](int qt) {
            // rare: padded keys and `eye |`: S[q][q] is allowed even when key q is padded - the diagonal tile takes the bias afterwards
            const bool dfix = fixdiag && qt == kt;
            const float s_init = dfix ? 0.f : kbv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = s_init; dpv[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = (qt * 32 + l31) * RS + ks * 32 + kh * 16;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(As, off), kfr[ks], s, 0, 0, 0);        // S[q][key]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bs, off), vfr[ks], dpv, 0, 0, 0);    // dP[q][key]
            }
            if (dfix) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] += (mrow(r, kh) == l31) ? 0.f : kbv;
            }
        };
        const int gtail = ((Lq - (nqt - 1) * 32) + 7) >> 3;            // valid 8-query groups of the last query tile (Lq % 8 == 0)
        __syncthreads();                               // wave 7 has its K^T operands: the K image is dead, the staging slots free
        for (int qt = 0; qt < nqt; ++qt) {
            if (active) {
                scoresA(qt);
                char* slot = stg + ((qt & 1) * CW + wave) * TILE + l31 * TS;
                if (qt == nqt - 1 && gtail != 4) {
                    if (gtail == 1) bwd_tile<1>(s, dpv, c2, ldl, As, Bs, slot, qt, kh, lane, dKt, dVt);
                    else if (gtail == 2) bwd_tile<2>(s, dpv, c2, ldl, As, Bs, slot, qt, kh, lane, dKt, dVt);
                    else bwd_tile<3>(s, dpv, c2, ldl, As, Bs, slot, qt, kh, lane, dKt, dVt);
                } else {
                    bwd_tile<4>(s, dpv, c2, ldl, As, Bs, slot, qt, kh, lane, dKt, dVt);
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
                                kt * 32, Lk, lane, 1.f);
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
extern "C" int mmfm_attn_probe_read(unsigned long long* host6, int reset) {
    static unsigned long long* h = new unsigned long long[APROBE_ROWS * 6];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(mmfm_attn_probe_acc), sizeof(unsigned long long) * APROBE_ROWS * 6);
    for (int i = 0; i < 6; ++i) host6[i] = 0;
    for (int r = 0; r < APROBE_ROWS; ++r) for (int i = 0; i < 6; ++i) host6[i] += h[r * 6 + i];
    if (reset) { for (int i = 0; i < APROBE_ROWS * 6; ++i) h[i] = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(mmfm_attn_probe_acc), h, sizeof(unsigned long long) * APROBE_ROWS * 6); }
    return 0;
}
#endif

// Shapes the fast kernels take.  Returns -1000 when the general kernels must run.
int mmfm_attn_fast_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st) {
    static const bool off = [] { const char* e = getenv("MMFM_ATTN_FAST"); return e && atoi(e) == 0; }();
    if (off || d.dh != 32 || (d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP))) return -1000;
    const int nqt = (d.Lq + 31) / 32, nkt = (d.Lk + 31) / 32;
    if (d.Lq % 8 || d.Lk % 8 || nqt > 8 || nkt > B_CW || nkt > nqt) return -1000;
    const bool drop = d.drop_p.p > 0.f && d.drop_p.state != nullptr;
    if (drop && d.drop_p.p >= 1.f) return -1000;
    const bool al = d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 && (uintptr_t)d.q % 16 == 0 &&
                    (uintptr_t)d.k % 16 == 0 && (uintptr_t)d.v % 16 == 0 && (uintptr_t)d.o % 16 == 0;
    const bool alb = !backward || (d.lddo % 8 == 0 && d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0 && (uintptr_t)d.d_o % 16 == 0 &&
                                   (uintptr_t)d.dq % 16 == 0 && (uintptr_t)d.dk % 16 == 0 && (uintptr_t)d.dv % 16 == 0);
    if (!al) return -1000;
    // every untiled bf16 kernel takes the same dropout decisions (attn_common.h Drop16), so the two directions are picked
    // independently: with attention dropout the backward is the general single-pass kernel (its hash evaluation shares the work
    // between neighbouring key lanes; ported here it spilled, and the int8-MFMA alternative measured 6 % slower in the step)
    if (backward && (drop || !alb)) return -1000;
    const int grid = d.B * d.heads;
    if (!backward) {
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
        MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16, dh 32)");
        return 0;
    }
    const size_t lds = bwd_fast_lds(d.Lq, d.Lk);
    auto kern = attn_bwd_fast_kernel;
    if (int rc = opt_in(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(B_NW * 64), lds, st, d);
    MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16, dh 32)");
    return 0;
}
