// bf16 attention, dh = 32, heads of up to 224 keys / 256 queries (the d_model-256 configurations: L = 200): the straight-line
// forward and the single-pass backward that the step spends its attention time in.  Same contract as attention_bf16.hip
// (masks: key padding and DIAG; CAUSAL / SEP stay with the general kernels), reference: mm_utils.py:97-152.
//
// Round 4: attention-probability dropout no longer hashes inside these kernels.  Both were VALU-issue bound and the counter
// hash was 27 % of the forward's vector instructions (479 of 1,768) and ~60 us of every backward launch.  Now a generator kernel
// draws the keep decisions ONCE per launch as bit tiles (mmfm_attn_desc.keepbits, layout below) - 32 decisions per lane-word by
// the binary-expansion trick, i.e. at the full width of the vector unit instead of two decisions per 7-instruction hash - and
//   * the forward reads a tile's 16 lane masks with two scalar loads (s_load_dwordx16) and applies them with one
//     v_cndmask_b32 per probability: no compare, no hash, no vector register;
//   * the backward lane (= key) fetches its seven words with the prologue loads and expands a bit per element.
// Without a keepbits buffer, calls with attention dropout go to the general kernels (attention_bf16.hip), which hash.
//
// Keep-bit layout: uint32 words [b * heads + h][qt][kt][32]; word w = 2 r + kh of tile (qt, kt) belongs to key
// 32 kt + mrow(r, kh) (the key that accumulator register r holds in lane half kh when the lane is the query), bit j of it to
// query 32 qt + j.  So the forward's lane mask for register r is the 64-bit pair (words 2 r, 2 r + 1) as it lies in memory, and
// a backward lane (key 32 kt + l, l = mrow(r, kh)) needs word 2 r + kh of each of its query tiles.
//
// Kept from round 3: masks ride on the MFMA (the score accumulator starts at bias[key], 0 or -inf), the tile body is instantiated
// per number of valid 8-row groups, lazy rescaling of the running maximum, one query tile per wave with as many
// waves as the head has tiles.
#include "attn_common.h"
#include <algorithm>
#include <stdlib.h>

using namespace attn;

namespace {

constexpr int KEEP_BITS = 10;         // the keep probability is honoured to 2^-10 (mmfm_attn_keep_prob)

// ---------------------------------------------------------------------------------------------- keep-bit generator
// One thread per 32-decision word.  Bernoulli(keep) bits from uniform words by the binary expansion keep = 0.b1 b2 ... bn:
// walking the bits from the least significant, r = b ? (u | r) : (u & r) halves the distance to the next digit each time, so after
// the walk every bit of r is set with probability keep (to 2^-n), independently per bit position.  n uniform words per 32 decisions
// instead of 16 hashes: 2.4 vector instructions per decision here against 5.5 in the attention kernels' own lanes.
struct KeepArgs {
    uint32_t* bits;
    const uint32_t* state;
    uint32_t site, thresh;        // thresh = round(keep * 2^KEEP_BITS) in [1, 2^KEEP_BITS - 1]
    uint32_t nwords;
    int nkt, Lk;
};
__global__ __launch_bounds__(256) void attn_keepbits_kernel(const KeepArgs a) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= a.nwords) return;
    // words of keys beyond the head's last key (the padded part of the last key tile) are never looked at: 11 % of them at L = 200
    const uint32_t w = gid & 31u, kt = (gid >> 5) % (uint32_t)a.nkt;
    if ((int)(32u * kt + ((w >> 1) & 3u) + 8u * (w >> 3) + 4u * (w & 1u)) >= a.Lk) return;
    const uint32_t k0 = mix32(a.state[0] + a.site * 0x9E3779B9u), k1 = mix32(a.state[1] ^ (a.site * 0x85EBCA6Bu + 0xC2B2AE35u));
    const uint32_t s = mix32(gid ^ k0);
    const uint32_t kb = k1 + __umul24(s >> 24, 0x9E3779u);       // v_mul_u32_u24 sees bits 0..23 only: the top byte enters here
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < KEEP_BITS; ++j) {
        if ((a.thresh & ((2u << j) - 1u)) == 0) continue;        // trailing zero digits leave r = 0 (uniform branch)
        uint32_t h = __umul24(s + (uint32_t)j * 0x3C6EF35Fu, 0x7FEB35u) + kb;
        h ^= h >> 13;
        h = __umul24(h, 0x46CA6Bu);
        h ^= h >> 16;
        r = ((a.thresh >> j) & 1u) ? (h | r) : (h & r);
    }
    a.bits[gid] = r;
}

// ---------------------------------------------------------------------------------------------- forward
// One key tile (32 keys x 32 queries, lane = query): probabilities of the scores `st` (key bias already inside) against the row's
// reference exponent, dropout, O^T += V^T P^T.  G = 8-key groups of the tile that exist (registers 4g .. 4g+3), mk = the tile's 16
// lane masks.  nm = -(reference): see the kernel for the two modes.
template <int G, bool DROP>
__device__ __forceinline__ void fwd_tile(const f32x16& st, float c2, float nm, float& l_run, f32x16& acc, const Masks16& mk, const char* Vs,
                                         int kt, int lane) {
    uint32_t pk[8];                                                                   // the sixteen probabilities, packed as they are made
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        if (r < 4 * G) {
            float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, nm));        // -inf scores (padded keys): exp2(-inf) = 0
            float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r + 1], c2, nm));
            ps0 = v_add(ps0, p0);
            ps1 = v_add(ps1, p1);
            if (DROP) {                                                              // the 1 / keep factor rides on the final normalisation
                p0 = v_keep(p0, mk.m[r]);
                p1 = v_keep(p1, mk.m[r + 1]);
            }
            // Opaque to the optimiser, no instruction inside.  (i) Without it every probability is converted to bf16 on its own, selected
            // as a 16-bit value and the halves permuted together (16 cvt + 16 select + 8 perm per tile instead of 16 select + 8 packed cvt).
            // (ii) The running sums pass through it too: left alone, the row-sum adds (a dependent chain) sink to the end of the tile, all
            // sixteen exponentials stay alive for them and the straight-line kernel spills 50 registers.
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(ps0), "+v"(ps1));
            pk[r >> 1] = pack2(p0, p1);
        } else {
            pk[r >> 1] = 0u;
        }
    }
    l_run = v_add(l_run, v_add(ps0, ps1));
    const uint4 lo = make_uint4(pk[0], pk[1], pk[2], pk[3]), hi = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, 64, kt * 32, 0, lane), __builtin_bit_cast(bf16x8v, lo), acc, 0, 0, 0);
    if (G > 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag(Vs, 64, kt * 32 + 16, 0, lane), __builtin_bit_cast(bf16x8v, hi), acc, 0, 0, 0);
}
template <int G>
__device__ __forceinline__ float tile_max(const f32x16& st) {
    float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
    for (int r = 3; r < 4 * G; ++r) mx = fmaxf(mx, st[r]);
    return xhalf_max(mx);                                      // identical in both lane halves
}

constexpr int F_KRS = 80, F_VRS = 64, F_ORS = 80, F_MAXKT = 7;
constexpr float F_OVERFLOW = 1.2676506e30f;                    // 2^100: a row sum beyond it sends the wave to the exact pass

// NKT = key tiles of the head when known at compile time (7: the L = 200 / 224 step shapes), 0 = read from the descriptor.
template <int NW, bool DROP, int NKT>
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_fast_kernel(const mmfm_attn_desc d, const float keep_scale) {
    constexpr int NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LkP = NKT ? NKT * 32 : (Lk + 31) & ~31;
    const int nqt = (Lq + 31) >> 5, nkt = NKT ? NKT : LkP >> 5;
    char* Ks = smem;
    char* Vs = Ks + LkP * F_KRS;
    float* kbias = reinterpret_cast<float*>(Vs + LkP * F_VRS);
    char* ost = reinterpret_cast<char*>(kbias + LkP);
    int* wflag = reinterpret_cast<int*>(ost + NW * 32 * F_ORS);          // per-wave "a key of this head is padded" votes (no static LDS:
                                                                         // it would shift the 16-B aligned carve-up, Guideline 17)
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * 32;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * 32;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * 32;

    // ONE memory round trip for the whole prologue: every global load of the workgroup (K and V chunks, the key-padding bytes, the
    // wave's own Q rows) is issued before the first wait.  Chunk c = t + NT j (j = 0, 1; LkP * 4 <= 2 NT for every NW the launcher
    // picks): row c >> 2, 16-B column c & 3; rows >= Lk are zero-filled (a NaN bit pattern left in LDS would survive the -inf bias /
    // the zero probability).
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
    const int wv = __any(pad) ? 1 : 0;                  // all 64 lanes vote before any divergence
    if (lane == 0) wflag[wave] = wv;
    __syncthreads();                                    // also: the images are complete
    int anypad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) anypad |= wflag[w];
    anypad = __builtin_amdgcn_readfirstlane(anypad);     // the same word in every lane: tell the compiler, or `dfix` is a divergent branch
    if (qt >= nqt) return;

    bf16x8v qf[2];
    qf[0] = __builtin_bit_cast(bf16x8v, qv[0]);
    qf[1] = __builtin_bit_cast(bf16x8v, qv[1]);
    const float c2 = d.scale * LOG2E;
    const bool fixdiag = anypad && (d.flags & MMFM_ATTN_DIAG);
    // the wave's keep-bit tiles: [bh][qt][kt] x 128 B, wave-uniform addresses
    const masks_ptr mkp = reinterpret_cast<masks_ptr>(reinterpret_cast<uintptr_t>(d.keepbits)) + ((size_t)bh_ * nqt + qt) * nkt;

    // S^T tile: rows = keys (registers), lane = query; the accumulator starts at the key bias
    auto score = [&](int kt) {
        // rare: padded keys in the head and `eye |` (a query always sees itself): the diagonal tile drops the key bias on the diagonal
        const bool dfix = fixdiag && kt == qt;
        f32x16 a;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 kb4 = *reinterpret_cast<const float4*>(kbias + kt * 32 + 8 * g + 4 * kh);
            a[4 * g + 0] = kb4.x; a[4 * g + 1] = kb4.y; a[4 * g + 2] = kb4.z; a[4 * g + 3] = kb4.w;
        }
        if (dfix) {
            asm volatile("" ::: "memory");               // keeps this a branch: as selects it costs every tile 16 instructions and 32 SGPRs
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = (mrow(r, kh) == l31) ? 0.f : a[r];
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Ks, (kt * 32 + l31) * F_KRS + ks * 32 + kh * 16), qf[ks], a, 0, 0, 0);
        return a;
    };
    const int gtail = ((Lk - (nkt - 1) * 32) + 7) >> 3;             // 1..4 valid groups in the last key tile (Lk % 8 == 0: launcher)
    float m_ref, l_run = 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto tile = [&](int kt, const f32x16& st, const Masks16& mk) {
        if (kt == nkt - 1 && gtail != 4) {
            if (gtail == 1) fwd_tile<1, DROP>(st, c2, -m_ref, l_run, acc, mk, Vs, kt, lane);
            else if (gtail == 2) fwd_tile<2, DROP>(st, c2, -m_ref, l_run, acc, mk, Vs, kt, lane);
            else fwd_tile<3, DROP>(st, c2, -m_ref, l_run, acc, mk, Vs, kt, lane);
        } else {
            fwd_tile<4, DROP>(st, c2, -m_ref, l_run, acc, mk, Vs, kt, lane);
        }
    };
    // FAST PASS.  The reference exponent of a row is the maximum of its FIRST key tile and stays put: no running maximum, no
    // rescaling of the output tile, nothing per tile but the probabilities themselves (the element-wise work of this kernel is what
    // bounds it: 16 instructions per score in round 3, of which the max / rescale bookkeeping and the copies it caused were 4).
    // Later scores may exceed the reference: fp32 (and the bf16 operand of P.V) carry 2^127, the LSE below is exact for any
    // reference, and a row whose sum passes 2^100 - a score 69 above everything in the first 32 keys, or a first tile with no
    // allowed key - sends the whole wave through the exact pass below instead.
    // Straight-line code over the key tiles: two score tiles and two mask tiles in flight, ping-pong (tile kt+1's MFMAs and scalar
    // loads are issued before the element-wise work of tile kt).
    {
        Masks16 mk[2];
        f32x16 sc[2];
        if (DROP) mk[0] = ld_masks(mkp);
        sc[0] = score(0);
        m_ref = v_max((nkt == 1 && gtail != 4) ? (gtail == 1 ? tile_max<1>(sc[0]) : gtail == 2 ? tile_max<2>(sc[0]) : tile_max<3>(sc[0]))
                                                : tile_max<4>(sc[0]), -1e30f / c2) * c2;
#pragma unroll
        for (int kt = 0; kt < F_MAXKT; ++kt) {
            if (kt + 1 < nkt) {
                sc[(kt + 1) & 1] = score(kt + 1);
                if (DROP) mk[(kt + 1) & 1] = ld_masks(mkp + (kt + 1));
            }
            tile(kt, sc[kt & 1], mk[kt & 1]);
            if (kt + 1 >= nkt) break;
        }
    }
    float l_tot = xhalf_sum(l_run);
    if (__any(!(l_tot < F_OVERFLOW))) {
        // EXACT PASS (rare; also taken by NaN inputs, once): running maximum per tile, output tile rescaled every tile.
        float m_run = -1e30f;
        l_run = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            const f32x16 st = score(kt);
            Masks16 mk;
            if (DROP) mk = ld_masks(mkp + kt);
            const float mt = tile_max<4>(st) * c2;          // padded rows of a ragged last tile carry the -inf bias: no effect on the maximum
            const float m_new = v_max(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
            m_ref = m_run;
            tile(kt, st, mk);
        }
        l_tot = xhalf_sum(l_run);
    }
    // everything the epilogue needs from the descriptor is read HERE, from the kernel-argument segment: held across the tile loop these
    // pointers and strides, together with two tiles of lane masks (64 scalar registers), overflow the scalar file into vector lanes
    const mmfm_attn_desc* kd = (const mmfm_attn_desc*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kd));
    const Drop dout = drop_init(kd->drop_o);
    const float inv = keep_scale / l_tot;
    if (kh == 0 && q < Lq) kd->lse[(size_t)bh_ * Lq + q] = m_ref * LN2 + __logf(l_tot);
    // O^T (rows = d in registers, lane = query) -> bf16 rows [query][d] through the wave's staging tile, output dropout on the way
    char* tl = ost + wave * 32 * F_ORS;
    const int heads = kd->heads, ldo = kd->ldo;
    uint16_t* og = reinterpret_cast<uint16_t*>(kd->o) + (size_t)b * Lq * ldo + h * 32;
    const uint64_t base = ((uint64_t)b * Lq + (uint64_t)q) * (uint64_t)(heads * 32) + (uint64_t)(h * 32);
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
            *reinterpret_cast<uint4*>(og + (size_t)(q0 + row) * ldo + 8 * c) = *reinterpret_cast<const uint4*>(tl + row * F_ORS + c * 16);
    }
}

size_t fwd_fast_lds(int Lk, int nw) {
    const int LkP = (Lk + 31) & ~31;
    return (size_t)LkP * (F_KRS + F_VRS + 4) + (size_t)nw * 32 * F_ORS + 64;
}

// ---------------------------------------------------------------------------------------------- backward (single pass)
// Compute waves 0..6 own one 32-key tile each (lane = key; dK^T, dV^T in accumulators) and walk the query tiles, dropping their
// packed dS^T tile into an LDS staging slot; wave 7 turns the staged tiles of a query tile into dQ.  Fixed order, no atomics.
// Round 4:
//   * the Q / dO images are DENSE 64-byte rows with the 16-byte chunk index XOR-ed by (row >> 2) & 3: the row reads (ds_read_b128,
//     S and dP operands) and the transposed reads (ds_read_b64_tr_b16, dK^T / dV^T operands: 4 rows x 64 B = all 64 banks once) are
//     both conflict free; the 80-byte rows of round 3 cost the transposed reads a third cycle (SQ_LDS_BANK_CONFLICT was 35 % of
//     SQ_LDS_IDX_ACTIVE);
//   * dropout decisions come from the forward's keep bits: a lane (= key) loads its word of every query tile with the prologue
//     loads (<= 8 dwords) and expands one bit per element (v_bfe_i32 -> and);
//   * tried and removed: an LDS-flag hand-off in place of the per-tile workgroup barrier (a compute wave publishes "tile qt staged" in
//     its own word and only waits for wave 7 to have drained the slot it overwrites, wave 7 waits for the seven words) so that the
//     compute waves need not run in lock step: bit-identical results, 2-3 % SLOWER on two boxes (339 / 335 us against 332 / 324).
constexpr int B_NW = 8, B_CW = 7, B_TS = 80, B_TILE = 32 * B_TS, B_RS = 64, B_QRS = 80;

__device__ __forceinline__ int img_off(int row, int chunk) { return row * B_RS + ((chunk ^ ((row >> 2) & 3)) << 4); }
// transposed operand out of a swizzled dense image: element j = image[rbase + 8 (j >> 2) + 4 h + (j & 3)][c]   (rbase % 16 == 0)
__device__ __forceinline__ bf16x8v trfrag_sw(const char* S, int rbase, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int r0 = 4 * (g >> 1) + q;                                    // 0..7; the second half reads row r0 + 8
    const int chunk = 2 * (g & 1) + (p >> 1), byte = 8 * (p & 1);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + (rbase + r0) * B_RS + ((chunk ^ (g >> 1)) << 4) + byte));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + (rbase + r0 + 8) * B_RS + ((chunk ^ ((g >> 1) + 2)) << 4) + byte));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8v, v);
}

// one query tile of a compute wave (lane = key, rows = queries).  GRP = 8-query groups of the tile that exist (registers 4g .. 4g+3 =
// queries 8g + 4 kh + 0..3).  wsh = the lane's keep word of this query tile, shifted so that bit (r & 3) + 8 (r >> 2) is register r's.
template <int GRP, bool DROP>
__device__ __forceinline__ void bwd_tile(const f32x16& s, const f32x16& dpv, float c2, const float* lse2, const float* dl, uint32_t wsh,
                                         const char* As, const char* Bs, char* slot, int qt, int kh, int lane, f32x16& dKt, f32x16& dVt) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        if (2 * s2 >= GRP) continue;
        const bool both = GRP > 2 * s2 + 1;
        float pd[8], ds[8];
        const int qb = qt * 32 + 16 * s2 + 4 * kh;                    // the half tile's queries: qb + 0..3 and qb + 8 + 0..3
#pragma unroll
        for (int e4 = 0; e4 < 2; ++e4) {
            if (e4 == 1 && !both) {
#pragma unroll
                for (int e = 4; e < 8; ++e) { pd[e] = 0.f; ds[e] = 0.f; }
                continue;
            }
            const float4 l4 = *reinterpret_cast<const float4*>(lse2 + qb + 8 * e4), d4 = *reinterpret_cast<const float4*>(dl + qb + 8 * e4);
            const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * e4 + i, r = 8 * s2 + e;
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c2, -lq[i]));
                float pm = p;
                if (DROP) {
                    int m = __builtin_amdgcn_sbfe((int)wsh, (r & 3) + 8 * (r >> 2), 1);     // 0 or ~0: v_bfe_i32
                    asm volatile("" : "+v"(m));                        // opaque: else the optimiser turns bfe + and into test + compare + select
                    pm = __uint_as_float(__float_as_uint(p) & (uint32_t)m);
                }
                pd[e] = pm;                                            // kept probabilities (unscaled): dV's operand
                ds[e] = __builtin_fmaf(pm, dpv[r], -v_mul(p, dq_[i])); // dS / dropout scale = p (m dP - delta / scale)
            }
        }
        const bf16x8v pf = pack8(pd), sf = pack8(ds);
        // dS^T[key = lane][q]: elements 0..3 are queries 16*s2 + 4*kh + 0..3, elements 4..7 the same + 8
        const uint4 sw = __builtin_bit_cast(uint4, sf);
        *reinterpret_cast<uint2*>(slot + (16 * s2 + 4 * kh) * 2) = make_uint2(sw.x, sw.y);
        *reinterpret_cast<uint2*>(slot + (16 * s2 + 8 + 4 * kh) * 2) = make_uint2(sw.z, sw.w);
        dVt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_sw(Bs, qt * 32 + 16 * s2, lane), pf, dVt, 0, 0, 0);
        dKt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_sw(As, qt * 32 + 16 * s2, lane), sf, dKt, 0, 0, 0);
    }
    if (GRP <= 2) {                                    // the dQ wave reads whole tiles: the second half must not be stale
        *reinterpret_cast<uint2*>(slot + (16 + 4 * kh) * 2) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(slot + (24 + 4 * kh) * 2) = make_uint2(0u, 0u);
    }
}

// NQT = query tiles of the head when known at compile time (7: the L = 200 / 224 step shapes - the query-tile loop is then straight-line
// code and every LDS address in it a lane constant plus an immediate), 0 = read from the descriptor.
template <bool DROP, int NQT>
__global__ __launch_bounds__(B_NW * 64, 4) void attn_bwd_fast_kernel(const mmfm_attn_desc d, const float keep_scale) {
    constexpr int NW = B_NW, CW = B_CW, TS = B_TS, TILE = B_TILE, RS = B_RS, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, lane = t & 63, kh = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int bh_ = attn_xcd_remap((int)blockIdx.x, (int)gridDim.x, d.flags);
    const int b = bh_ / d.heads, h = bh_ % d.heads;
    const int Lq = d.Lq, Lk = d.Lk, LqP = NQT ? NQT * 32 : (Lq + 31) & ~31, LkP = NQT ? NQT * 32 : (Lk + 31) & ~31;    // NQT: the carve-up
    const int nqt = NQT ? NQT : LqP / 32, nkt = ((Lk + 31) & ~31) / 32;        // nkt <= min(CW, nqt) (launcher)            // is a constant
    char* As = smem;                                  // Q image (dense swizzled rows)
    char* Bs = As + LqP * RS;                         // dO image (output dropout applied)
    float* lse2 = reinterpret_cast<float*>(Bs + LqP * RS);   // per query: lse * log2 e
    float* dl = lse2 + LqP;                           // per query: delta / dropout scale
    float* kbias = dl + LqP;
    char* stg = reinterpret_cast<char*>(kbias + LkP); // [2][CW][32 keys x TS] dS^T tiles; first the K image (prologue only)
    char* sc7 = stg + 2 * CW * TILE;                  // [32 x B_QRS] dQ transpose tile of wave 7
    uint32_t* flags = reinterpret_cast<uint32_t*>(sc7 + 32 * B_QRS);      // [0..7]: per-wave pad votes
    const uint16_t* qg = reinterpret_cast<const uint16_t*>(d.q) + (size_t)b * Lq * d.ldq + h * 32;
    const uint16_t* kg = reinterpret_cast<const uint16_t*>(d.k) + (size_t)b * Lk * d.ldk + h * 32;
    const uint16_t* vg = reinterpret_cast<const uint16_t*>(d.v) + (size_t)b * Lk * d.ldv + h * 32;
    const uint16_t* og = reinterpret_cast<const uint16_t*>(d.o) + (size_t)b * Lq * d.ldo + h * 32;
    const uint16_t* dog = reinterpret_cast<const uint16_t*>(d.d_o) + (size_t)b * Lq * d.lddo + h * 32;
    // the softmax scale and the dropout scale stay out of the per-element algebra: dK / dQ / dV are scaled once, when stored
    const float osc_dk = d.scale * keep_scale, osc_dv = keep_scale, inv_keep = 1.f / keep_scale;
    const Drop dout = drop_init(d.drop_o);

    // ONE memory round trip for the prologue (see the forward): all of Q, K, d_o, o of the head, the LSE row, the key-padding bytes,
    // the compute waves' own K / V operand rows and keep words are requested before the first wait.  Chunk c = t + 512 j (j = 0, 1):
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
    // keep words of (query tile i, this wave's key tile): key l31 = mrow(r, kh') sits in word 2 r + kh' of its tile.  The word of
    // tile 0 comes with the prologue loads, tile qt + 1's is requested while tile qt is worked on (L2-resident: the forward wrote them)
    const uint32_t* kbp = reinterpret_cast<const uint32_t*>(d.keepbits) + ((size_t)bh_ * nqt * nkt + min(kt_own, nkt - 1)) * 32 +
                          2 * ((l31 & 3) + 4 * (l31 >> 3)) + ((l31 >> 2) & 1);
    const int kw_stride = nkt * 32;
    uint32_t kw_next = 0u;
    if (DROP) kw_next = kbp[0];
    float lsev = 0.f;
    uint8_t kpv = 1;
    if (t < Lq) lsev = d.lse[(size_t)bh_ * Lq + t];                               // NT = 512 >= LqP >= LkP
    if (t < Lk && d.keypad != nullptr) kpv = d.keypad[(size_t)b * Lk + t];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = t + NT * j, row = c >> 2, col = c & 3;
        if (row < LqP) {
            *reinterpret_cast<uint4*>(As + img_off(row, col)) = qc[j];
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
            if (col == 0) dl[row] = part * inv_keep;
            *reinterpret_cast<uint4*>(Bs + img_off(row, col)) = __builtin_bit_cast(uint4, pack8(gd));
        }
        if (row < LkP) *reinterpret_cast<uint4*>(stg + img_off(row, col)) = kc[j];     // K image, for wave 7's K^T operands only
    }
    if (t < LqP) lse2[t] = lsev * LOG2E;
    const bool kok = t < Lk && kpv != 0;
    if (t < LkP) kbias[t] = kok ? 0.f : -INFINITY;
    const int pad = (t < Lk && !kok) ? 1 : 0;
    const int wv = __any(pad) ? 1 : 0;
    if (lane == 0) flags[wave] = (uint32_t)wv;
    __syncthreads();
    int anypad = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) anypad |= (int)flags[w];
    anypad = __builtin_amdgcn_readfirstlane(anypad);
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
        // lane-constant part of the operand row addresses (row 32 qt + l31 of a swizzled image: + qt * 2048 per tile)
        const int roff0 = l31 * RS + (((0 + kh) ^ ((l31 >> 2) & 3)) << 4), roff1 = l31 * RS + (((2 + kh) ^ ((l31 >> 2) & 3)) << 4);
        auto scoresA = [&](int qt) {
            const char* Aq = As + qt * (32 * RS);
            const char* Bq = Bs + qt * (32 * RS);
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (!anypad) {
                // no padded key in this head (the usual case): both products start from the literal zero, no register is initialised
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Aq, roff0), kfr[0], zero, 0, 0, 0);               // S[q][key]
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bq, roff0), vfr[0], zero, 0, 0, 0);             // dP[q][key]
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Aq, roff1), kfr[1], s, 0, 0, 0);
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bq, roff1), vfr[1], dpv, 0, 0, 0);
            } else {
                // the key bias (0 / -inf) is the S accumulator's initial value; with `eye |` the diagonal tile keeps S[q][q] finite
                const bool dfix = (d.flags & MMFM_ATTN_DIAG) && qt == kt;
                int lv = l31;
                asm volatile("" : "+v"(lv));            // keeps the diagonal selects inside this (rare) branch instead of 16 hoisted registers
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (dfix && mrow(r, kh) == lv) ? 0.f : kbv;
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Aq, roff0), kfr[0], s, 0, 0, 0);
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bq, roff0), vfr[0], zero, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Aq, roff1), kfr[1], s, 0, 0, 0);
                dpv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(Bq, roff1), vfr[1], dpv, 0, 0, 0);
            }
        };
        const int gtail = ((Lq - (nqt - 1) * 32) + 7) >> 3;            // valid 8-query groups of the last query tile (Lq % 8 == 0)
        __syncthreads();                               // wave 7 has its K^T operands: the K image is dead, the staging slots free
        auto step = [&](int qt) {
            if (active) {
                scoresA(qt);
                uint32_t wq = 0u;
                if (DROP) {
                    wq = kw_next >> (4 * kh);          // bit (r & 3) + 8 (r >> 2) is now register r's query
                    kw_next = kbp[(size_t)min(qt + 1, nqt - 1) * kw_stride];
                }
                char* slot = stg + ((qt & 1) * CW + wave) * TILE + l31 * TS;
                if (qt == nqt - 1 && gtail != 4) {
                    if (gtail == 1) bwd_tile<1, DROP>(s, dpv, c2, lse2, dl, wq, As, Bs, slot, qt, kh, lane, dKt, dVt);
                    else if (gtail == 2) bwd_tile<2, DROP>(s, dpv, c2, lse2, dl, wq, As, Bs, slot, qt, kh, lane, dKt, dVt);
                    else bwd_tile<3, DROP>(s, dpv, c2, lse2, dl, wq, As, Bs, slot, qt, kh, lane, dKt, dVt);
                } else {
                    bwd_tile<4, DROP>(s, dpv, c2, lse2, dl, wq, As, Bs, slot, qt, kh, lane, dKt, dVt);
                }
            }
            __syncthreads();                           // staging buffer (qt & 1) is complete; buffer ((qt+1) & 1) has been consumed
        };
        if constexpr (NQT != 0) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) step(qt);
        } else {
            for (int qt = 0; qt < nqt; ++qt) step(qt);
        }
        // dK / dV leave through this wave's staging slot of the buffer the LAST query tile does not use (tile nqt - 2's: drained
        // before the last barrier)
        if (active) {
            char* scr = stg + ((nqt & 1) * CW + wave) * TILE;
            const f32x16 dk1[1] = {dKt}, dv1[1] = {dVt};
            store_tile_T<32, 1>(scr, TS, dk1, reinterpret_cast<uint16_t*>(d.dk) + (size_t)b * Lk * d.lddk + h * 32, d.lddk, kt * 32, Lk, lane, osc_dk);
            store_tile_T<32, 1>(scr, TS, dv1, reinterpret_cast<uint16_t*>(d.dv) + (size_t)b * Lk * d.lddv + h * 32, d.lddv, kt * 32, Lk, lane, osc_dv);
        }
    } else {
        // ---------------- wave 7: K^T operands of every key tile, hardware-transposed out of the K image and kept for the
        // whole kernel; then dQ of query tile qt from the staged dS^T tiles
        bf16x8v kT[CW][2];
#pragma unroll
        for (int kt = 0; kt < CW; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) kT[kt][s2] = trfrag_sw(stg, min(kt, nkt - 1) * 32 + 16 * s2, lane);
        __syncthreads();
        uint16_t* dqg = reinterpret_cast<uint16_t*>(d.dq) + (size_t)b * Lq * d.lddq + h * 32;
        auto step = [&](int qt) {
            __syncthreads();
            const char* buf = stg + (qt & 1) * CW * TILE;
            f32x16 dQt[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) dQt[0][r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < CW; ++kt) {
                if (kt < nkt) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
                        dQt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kT[kt][s2], trfrag(buf + kt * TILE, TS, 16 * s2, 0, lane), dQt[0], 0, 0, 0);
                }
            }
            store_tile_T<32, 1>(sc7, B_QRS, dQt, dqg, d.lddq, qt * 32, Lq, lane, osc_dk);
        };
        if constexpr (NQT != 0) {
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) step(qt);
        } else {
            for (int qt = 0; qt < nqt; ++qt) step(qt);
        }
    }
}

size_t bwd_fast_lds(int Lq, int Lk) {
    const int LqP = (Lq + 31) & ~31, LkP = (Lk + 31) & ~31;
    return (size_t)2 * LqP * B_RS + (size_t)2 * LqP * 4 + (size_t)LkP * 4 + (size_t)2 * B_CW * B_TILE + (size_t)32 * B_QRS + 64;
}

}  // namespace

// keep probability the keep-bit path applies for a requested drop probability p (quantised to 2^-KEEP_BITS)
static uint32_t keep_thresh(float p) {
    const int one = 1 << KEEP_BITS;
    const int t = (int)lrintf((1.f - p) * (float)one);
    return (uint32_t)std::min(std::max(t, 1), one - 1);
}
extern "C" float mmfm_attn_keep_prob(float p) { return p <= 0.f ? 1.f : (float)keep_thresh(p) / (float)(1 << KEEP_BITS); }
// bit tiles, then (dh = 64 kernels, attention_long.hip) one float per (b, head, query) for the backward's delta
extern "C" int64_t mmfm_attn_keepbits_bytes(int B, int heads, int Lq, int Lk) {
    const int64_t tiles = (int64_t)B * heads * ((Lq + 31) / 32) * ((Lk + 31) / 32) * 128;
    return tiles + (((int64_t)B * heads * Lq * 4 + 127) & ~(int64_t)127);
}

// the generator launch of a forward with attention dropout on the keep-bit path
int mmfm_attn_keepbits_launch(const mmfm_attn_desc& d, hipStream_t st) {
    KeepArgs a;
    a.bits = reinterpret_cast<uint32_t*>(d.keepbits);
    a.state = reinterpret_cast<const uint32_t*>(d.drop_p.state);
    a.site = d.drop_p.site;
    a.thresh = keep_thresh(d.drop_p.p);
    a.nkt = (d.Lk + 31) / 32;
    a.Lk = d.Lk;
    a.nwords = (uint32_t)((int64_t)d.B * d.heads * ((d.Lq + 31) / 32) * a.nkt * 32);
    hipLaunchKernelGGL(attn_keepbits_kernel, dim3((a.nwords + 255) / 256), dim3(256), 0, st, a);
    MMFM_LAUNCH_CHECK("mmfm_attn_fwd(keep bits)");
    return 0;
}

// Shapes the fast kernels take.  Returns -1000 when the general kernels must run.
int mmfm_attn_fast_launch(const mmfm_attn_desc& d, bool backward, hipStream_t st) {
    static const bool off = [] { const char* e = getenv("MMFM_ATTN_FAST"); return e && atoi(e) == 0; }();
    if (off || d.dh != 32 || (d.flags & (MMFM_ATTN_CAUSAL | MMFM_ATTN_SEP))) return -1000;
    const int nqt = (d.Lq + 31) / 32, nkt = (d.Lk + 31) / 32;
    if (d.Lq % 8 || d.Lk % 8 || nqt > 8 || nkt > B_CW || nkt > nqt) return -1000;
    const bool drop = d.drop_p.p > 0.f && d.drop_p.state != nullptr;
    if (drop && d.drop_p.p >= 1.f) return -1000;
    // with attention dropout the pair needs the keep-bit workspace (one decision source for both directions); without it the general
    // kernels hash, forward and backward alike
    if (drop && d.keepbits == nullptr) return -1000;
    const bool al = d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 8 == 0 && (uintptr_t)d.q % 16 == 0 &&
                    (uintptr_t)d.k % 16 == 0 && (uintptr_t)d.v % 16 == 0 && (uintptr_t)d.o % 16 == 0 && (uintptr_t)d.keepbits % 128 == 0;
    if (!al) return -1000;
    const bool alb = !backward || (d.lddo % 8 == 0 && d.lddq % 8 == 0 && d.lddk % 8 == 0 && d.lddv % 8 == 0 && (uintptr_t)d.d_o % 16 == 0 &&
                                   (uintptr_t)d.dq % 16 == 0 && (uintptr_t)d.dk % 16 == 0 && (uintptr_t)d.dv % 16 == 0);
    if (!alb) {
        // the forward of this shape took its decisions from the keep bits; the general backward would hash different ones
        if (drop) return mmfm_set_error(-1, "mmfm_attn_bwd(bf16, dh 32): gradient tensors must be 16-byte aligned with leading dims %% 8 == 0 when "
                                            "attention dropout runs on the keep-bit path (mmfm_attn_desc.keepbits)");
        return -1000;
    }
    const int grid = d.B * d.heads;
    const float keep_scale = drop ? 1.f / mmfm_attn_keep_prob(d.drop_p.p) : 1.f;
    if (!backward) {
        if (drop) { if (int rc = mmfm_attn_keepbits_launch(d, st)) return rc; }
        const int nw = nqt <= 4 ? 4 : (nqt == 7 ? 7 : (nqt <= 6 ? 6 : 8));
        const size_t lds = fwd_fast_lds(d.Lk, nw);
#define FWDF3(NWV, DRP, NKT)                                                                                        \
        {                                                                                                           \
            auto kern = attn_fwd_fast_kernel<NWV, DRP, NKT>;                                                        \
            if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(kern), lds, "mmfm_attn_fwd(bf16, dh 32)")) return rc; \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, st, d, keep_scale);                           \
        }
#define FWDF(NWV)                                                                                                   \
        {                                                                                                           \
            if (drop) { if (nkt == 7 && NWV >= 7) FWDF3(NWV, true, 7) else FWDF3(NWV, true, 0) }                    \
            else { if (nkt == 7 && NWV >= 7) FWDF3(NWV, false, 7) else FWDF3(NWV, false, 0) }                       \
        }
        if (nw == 4) FWDF(4) else if (nw == 6) FWDF(6) else if (nw == 7) FWDF(7) else FWDF(8)
#undef FWDF
#undef FWDF3
        MMFM_LAUNCH_CHECK("mmfm_attn_fwd(bf16, dh 32)");
        return 0;
    }
    const size_t lds = nqt == 7 ? bwd_fast_lds(224, 224) : bwd_fast_lds(d.Lq, d.Lk);
#define BWDF3(DRP, NQ)                                                                                              \
    {                                                                                                               \
        auto kern = attn_bwd_fast_kernel<DRP, NQ>;                                                                  \
        if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(kern), lds, "mmfm_attn_bwd(bf16, dh 32)")) return rc; \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(B_NW * 64), lds, st, d, keep_scale);                              \
    }
    if (drop) { if (nqt == 7) BWDF3(true, 7) else BWDF3(true, 0) }
    else { if (nqt == 7) BWDF3(false, 7) else BWDF3(false, 0) }
#undef BWDF3
    MMFM_LAUNCH_CHECK("mmfm_attn_bwd(bf16, dh 32)");
    return 0;
}
