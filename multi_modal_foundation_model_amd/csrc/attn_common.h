// Device helpers shared by the bf16 attention kernels (attention_bf16.hip, attention_fast.hip): operand fragments out of
// row-major LDS images, the accumulator row map, head staging, transposed tile stores.
#pragma once
#include "common.h"

namespace attn {

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

// Accumulator tile (rows = head dim d, lane = token) -> bf16 rows [token][d] written IN PLACE into the wave's own,
// already consumed 32-row tile of an LDS image (RS-byte rows), then streamed to global as 16-B row chunks.
template <int DH, int DT>
__device__ __forceinline__ void store_tile_T(char* tile, int RS, const f32x16 (&acc)[DT], uint16_t* outg, int ld, int row0, int nrows_total,
                                             int lane, float osc) {
    const int l31 = lane & 31, kh = lane >> 5;
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dcol = i * 32 + 8 * g + 4 * kh;
            if (dcol < DH) {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
                bf16x4v v;
                v[0] = (__bf16)(acc[i][4 * g + 0] * osc); v[1] = (__bf16)(acc[i][4 * g + 1] * osc);
                v[2] = (__bf16)(acc[i][4 * g + 2] * osc); v[3] = (__bf16)(acc[i][4 * g + 3] * osc);
                *reinterpret_cast<uint2*>(tile + l31 * RS + dcol * 2) = __builtin_bit_cast(uint2, v);
            }
        }
    wave_lds_fence();
    constexpr int C8 = DH / 8;
    for (int idx = lane; idx < 32 * C8; idx += 64) {
        const int row = idx / C8, c = idx % C8;
        if (row0 + row < nrows_total)
            *reinterpret_cast<uint4*>(outg + (size_t)(row0 + row) * ld + 8 * c) = *reinterpret_cast<const uint4*>(tile + row * RS + c * 16);
    }
}

// ---- attention-probability dropout, shared by the general and the fast kernels
// Dropout on the probabilities.  One hash decides a PAIR of keys (even key: low 16 bits, odd key: high 16 bits, each against
// the top 16 bits of the threshold).  The hash is two-level: a full-strength 2 x 32-bit ROW key per (batch, head, query)
// - computed once per query tile (forward, dQ phase: lane = query) or once per workgroup into an LDS table (dK/dV phase:
// lane = key) - and a 7-instruction xorshift / 24-bit-multiply mix of (pair index ^ key A, key B) per pair.  Forward and
// both backward phases evaluate the same function, so no mask is stored.  v_mul_u32_u24 issues at the full VALU rate,
// v_mul_lo_u32 at a quarter of it, and a step draws ~10^9 of these decisions: the round-1 version (11 instructions per pair
// on a linear pair index, plus a 32-bit multiply to form that index in the dK/dV phase) was ~40 % of the forward tile's VALU work.
struct Drop16 {
    uint32_t k0, k1, t16;
    float scale;
    bool on;
    __device__ __forceinline__ void rowkeys(uint32_t rowid, uint32_t& ka, uint32_t& kb) const {       // rowid = bh * Lq + q
        ka = mix32(rowid ^ k0);
        kb = mix32((rowid + 0x9E3779B9u) ^ k1);
    }
    // jx = (key >> 1) ^ ka
    __device__ __forceinline__ uint32_t hash(uint32_t jx, uint32_t kb) const {
        uint32_t h = __umul24(jx, 0x7FEB35u) + kb;
        h ^= h >> 13;
        h = __umul24(h, 0x46CA6Bu);
        h ^= h >> 16;
        return h;
    }
};
__device__ __forceinline__ Drop16 drop16_init(mmfm_dropout d) {
    const Drop b = drop_init(d);
    Drop16 r;
    r.k0 = b.k0; r.k1 = b.k1; r.t16 = b.thresh >> 16; r.scale = b.scale; r.on = b.on();
    return r;
}

// ---- element-wise helpers of the keep-bit kernels (attention_fast.hip, attention_long.hip)
// NO vector instruction of these kernels lives in inline asm.  Round 4 tried (v_add / v_max3 / v_cndmask / v_bfe in asm, to keep the
// compiler from packing fp32 pairs or rewriting a bit test): every launch returned garbage.  gfx950 leaves several read-after-write
// waits to software (an MFMA's result, a transcendental's, a permlane's) and the compiler inserts them only around instructions it
// can see.  What shapes the code instead: -fno-slp-vectorize for this file (Makefile: v_pk_*_f32 is 8 issue cycles for two results,
// no gain beside MFMAs), __builtin_amdgcn_inverse_ballot_w64 for "select by a scalar lane mask" (one v_cndmask_b32 with an SGPR pair),
// and EMPTY asm statements as optimisation barriers only.
__device__ __forceinline__ float v_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float v_add(float a, float b) { return a + b; }
__device__ __forceinline__ float v_mul(float a, float b) { return a * b; }
// p where the lane's bit of `mask` is set, else 0
__device__ __forceinline__ float v_keep(float p, uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask) ? p : 0.f; }
__device__ __forceinline__ float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return v_max(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


// the 16 lane masks of one (query tile, key tile): register r's mask is the 64-bit pair (words 2 r, 2 r + 1) of the keep-bit tile
struct Masks16 { uint64_t m[16]; };
typedef const Masks16 __attribute__((address_space(4))) * masks_ptr;      // constant address space: scalar loads
__device__ __forceinline__ Masks16 ld_masks(masks_ptr p) {
    Masks16 r;
#pragma unroll
    for (int i = 0; i < 16; ++i) r.m[i] = p->m[i];
    return r;
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {          // one v_cvt_pk_bf16_f32
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
    bf16x2v v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}

}  // namespace attn
