// Shared device/host helpers for the mmfm HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>

#include "../../include/mmfm.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __bf16 bf16_t;

// ------------------------------------------------------------------ errors (host)
int mmfm_set_error(int code, const char* fmt, ...);

#define MMFM_REQUIRE(cond, ...)                      \
    do {                                             \
        if (!(cond)) return mmfm_set_error(-1, __VA_ARGS__); \
    } while (0)

#define MMFM_LAUNCH_CHECK(name)                                                         \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess)                                                          \
            return mmfm_set_error((int)e__, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ bf16 <-> f32
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    bf16_t b = (bf16_t)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return *reinterpret_cast<uint16_t*>(&b);
}

// ------------------------------------------------------------------ streaming (non-temporal) stores
// Activation outputs are written once and read by a LATER kernel: a plain store write-allocates in the XCD's 4 MB L2 and
// evicts the operand panels co-resident workgroups are about to re-read.  Measured on the bf16 GEMM at B = 1024 (x.W^T,
// qkv): 201 -> 163 us with the output stored non-temporally; up-projection 133 -> 107 us.
typedef __attribute__((ext_vector_type(4))) unsigned int mmfm_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int mmfm_u32x2;
typedef __attribute__((ext_vector_type(4))) float mmfm_f32x4;
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
    __builtin_nontemporal_store(__builtin_bit_cast(mmfm_u32x4, v), reinterpret_cast<mmfm_u32x4*>(p));
}
__device__ __forceinline__ void st_stream(uint2* p, uint2 v) {
    __builtin_nontemporal_store(__builtin_bit_cast(mmfm_u32x2, v), reinterpret_cast<mmfm_u32x2*>(p));
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
    __builtin_nontemporal_store(__builtin_bit_cast(mmfm_f32x4, v), reinterpret_cast<mmfm_f32x4*>(p));
}

template <typename T> struct io;
template <> struct io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ void st4s(float* p, float4 v) { st_stream(reinterpret_cast<float4*>(p), v); }
};
template <> struct io<uint16_t> {  // bf16 storage
    static __device__ __forceinline__ float ld(const uint16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f2bf(v); }
    static __device__ __forceinline__ float4 ld4(const uint16_t* p) {
        uint2 u = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st4(uint16_t* p, float4 v) {
        uint2 u;
        u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
        u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
        *reinterpret_cast<uint2*>(p) = u;
    }
    static __device__ __forceinline__ void st4s(uint16_t* p, float4 v) {
        uint2 u;
        u.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
        u.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
        st_stream(reinterpret_cast<uint2*>(p), u);
    }
};

// ------------------------------------------------------------------ counter-based dropout RNG
// keep(idx) is a pure function of (state[0], state[1], site, idx): the backward pass regenerates
// the forward's mask instead of storing it.  `state` lives in device memory so a captured
// hipGraph sees a fresh stream every replay (mmfm_rng_advance bumps state[1]).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
struct Drop {
    uint32_t k0, k1, thresh;
    float scale;
    __device__ __forceinline__ bool on() const { return thresh != 0; }
    __device__ __forceinline__ bool keep(uint64_t idx) const {
        uint32_t h = mix32(mix32((uint32_t)idx ^ k0) + k1 + (uint32_t)(idx >> 32) * 0x9E3779B9u);
        return h >= thresh;
    }
    __device__ __forceinline__ float apply(float v, uint64_t idx) const {
        return on() ? (keep(idx) ? v * scale : 0.f) : v;
    }
};
__device__ __forceinline__ Drop drop_init(mmfm_dropout d) {
    Drop r;
    if (d.p <= 0.f || d.state == nullptr) { r.k0 = r.k1 = r.thresh = 0; r.scale = 1.f; return r; }
    const uint32_t* s = reinterpret_cast<const uint32_t*>(d.state);
    r.k0 = mix32(s[0] + d.site * 0x9E3779B9u);
    r.k1 = mix32(s[1] ^ (d.site * 0x85EBCA6Bu + 0xC2B2AE35u));
    double t = (double)d.p * 4294967296.0;
    r.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    r.scale = 1.f / (1.f - d.p);
    return r;
}

// ------------------------------------------------------------------ attention workgroup -> (batch, head)
// One workgroup per (b, head) reads dh-wide slices of [.., heads*dh] rows: with dh = 32 a 128-B line holds TWO heads, and the
// dispatcher deals consecutive workgroups round-robin over the 8 XCDs, so the heads of one sample landed on 8 different L2s and
// every line was fetched from HBM once per head that touches it (PMC round 1: 631 MB fetched for 315 MB of q/k/v).  The bijective
// XCD remap (cdna_hip_programming.md T1) gives every XCD a contiguous range of (b, head) ids: the heads of a sample run on one
// XCD, back to back, and share its L2.  MMFM_ATTN_NO_REMAP bit (flags & 0x100) restores the plain order for A/B runs.
__device__ __forceinline__ int attn_xcd_remap(int bid, int nwg, int flags) {
    if (flags & 0x100) return bid;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------ activations
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float softsign_f(float x) { return x / (1.f + fabsf(x)); }
__device__ __forceinline__ float softsign_grad(float x) { float d = 1.f + fabsf(x); return 1.f / (d * d); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
