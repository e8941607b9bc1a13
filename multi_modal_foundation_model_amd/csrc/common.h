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

// Dynamic-LDS opt-in above 64 KB (hipFuncAttributeMaxDynamicSharedMemorySize), remembered per (device, kernel): the attribute belongs
// to the pair, so a process that drives several GPUs opts in on each (api.hip).  `what` names the entry point in the error text.
int mmfm_lds_opt_in(const void* kern, size_t bytes, const char* what);

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
// One hash decides a PAIR of neighbouring elements (even index: low 16 bits, odd index: high 16 bits, each compared with
// the top 16 bits of the threshold: p is honoured to 2^-16).  The mixer is xorshift / 24-bit multiply: v_mul_u32_u24 issues at
// the full VALU rate, v_mul_lo_u32 at a quarter of it, and a step draws ~10^9 decisions.  (Round 1 used two murmur-style
// 32-bit finalisers per ELEMENT: 4 quarter-rate multiplies + 12 ops = ~110 cycles per decision at one wave per SIMD, which
// was 7k cycles of every 128x128 GEMM tile epilogue with dropout and 14k cycles of every fused-MLP row pass.)
struct Drop {
    uint32_t k0, k1, thresh, t16;
    float scale;
    __device__ __forceinline__ bool on() const { return thresh != 0; }
    __device__ __forceinline__ uint32_t hash(uint32_t pidx) const {
        uint32_t h = pidx ^ k0;
        h ^= h >> 16;
        // v_mul_u32_u24 sees bits 0..23 only: the top byte enters through its own multiply (an additive, non-cancelling term).
        // Without it hash(p) == hash(p ^ (d << 24 | d << 8)) for every d: tensors beyond 2^25 elements repeated their masks.
        h = __umul24(h, 0x7FEB35u) + (__umul24(h >> 24, 0x9E3779u) + k1);
        h ^= h >> 13;
        h = __umul24(h, 0x46CA6Bu) ^ (h >> 9);
        h ^= h >> 16;
        return h;
    }
    // hash of the pair that holds element idx (idx and idx ^ 1 share it)
    __device__ __forceinline__ uint32_t pair(uint64_t idx) const {
        return hash((uint32_t)(idx >> 1) + __umul24((uint32_t)(idx >> 33), 0x9E3779u));
    }
    __device__ __forceinline__ bool lo(uint32_t h) const { return (h & 0xffffu) >= t16; }     // even element of the pair
    __device__ __forceinline__ bool hi(uint32_t h) const { return (h >> 16) >= t16; }         // odd element
    __device__ __forceinline__ bool keep(uint64_t idx) const {
        const uint32_t h = pair(idx);
        return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= t16;
    }
    __device__ __forceinline__ float apply(float v, uint64_t idx) const {
        return on() ? (keep(idx) ? v * scale : 0.f) : v;
    }
    // elements idx (even) and idx + 1 with one hash
    __device__ __forceinline__ void apply2(float& v0, float& v1, uint64_t idx_even) const {
        const uint32_t h = pair(idx_even);
        v0 = lo(h) ? v0 * scale : 0.f;
        v1 = hi(h) ? v1 * scale : 0.f;
    }
};
__device__ __forceinline__ Drop drop_init(mmfm_dropout d) {
    Drop r;
    if (d.p <= 0.f || d.state == nullptr) { r.k0 = r.k1 = r.thresh = r.t16 = 0; r.scale = 1.f; return r; }
    const uint32_t* s = reinterpret_cast<const uint32_t*>(d.state);
    r.k0 = mix32(s[0] + d.site * 0x9E3779B9u);
    r.k1 = mix32(s[1] ^ (d.site * 0x85EBCA6Bu + 0xC2B2AE35u));
    double t = (double)d.p * 4294967296.0;
    r.thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    r.t16 = r.thresh >> 16;
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
// bf16-mode GELU: Phi(x) = 0.5 + x~ P(x~^2), x~ = clamp(x, -4, 4), P a degree-9 near-minimax polynomial: no transcendental,
// every step is a packed fp32 op on a PAIR of values (v_pk_mul_f32 / v_pk_fma_f32).  |Phi error| <= 3.4e-5 everywhere
// (|gelu error| <= 1.9e-5 on [-4, 4], |x| * 3e-5 beyond), against the 2^-9 relative step of the bf16 value it feeds.
// erff + expf (gelu_erf / gelu_erf_grad below, kept for the fp32 parity kernels) are ~60 VALU ops and 2-3 quarter-rate
// transcendentals per element; this is 7 ops per element.  Measured: see DESIGN.md 3b.
typedef float mmfm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mmfm_f32x2 splat2(float v) { mmfm_f32x2 r; r.x = v; r.y = v; return r; }
__device__ __forceinline__ mmfm_f32x2 phi2(mmfm_f32x2 x) {
    mmfm_f32x2 xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -4.f, 4.f);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -4.f, 4.f);
    const mmfm_f32x2 t = xc * xc;
    mmfm_f32x2 p = splat2(-4.407132645e-12f);
    p = __builtin_elementwise_fma(p, t, splat2(4.129891984e-10f));
    p = __builtin_elementwise_fma(p, t, splat2(-1.754582968e-08f));
    p = __builtin_elementwise_fma(p, t, splat2(4.542657450e-07f));
    p = __builtin_elementwise_fma(p, t, splat2(-8.172721209e-06f));
    p = __builtin_elementwise_fma(p, t, splat2(1.105528936e-04f));
    p = __builtin_elementwise_fma(p, t, splat2(-1.176239806e-03f));
    p = __builtin_elementwise_fma(p, t, splat2(9.960514493e-03f));
    p = __builtin_elementwise_fma(p, t, splat2(-6.648434699e-02f));
    p = __builtin_elementwise_fma(p, t, splat2(3.989418149e-01f));
    return __builtin_elementwise_fma(xc, p, splat2(0.5f));
}
__device__ __forceinline__ mmfm_f32x2 gelu2(mmfm_f32x2 x) { return x * phi2(x); }
__device__ __forceinline__ mmfm_f32x2 gelu_grad2(mmfm_f32x2 x) {             // Phi(x) + x phi(x)
    const mmfm_f32x2 t = x * x * splat2(-0.72134752044448170f);
    mmfm_f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    return __builtin_elementwise_fma(x * splat2(0.39894228040143268f), e, phi2(x));
}
__device__ __forceinline__ void gelu_both2(mmfm_f32x2 x, mmfm_f32x2& g, mmfm_f32x2& dg) {     // gelu(x) and gelu'(x), one Phi
    const mmfm_f32x2 ph = phi2(x), t = x * x * splat2(-0.72134752044448170f);
    mmfm_f32x2 e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    g = x * ph;
    dg = __builtin_elementwise_fma(x * splat2(0.39894228040143268f), e, ph);
}
__device__ __forceinline__ float gelu_poly(float x) { return gelu2(splat2(x)).x; }
__device__ __forceinline__ float gelu_poly_grad(float x) { return gelu_grad2(splat2(x)).x; }
// in place on an even-length array
template <int N> __device__ __forceinline__ void gelu_n(float* v) {
#pragma unroll
    for (int i = 0; i < N; i += 2) { mmfm_f32x2 a; a.x = v[i]; a.y = v[i + 1]; a = gelu2(a); v[i] = a.x; v[i + 1] = a.y; }
}
template <int N> __device__ __forceinline__ void mul_gelu_grad_n(float* v, const float* u) {      // v *= gelu'(u)
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        mmfm_f32x2 a; a.x = u[i]; a.y = u[i + 1];
        a = gelu_grad2(a);
        v[i] *= a.x; v[i + 1] *= a.y;
    }
}
__device__ __forceinline__ float softsign_f(float x) { return x / (1.f + fabsf(x)); }
__device__ __forceinline__ float softsign_grad(float x) { float d = 1.f + fabsf(x); return 1.f / (d * d); }
// the same derivative from the activation's OUTPUT y = s * x / (1 + |x|):  1 / (1 + |x|) = 1 - |y| / s  (act 5: the tokeniser's
// backward reads the activation it needs anyway instead of a second, saved [rows, 1336] pre-activation tensor)
// Error bound (tests/test_kernels_gpu.py::test_gemm_act5_softsign_grad_from_output_error_bound): y is stored in bf16, so r = 1 - |y| / s carries
// an absolute error of up to 2^-9 and the factor r^2 a RELATIVE error of about 0.6 % x (1 + |x|) - 5 % at |x| = 8, the whole value beyond |x| ~ 170
// (y rounds to s, the gradient reads 0 where the true factor is < 4e-5).  Spike-count pre-activations of the tokenisers sit at |x| of a few units.
__device__ __forceinline__ float softsign_grad_from_out(float y, float inv_s) { const float r = 1.f - fabsf(y) * inv_s; return r * r; }

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
