// bf16 MFMA GEMM (throughput mode) behind mmfm_gemm: bf16 storage, fp32 accumulate,
// v_mfma_f32_32x32x16_bf16.  128x128 output tile, BK = 64, 4 wavefronts (2x2), each wave a 64x64
// sub-tile = 2x2 MFMA tiles (64 accumulator VGPRs); 36 KB of LDS -> up to 4 workgroups per CU, so
// wave-level parallelism hides the global-load latency of the register-staged prefetch.
//
// Two LDS tile images, chosen per operand from its contiguous axis in global memory:
//   KC (reduction contiguous: x[M,K], W[N,K]):  [128 rows][64 k + 8 pad] bf16, 144-B rows.  The MFMA
//      operand (8 consecutive k of one row) is one ds_read_b128; the 16-B pad walks consecutive rows
//      across all sixteen 16-B slots of the 256-B bank row -> conflict-free.
//   RC (row contiguous: dY^T and X for dW, W for dX):  [64 k][128 rows] bf16, 256-B rows, byte offset
//      XOR ((k & 3) << 6).  The operand needs 8 k-strided values per lane: two ds_read_b64_tr_b16
//      (hardware 4x16 transpose); the XOR spreads the four k-rows of a read over the four 64-B quarters
//      of the bank row -> conflict-free.  Global loads stay 16 B/lane along the contiguous axis.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int NTHREADS = 256;
// BK (k-tile depth) is a template parameter: 64 (36 KB LDS, up to 4 workgroups/CU) or 128 (70 KB, 2/CU, half the
// load-latency rounds for the K = 256 shapes).  KC image rows are BK + 8 elements: 144 B / 272 B, both conflict-free.
template <int BK> struct Geo {
    static constexpr int KC_LD = BK + 8;
    static constexpr int TILE_BYTES = 128 * KC_LD * 2;     // >= RC image BK * 256
    static constexpr int NP = BK / 16;                     // uint4 per thread per operand tile
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8v;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 8 consecutive bf16 starting at src, of which the first `nvalid` (<= 8) exist; align = 8/4/1 elements
__device__ __forceinline__ uint4 load8(const uint16_t* __restrict__ src, int nvalid, int align) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (nvalid >= 8 && align == 8) return *reinterpret_cast<const uint4*>(src);
    if (nvalid >= 8 && align == 4) {
        const uint2 a = *reinterpret_cast<const uint2*>(src), b = *reinterpret_cast<const uint2*>(src + 4);
        return make_uint4(a.x, a.y, b.x, b.y);
    }
    if (nvalid <= 0) return v;
    uint16_t e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = (j < nvalid) ? src[j] : (uint16_t)0;
    v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
    v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
    return v;
}

template <bool RC, int BK>
__device__ __forceinline__ void g2r(uint4 (&r)[BK / 16], const uint16_t* __restrict__ base, int ld, int row0, int k0, int rows, int kend,
                                    int align, int t) {
    constexpr int NP = BK / 16, LPR = BK / 8, RPP = 256 / LPR;      // lanes per row, rows per pass
    if (!RC) {          // reduction contiguous: LPR lanes x 16 B = one BK*2-byte row segment
        const int kq = t % LPR, r0 = t / LPR, k = k0 + 8 * kq;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int row = row0 + r0 + RPP * p;
            r[p] = (row < rows) ? load8(base + (size_t)row * ld + k, kend - k, align) : make_uint4(0u, 0u, 0u, 0u);
        }
    } else {            // row contiguous: 16 lanes x 16 B = 256 B of one k-row
        const int cq = t & 15, kk0 = t >> 4, col = row0 + 8 * cq;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = k0 + kk0 + 16 * p;
            r[p] = (k < kend) ? load8(base + (size_t)k * ld + col, rows - col, align) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

template <bool RC, int BK>
__device__ __forceinline__ void r2s(char* __restrict__ S, const uint4 (&r)[BK / 16], int t) {
    constexpr int NP = BK / 16, LPR = BK / 8, RPP = 256 / LPR, KC_LD = Geo<BK>::KC_LD;
    if (!RC) {
        const int kq = t % LPR, r0 = t / LPR;
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<uint4*>(S + (r0 + RPP * p) * (KC_LD * 2) + kq * 16) = r[p];
    } else {
        const int cq = t & 15, kk0 = t >> 4;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int kk = kk0 + 16 * p;
            *reinterpret_cast<uint4*>(S + kk * 256 + ((cq * 16) ^ ((kk & 3) << 6))) = r[p];
        }
    }
}

// MFMA operand for rows [rowbase, rowbase+32) and k in [16*ks, 16*ks+16): lane (r = lane%32, h = lane/32)
// holds row rowbase + r, k = 16*ks + 8*h + 0..7
template <bool RC, int BK>
__device__ __forceinline__ bf16x8v frag(const char* __restrict__ S, int rowbase, int ks, int lane) {
    if (!RC) {
        const int r = lane & 31, h = lane >> 5;
        const uint4 v = *reinterpret_cast<const uint4*>(S + (rowbase + r) * (Geo<BK>::KC_LD * 2) + ks * 32 + h * 16);
        return __builtin_bit_cast(bf16x8v, v);
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(row) block; lane 4q+p supplies row q's address at
        // columns 4p..4p+3 and lane i receives column i with the 4 k values in order.
        const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
        const int col = rowbase + 16 * (g & 1) + 4 * p;
        const int kb = ks * 16 + 8 * (g >> 1);
        const int off0 = (kb + q) * 256 + ((col * 2) ^ (q << 6));            // (kb+q) & 3 == q
        const int off1 = (kb + 4 + q) * 256 + ((col * 2) ^ (q << 6));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + off0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(S + off1));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(bf16x8v, v);
    }
}

template <typename TO>
__device__ __forceinline__ void epilogue_store(const mmfm_gemm_desc& d, const Drop& dr, float v, int m, int n) {
    typedef io<uint16_t> I16;
    if (d.bias) v += d.bias[n];
    if (d.pre_out) I16::st(reinterpret_cast<uint16_t*>(d.pre_out) + (size_t)m * d.ldc + n, v);
    if (d.act == 1) v = gelu_poly(v);
    else if (d.act == 2) v = softsign_f(v) * d.act_scale;
    if (d.gradmul_pre) {
        const float u = I16::ld(reinterpret_cast<const uint16_t*>(d.gradmul_pre) + (size_t)m * d.ldc + n);
        v *= (d.act == 3) ? gelu_poly_grad(u) : (d.act == 4 ? softsign_grad(u) : softsign_grad_from_out(u, 1.f / d.act_scale)) * d.act_scale;
    }
    v = dr.apply(v, (uint64_t)m * (uint64_t)d.N + (uint64_t)n);
    if (d.residual) v += I16::ld(reinterpret_cast<const uint16_t*>(d.residual) + (size_t)m * d.ldr + n);
    io<TO>::st(reinterpret_cast<TO*>(d.C) + (size_t)m * d.ldc + n, v);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8e;
__device__ __forceinline__ bf16x8e pack8f(const float* v) {
    bf16x8e o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    return o;
}
__device__ __forceinline__ void unpack8(uint4 w, float* u) {
    u[0] = __uint_as_float(w.x << 16); u[1] = __uint_as_float(w.x & 0xffff0000u);
    u[2] = __uint_as_float(w.y << 16); u[3] = __uint_as_float(w.y & 0xffff0000u);
    u[4] = __uint_as_float(w.z << 16); u[5] = __uint_as_float(w.z & 0xffff0000u);
    u[6] = __uint_as_float(w.w << 16); u[7] = __uint_as_float(w.w & 0xffff0000u);
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Epilogue of one output tile (see the comment inside).  `smem` is the workgroup's operand LDS, free at this point.
// 8 bf16 of an epilogue chunk: one 16-B access, or - rows only 8-B aligned / ragged last chunk (N % 8 == 4, e.g. the
// 668-neuron head and token-embedding shapes) - two 8-B halves of which the second exists only if `hi`
__device__ __forceinline__ uint4 ldg8(const uint16_t* p, bool a16, bool hi, bool nt) {
    if (a16 && nt) return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const mmfm_u32x4*>(p)));
    if (a16) return *reinterpret_cast<const uint4*>(p);
    const uint2 a = *reinterpret_cast<const uint2*>(p);
    const uint2 b = hi ? *reinterpret_cast<const uint2*>(p + 4) : make_uint2(0u, 0u);
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ void stg8(uint16_t* p, uint4 v, bool a16, bool hi, bool nt) {
    if (nt) {            // streaming store (see common.h): the tile is not read again by this kernel
        if (a16) { st_stream(reinterpret_cast<uint4*>(p), v); return; }
        st_stream(reinterpret_cast<uint2*>(p), make_uint2(v.x, v.y));
        if (hi) st_stream(reinterpret_cast<uint2*>(p + 4), make_uint2(v.z, v.w));
        return;
    }
    if (a16) { *reinterpret_cast<uint4*>(p) = v; return; }
    *reinterpret_cast<uint2*>(p) = make_uint2(v.x, v.y);
    if (hi) *reinterpret_cast<uint2*>(p + 4) = make_uint2(v.z, v.w);
}

// EPI_LOADS = false: the instantiation of the persistent bf16-output kernel, launched only without a saved pre-activation /
// residual operand (their registers would sit on top of the next tile's prefetched slice)
template <typename TO, bool EPI_LOADS = true>
__device__ __forceinline__ void epilogue_tile(const mmfm_gemm_desc& d, f32x16 (&acc)[2][2], char* smem, int m0, int n0, int z, int vec_epi,
                                              int t, int wm, int wn, int kh, int l31) {
    // ---- epilogue.  Fast path (row-aligned shapes): the fp32 tile is staged through LDS half a tile at a time
    // (64 rows x 128 cols x 4 B = 32 KB, reusing the operand buffers) so that each thread owns 8 consecutive
    // columns of a row: bias/pre-activation/residual/output move as 16-B (bf16) or 2x16-B (fp32) accesses.
    const bool split = d.splits > 1;
    float* Cf = reinterpret_cast<float*>(d.C) + (split ? (size_t)z * d.slab_stride : 0);
    const Drop dr = drop_init(d.drop);
    if (vec_epi & 1) {
        float* stage = reinterpret_cast<float*>(smem);
        constexpr int SLDW = 132;
        constexpr bool BF_OUT = sizeof(TO) == 2;
        const bool bf_path = BF_OUT && !split;
        const bool a16 = !(vec_epi & 2);
        // every chunk of a thread has the same 8 columns (chunk & 15 == t & 15): bias is fetched once
        const int col = (t & 15) * 8, n = n0 + col;
        const bool n_ok = n < d.N, hi = n + 8 <= d.N;
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = 0.f;
        if (d.bias && n_ok && !split) {
            const float4 b0 = *reinterpret_cast<const float4*>(d.bias + n);
            const float4 b1 = hi ? *reinterpret_cast<const float4*>(d.bias + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            bs[0] = b0.x; bs[1] = b0.y; bs[2] = b0.z; bs[3] = b0.w; bs[4] = b1.x; bs[5] = b1.y; bs[6] = b1.z; bs[7] = b1.w;
        }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            // the half's epilogue operands (saved pre-activation, residual) are requested BEFORE the tile is staged: their
            // latency runs under the two barriers and the LDS round trip instead of in front of every chunk (one 16-B load,
            // use, next load ... was 4 exposed round trips per half: dg GEMM 193 us, down-projection 145 us at B = 1024)
            uint4 gu[4], gr[4];
            if (EPI_LOADS && bf_path) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int m = m0 + half * 64 + ((t + 256 * c) >> 4);
                    gu[c] = gr[c] = make_uint4(0u, 0u, 0u, 0u);
                    if (m < d.M && n_ok) {
                        if (d.gradmul_pre) gu[c] = ldg8(reinterpret_cast<const uint16_t*>(d.gradmul_pre) + (size_t)m * d.ldc + n, a16, hi, (vec_epi & 32768) != 0);
                        if (d.residual) gr[c] = ldg8(reinterpret_cast<const uint16_t*>(d.residual) + (size_t)m * d.ldr + n, a16, hi, (vec_epi & 32768) != 0);
                    }
                }
            }
            __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stage[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * SLDW + wn * 64 + j * 32 + l31] = acc[i][j][r];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = (t + 256 * c) >> 4;
                const int m = m0 + half * 64 + row;
                if (m >= d.M || !n_ok) continue;          // chunks are all-in, all-out or (half mode) 4 columns in
                float v[8];
                const float4 s0 = *reinterpret_cast<const float4*>(stage + row * SLDW + col);
                const float4 s1 = *reinterpret_cast<const float4*>(stage + row * SLDW + col + 4);
                v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w; v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
                const size_t off = (size_t)m * d.ldc + n;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bs[e];
                if (split || !BF_OUT) {
                    float* dst = (split ? Cf : reinterpret_cast<float*>(d.C)) + off;
                    if (vec_epi & 16384) {
                        st_stream(reinterpret_cast<float4*>(dst), make_float4(v[0], v[1], v[2], v[3]));
                        if (hi) st_stream(reinterpret_cast<float4*>(dst + 4), make_float4(v[4], v[5], v[6], v[7]));
                    } else {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        if (hi) *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                    continue;
                }
                if (d.pre_out) stg8(reinterpret_cast<uint16_t*>(d.pre_out) + off, __builtin_bit_cast(uint4, pack8f(v)), a16, hi, (vec_epi & 8192) != 0);
                if (d.act == 1) gelu_n<8>(v);
                else if (d.act == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = softsign_f(v[e]) * d.act_scale;
                }
                if (EPI_LOADS && d.gradmul_pre) {
                    float u[8];
                    unpack8(gu[c], u);
                    if (d.act == 3) mul_gelu_grad_n<8>(v, u);
                    else if (d.act == 4) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= softsign_grad(u[e]) * d.act_scale;
                    } else {
                        const float inv_s = 1.f / d.act_scale;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= softsign_grad_from_out(u[e], inv_s) * d.act_scale;
                    }
                }
                if (dr.on()) {
                    const uint64_t base = (uint64_t)m * (uint64_t)d.N + (uint64_t)n;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) dr.apply2(v[e], v[e + 1], base + e);      // n and N are even here, so base is
                }
                if (EPI_LOADS && d.residual) {
                    float u[8];
                    unpack8(gr[c], u);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += u[e];
                }
                stg8(reinterpret_cast<uint16_t*>(d.C) + off, __builtin_bit_cast(uint4, pack8f(v)), a16, hi, (vec_epi & 4096) != 0);
            }
        }
        return;
    }
    if (split) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int n = n0 + wn * 64 + j * 32 + l31;
                    if (m < d.M && n < d.N) Cf[(size_t)m * d.ldc + n] = acc[i][j][r];
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (m < d.M && n < d.N) epilogue_store<TO>(d, dr, acc[i][j][r], m, n);
            }
}

// Work item = output tile x K-split; item order is XCD-aware (xcd_remap): items that share an operand panel run on one
// XCD at about the same time.
// fp32-output launches (split-K weight-gradient slabs, fp32 heads) are PERSISTENT: a workgroup walks items
// w = blockIdx.x, += gridDim.x and fetches the first K-slice of the NEXT item into the staging registers before the
// epilogue of the current one, so the slab stores overlap the next loads (measured on MI355X at B=1024: dW GEMMs 68 -> 47,
// 130 -> 90 us).  bf16-output launches stay one item per workgroup: their epilogue LOADS (bias, residual, saved
// pre-activation) are issued after the prefetch and vmcnt retires in order, so the epilogue would wait for the whole
// prefetch; hoisting those loads above the prefetch costs 40 VGPRs, spills, and measured 30 % slower (194 -> 261 us).
template <bool ARC, bool BRC, typename TO, int BK, bool PERSIST = (sizeof(TO) == 4)>
__global__ __launch_bounds__(NTHREADS, BK == 64 ? 3 : 2) void gemm_bf16_kernel(const mmfm_gemm_desc d, const int alignA, const int alignB,
                                                                              const int vec_epi, const int total_items) {
    constexpr int TILE_BYTES = Geo<BK>::TILE_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILE_BYTES];
    char* As = smem;
    char* Bs = smem + TILE_BYTES;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, l31 = lane & 31;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int tiles = ((d.M + BM - 1) / BM) * tiles_n;
    const uint16_t* A = reinterpret_cast<const uint16_t*>(d.A);
    const uint16_t* B = reinterpret_cast<const uint16_t*>(d.B);

    auto decode = [&](int w, int& m0, int& n0, int& z, int& kbeg, int& kend) {
        const int item = xcd_remap(w, total_items);     // = z * tiles + tile: neighbours share the K range and a panel
        z = item / tiles;
        const int tile = item - z * tiles;
        m0 = (tile / tiles_n) * BM;
        n0 = (tile % tiles_n) * BN;
        kbeg = z * d.kchunk;
        kend = min(d.K, kbeg + d.kchunk);
    };

    int m0, n0, z, kbeg, kend;
    int w = blockIdx.x;
    if (w >= total_items) return;
    decode(w, m0, n0, z, kbeg, kend);
    uint4 ra[BK / 16], rb[BK / 16];
    g2r<ARC, BK>(ra, A, d.lda, m0, kbeg, d.M, kend, alignA, t);
    g2r<BRC, BK>(rb, B, d.ldb, n0, kbeg, d.N, kend, alignB, t);
    for (;;) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // bias gradient riding on the dW launch: the workgroups of the first tile column add up the columns of the A
        // (= dY^T) slices they stage; 8 columns x (BK/16) k-rows per thread and slice, read back from the thread's own
        // chunks of the LDS image (no hazard), reduced over the 16 k-row groups after the last slice.
        const bool do_cs = ARC && d.colsum != nullptr && n0 == 0;
        float cs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] = 0.f;
        const int wn_ = w + gridDim.x;
        const bool more = PERSIST && wn_ < total_items;
        int m0n = 0, n0n = 0, zn = 0, kbegn = 0, kendn = 0;
        if (more) decode(wn_, m0n, n0n, zn, kbegn, kendn);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            __syncthreads();                   // the previous slice's MFMA reads (or the previous epilogue) are done
            r2s<ARC, BK>(As, ra, t);
            r2s<BRC, BK>(Bs, rb, t);
            __syncthreads();
            {
                if (k0 + BK < kend) {          // next K-slice of this tile
                    g2r<ARC, BK>(ra, A, d.lda, m0, k0 + BK, d.M, kend, alignA, t);
                    g2r<BRC, BK>(rb, B, d.ldb, n0, k0 + BK, d.N, kend, alignB, t);
                } else if (more) {             // first K-slice of the next item: in flight during this item's epilogue
                    g2r<ARC, BK>(ra, A, d.lda, m0n, kbegn, d.M, kendn, alignA, t);
                    g2r<BRC, BK>(rb, B, d.ldb, n0n, kbegn, d.N, kendn, alignB, t);
                }
            }
            if (ARC && do_cs) {
                const int cq = t & 15, kk0 = t >> 4;
#pragma unroll
                for (int p = 0; p < BK / 16; ++p) {
                    const int kk = kk0 + 16 * p;
                    float u[8];
                    unpack8(*reinterpret_cast<const uint4*>(As + kk * 256 + ((cq * 16) ^ ((kk & 3) << 6))), u);
#pragma unroll
                    for (int j = 0; j < 8; ++j) cs[j] += u[j];
                }
            }
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const bf16x8v a0 = frag<ARC, BK>(As, wm * 64, ks, lane), a1 = frag<ARC, BK>(As, wm * 64 + 32, ks, lane);
                const bf16x8v b0 = frag<BRC, BK>(Bs, wn * 64, ks, lane), b1 = frag<BRC, BK>(Bs, wn * 64 + 32, ks, lane);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if (ARC && do_cs) {
            __syncthreads();                   // the last slice's operand reads are done: smem is free
            float* red = reinterpret_cast<float*>(smem);       // [16 k-row groups][128 columns]
            float* mine = red + (t >> 4) * 128 + (t & 15) * 8;
            *reinterpret_cast<float4*>(mine) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4*>(mine + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
            __syncthreads();
            if (t < 128 && m0 + t < d.M) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += red[i * 128 + t];
                d.colsum[(d.splits > 1 ? (size_t)z * d.slab_stride : 0) + m0 + t] = sum;
            }
        }
        epilogue_tile<TO, !(PERSIST && sizeof(TO) == 2)>(d, acc, smem, m0, n0, z, vec_epi, t, wm, wn, kh, l31);
        if (!more) break;
        w = wn_; m0 = m0n; n0 = n0n; z = zn; kbeg = kbegn; kend = kendn;
    }
}

int align_of(const void* p, int ld) {
    if (ld % 8 == 0 && (uintptr_t)p % 16 == 0) return 8;
    if (ld % 4 == 0 && (uintptr_t)p % 8 == 0) return 4;
    return 1;
}

}  // namespace

int mmfm_gemm_big_launch(const mmfm_gemm_desc* dp, hipStream_t st);   // gemm_big.hip: 256 x 256 tiles for the compute-bound shapes
int mmfm_gemm_dw_launch(const mmfm_gemm_desc* dp, hipStream_t st);    // gemm_dw.hip: the HBM-bound weight-gradient stream

int mmfm_gemm_bf16_launch(const mmfm_gemm_desc* dp, hipStream_t st) {
    const mmfm_gemm_desc d = *dp;
    {
        const int rb = mmfm_gemm_big_launch(dp, st);
        if (rb != -1000) return rb;
        const int rw = mmfm_gemm_dw_launch(dp, st);
        if (rw != -1000) return rw;
    }
    MMFM_REQUIRE(d.splits == 1 || d.kchunk % 64 == 0, "mmfm_gemm(bf16): kchunk %d must be a multiple of 64", d.kchunk);
    static const int bk_env = [] { const char* e = getenv("MMFM_GEMM_BK"); return e ? atoi(e) : 0; }();
    // deeper k-tiles when the reduction is short (K = 256/512 block GEMMs): fewer serialized load-latency rounds
    const int kspan = d.splits > 1 ? d.kchunk : d.K;
    // measured on MI355X (scripts/gemm_bench.py): BK = 128 is 0-15 % SLOWER than BK = 64 at 3 workgroups/CU for the
    // K = 256/512 shapes (the kernel is bound by ds_write + L1 + MFMA issue, not by load-latency rounds) -> default 64
    (void)kspan;
    const int BKsel = bk_env == 128 ? 128 : 64;
    const int tiles = cdiv(d.M, BM) * cdiv(d.N, BN);
    const int aA = align_of(d.A, d.lda), aB = align_of(d.B, d.ldb);
    const int total_items = tiles * d.splits;
    static const int wg_per_cu = [] { const char* e = getenv("MMFM_GEMM_WG_PER_CU"); const int v = e ? atoi(e) : 3; return v > 0 ? v : 3; }();
    const bool f32out = d.c_f32 || d.splits > 1;
    // vector epilogue needs 16-B aligned 8-column chunks of every tensor it touches
    auto al16 = [](const void* p) { return p == nullptr || (uintptr_t)p % 16 == 0; };
    const int vec8 = (d.N % 8 == 0) && (d.ldc % 8 == 0) && al16(d.C) && al16(d.pre_out) && al16(d.gradmul_pre) && al16(d.bias) &&
                     (!d.residual || (d.ldr % 8 == 0 && al16(d.residual))) && (!d.splits || d.splits == 1 || d.slab_stride % 4 == 0);
    // half mode: 4-column granularity (N = 668): bf16 rows are 8-B aligned, fp32 rows 16-B aligned
    auto al8 = [](const void* p) { return p == nullptr || (uintptr_t)p % 8 == 0; };
    const bool f32c = d.c_f32 || d.splits > 1;
    const int vec4 = (d.N % 4 == 0) && (d.ldc % 4 == 0) && (f32c ? al16(d.C) : al8(d.C)) && al8(d.pre_out) && al8(d.gradmul_pre) && al16(d.bias) &&
                     (!d.residual || (d.ldr % 4 == 0 && al8(d.residual))) && (!d.splits || d.splits == 1 || d.slab_stride % 4 == 0);
    const int vec = vec8 ? 1 : (vec4 ? 3 : 0);
    // MMFM_GEMM_NT: bit 0 = bf16 output C, bit 1 = saved pre-activation, bit 2 = fp32 output / split-K slabs stored non-temporally, bit 3 = residual / saved
    // pre-activation LOADED non-temporally
    static const int nt_env = [] { const char* e = getenv("MMFM_GEMM_NT"); return e ? atoi(e) : 3; }();
    const int vecf = vec | ((nt_env & 1) ? 4096 : 0) | ((nt_env & 2) ? 8192 : 0) | ((nt_env & 4) ? 16384 : 0) | ((nt_env & 8) ? 32768 : 0);
    // bf16-output launches without epilogue operands and with a short reduction are persistent too (the next tile's first
    // K-slice is in flight during the epilogue): qkv x.W^T 160 -> 151 us, its dX 139 -> 131 us in the B = 1024 step; with
    // K >= 1336 (token embedding) the prefetch registers spill and it measured 30 % SLOWER, hence the K bound.
    // MMFM_GEMM_PERSIST_BF16 = 0 turns it off.
    static const int persist_bf16 = [] { const char* e = getenv("MMFM_GEMM_PERSIST_BF16"); return e ? atoi(e) : 1; }();
    const bool pb = !f32out && persist_bf16 && !d.gradmul_pre && !d.residual && !d.pre_out && (vec & 1) && d.K <= 768;
    dim3 grid((f32out || pb) ? std::min(total_items, 256 * wg_per_cu) : total_items), block(NTHREADS);   // persistent: 3 resident workgroups per CU
    // occupancy probe: unused dynamic LDS bytes per workgroup (60000 -> one workgroup per CU less, 100000 -> one per CU)
    static const int pad_lds = [] { const char* e = getenv("MMFM_GEMM_PAD_LDS"); return e ? atoi(e) : 0; }();
#define LAUNCH2(ARC, BRC, BKV)                                                                                    \
    if (f32out) hipLaunchKernelGGL((gemm_bf16_kernel<ARC, BRC, float, BKV>), grid, block, pad_lds, st, d, aA, aB, vecf, total_items);  \
    else if (pb) hipLaunchKernelGGL((gemm_bf16_kernel<ARC, BRC, uint16_t, BKV, true>), grid, block, pad_lds, st, d, aA, aB, vecf, total_items); \
    else hipLaunchKernelGGL((gemm_bf16_kernel<ARC, BRC, uint16_t, BKV>), grid, block, pad_lds, st, d, aA, aB, vecf, total_items);
#define LAUNCH(ARC, BRC) if (BKsel == 128) { LAUNCH2(ARC, BRC, 128) } else { LAUNCH2(ARC, BRC, 64) }
    if (d.a_kcontig && d.b_kcontig) { LAUNCH(false, false) }
    else if (d.a_kcontig && !d.b_kcontig) { LAUNCH(false, true) }
    else { LAUNCH(true, true) }
#undef LAUNCH
#undef LAUNCH2
    MMFM_LAUNCH_CHECK("mmfm_gemm(bf16)");
    return 0;
}
