// Weight-gradient GEMM behind mmfm_gemm: C[M,N] (fp32, one slab per K-split) = A[K,M]^T . B[K,N], both operands ROW-contiguous
// (A = dY, B = the layer input, K = B*L tokens), plus the bias gradient (column sums of A) riding on the same pass.
// Reference sites: the autograd of every nn.Linear on the path (mm_utils.py:46-52,88-95; encoder_embeddings.py:50-54,
// decoder_embeddings.py:50-54): dW = dY^T X, db = sum_rows dY.
//
// The product is HBM-bound (M, N are 256..768 against K = 204,800: 128 flop per byte at best, a quarter of what the MFMA pipe
// could take), so the kernel is built around the stream, not the arithmetic:
//   * 128 x 256 output tile per workgroup (128 x 128 when N <= 128), 4 waves (64 x 128 each), ONE workgroup per CU and - the engine
//     sizes the split count with mmfm_gemm_dw_tiles so - one (tile, K-slab) item per workgroup: 256 slabs of 128 KB per launch are
//     written here and read by mmfm_reduce_slabs, against 32-48 MB each way for the 3-workgroups-per-CU grid of gemm_bf16.hip;
//   * operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass) through a SIX-stage ring
//     of 24 KB k-tiles (32 deep): 120 KB per CU in flight, one s_barrier per k-tile, counted waits (the four younger stages of a wave
//     stay in flight).  The buffer descriptor's bounds check zero-fills rows beyond K and columns beyond the tile, and the ring
//     simply keeps requesting past the slab's end with an out-of-range offset, so the loop has one shape;
//   * LDS image per operand [k][cols] with byte offset XOR ((k & 3) << 6) applied on the SOURCE side of the DMA; the MFMA
//     operands (8 k-strided values per lane) are two ds_read_b64_tr_b16 each, conflict-free, issued as inline asm (behind a plain
//     LDS load hipcc waits vmcnt(0) for every DMA in flight);
//   * items that share a K-slab are neighbours on one XCD: the PMC fetch count is 1.08 x the operand bytes, L2 hit rate 64 %;
//   * column sums: every wave also multiplies its A operands by a ones operand (2 extra MFMAs per k-step; unconditional, because a
//     branch around them made hipcc shuttle the accumulators between VGPRs and AGPRs) and the waves of the first tile column
//     store the result - no VALU pass over the tile.
// Measured (scripts/dw_bench.py, dw_slope.py, dw_pattern.py; B = 1024 shapes, operands beyond the 256 MB Infinity Cache): 3.3-3.8 TB/s of
// operand bytes where several tiles share a slab, 5.5 TB/s with one tile per slab, against 2.9-3.4 TB/s for the 128-tile kernel.
// Tried on top and measured no better (removed): the tiles of a slab walking its k-tiles in rotated order (no concurrent requests
// for a line), two of the four waves - or a fifth wave - issuing nothing but L2 warm-up loads for the slab's full-width k-tiles
// 12-20 steps ahead, 128-wide tiles with an 8-stage ring, 64-deep k-tiles.
#include "common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(8))) __bf16 opnd;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int TM = 128, DNT = 256;
constexpr uint32_t OOB = 0x80000000u;     // beyond every descriptor's num_records (< 2^31, checked on the host): the load returns zeros

// TN = output-tile width (the A panel is 128 wide), BK = k-tile depth; the ring takes what fits in 160 KB
// (MMFM_DW_LDS_KB: diagnostic builds with a smaller ring, scripts/probe/build_dw_lds.sh)
#ifndef MMFM_DW_LDS_KB
#define MMFM_DW_LDS_KB 160
#endif
template <int TN, int BK> struct Geo {
    static constexpr int OPA = BK * TM * 2, OPB = BK * TN * 2, STB = OPA + OPB;
    static constexpr int NST = (MMFM_DW_LDS_KB * 1024) / STB < 8 ? (MMFM_DW_LDS_KB * 1024) / STB : 8;
    static_assert(NST >= 3, "ring needs three stages");
    static constexpr int LDS = NST * STB;
    static constexpr int NA = BK / 16;            // DMA instructions per wave and k-tile, A (1 KB = 4 rows x 256 B each)
    static constexpr int NB = NA * (TN / TM);     // ... and B (1 KB = 512 / TN-bytes rows)
    static constexpr int NJ = TN / 64;            // 32-column accumulator tiles per wave along N (wave grid 2 x 2)
};

struct DwArgs {
    mmfm_gemm_desc d;
    int tiles_n, ntiles, items;
};

__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ opnd mk_opnd(u32x2 lo, u32x2 hi) {
    u32x4 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return __builtin_bit_cast(opnd, v);
}

#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LDS_WAIT0 do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// Two independent weight gradients in ONE launch (mmfm_gemm_pair): workgroups [0, first) take problem p.a, the rest p.b, one item each.
// Each problem then makes half as many K-slabs as it would alone on the whole chip: half the slab bytes written here and read by the
// reductions (256 x 128 KB per launch either way), one launch less.
struct DwPair {
    DwArgs a, b;
    int first;           // items of problem a (a multiple of 8: the XCD-aware item order works per problem); 0 = single launch of a
};

template <int TN, int BK>
__global__ __launch_bounds__(DNT) void gemm_dw_kernel(const DwPair pr) {
    const bool second = pr.first > 0 && (int)blockIdx.x >= pr.first;
    const DwArgs& a = second ? pr.b : pr.a;
    const int w0 = second ? (int)blockIdx.x - pr.first : (int)blockIdx.x;
    const int wstep = pr.first > 0 ? (1 << 30) : (int)gridDim.x;
    typedef Geo<TN, BK> G;
    constexpr int NLW = 4;                                       // every wave issues its quarter of the DMA loads
    constexpr int NST = G::NST, NJ = G::NJ, RBB = TN * 2;       // B image row bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mmfm_gemm_desc& d = a.d;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t ldaB = (uint32_t)d.lda * 2u, ldbB = (uint32_t)d.ldb * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.A), 0, (int)((int64_t)d.K * ldaB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(d.B), 0, (int)((int64_t)d.K * ldbB), 0x00020000);

    // DMA source: one wave instruction fills 1 KB of an image = RPI k-rows; lane l sits at (k-row l / PPR, 16-B slot l % PPR) and
    // fetches the source chunk that the swizzle (slot ^ ((k & 3) << 2)) maps to that slot.  Wave w moves row blocks NW j + w.
    const int dkA = lane >> 4, dcA = (lane & 15) ^ (dkA << 2);
    constexpr int PPR = TN / 8, RPI = 64 / PPR;                  // B: pieces per row, rows per instruction (4 or 2)
    const int dkB = lane / PPR;
    const int dcB = (lane % PPR) ^ ((((RPI * wave) & 3) + dkB) << 2);        // (k & 3) of block NLW j + w, row dkB: RPI (NLW j + w) + dkB
    // operand reads (ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(col) block, lane 4q+p supplies k-row q's address at columns
    // 4p..4p+3, lane i receives column i with its four k values): group g -> columns 16 (g & 1), k 8 (g >> 1)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t ra[2], rb[NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = lds0 + (uint32_t)((8 * (g >> 1) + q) * 256 + (((128 * wm + 64 * i) ^ (q << 6)) + 32 * (g & 1) + 8 * p));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        rb[j] = lds0 + (uint32_t)(G::OPA + (8 * (g >> 1) + q) * RBB + (((TN * wn + 64 * j) ^ (q << 6)) + 32 * (g & 1) + 8 * p));
    u32x4 ones4;
    ones4[0] = ones4[1] = ones4[2] = ones4[3] = 0x3F803F80u;
    const opnd ones = __builtin_bit_cast(opnd, ones4);

    for (int w = w0; w < a.items; w += wstep) {
        const int item = xcd_order(w, a.items);           // = z * ntiles + tile: neighbours on an XCD share the K-slab
        const int z = item / a.ntiles, tile = item - z * a.ntiles;
        const int m0 = (tile / a.tiles_n) * TM, n0 = (tile % a.tiles_n) * TN;
        const int kbeg = d.splits > 1 ? z * d.kchunk : 0;
        const int kend = d.splits > 1 ? min(d.K, kbeg + d.kchunk) : d.K;
        const int nk = (kend - kbeg + BK - 1) / BK;
        const bool do_cs = d.colsum != nullptr && n0 == 0 && wn == 0;

        const uint32_t srcA = (m0 + 8 * dcA < d.M) ? (uint32_t)dkA * ldaB + (uint32_t)(m0 + 8 * dcA) * 2u : OOB;
        const uint32_t srcB = (n0 + 8 * dcB < d.N) ? (uint32_t)dkB * ldbB + (uint32_t)(n0 + 8 * dcB) * 2u : OOB;
        // stage kt % NST of the ring <- k-tile kt (zeros once kt >= nk)
        auto issue = [&](int kt) {
            char* buf = smem + (kt % NST) * G::STB;
            const bool live = kt < nk;
            const uint32_t k0 = (uint32_t)(kbeg + kt * BK);
#pragma unroll
            for (int j = 0; j < G::NA; ++j) {
                const uint32_t kr = k0 + (uint32_t)(4 * (NLW * j + wave));
                const uint32_t va = live ? srcA + kr * ldaB : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t*)(buf + (NLW * j + wave) * 1024), 16, (int)va, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < G::NB; ++j) {
                const uint32_t kr = k0 + (uint32_t)(RPI * (NLW * j + wave));
                const uint32_t vb = live ? srcB + kr * ldbB : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t*)(buf + G::OPA + (NLW * j + wave) * 1024), 16, (int)vb, 0, 0, 0);
            }
        };
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue(s);

        f32x16 acc[2][NJ], cs[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                cs[i][r] = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j][r] = 0.f;
            }
        }

        for (int kt = 0; kt < nk; ++kt) {
            vm_wait<(NST - 2) * (G::NA + G::NB)>();                  // this wave's share of k-tile kt has landed (the younger stages may be in flight)
            __builtin_amdgcn_s_barrier();                            // ... and everybody's; every wave is done reading k-tile kt-1
            __builtin_amdgcn_sched_barrier(0);
            issue(kt + NST - 1);                                     // into the buffer of k-tile kt-1
            const uint32_t sb = (uint32_t)(kt % NST) * G::STB;
            uint32_t pa[2], pb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) pa[i] = ra[i] + sb;
#pragma unroll
            for (int j = 0; j < NJ; ++j) pb[j] = rb[j] + sb;
            u32x2 FA[2][2][2], FB[2][NJ][2];                         // [set][tile][lo / hi]
#define READ_SET(S, KS)                                                                                   \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                               \
                TR_READ(FA[S][i][0], pa[i], (KS) * 16 * 256); TR_READ(FA[S][i][1], pa[i], (KS) * 16 * 256 + 4 * 256); \
            }                                                                                             \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                              \
                TR_READ(FB[S][j][0], pb[j], (KS) * 16 * RBB); TR_READ(FB[S][j][1], pb[j], (KS) * 16 * RBB + 4 * RBB); \
            }
#define MMA_SET(S)                                                                                        \
            {                                                                                             \
                const opnd fa0 = mk_opnd(FA[S][0][0], FA[S][0][1]), fa1 = mk_opnd(FA[S][1][0], FA[S][1][1]); \
                _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                          \
                    const opnd fb = mk_opnd(FB[S][j][0], FB[S][j][1]);                                    \
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0][j], 0, 0, 0);     \
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1][j], 0, 0, 0);     \
                }                                                                                         \
                cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, ones, cs[0], 0, 0, 0);               \
                cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, ones, cs[1], 0, 0, 0);               \
            }
            READ_SET(0, 0)
            LDS_WAIT0;
            READ_SET(1, 1)
            MMA_SET(0)
            LDS_WAIT0;
            if (BK == 64) {
                READ_SET(0, 2)
                MMA_SET(1)
                LDS_WAIT0;
                READ_SET(1, 3)
                MMA_SET(0)
                LDS_WAIT0;
            }
            MMA_SET(1)
#undef READ_SET
#undef MMA_SET
        }
        // the zero-fill requests past the slab's end still target the ring: drain them before the next item's first k-tiles
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        // ---- slab store: register r of tile (i, j) = row m0 + 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 h, column n0 + (TN/2) wn + 32 j + lane & 31:
        // a half-wave writes one whole 128-B line per register
        float* Cf = reinterpret_cast<float*>(d.C) + (d.splits > 1 ? (size_t)z * d.slab_stride : 0);
        const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + (TN / 2) * wn + 32 * j + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < d.M && n < d.N) Cf[(size_t)m * d.ldc + n] = acc[i][j][r];
                }
            }
        if (do_cs && l31 == 0) {
            float* csum = d.colsum + (d.splits > 1 ? (size_t)z * d.slab_stride : 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < d.M) csum[m] = cs[i][r];
                }
        }
    }
}

template <int TN>
DwArgs make_args(const mmfm_gemm_desc& d) {
    DwArgs a;
    a.d = d;
    a.tiles_n = cdiv(d.N, TN);
    a.ntiles = cdiv(d.M, TM) * a.tiles_n;
    a.items = a.ntiles * std::max(1, d.splits);
    return a;
}

template <int TN, int BK>
int launch(const mmfm_gemm_desc& d, const mmfm_gemm_desc* d2, hipStream_t st) {
    typedef Geo<TN, BK> G;
    if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(gemm_dw_kernel<TN, BK>), G::LDS, "mmfm_gemm(bf16, dW stream)")) return rc;
    DwPair p;
    p.a = make_args<TN>(d);
    p.b = p.a;
    p.first = 0;
    int grid = std::min(p.a.items, 256);
    if (d2) {
        p.b = make_args<TN>(*d2);
        p.first = p.a.items;
        grid = p.a.items + p.b.items;
    }
    hipLaunchKernelGGL((gemm_dw_kernel<TN, BK>), dim3(grid), dim3(DNT), G::LDS, st, p);
    MMFM_LAUNCH_CHECK("mmfm_gemm(bf16, dW stream)");
    return 0;
}

}  // namespace

// 256-wide tiles halve the operand re-reads of a long stream; a short reduction (the reference's batch of 16: K = 3,200) wants the
// parallelism of twice as many 128-wide tiles instead (B = 16 step: 4.92 -> 4.64 ms)
static bool dw_wide(int N, int K) {
    static const int tn_env = [] { const char* e = getenv("MMFM_GEMM_DW_TN"); return e ? atoi(e) : 0; }();
    if (tn_env == 128 || N <= 128) return false;
    return tn_env == 256 || K >= 32768;
}

// number of (tile) items per K-slab the streaming kernel makes of an [M, N] gradient over K rows (the engine sizes the split count with it)
extern "C" int mmfm_gemm_dw_tiles(int M, int N, int K) { return cdiv(M, TM) * cdiv(N, dw_wide(N, K) ? 256 : 128); }

static bool dw_eligible(const mmfm_gemm_desc& d) {
    static const int on = [] { const char* e = getenv("MMFM_GEMM_DW"); return e ? atoi(e) : 1; }();
    const bool f32out = d.c_f32 || d.splits > 1;
    if (!on || d.dtype != MMFM_BF16 || !f32out || d.a_kcontig || d.b_kcontig) return false;
    if (d.bias || d.pre_out || d.gradmul_pre || d.residual || d.act || (d.drop.p > 0.f)) return false;
    // 16-B pieces: rows 16-B aligned; a ragged last piece (M or N not a multiple of 8) must still lie inside its row (padded leading dimension) -
    // the columns it adds are computed and dropped
    if (d.lda % 8 || d.ldb % 8 || d.lda < (d.M + 7) / 8 * 8 || d.ldb < (d.N + 7) / 8 * 8 || ((uintptr_t)d.A & 15) || ((uintptr_t)d.B & 15) || ((uintptr_t)d.C & 3)) return false;
    if (d.splits > 1 && (d.kchunk % 64 || d.kchunk <= 0)) return false;
    if (((int64_t)d.K + 9 * 64) * std::max(d.lda, d.ldb) * 2 >= (int64_t)1 << 31) return false;          // 32-bit buffer offsets, ring run-out included
    return true;
}

// returns -1000 when the launch belongs to the general kernel of gemm_bf16.hip
int mmfm_gemm_dw_launch(const mmfm_gemm_desc* dp, hipStream_t st) {
    const mmfm_gemm_desc& d = *dp;
    if (!dw_eligible(d)) return -1000;
    return dw_wide(d.N, d.K) ? launch<256, 32>(d, nullptr, st) : launch<128, 32>(d, nullptr, st);
}

// both descriptors in one launch when both belong to the streaming kernel with the same tile width, every item fits the grid once and the
// first problem's item count is a multiple of 8; -1000 otherwise (the caller then issues them one after the other)
int mmfm_gemm_dw_pair_launch(const mmfm_gemm_desc* ap, const mmfm_gemm_desc* bp, hipStream_t st) {
    const mmfm_gemm_desc &a = *ap, &b = *bp;
    if (!dw_eligible(a) || !dw_eligible(b)) return -1000;
    const bool wide = dw_wide(a.N, a.K);
    if (wide != dw_wide(b.N, b.K)) return -1000;
    const int ia = mmfm_gemm_dw_tiles(a.M, a.N, a.K) * std::max(1, a.splits), ib = mmfm_gemm_dw_tiles(b.M, b.N, b.K) * std::max(1, b.splits);
    if (ia % 8 || ia + ib > 512) return -1000;
    return wide ? launch<256, 32>(a, &b, st) : launch<128, 32>(a, &b, st);
}
