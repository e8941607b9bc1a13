// Compute-bound bf16 GEMM behind mmfm_gemm: C[M,N] = epi(A[M,K] . B[N,K]^T), both operands reduction-contiguous (an nn.Linear
// forward, or its dX against the transposed weight copy), K a multiple of 64 and >= 512.  Reference sites: the token embedding
// 668 -> 1336 -> 256 (encoder_embeddings.py:50-54, decoder_embeddings.py:50-54) and every linear of the d_model-512 configuration
// (mm_utils.py:46-52,88-95).  The 128 x 128 kernel of gemm_bf16.hip stays for the K = 256 shapes (HBM-bound) and the dW products.
//
// Structure (cdna_hip_programming.md section 5, "glds vs register staging" / "The 256^2 8-phase template", in its two-buffer form):
//   * 256 x 256 x 64 tiles, 512 threads = 8 waves, ONE workgroup per CU (160 KB of LDS: 2 x (32 + 32) KB operand buffers +
//     8 x 4 KB epilogue staging), persistent: a workgroup walks tiles w = blockIdx.x, += gridDim.x in the XCD-aware order.
//   * operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  The image is lane-linear
//     per wave instruction (8 rows x 128 B); the bank swizzle - 16-B chunk c of row r sits at chunk c ^ ((r >> 1) & 7), which makes
//     the ds_read_b128 operand reads of 32 consecutive rows conflict-free - is applied to the per-lane SOURCE address.
//   * one flat pipeline over (tile, k-tile) pairs g = 0, 1, 2, ...: k-tile g is multiplied out of buffer g & 1 while k-tile g+1 is
//     in flight; the loads of g+2 are issued right behind the barrier that frees the buffer.  At a tile's last k-tile the loads in
//     flight belong to the NEXT output tile, so its first two k-tiles land under this tile's epilogue.  Waits are counted
//     (s_waitcnt vmcnt(8): this wave's eight loads of g+1 may stay in flight), barriers are raw s_barrier.
//   * the product is computed TRANSPOSED (rowchain.h): weight rows are the MFMA A operand, token rows the B operand, so an
//     accumulator tile has the token on the lane and four consecutive features per register group - bias / activation / dropout /
//     residual run in that layout and the output leaves as whole 128-B lines through a 4 KB per-wave staging area.
//     Wave w owns features 128 (w & 1) .. + 127 and tokens 64 (w >> 1) .. + 63: 4 x 2 accumulator tiles (128 VGPRs), 6 operand
//     reads per 8 MFMAs.
#include "rowchain.h"
#include <stdlib.h>
#include <algorithm>

using namespace rowchain;

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int TB = 256, BK = 64, NT = 512;
constexpr int TILE_B = TB * BK * 2;            // one operand tile: 32 KB
constexpr int STG0 = 4 * TILE_B;               // staging behind the two double-buffered operand pairs
constexpr int LDS_ALL = STG0 + 8 * STG_BYTES;  // 160 KB

__device__ __forceinline__ opnd tfrag(const char* tile, int row, int c) {
    return as_opnd(*reinterpret_cast<const uint4*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)));
}

__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

struct BigArgs {
    mmfm_gemm_desc d;
    int tiles_n, ntiles, nk;
    int nt_c, nt_pre;            // non-temporal stores for C / pre_out
};

// the eight LDS-DMA loads of one k-tile: thread t moves 16-B chunk (t & 7) ^ ((t >> 4) & 7) of rows 64 i + (t >> 3), i = 0..3,
// of the token tile and of the weight tile (row pointers are per-thread constants of the output tile; k advances by 64 elements)
__device__ __forceinline__ void issue_ktile(char* buf, const uint16_t* const (&pa)[4], const uint16_t* const (&pb)[4], int koff, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pa[i] + koff), (lds_void_t*)(buf + (i * 512 + 64 * wave) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(pb[i] + koff), (lds_void_t*)(buf + TILE_B + (i * 512 + 64 * wave) * 16), 16, 0, 0);
}

// operand reads as inline asm: behind a plain LDS load hipcc waits vmcnt(0) for every LDS-DMA in flight (it cannot tell which
// buffer the load touches) and the prefetch of the next k-tile would drain at the top of every k-tile.  The reads are ordered by
// the hand-placed waits below; `off` is an immediate (tile, row block), the per-lane address carries buffer, row and swizzled chunk.
#define LDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LDS_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(NT, 2) void gemm_big_kernel(const BigArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const mmfm_gemm_desc& d = a.d;
    const int t = threadIdx.x, lane = t & 63, m = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wnh = wave & 1, wmq = wave >> 1;
    char* stg = smem + STG0 + wave * STG_BYTES;
    const uint16_t* A = reinterpret_cast<const uint16_t*>(d.A);
    const uint16_t* B = reinterpret_cast<const uint16_t*>(d.B);
    const int nk = a.nk;
    const int my_tiles = blockIdx.x < a.ntiles ? (a.ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    if (my_tiles == 0) return;
    const int G = my_tiles * nk;

    // per-thread source rows of the tile that is being LOADED (runs up to two k-tiles ahead of the tile being multiplied)
    const int ldrow = t >> 3, ldc8 = 8 * ((t & 7) ^ ((t >> 4) & 7));
    const uint16_t* pa[4];
    const uint16_t* pb[4];
    auto set_rows = [&](int it) {
        const int tile = xcd_order((int)blockIdx.x + it * (int)gridDim.x, a.ntiles);
        const int m0 = (tile / a.tiles_n) * TB, n0 = (tile % a.tiles_n) * TB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pa[i] = A + (size_t)min(m0 + 64 * i + ldrow, d.M - 1) * d.lda + ldc8;       // rows beyond the matrix: clamped, results dropped
            pb[i] = B + (size_t)min(n0 + 64 * i + ldrow, d.N - 1) * d.ldb + ldc8;
        }
    };
    int ld_g = 0, ld_it = 0, ld_kt = 0;           // next (tile, k-tile) to request
    auto issue_next = [&]() {
        if (ld_g >= G) return false;
        if (ld_kt == 0) set_rows(ld_it);
        issue_ktile(smem + (ld_g & 1) * 2 * TILE_B, pa, pb, ld_kt * BK, wave);
        ++ld_g;
        if (++ld_kt == nk) { ld_kt = 0; ++ld_it; }
        return true;
    };
    issue_next();
    bool tail1 = issue_next();                    // is a younger k-tile in flight behind the one we are about to wait for?

    const Drop dr = drop_init(d.drop);
    const GBuf Cb = gbuf(d.C, (int64_t)d.M * d.ldc * 2), Pb = gbuf(d.pre_out, (int64_t)d.M * d.ldc * 2);
    const GBuf Gb = gbuf(d.gradmul_pre, (int64_t)d.M * d.ldc * 2), Rb = gbuf(d.residual, (int64_t)d.M * d.ldr * 2);

    // operand read addresses: row 128 wnh + 32 i + m (weights) / 64 wmq + 32 j + m (tokens); the swizzle term of a row depends on
    // m alone (32 i, 64 wmq, 128 wnh are multiples of 16), so chunk 2 ks + h of every row block sits at the same per-lane offset
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t sw = (uint32_t)((m >> 1) & 7);
    const uint32_t xrow = lds0 + (uint32_t)(64 * wmq + m) * 128u, wrow = lds0 + (uint32_t)(128 * wnh + m) * 128u;
    uint32_t coff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) coff[ks] = (((uint32_t)(2 * ks + h)) ^ sw) << 4;

    f32x16 acc[4][2];
    int g = 0;
    for (int it = 0; it < my_tiles; ++it) {
        const int tile = xcd_order((int)blockIdx.x + it * (int)gridDim.x, a.ntiles);
        const int m0 = (tile / a.tiles_n) * TB, n0 = (tile % a.tiles_n) * TB;
        // the accumulators start at the bias (feature of register r of tile i: n0 + 128 wnh + 32 i + 8 (r >> 2) + 4 h + (r & 3))
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 z = zero16();
            if (d.bias) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + 128 * wnh + 32 * i + 8 * q + 4 * h;
                    if (n < d.N) {
                        const float4 b4 = *reinterpret_cast<const float4*>(d.bias + n);
                        z[4 * q + 0] = b4.x; z[4 * q + 1] = b4.y; z[4 * q + 2] = b4.z; z[4 * q + 3] = b4.w;
                    }
                }
            }
            acc[i][0] = z;
            acc[i][1] = z;
        }
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // k-tile g has landed in every wave's eyes: own loads retired (the eight of g+1 may stay in flight), then the barrier.
            // (the bias loads above were waited for by hipcc with vmcnt(0): the first k-tile of a tile sees an empty queue anyway)
            if (tail1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t boff = (uint32_t)(g & 1) * (2u * TILE_B);
            // two operand sets, ping-pong: the reads of k-step ks+1 are in flight under the eight MFMAs of k-step ks
            uint4 w0[4], x0[2], w1[4], x1[2];
#define READ_SET(W, X, KS)                                                              \
            {                                                                           \
                const uint32_t aw_ = wrow + boff + coff[KS], ax_ = xrow + boff + coff[KS]; \
                LDS_READ(W[0], aw_, TILE_B); LDS_READ(X[0], ax_, 0); LDS_READ(X[1], ax_, 4096);     \
                LDS_READ(W[1], aw_, TILE_B + 4096); LDS_READ(W[2], aw_, TILE_B + 8192); LDS_READ(W[3], aw_, TILE_B + 12288); \
            }
#define MMA_SET(W, X)                                                                   \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                               \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma(as_opnd(W[i]), as_opnd(X[j]), acc[i][j]);
            READ_SET(w0, x0, 0)
            LDS_WAIT(0);
            READ_SET(w1, x1, 1)
            MMA_SET(w0, x0)
            LDS_WAIT(0);
            READ_SET(w0, x0, 2)
            MMA_SET(w1, x1)
            LDS_WAIT(0);
            READ_SET(w1, x1, 3)
            MMA_SET(w0, x0)
            LDS_WAIT(0);
            // every operand read of this buffer has returned: behind the barrier the buffer is free for k-tile g + 2
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            tail1 = issue_next();
            MMA_SET(w1, x1)
#undef READ_SET
#undef MMA_SET
        }
        // ---------------- epilogue of tile (m0, n0): the next tile's first two k-tiles are in flight meanwhile
        // (hipcc drains vmcnt around the ordinary loads / stores below, which costs the overlap in the gradmul / residual cases only).
        // One sweep per operation over the two accumulator tiles of a 128-B line group keeps the live registers at the accumulators
        // plus one staged tile.
        const uint32_t ldcb = d.ldc * 2, ldrb = d.ldr * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t wrow0 = (uint32_t)(m0 + 64 * wmq + 32 * j);
            const uint32_t row = wrow0 + m;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {                 // pairs of feature tiles = one 128-B line group
                const int ncol0 = n0 + 128 * wnh + 64 * ip;   // first feature of the pair
                const int nchunk = min(8, max(0, (d.N - ncol0) >> 3));
                if (nchunk <= 0) continue;
                if (d.pre_out) {
                    stage_tile(stg, 0, m, h, acc[2 * ip][j]);
                    stage_tile(stg, 1, m, h, acc[2 * ip + 1][j]);
                    if (a.nt_pre) flush_lines<true>(stg, Pb, wrow0, ldcb, (uint32_t)ncol0 * 2u, lane, nchunk);
                    else flush_lines<false>(stg, Pb, wrow0, ldcb, (uint32_t)ncol0 * 2u, lane, nchunk);
                }
                if (d.act == 1) { gelu16(acc[2 * ip][j]); gelu16(acc[2 * ip + 1][j]); }
                else if (d.act == 2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[2 * ip + e][j][r];
                            acc[2 * ip + e][j][r] = v * __builtin_amdgcn_rcpf(1.f + fabsf(v)) * d.act_scale;      // v_rcp_f32: 1 ulp, far below bf16
                        }
                } else if (d.gradmul_pre) {
                    const Lines L = fetch_lines(Gb, wrow0, ldcb, (uint32_t)ncol0 * 2u, lane);
                    stage_lines(stg, L, lane);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x16 u = unstage_tile(stg, e, m, h);
                        f32x16& v = acc[2 * ip + e][j];
                        if (d.act == 3) {
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                mmfm_f32x2 uu; uu.x = u[r]; uu.y = u[r + 1];
                                uu = gelu_grad2(uu);
                                v[r] *= uu.x; v[r + 1] *= uu.y;
                            }
                        } else if (d.act == 4) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float rd = __builtin_amdgcn_rcpf(1.f + fabsf(u[r]));
                                v[r] *= rd * rd * d.act_scale;
                            }
                        } else {
                            const float inv_s = 1.f / d.act_scale;
#pragma unroll
                            for (int r = 0; r < 16; ++r) v[r] *= softsign_grad_from_out(u[r], inv_s) * d.act_scale;
                        }
                    }
                }
                if (dr.on()) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            float v0 = acc[2 * ip + e][j][r], v1 = acc[2 * ip + e][j][r + 1];
                            dr.apply2(v0, v1, (uint64_t)row * (uint64_t)d.N + (uint64_t)(ncol0 + 32 * e + 8 * (r >> 2) + 4 * h + (r & 3)));
                            acc[2 * ip + e][j][r] = v0; acc[2 * ip + e][j][r + 1] = v1;
                        }
                }
                if (d.residual) {
                    const Lines L = fetch_lines(Rb, wrow0, ldrb, (uint32_t)ncol0 * 2u, lane);
                    stage_lines(stg, L, lane);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const f32x16 u = unstage_tile(stg, e, m, h);
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[2 * ip + e][j][r] += u[r];
                    }
                }
                stage_tile(stg, 0, m, h, acc[2 * ip][j]);
                stage_tile(stg, 1, m, h, acc[2 * ip + 1][j]);
                if (a.nt_c) flush_lines<true>(stg, Cb, wrow0, ldcb, (uint32_t)ncol0 * 2u, lane, nchunk);
                else flush_lines<false>(stg, Cb, wrow0, ldcb, (uint32_t)ncol0 * 2u, lane, nchunk);
            }
        }
        // ordinary loads / stores were issued since the last LDS-DMA: the counted wait of the next k-tile would have to count them too
        tail1 = false;
    }
}

}  // namespace

// returns -1000 when the shape belongs to the 128 x 128 kernel
int mmfm_gemm_big_launch(const mmfm_gemm_desc* dp, hipStream_t st) {
    const mmfm_gemm_desc& d = *dp;
    static const int on = [] { const char* e = getenv("MMFM_GEMM_BIG"); return e ? atoi(e) : 1; }();
    static const int kmin = [] { const char* e = getenv("MMFM_GEMM_BIG_KMIN"); return e ? atoi(e) : 512; }();
    if (!on || d.dtype != MMFM_BF16 || d.c_f32 || d.splits > 1 || !d.a_kcontig || !d.b_kcontig || d.colsum) return -1000;
    if (d.K % BK || d.K < kmin || d.N < 128 || d.M < 1024 || d.N % 8) return -1000;
    // too few 256-wide tiles to occupy the chip (the reference's batch of 16: M = 3,200 rows x N = 256 = 13 tiles, 23-34 us against ~15 us
    // for the 50 tiles of the 128-tile kernel)
    // (MMFM_GEMM_BIG_MIN_TILES is read per call: the kernel's own test lowers it for its small shapes)
    const char* mt = getenv("MMFM_GEMM_BIG_MIN_TILES");
    if (cdiv(d.M, TB) * cdiv(d.N, TB) < (mt ? atoi(mt) : 96)) return -1000;
    auto al16 = [](const void* p) { return p == nullptr || (uintptr_t)p % 16 == 0; };
    if (d.lda % 8 || d.ldb % 8 || d.ldc % 8 || (d.residual && d.ldr % 8) || !al16(d.A) || !al16(d.B) || !al16(d.C) || !al16(d.pre_out) ||
        !al16(d.gradmul_pre) || !al16(d.residual) || !al16(d.bias))
        return -1000;
    if ((int64_t)d.M * std::max(d.ldc, d.ldr) * 2 >= (int64_t)1 << 31) return -1000;        // 32-bit buffer offsets of the epilogue
    if (int rc = mmfm_lds_opt_in(reinterpret_cast<const void*>(gemm_big_kernel), LDS_ALL, "mmfm_gemm(bf16, 256 tile)")) return rc;
    static const int nt_env = [] { const char* e = getenv("MMFM_GEMM_NT"); return e ? atoi(e) : 3; }();
    BigArgs a;
    a.d = d;
    a.tiles_n = cdiv(d.N, TB);
    a.ntiles = cdiv(d.M, TB) * a.tiles_n;
    a.nk = d.K / BK;
    a.nt_c = nt_env & 1;
    a.nt_pre = (nt_env >> 1) & 1;
    hipLaunchKernelGGL(gemm_big_kernel, dim3(std::min(a.ntiles, 256)), dim3(NT), LDS_ALL, st, a);
    MMFM_LAUNCH_CHECK("mmfm_gemm(bf16, 256 tile)");
    return 0;
}
