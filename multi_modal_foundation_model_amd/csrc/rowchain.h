// Row-owner building blocks (bf16 throughput mode, model width 256) shared by rowgemm.hip and mlp_fused.hip.
//
// Every transformer op on the path except attention is row-local, and the rows are many (R = B*L = 204,800 at the bench
// batch) while the weights are tiny (<= 0.4 MB per matrix).  So a wavefront OWNS 32 token rows: it keeps them in registers
// as MFMA operands / accumulators through a whole chain of ops (LayerNorm -> GEMM -> activation -> GEMM -> residual ...),
// and the weights stream past it through LDS in 16 KB chunks shared by the workgroup's waves.  Activations cross HBM once
// per chain instead of once per op, and a row's statistics (LayerNorm forward and backward) never leave its wave.
//
// Orientation: every product is computed TRANSPOSED, D[n][m] = sum_k W[n][k] * X[m][k], with the weight rows as the MFMA
// A operand and the token rows as the B operand.  A 32x32 accumulator tile then has the token on the lane (m = lane & 31)
// and, in its 16 registers, n = 8*(r>>2) + 4*(lane>>5) + (r&3): four CONSECUTIVE output features per register group, so
//   * a row's full output (all n) lives in lanes m and m+32 -> row statistics are in-lane sums plus one lane exchange;
//   * outputs leave as 16-B row-major pieces (v_permlane32_swap pairs the two half-waves' 8-B groups);
//   * the tile, converted to bf16, IS the B operand of the next product that sums over n (cdna_hip_programming.md S3,
//     "An accumulator tile as the next MFMA's operand"): the chain needs no LDS round trip and no lane movement.  Its k
//     order inside a 16-deep step is permuted (element j of lane half h = k 8*(j>>2) + 4*h + (j&3)); the weight chunks
//     that multiply such operands are staged with the same permutation (PERM images below).
//
// Weight rings.  The row GEMMs (rowgemm.hip: two workgroups per CU, HBM-bound) use the register-staged ring described next; the MLP
// kernels (mlp_fused.hip: one workgroup per CU, every latency exposed) the asynchronous LDS-DMA ring further down.
// Register-staged ring: two 16 KB LDS slots per workgroup.  Chunk c+2 is in flight from L2 in staging registers while chunk c+1
// is written to the free slot and chunk c is multiplied; ONE __syncthreads() per chunk (16 MFMAs per wave).  All loads
// are ordinary loads, so hipcc's own s_waitcnt bookkeeping orders everything - and to let it COUNT (vmcnt(N), not
// vmcnt(0): the vector-memory counter retires in order, so a conservative wait for a weight chunk would also wait for
// the output stores issued after it) the chunk loops are straight-line code: activations move through raw buffer
// loads / stores whose bounds check replaces every `if (row < R)` (out-of-range loads return 0, stores are dropped, a
// NULL optional tensor is a zero-sized buffer), chunk indices are clamped instead of guarded, and biases live in LDS.
#pragma once
#include "common.h"

namespace rowchain {

typedef __attribute__((ext_vector_type(8))) __bf16 opnd;      // one MFMA A/B operand: 8 bf16 (4 VGPRs)
constexpr int CHUNK = 16384;                                    // bytes per weight chunk / LDS slot
constexpr int LDS_BYTES = 2 * CHUNK;

// Where a chunk comes from.  kind 0: [32 rows][256 k] read by operands in natural k order;  kind 1: the same block for
// operands that were accumulator tiles (PERM);  kind 2: [256 rows][32 k] PERM (all eight 32-row tiles x one 32-deep k slice).
struct WChunk {
    const uint16_t* base;      // element (row 0, k 0) of the block; rows are K-contiguous, 16-B aligned
    int ld;                    // row stride in elements (multiple of 8)
    int kind;                  // all 32 (kind 0, 1) / 256 (kind 2) rows of the block exist
};

__device__ __forceinline__ opnd as_opnd(uint4 v) { return __builtin_bit_cast(opnd, v); }
__device__ __forceinline__ uint4 as_u4(opnd v) { return __builtin_bit_cast(uint4, v); }

// ---- bounds-checked activation tensors (raw buffer descriptors; byte offsets < 2^31, checked on the host)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
struct GBuf { __amdgpu_buffer_rsrc_t rs; };
__device__ __forceinline__ GBuf gbuf(const void* p, int64_t bytes) {
    GBuf b;
    b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, p ? (int)bytes : 0, 0x00020000);
    return b;
}
__device__ __forceinline__ uint4 ld16(const GBuf& b, uint32_t off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b.rs, (int)off, 0, 0));
}
template <bool NTS>
__device__ __forceinline__ void st16(const GBuf& b, uint32_t off, uint4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b.rs, (int)off, 0, NTS ? 2 : 0);
}
__device__ __forceinline__ float ld4f(const GBuf& b, uint32_t off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.rs, (int)off, 0, 0));
}
__device__ __forceinline__ void st4f(const GBuf& b, uint32_t off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), b.rs, (int)off, 0, 0);
}

// NT threads move one chunk (1024 16-B pieces) global -> registers -> LDS slot.
// At one wave per SIMD every vector instruction costs 4 cycles, and the first version spent ~1,100 cycles per ring step on
// address arithmetic alone (measured: scripts/probe/mlp_stamp.py).  So: the loads go through ONE buffer descriptor per chunk
// (wave-uniform base) with a per-thread byte offset that depends only on the chunk kind and row stride plus a SCALAR offset
// per piece; the LDS addresses are two per-thread constants per kind (the XOR swizzles alternate between two values as the
// piece index steps through the rows) plus immediates.
template <int NT>
__device__ __forceinline__ void stage_load(uint4 (&r)[1024 / NT], const WChunk& c, int t) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(c.base), 0, 0x7fffffff, 0x00020000);
    const int ldb = c.ld * 2;
    if (c.kind == 2) {                     // [256 rows][32 k]: piece p = t + NT q -> row p / 4 = t / 4 + (NT / 4) q, col t % 4
        const int v0 = (t >> 2) * ldb + (t & 3) * 16;
#pragma unroll
        for (int q = 0; q < 1024 / NT; ++q) r[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, v0, q * (NT / 4) * ldb, 0));
    } else {                               // [32 rows][256 k]: row t / 32 + (NT / 32) q, col t % 32
        const int v0 = (t >> 5) * ldb + (t & 31) * 16;
#pragma unroll
        for (int q = 0; q < 1024 / NT; ++q) r[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, v0, q * (NT / 32) * ldb, 0));
    }
}
template <int NT>
__device__ __forceinline__ void stage_write(const uint4 (&r)[1024 / NT], int kind, char* slot, int t) {
    constexpr int PP = 1024 / NT;
    if (kind == 0) {                         // 512-B rows, 16-B chunk index XOR (row & 15): conflict-free ds_read_b128
        // row = r0 + (NT/32) q with r0 = t / 32 < NT / 32 <= 16: (row & 15) = r0 ^ ((NT/32) q & 15) when NT/32 is 8 or 16
        const int r0 = t >> 5, col = t & 31;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int row = r0 + (NT / 32) * q;
            *reinterpret_cast<uint4*>(slot + row * 512 + ((col ^ (row & 15)) << 4)) = r[q];
        }
    } else if (kind == 1) {                  // PERM: 2x2 transpose of the 8-B units of each aligned 32-B pair
        const int r0 = t >> 5, col = t & 31, c0 = col & ~1, e = col & 1;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int row = r0 + (NT / 32) * q;
            *reinterpret_cast<uint2*>(slot + row * 512 + ((c0 ^ (row & 15)) << 4) + 8 * e) = make_uint2(r[q].x, r[q].y);
            *reinterpret_cast<uint2*>(slot + row * 512 + (((c0 + 1) ^ (row & 15)) << 4) + 8 * e) = make_uint2(r[q].z, r[q].w);
        }
    } else {                                 // 64-B rows, chunk index XOR ((row >> 2) & 3), PERM
        // row = t / 4 + (NT/4) q: (row >> 2) & 3 = (t >> 4) & 3 for NT/4 a multiple of 16 -> ONE swizzle value per thread
        const int r0 = t >> 2, col = t & 3, c0 = col & 2, e = col & 1, sw = (r0 >> 2) & 3;
        char* a0 = slot + r0 * 64 + ((c0 ^ sw) << 4) + 8 * e;
        char* a1 = slot + r0 * 64 + (((c0 + 1) ^ sw) << 4) + 8 * e;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            *reinterpret_cast<uint2*>(a0 + q * (NT / 4) * 64) = make_uint2(r[q].x, r[q].y);
            *reinterpret_cast<uint2*>(a1 + q * (NT / 4) * 64) = make_uint2(r[q].z, r[q].w);
        }
    }
}

// Two-slot weight ring (see the header comment) as macros over plain locals of the kernel (a struct holding the staging
// registers ended up in scratch memory): `SRC(g)` maps the workgroup's g-th chunk to its WChunk; chunk indices are clamped
// to the last one so that the loop body has no guards (the surplus copies at the very end are never read).
//   RING_DECL(NT)                    locals: staging registers, chunk counter
//   RING_START(smem, total, SRC)     first chunk into slot 0, second in flight
//   RING_SYNC_WRITE(SRC)             barrier; chunk cc readable; chunk cc+1 written to the other slot
//   RING_FETCH(SRC, slot)            chunk cc+2 goes in flight; `slot` = readable slot; ++cc
#define RING_DECL(NTV) uint4 ring_r[1024 / (NTV)]; int ring_cc = 0, ring_last = 0; char* ring_smem = nullptr; constexpr int RING_NT = (NTV)
#define RING_START(SMEM, TOTAL, SRC)                                                         \
    do {                                                                                     \
        ring_smem = (SMEM); ring_last = (TOTAL) - 1;                                         \
        { const WChunk c0_ = SRC(0); stage_load<RING_NT>(ring_r, c0_, t); stage_write<RING_NT>(ring_r, c0_.kind, ring_smem, t); } \
        { const WChunk c1_ = SRC(min(1, ring_last)); stage_load<RING_NT>(ring_r, c1_, t); }  \
    } while (0)
#define RING_SYNC_WRITE(SRC)                                                                 \
    do {                                                                                     \
        __syncthreads();                                                                     \
        const WChunk cw_ = SRC(min(ring_cc + 1, ring_last));                                 \
        stage_write<RING_NT>(ring_r, cw_.kind, ring_smem + ((ring_cc + 1) & 1) * CHUNK, t);  \
    } while (0)
#define RING_FETCH(SRC, SLOT)                                                                \
    do {                                                                                     \
        const WChunk cf_ = SRC(min(ring_cc + 2, ring_last));                                 \
        stage_load<RING_NT>(ring_r, cf_, t);                                                 \
        SLOT = ring_smem + (ring_cc & 1) * CHUNK;                                            \
        ++ring_cc;                                                                           \
    } while (0)
#define RING_STEP(SRC, SLOT) do { RING_SYNC_WRITE(SRC); RING_FETCH(SRC, SLOT); } while (0)

// The same ring with 32 KB chunks made of two 16 KB sub-blocks (possibly of different matrices): twice the MFMAs per barrier.
struct WChunk2 { WChunk s[2]; };
constexpr int CHUNK2 = 2 * CHUNK;
template <int NT>
__device__ __forceinline__ void stage_load2(uint4 (&r)[2048 / NT], const WChunk2& c, int t) {
    constexpr int PP = 1024 / NT;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int p = t + NT * q;
            const int row = c.s[sub].kind == 2 ? (p >> 2) : (p >> 5);
            const int col = c.s[sub].kind == 2 ? (p & 3) : (p & 31);
            r[sub * PP + q] = *reinterpret_cast<const uint4*>(c.s[sub].base + (size_t)row * c.s[sub].ld + 8 * col);
        }
}
template <int NT>
__device__ __forceinline__ void stage_write2(const uint4 (&r)[2048 / NT], const WChunk2& c, char* slot, int t) {
    constexpr int PP = 1024 / NT;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        uint4 rr[PP];
#pragma unroll
        for (int q = 0; q < PP; ++q) rr[q] = r[sub * PP + q];
        stage_write<NT>(rr, c.s[sub].kind, slot + sub * CHUNK, t);
    }
}
#define RING2_DECL(NTV) uint4 ring_r[2048 / (NTV)]; int ring_cc = 0, ring_last = 0; char* ring_smem = nullptr; constexpr int RING_NT = (NTV)
#define RING2_START(SMEM, TOTAL, SRC)                                                        \
    do {                                                                                     \
        ring_smem = (SMEM); ring_last = (TOTAL) - 1;                                         \
        { const WChunk2 c0_ = SRC(0); stage_load2<RING_NT>(ring_r, c0_, t); stage_write2<RING_NT>(ring_r, c0_, ring_smem, t); } \
        { const WChunk2 c1_ = SRC(min(1, ring_last)); stage_load2<RING_NT>(ring_r, c1_, t); } \
    } while (0)
#define RING2_SYNC_WRITE(SRC)                                                                \
    do {                                                                                     \
        __syncthreads();                                                                     \
        const WChunk2 cw_ = SRC(min(ring_cc + 1, ring_last));                                \
        stage_write2<RING_NT>(ring_r, cw_, ring_smem + ((ring_cc + 1) & 1) * CHUNK2, t);     \
    } while (0)
#define RING2_FETCH(SRC, SLOT)                                                               \
    do {                                                                                     \
        const WChunk2 cf_ = SRC(min(ring_cc + 2, ring_last));                                \
        stage_load2<RING_NT>(ring_r, cf_, t);                                                \
        SLOT = ring_smem + (ring_cc & 1) * CHUNK2;                                           \
        ++ring_cc;                                                                           \
    } while (0)
#define RING2_STEP(SRC, SLOT) do { RING2_SYNC_WRITE(SRC); RING2_FETCH(SRC, SLOT); } while (0)

// ---- asynchronous weight ring: THREE 16 KB slots filled by LDS-DMA (buffer_load_dwordx4 ... lds) -----------------------------
// No staging registers, no ds_write pass, and chunk cc+2 is requested while chunk cc is multiplied (the register-staged rings above
// pay a chunk's L2 latency or 16-32 staging registers for that distance).  The DMA and the operand reads are inline asm: hipcc orders
// a plain LDS access behind EVERY LDS-DMA it knows to be in flight (s_waitcnt vmcnt(0)), which would drain the prefetch at each
// bias / staging access of the loop.  Ordering is by hand instead:
//   step cc:  s_waitcnt vmcnt(PP)   the PP = 1024 / NT loads of chunk cc+1 (issued last step) may stay in flight; vector-memory
//                                   operations retire in order, so chunk cc's have landed (conservative when the kernel issued
//                                   other loads / stores since: those and part of chunk cc+1 are waited for too)
//             s_barrier             every wave's share of chunk cc is in LDS, and every wave is done reading chunk cc-1
//             issue chunk cc+2      into the slot of chunk cc-1
//             multiply chunk cc
// The swizzles of the kind-0 / kind-2 images are applied to the per-lane SOURCE address (the LDS side of a DMA is lane-linear);
// chunks whose operands are accumulator tiles need the PERM order in memory already: mmfm_prep_weights writes such copies.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
constexpr int RINGA_SLOTS = 3;
struct AChunk {
    __amdgpu_buffer_rsrc_t rs;     // the whole weight matrix (bf16, rows K-contiguous)
    uint32_t off;                  // byte offset of the block's (row 0, k 0): wave-uniform
    uint32_t ldb;                  // row stride in bytes
    int kind;                      // 0: [32 rows][256 k]   2: [256 rows][32 k], 8-B units already in PERM order
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wbuf(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// per-thread source offsets: kind 0 - piece p = t + NT q sits at row p / 32, slot p % 32 and holds source chunk slot ^ (row & 15)
// (two values: with NT / 32 = 8 rows per instruction, row & 15 alternates between r0 and r0 ^ 8); kind 2 - row p / 4, slot p % 4,
// source chunk slot ^ ((row >> 2) & 3), the same for every q
template <int NT> struct ALane { uint32_t k0[2], k2; };
template <int NT>
__device__ __forceinline__ ALane<NT> alane_init(int t, uint32_t ldb0, uint32_t ldb2) {
    ALane<NT> a;
    const uint32_t r0 = (uint32_t)t >> 5, c = (uint32_t)t & 31;
    a.k0[0] = r0 * ldb0 + ((c ^ (r0 & 15)) << 4);
    a.k0[1] = r0 * ldb0 + ((c ^ ((r0 + NT / 32) & 15)) << 4);
    a.k2 = ((uint32_t)t >> 2) * ldb2 + ((((uint32_t)t & 3) ^ (((uint32_t)t >> 4) & 3)) << 4);
    return a;
}
__device__ __forceinline__ void dma16(uint32_t lds_dst, uint32_t voff, __amdgpu_buffer_rsrc_t rs, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
// all NT threads: request one chunk into the slot at LDS byte address `slot` (wave-uniform)
template <int NT>
__device__ __forceinline__ void dma_chunk(uint32_t slot, const AChunk& c, const ALane<NT>& a, int wave) {
    constexpr int PP = 1024 / NT;
#pragma unroll
    for (int q = 0; q < PP; ++q) {
        const uint32_t dst = slot + (uint32_t)(q * NT * 16) + (uint32_t)wave * 1024u;
        if (c.kind == 2) dma16(dst, a.k2, c.rs, c.off + (uint32_t)(q * (NT / 4)) * c.ldb);
        else dma16(dst, a.k0[(NT == 256) ? (q & 1) : 0], c.rs, c.off + (uint32_t)(q * (NT / 32)) * c.ldb);
    }
}
// one of the PP = 1024 / NT requests of a chunk (q = 0 .. PP-1): lets a kernel place them between its MFMA groups - a request costs
// ~130 issue cycles, which disappear in the shadow of four MFMAs but add up to a quarter of a ring step when issued in a row
template <int NT>
__device__ __forceinline__ void dma_piece(uint32_t slot, const AChunk& c, const ALane<NT>& a, int wave, int q) {
    const uint32_t dst = slot + (uint32_t)(q * NT * 16) + (uint32_t)wave * 1024u;
    if (c.kind == 2) dma16(dst, a.k2, c.rs, c.off + (uint32_t)(q * (NT / 4)) * c.ldb);
    else dma16(dst, a.k0[(NT == 256) ? (q & 1) : 0], c.rs, c.off + (uint32_t)(q * (NT / 32)) * c.ldb);
}
template <int N> __device__ __forceinline__ void vm_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

//   RINGA_DECL(NT)                   locals
//   RINGA_START(lds0, total, SRC)    chunks 0 and 1 requested (SRC(g) -> AChunk of the workgroup's g-th chunk)
//   RINGA_SYNC / RINGA_PIECE         one ring step (below)
#define RINGA_DECL(NTV) int ring_cc = 0, ring_last = 0; uint32_t ring_lds = 0; constexpr int RING_NT = (NTV)
#define RINGA_START(LDS0, TOTAL, SRC)                                                        \
    do {                                                                                     \
        ring_lds = (LDS0); ring_last = (TOTAL) - 1;                                          \
        dma_chunk<RING_NT>(ring_lds, SRC(0), ring_al, wave);                                 \
        dma_chunk<RING_NT>(ring_lds + CHUNK, SRC(min(1, ring_last)), ring_al, wave);         \
    } while (0)
// the same ring with 32 KB chunks of two 16 KB sub-blocks (three slots = 96 KB): SRC(g) -> AChunk2
struct AChunk2 { AChunk s[2]; };
#define RINGA2_START(LDS0, TOTAL, SRC)                                                       \
    do {                                                                                     \
        ring_lds = (LDS0); ring_last = (TOTAL) - 1;                                          \
        { const AChunk2 c_ = SRC(0); dma_chunk<RING_NT>(ring_lds, c_.s[0], ring_al, wave); dma_chunk<RING_NT>(ring_lds + CHUNK, c_.s[1], ring_al, wave); } \
        { const AChunk2 c_ = SRC(min(1, ring_last)); dma_chunk<RING_NT>(ring_lds + CHUNK2, c_.s[0], ring_al, wave); dma_chunk<RING_NT>(ring_lds + CHUNK2 + CHUNK, c_.s[1], ring_al, wave); } \
    } while (0)
// A ring step: RINGA_SYNC(SRC, SLOT, EXTRA) waits / synchronises (SLOT = LDS byte address of chunk cc; ++cc) and names the chunk to
// request (ring_nc -> slot ring_nd); the kernel then places RINGA_PIECE(q), q = 0 .. PP-1, between its MFMA groups.  All PP pieces
// must be issued before the next RINGA_SYNC.  EXTRA = vector-memory operations (stores) the kernel is KNOWN to have issued since
// chunk cc's request, on top of chunk cc+1's: they may stay in flight too (only where that count is static; too large a value
// would let chunk cc itself be outstanding).
#define RINGA_SYNC(SRC, SLOT, EXTRA)                                                         \
        vm_wait_n<1024 / RING_NT + (EXTRA)>();                                               \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        const AChunk ring_nc = SRC(min(ring_cc + 2, ring_last));                             \
        const uint32_t ring_nd = ring_lds + (uint32_t)((ring_cc + 2) % RINGA_SLOTS) * CHUNK; \
        SLOT = ring_lds + (uint32_t)(ring_cc % RINGA_SLOTS) * CHUNK;                         \
        ++ring_cc
#define RINGA_PIECE(Q) dma_piece<RING_NT>(ring_nd, ring_nc, ring_al, wave, (Q))
// the same without the wait: for kernels whose EXTRA depends on a wave-uniform condition (they issue vm_wait_n<...>() themselves first)
#define RINGA_SYNC_NOWAIT(SRC, SLOT)                                                         \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        const AChunk ring_nc = SRC(min(ring_cc + 2, ring_last));                             \
        const uint32_t ring_nd = ring_lds + (uint32_t)((ring_cc + 2) % RINGA_SLOTS) * CHUNK; \
        SLOT = ring_lds + (uint32_t)(ring_cc % RINGA_SLOTS) * CHUNK;                         \
        ++ring_cc
// ... and for the 32 KB chunks: pieces q = 0 .. 2 PP - 1 (sub-block q / PP)
#define RINGA2_SYNC(SRC, SLOT)                                                               \
        vm_wait_n<2048 / RING_NT>();                                                         \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        const AChunk2 ring_nc = SRC(min(ring_cc + 2, ring_last));                            \
        const uint32_t ring_nd = ring_lds + (uint32_t)((ring_cc + 2) % RINGA_SLOTS) * CHUNK2; \
        SLOT = ring_lds + (uint32_t)(ring_cc % RINGA_SLOTS) * CHUNK2;                        \
        ++ring_cc
#define RINGA2_PIECE(Q) dma_piece<RING_NT>(ring_nd + (uint32_t)((Q) / (1024 / RING_NT)) * CHUNK, ring_nc.s[(Q) / (1024 / RING_NT)], ring_al, wave, (Q) % (1024 / RING_NT))

// operand reads of the asynchronous ring (inline asm, see above).  kind-0 image: operand S of lane (m, h) sits at
//   m * 512 + (((2 S + h) ^ (m & 15)) << 4) = m * 512 + ((h ^ (m & 1)) << 4) + (((S & 7) << 5) ^ ((m & 14) << 4)) + (S >> 3) * 256
// -> eight per-lane constants + an immediate; kind-2 image: tile t2, k-step s at (32 t2 + m) * 64 + (((2 s + h) ^ ((m >> 2) & 3)) << 4)
// -> two per-lane constants + t2 * 2048.
struct AFrag { uint32_t a[8], b[2]; };
__device__ __forceinline__ AFrag afrag_init(int m, int h) {
    AFrag f;
    const uint32_t base = (uint32_t)m * 512u + ((uint32_t)(h ^ (m & 1)) << 4), A = (uint32_t)(m & 14) << 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) f.a[s] = base + (((uint32_t)s << 5) ^ A);
#pragma unroll
    for (int s = 0; s < 2; ++s) f.b[s] = (uint32_t)m * 64u + ((uint32_t)((2 * s + h) ^ ((m >> 2) & 3)) << 4);
    return f;
}
#define ALDS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define ALDS_WAIT0 do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ f32x16 mfma_u4(uint4 a, opnd b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(opnd, a), b, c, 0, 0, 0); }
// 16 MFMAs of a kind-0 chunk at LDS address `slot` against 16 operands; two sets of D weight operands, ping-pong, COUNTED waits:
// two sets of reads are kept in flight and a group's MFMAs wait only for their own set (LDS returns in order), so the read
// latency hides under the previous group's MFMAs (with lgkmcnt(0) after every group: 966 instead of ~600 cycles per chunk).
// (No scalar loads may be pending here - they share the counter and return out of order; the kernels load their arguments up front.)
#define ALDS_WAITN(N) do { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// `after(g)` runs behind the MFMAs of group g (the kernels issue one ring request there)
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
template <int D = 8, typename HOOK = NoHook>
__device__ __forceinline__ f32x16 mma16a(uint32_t slot, const AFrag& f, const opnd* x, f32x16 acc, HOOK after = HOOK()) {
    constexpr int NG = 16 / D;
    uint4 w[2][D];
#define MMA16A_READ(SET, G)                                                                                  \
    _Pragma("unroll") for (int s = 0; s < D; ++s) {                                                          \
        const int S = (G) * D + s;                                                                           \
        const uint32_t ad = f.a[S & 7] + slot;                                                               \
        if (S < 8) ALDS_READ(w[SET][s], ad, 0); else ALDS_READ(w[SET][s], ad, 256);                          \
    }
    MMA16A_READ(0, 0)
    if (NG > 1) { MMA16A_READ(1, 1) }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) ALDS_WAITN(D); else ALDS_WAITN(0);
#pragma unroll
        for (int s = 0; s < D; ++s) acc = mfma_u4(w[g & 1][s], x[g * D + s], acc);
        __builtin_amdgcn_sched_barrier(0);
        after(g);
        if (g + 2 < NG) { if (g & 1) { MMA16A_READ(1, g + 2) } else { MMA16A_READ(0, g + 2) } }
    }
#undef MMA16A_READ
    return acc;
}
// weight operand (tile t2 of 8, k-step s of 2) of a kind-2 chunk at LDS address `slot`
#define ALDS_READ_B(dst, slot, f, t2, s) do { const uint32_t ad_ = (f).b[s] + (slot); ALDS_READ(dst, ad_, (t2) * 2048); } while (0)
#pragma clang diagnostic pop

// weight operand of k-step S (0..15) from a [32][256] image: lane (i = lane & 31, h = lane >> 5)
__device__ __forceinline__ opnd wfragA(const char* slot, int S, int i, int h) {
    return as_opnd(*reinterpret_cast<const uint4*>(slot + i * 512 + (((2 * S + h) ^ (i & 15)) << 4)));
}
// weight operand of (tile t2 of 8, k-step s of 2) from a [256][32] image
__device__ __forceinline__ opnd wfragB(const char* slot, int t2, int s, int i, int h) {
    const int row = 32 * t2 + i;
    return as_opnd(*reinterpret_cast<const uint4*>(slot + row * 64 + (((2 * s + h) ^ ((row >> 2) & 3)) << 4)));
}

__device__ __forceinline__ f32x16 mfma(opnd a, opnd b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// accumulator tile -> the two operands (k-steps) it provides to a product that sums over its n index
__device__ __forceinline__ void acc_to_opnd(const f32x16& v, opnd& o0, opnd& o1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { o0[j] = (__bf16)v[j]; o1[j] = (__bf16)v[8 + j]; }
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 p; p[0] = (__bf16)a; p[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float lo_f(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- accumulator-layout global I/O: lane (m, h) moves 16 B = features 32t + 16p + 8h .. +7 of its row, p = 0, 1
// (`rowoff` = byte offset of the lane's row; rows beyond the tensor are dropped / read as zero by the bounds check)
template <bool NTS>
__device__ __forceinline__ void store_tile(const GBuf& b, uint32_t rowoff, int t, int h, const f32x16& v) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint32_t a0 = pack2(v[8 * p + 0], v[8 * p + 1]), a1 = pack2(v[8 * p + 2], v[8 * p + 3]);
        uint32_t b0 = pack2(v[8 * p + 4], v[8 * p + 5]), b1 = pack2(v[8 * p + 6], v[8 * p + 7]);
        auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        st16<NTS>(b, rowoff + 2 * (32 * t + 16 * p + 8 * h), make_uint4(r0[0], r1[0], r0[1], r1[1]));
    }
}
struct RawTile { uint4 o[2]; };                 // a tile as loaded (lets the load be issued long before its use)
__device__ __forceinline__ RawTile load_tile_raw(const GBuf& b, uint32_t rowoff, int t, int h) {
    RawTile r;
#pragma unroll
    for (int p = 0; p < 2; ++p) r.o[p] = ld16(b, rowoff + 2 * (32 * t + 16 * p + 8 * h));
    return r;
}
__device__ __forceinline__ f32x16 tile_f32(const RawTile& rt) {
    f32x16 v;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const uint4 o = rt.o[p];
        auto r0 = __builtin_amdgcn_permlane32_swap(o.x, o.z, false, false);      // inverse of the store's exchange
        auto r1 = __builtin_amdgcn_permlane32_swap(o.y, o.w, false, false);
        v[8 * p + 0] = lo_f(r0[0]); v[8 * p + 1] = hi_f(r0[0]); v[8 * p + 2] = lo_f(r1[0]); v[8 * p + 3] = hi_f(r1[0]);
        v[8 * p + 4] = lo_f(r0[1]); v[8 * p + 5] = hi_f(r0[1]); v[8 * p + 6] = lo_f(r1[1]); v[8 * p + 7] = hi_f(r1[1]);
    }
    return v;
}
__device__ __forceinline__ f32x16 load_tile(const GBuf& b, uint32_t rowoff, int t, int h) { return tile_f32(load_tile_raw(b, rowoff, t, h)); }
// feature index of register r of tile t for lane half h
__device__ __forceinline__ int feat(int t, int r, int h) { return 32 * t + 8 * (r >> 2) + 4 * h + (r & 3); }

// v[r] += vec[feature]  (bias, kept in LDS): four 16-B reads per tile
__device__ __forceinline__ void add_vec(f32x16& v, const float* vec, int t, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4*>(vec + 32 * t + 8 * q + 4 * h);
        v[4 * q + 0] += b.x; v[4 * q + 1] += b.y; v[4 * q + 2] += b.z; v[4 * q + 3] += b.w;
    }
}

// workgroup copies a bias vector into LDS (zeros when there is none); caller synchronises
__device__ __forceinline__ void stage_vec(float* dst, const float* src, int n, int t, int nthreads) {
    for (int i = t; i < n; i += nthreads) dst[i] = src ? src[i] : 0.f;
}

__device__ __forceinline__ float xhalf(float v) { return v + __shfl_xor(v, 32); }   // row total: lanes m and m+32

// ---- operand-layout (natural k order) row loads: lane (m, h) holds k = 16s + 8h + 0..7 of its row in x[s]
template <int NS>
__device__ __forceinline__ void load_rows(opnd (&x)[NS], const GBuf& b, uint32_t rowoff, int h) {
#pragma unroll
    for (int s = 0; s < NS; ++s) x[s] = as_opnd(ld16(b, rowoff + 2 * (16 * s + 8 * h)));
}
template <int NS>
__device__ __forceinline__ void store_rows(const GBuf& b, uint32_t rowoff, int h, const opnd (&x)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) st16<false>(b, rowoff + 2 * (16 * s + 8 * h), as_u4(x[s]));
}
__device__ __forceinline__ void unpack8f(opnd v, float* f) {
    const uint4 w = as_u4(v);
    f[0] = lo_f(w.x); f[1] = hi_f(w.x); f[2] = lo_f(w.y); f[3] = hi_f(w.y);
    f[4] = lo_f(w.z); f[5] = hi_f(w.z); f[6] = lo_f(w.w); f[7] = hi_f(w.w);
}
__device__ __forceinline__ opnd pack8o(const float* f) {
    opnd o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)f[j];
    return o;
}
// LayerNorm without the affine part on a 256-wide row held as 16 operands: x <- (x - mean) * rstd; returns rstd.
// (gamma is folded into the prepared weights and beta into the prepared bias: mmfm_prep_weights.)
__device__ __forceinline__ float ln_rows(opnd (&x)[16], float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float f[8]; unpack8f(x[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += f[j];
    }
    const float mu = xhalf(s) * (1.f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float f[8]; unpack8f(x[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = f[j] - mu; q += d * d; }
    }
    const float rs = rsqrtf(xhalf(q) * (1.f / 256.f) + eps);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float f[8]; unpack8f(x[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (f[j] - mu) * rs;
        x[i] = pack8o(f);
    }
    return rs;
}

// ---- dropout of the fused MLP kernels (forward epilogue, backward prologue): a lane owns ONE token row, so the hash is two-level like the
// attention kernels' (attn_common.h Drop16): a full-strength 2 x 32-bit key per ROW, made once per row pass, and a 7-instruction
// xorshift / 24-bit-multiply mix of (feature pair ^ key A, key B) per pair of neighbouring features (even feature: low 16 bits, odd: high 16
// bits, each against the top 16 bits of the threshold).  Round 3 ran the flat counter hash (row * 256 + feature: ~15 instructions per
// pair, it has to digest a 26-bit counter) 64 times per lane in every backward pass - 14 % of that kernel.  Forward and backward
// evaluate the same function; the un-fused kernels (GEMM epilogue + mmfm_dropout_apply, used below 12 k rows) keep the flat counter hash,
// so the two plans draw different (equally distributed) masks.
struct RowDrop {
    uint32_t ka, kb, t16;
    float scale;
    __device__ __forceinline__ uint32_t pair(uint32_t fp) const {          // fp = feature >> 1 (< 128)
        uint32_t h = __umul24(fp ^ ka, 0x7FEB35u) + kb;
        h ^= h >> 13;
        h = __umul24(h, 0x46CA6Bu);
        h ^= h >> 16;
        return h;
    }
};
__device__ __forceinline__ RowDrop rowdrop_init(const Drop& dr, uint32_t row) {
    RowDrop r;
    r.ka = mix32(row ^ dr.k0);
    r.kb = mix32((row + 0x9E3779B9u) ^ dr.k1);
    r.t16 = dr.t16;
    r.scale = dr.scale;
    return r;
}
// ... on an accumulator tile of output tile t2 (registers i, i+1 with i even are neighbouring features: one hash per pair)
__device__ __forceinline__ void drop16(const RowDrop& rd, f32x16& v, int t2, int h) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        const uint32_t hh = rd.pair((uint32_t)feat(t2, i, h) >> 1);
        v[i] = (hh & 0xffffu) >= rd.t16 ? v[i] * rd.scale : 0.f;
        v[i + 1] = (hh >> 16) >= rd.t16 ? v[i + 1] * rd.scale : 0.f;
    }
}
// ... and on 8 consecutive features k0 .. k0+7 (k0 a multiple of 8)
__device__ __forceinline__ void drop8(const RowDrop& rd, float (&f)[8], int k0) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const uint32_t hh = rd.pair((uint32_t)(k0 + j) >> 1);
        f[j] = (hh & 0xffffu) >= rd.t16 ? f[j] * rd.scale : 0.f;
        f[j + 1] = (hh >> 16) >= rd.t16 ? f[j + 1] * rd.scale : 0.f;
    }
}

// ---- GELU on accumulator tiles: the packed polynomial of common.h (phi2), two registers per instruction
__device__ __forceinline__ void gelu16(f32x16& U) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        mmfm_f32x2 a; a.x = U[i]; a.y = U[i + 1];
        a = gelu2(a);
        U[i] = a.x; U[i + 1] = a.y;
    }
}
__device__ __forceinline__ void gelu_pair(f32x16& U, int i) {                  // registers i, i + 1 (i even)
    mmfm_f32x2 a; a.x = U[i]; a.y = U[i + 1];
    a = gelu2(a);
    U[i] = a.x; U[i + 1] = a.y;
}
// G = gelu(U), D *= gelu'(U)
__device__ __forceinline__ void gelu_fwd_bwd16(const f32x16& U, f32x16& G, f32x16& D) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        mmfm_f32x2 a, g, dg; a.x = U[i]; a.y = U[i + 1];
        gelu_both2(a, g, dg);
        G[i] = g.x; G[i + 1] = g.y;
        D[i] *= dg.x; D[i + 1] *= dg.y;
    }
}

// 16 MFMAs of one [32][256] chunk against 16 operands, weight operands fetched 8 at a time (the ds_read latency of a
// 2-deep fetch was 40 % of the loop: 830 cycles instead of 512 per chunk in the probe)
template <int DEPTH = 8>
__device__ __forceinline__ f32x16 mma16(const char* slot, const opnd* x, f32x16 acc, int m, int h) {
#pragma unroll
    for (int part = 0; part < 16 / DEPTH; ++part) {
        opnd wf[DEPTH];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) wf[s] = wfragA(slot, DEPTH * part + s, m, h);
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) acc = mfma(wf[s], x[DEPTH * part + s], acc);
    }
    return acc;
}

// one register pair (i, i+1; i even) of gelu_fwd_bwd16
__device__ __forceinline__ void gelu_fb_pair(const f32x16& U, f32x16& G, f32x16& D, int i) {
    mmfm_f32x2 a, g, dg; a.x = U[i]; a.y = U[i + 1];
    gelu_both2(a, g, dg);
    G[i] = g.x; G[i + 1] = g.y;
    D[i] *= dg.x; D[i + 1] *= dg.y;
}

// ---- full-line global stores / loads through a per-wave LDS staging area -----------------------------------------------
// In the accumulator (and the operand) layout only the two lanes m, m+32 hold data of row m, so a direct 16-B-per-lane
// access touches 32 rows x 32 B per instruction.  Measured on MI355X (scripts/probe/rowchain_probe.hip): such partial-line
// STORES issue at ~1.3 TB/s chip-wide and were 60 % of the first version's time.  Outputs therefore make one trip through
// a 4 KB per-wave staging area [32 rows][128 B] (16-B chunk index XOR ((row >> 1) & 7): conflict-free both ways) and leave
// as whole 128-B lines, 8 rows per instruction.  Same-wave LDS accesses execute in order: no barrier, no wait needed.
constexpr int STG_BYTES = 4096;
__device__ __forceinline__ int stg_off(int row, int c16) { return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4); }

// accumulator tile -> staging columns 64*j .. 64*j+63 (bytes) of the wave's rows (j = 0, 1)
__device__ __forceinline__ void stage_tile(char* stg, int j, int m, int h, const f32x16& v) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        uint32_t a0 = pack2(v[8 * p + 0], v[8 * p + 1]), a1 = pack2(v[8 * p + 2], v[8 * p + 3]);
        uint32_t b0 = pack2(v[8 * p + 4], v[8 * p + 5]), b1 = pack2(v[8 * p + 6], v[8 * p + 7]);
        auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        *reinterpret_cast<uint4*>(stg + stg_off(m, 4 * j + 2 * p + h)) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}
// operand (natural k order) -> staging: operand s (0..3 within the 64-feature quarter)
__device__ __forceinline__ void stage_opnd(char* stg, int s, int m, int h, opnd v) {
    *reinterpret_cast<uint4*>(stg + stg_off(m, 2 * s + h)) = as_u4(v);
}
// staging -> global: rows wrow0 .. wrow0+31 of a tensor with `ldb` bytes per row, 128 bytes starting at byte column `colb`;
// `nchunk` (1..8) = 16-B chunks of the 128 that exist (ragged last tile pair)
template <bool NTS>
__device__ __forceinline__ void flush_lines(const char* stg, const GBuf& b, uint32_t wrow0, uint32_t ldb, uint32_t colb, int lane, int nchunk = 8) {
    const int c = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + (lane >> 3);
        const uint4 v = *reinterpret_cast<const uint4*>(stg + stg_off(row, c));
        st16<NTS>(b, c < nchunk ? (wrow0 + row) * ldb + colb + 16 * c : 0xfffffff0u, v);
    }
}
// global -> staging (whole lines), for later reads in either layout
#ifndef MMFM_ACT_LOAD_AUX
#define MMFM_ACT_LOAD_AUX 0
#endif
struct Lines { uint4 v[4]; };
__device__ __forceinline__ Lines fetch_lines(const GBuf& b, uint32_t wrow0, uint32_t ldb, uint32_t colb, int lane) {
    Lines L;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        L.v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b.rs, (int)((wrow0 + 8 * i + (lane >> 3)) * ldb + colb + 16 * (lane & 7)), 0, MMFM_ACT_LOAD_AUX));
    return L;
}
__device__ __forceinline__ void stage_lines(char* stg, const Lines& L, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(stg + stg_off(8 * i + (lane >> 3), lane & 7)) = L.v[i];
}
// staging -> accumulator-layout tile j (0, 1) in fp32
__device__ __forceinline__ f32x16 unstage_tile(const char* stg, int j, int m, int h) {
    RawTile rt;
#pragma unroll
    for (int p = 0; p < 2; ++p) rt.o[p] = *reinterpret_cast<const uint4*>(stg + stg_off(m, 4 * j + 2 * p + h));
    return tile_f32(rt);
}
__device__ __forceinline__ opnd unstage_opnd(const char* stg, int s, int m, int h) {
    return as_opnd(*reinterpret_cast<const uint4*>(stg + stg_off(m, 2 * s + h)));
}
// a [32 rows][64*NQ features] block in operand layout -> global, whole lines (NQ quarters of 64 features)
template <int NQ, bool NTS>
__device__ __forceinline__ void store_rows_lines(char* stg, const GBuf& b, uint32_t wrow0, uint32_t ldb, int lane, int m, int h, const opnd (&x)[4 * NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int s = 0; s < 4; ++s) stage_opnd(stg, s, m, h, x[4 * q + s]);
        flush_lines<NTS>(stg, b, wrow0, ldb, 128u * q, lane);
    }
}
// global -> operand layout, whole lines: up to four quarters (64 registers) of loads in flight, transposed through the
// staging area as they arrive
template <int NQ>
__device__ __forceinline__ void load_rows_lines(char* stg, opnd (&x)[4 * NQ], const GBuf& b, uint32_t wrow0, uint32_t ldb, int lane, int m, int h) {
    constexpr int WIN = NQ < 4 ? NQ : 4;
    Lines L[WIN];
#pragma unroll
    for (int q = 0; q < WIN; ++q) L[q] = fetch_lines(b, wrow0, ldb, 128u * q, lane);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        stage_lines(stg, L[q % WIN], lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) x[4 * q + s] = unstage_opnd(stg, s, m, h);
        if (q + WIN < NQ) L[q % WIN] = fetch_lines(b, wrow0, ldb, 128u * (q + WIN), lane);
    }
}

}  // namespace rowchain
