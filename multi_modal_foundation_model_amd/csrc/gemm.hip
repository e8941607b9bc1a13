// GEMM for every nn.Linear forward/backward on the path (see include/mmfm.h, mmfm_gemm).
//
// fp32 (parity) path: v_mfma_f32_32x32x2_f32 — bit-for-bit a k-ordered fmaf chain per output
// element, at the fp32 vector rate (157 TF dense on MI355X).  128x128x32 tile, 4 wavefronts
// (2x2), each wave a 64x64 sub-tile = 2x2 MFMA tiles (64 accumulator VGPRs).
//
// Operand tiles live in LDS K-MAJOR ([k][row], row contiguous) whatever the global layout,
// so the MFMA operand read `tile[k = 2s + lane/32][row0 + lane%32]` is 32 consecutive dwords
// per half-wave: conflict-free.  The loader picks its thread->element map from the operand's
// contiguous axis so global reads stay 16 B/lane and coalesced:
//   K-contiguous (x[M,K], W[N,K]):    8 lanes x float4 = one 128-B row segment, transposed on the
//                                     LDS write (ds_write_b32 x4);
//   row-contiguous (dY^T, W for dX):  32 lanes x float4 = 512 B of one k-row, ds_write_b128.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = 132;  // LDS row stride (floats): 528 B keeps ds_write_b128 rows 16-B aligned
constexpr int NTHREADS = 256;

// ---- global -> registers (one 128 x 32 operand tile = 4 float4 per thread)
template <bool KC>
__device__ __forceinline__ void g2r(float4 (&r)[4], const float* __restrict__ base, int ld, int row0, int k0,
                                    int rows, int kend, bool vec, int t) {
    if (KC) {
        const int kq = t & 7, r0 = t >> 3;
        const int k = k0 + 4 * kq;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = row0 + r0 + 32 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows && k < kend) {
                const float* src = base + (size_t)row * ld + k;
                if (vec && k + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (k + 1 < kend) v.y = src[1];
                    if (k + 2 < kend) v.z = src[2];
                    if (k + 3 < kend) v.w = src[3];
                }
            }
            r[p] = v;
        }
    } else {
        const int rq = t & 31, kk0 = t >> 5;
        const int row = row0 + 4 * rq;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = k0 + kk0 + 8 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend && row < rows) {
                const float* src = base + (size_t)k * ld + row;
                if (vec && row + 3 < rows) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (row + 1 < rows) v.y = src[1];
                    if (row + 2 < rows) v.z = src[2];
                    if (row + 3 < rows) v.w = src[3];
                }
            }
            r[p] = v;
        }
    }
}

// ---- registers -> LDS tile [BK][LDT]
template <bool KC>
__device__ __forceinline__ void r2s(float* __restrict__ S, const float4 (&r)[4], int t) {
    if (KC) {
        const int kq = t & 7, r0 = t >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = r0 + 32 * p;
            S[(4 * kq + 0) * LDT + row] = r[p].x;
            S[(4 * kq + 1) * LDT + row] = r[p].y;
            S[(4 * kq + 2) * LDT + row] = r[p].z;
            S[(4 * kq + 3) * LDT + row] = r[p].w;
        }
    } else {
        const int rq = t & 31, kk0 = t >> 5;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(&S[(kk0 + 8 * p) * LDT + 4 * rq]) = r[p];
    }
}

// ---- fused epilogue for one element
template <typename TO>
__device__ __forceinline__ void epilogue_store(const mmfm_gemm_desc& d, const Drop& dr, float v, int m, int n) {
    typedef io<TO> O;
    if (d.bias) v += d.bias[n];
    if (d.pre_out) O::st(reinterpret_cast<TO*>(d.pre_out) + (size_t)m * d.ldc + n, v);
    if (d.act == 1) v = gelu_erf(v);
    else if (d.act == 2) v = softsign_f(v) * d.act_scale;
    if (d.gradmul_pre) {
        const float u = O::ld(reinterpret_cast<const TO*>(d.gradmul_pre) + (size_t)m * d.ldc + n);
        // act kinds 3/4 = multiply by gelu'(u) / softsign'(u)*scale (backward through the activation)
        v *= (d.act == 3) ? gelu_erf_grad(u) : (d.act == 4 ? softsign_grad(u) : softsign_grad_from_out(u, 1.f / d.act_scale)) * d.act_scale;
    }
    v = dr.apply(v, (uint64_t)m * (uint64_t)d.N + (uint64_t)n);
    if (d.residual) v += O::ld(reinterpret_cast<const TO*>(d.residual) + (size_t)m * d.ldr + n);
    O::st(reinterpret_cast<TO*>(d.C) + (size_t)m * d.ldc + n, v);
}

// XCD-aware block id: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous chunk of tile ids; tiles of one m-row then reuse the same A panel out of one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(const mmfm_gemm_desc d, const int vecA, const int vecB) {
    __shared__ __attribute__((aligned(16))) float As[BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, l31 = lane & 31;
    const int tiles_n = (d.N + BN - 1) / BN;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const int z = blockIdx.y;
    const int kbeg = z * d.kchunk;
    const int kend = min(d.K, kbeg + d.kchunk);
    const float* A = reinterpret_cast<const float*>(d.A);
    const float* B = reinterpret_cast<const float*>(d.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[4], rb[4];
    g2r<AKC>(ra, A, d.lda, m0, kbeg, d.M, kend, vecA, t);
    g2r<BKC>(rb, B, d.ldb, n0, kbeg, d.N, kend, vecB, t);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();                       // previous tile's MFMA reads are done
        r2s<AKC>(As, ra, t);
        r2s<BKC>(Bs, rb, t);
        __syncthreads();
        if (k0 + BK < kend) {                  // prefetch the next tile under the MFMAs
            g2r<AKC>(ra, A, d.lda, m0, k0 + BK, d.M, kend, vecA, t);
            g2r<BKC>(rb, B, d.ldb, n0, k0 + BK, d.N, kend, vecB, t);
        }
        const float* a = As + wm * 64 + l31;
        const float* b = Bs + wn * 64 + l31;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = 2 * ks + kh;
            const float a0 = a[kk * LDT], a1 = a[kk * LDT + 32];
            const float b0 = b[kk * LDT], b1 = b[kk * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    // C/D map of the 32x32 MFMA: col = lane%32, row = (r&3) + 8*(r>>2) + 4*(lane/32)
    if (d.splits > 1) {
        float* C = reinterpret_cast<float*>(d.C) + (size_t)z * d.slab_stride;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int n = n0 + wn * 64 + j * 32 + l31;
                    if (m < d.M && n < d.N) C[(size_t)m * d.ldc + n] = acc[i][j][r];
                }
        return;
    }
    const Drop dr = drop_init(d.drop);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (m < d.M && n < d.N) epilogue_store<float>(d, dr, acc[i][j][r], m, n);
            }
}

__global__ void reduce_slabs_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n, int nslabs,
                                    int64_t stride, int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = accumulate ? dst[i] : 0.f;
        for (int k = 0; k < nslabs; ++k) s += src[(size_t)k * stride + i];
        dst[i] = s;
    }
}

// Bandwidth-shaped slab reduction (n % 4 == 0, 16-B aligned).  A block owns 256 consecutive floats (64 lanes x
// float4 = 1 KB, one coalesced wave access per slab) and its 4 waves take slabs k = wave, wave+4, ... of the
// slab group blockIdx.y, two loads in flight each; the 4 partial sums meet in LDS in a fixed order.
//   INPLACE: the group's sum is written back into the group's first slab (stage 1 of a two-stage reduction)
template <bool INPLACE>
__global__ __launch_bounds__(256) void reduce_slabs4_kernel(float* __restrict__ dst, float* __restrict__ src, int64_t n, int nslabs,
                                                            int64_t stride, int group, int accumulate) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int k0 = blockIdx.y * group, k1 = min(nslabs, k0 + group);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (i < n) {
        int k = k0 + wave;
        for (; k + 4 < k1; k += 8) {
            const float4 u = *reinterpret_cast<const float4*>(src + (size_t)k * stride + i);
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(k + 4) * stride + i);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (k < k1) {
            const float4 u = *reinterpret_cast<const float4*>(src + (size_t)k * stride + i);
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
    }
    red[wave][lane] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    __syncthreads();
    if (wave == 0 && i < n) {
        float4 o;
        o.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
        o.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
        o.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
        o.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
        float* out = INPLACE ? src + (size_t)k0 * stride + i : dst + i;
        if (!INPLACE && accumulate) {
            const float4 p = *reinterpret_cast<const float4*>(out);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
        }
        *reinterpret_cast<float4*>(out) = o;
    }
}

// column sums: grid (ceil(N/64), S); block = 64 columns x 4 row lanes
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t R, int N, int ld, float* __restrict__ part,
                                                     int64_t rows_per) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float s = 0.f;
    if (c < N)
        for (int64_t r = r0 + rl; r < r1; r += 4) s += io<T>::ld(x + (size_t)r * ld + c);
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N)
        part[(size_t)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// vectorised column sums (N % 4 == 0): a wave covers 256 columns with 16-B (fp32) / 8-B (bf16) loads,
// the 4 waves of a block take consecutive rows, 4 rows in flight per wave.  grid (ceil(N/256), S)
template <typename T>
__global__ __launch_bounds__(256) void colsum4_kernel(const T* __restrict__ x, int64_t R, int N, int ld, float* __restrict__ part,
                                                      int64_t rows_per) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < N) {
        int64_t r = r0 + wave;
        for (; r + 12 < r1; r += 16) {
            const float4 a = io<T>::ld4(x + (size_t)r * ld + c), b = io<T>::ld4(x + (size_t)(r + 4) * ld + c);
            const float4 e = io<T>::ld4(x + (size_t)(r + 8) * ld + c), f = io<T>::ld4(x + (size_t)(r + 12) * ld + c);
            s.x += (a.x + b.x) + (e.x + f.x); s.y += (a.y + b.y) + (e.y + f.y);
            s.z += (a.z + b.z) + (e.z + f.z); s.w += (a.w + b.w) + (e.w + f.w);
        }
        for (; r < r1; r += 4) {
            const float4 a = io<T>::ld4(x + (size_t)r * ld + c);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < N) {
        float4 o;
        o.x = red[0][lane].x + red[1][lane].x + red[2][lane].x + red[3][lane].x;
        o.y = red[0][lane].y + red[1][lane].y + red[2][lane].y + red[3][lane].y;
        o.z = red[0][lane].z + red[1][lane].z + red[2][lane].z + red[3][lane].z;
        o.w = red[0][lane].w + red[1][lane].w + red[2][lane].w + red[3][lane].w;
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.y * N + c) = o;
    }
}

}  // namespace

int mmfm_gemm_bf16_launch(const mmfm_gemm_desc* d, hipStream_t st);  // gemm_bf16.hip

// argument checks and normalisation shared by mmfm_gemm and mmfm_gemm_pair (d = *dp on success)
static int gemm_check(const mmfm_gemm_desc* dp, mmfm_gemm_desc& d) {
    MMFM_REQUIRE(dp != nullptr, "mmfm_gemm: null descriptor");
    d = *dp;
    MMFM_REQUIRE(d.dtype == MMFM_F32 || d.dtype == MMFM_BF16, "mmfm_gemm: bad dtype %d", d.dtype);
    MMFM_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "mmfm_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    MMFM_REQUIRE(d.A && d.B && d.C, "mmfm_gemm: null operand");
    MMFM_REQUIRE(d.lda >= (d.a_kcontig ? d.K : d.M), "mmfm_gemm: lda %d too small", d.lda);
    MMFM_REQUIRE(d.ldb >= (d.b_kcontig ? d.K : d.N), "mmfm_gemm: ldb %d too small", d.ldb);
    MMFM_REQUIRE(d.ldc >= d.N, "mmfm_gemm: ldc %d < N %d", d.ldc, d.N);
    MMFM_REQUIRE(!(d.a_kcontig == 0 && d.b_kcontig == 1), "mmfm_gemm: layout (A row-contig, B k-contig) is not built");
    MMFM_REQUIRE(d.act >= 0 && d.act <= 5, "mmfm_gemm: bad act %d", d.act);
    MMFM_REQUIRE(!d.gradmul_pre || d.act >= 3, "mmfm_gemm: gradmul_pre needs act 3 (gelu'), 4 (softsign') or 5 (softsign' from the output)");
    MMFM_REQUIRE(d.gradmul_pre || d.act <= 2, "mmfm_gemm: act %d needs gradmul_pre", d.act);
    if (d.splits <= 1) {
        d.splits = 1;
        d.kchunk = d.K;
    } else {
        MMFM_REQUIRE(d.kchunk > 0 && d.kchunk % 64 == 0, "mmfm_gemm: kchunk %d must be a positive multiple of 64", d.kchunk);
        MMFM_REQUIRE((int64_t)d.splits * d.kchunk >= d.K, "mmfm_gemm: splits*kchunk < K");
        MMFM_REQUIRE(d.slab_stride >= (int64_t)d.M * d.ldc, "mmfm_gemm: slab_stride too small");
        MMFM_REQUIRE(!d.bias && !d.pre_out && !d.act && !d.residual && d.drop.p <= 0.f,
                     "mmfm_gemm: split-K writes raw partials, no epilogue");
    }
    MMFM_REQUIRE(!d.residual || d.ldr >= d.N, "mmfm_gemm: ldr too small");
    MMFM_REQUIRE(!d.colsum || (d.dtype == MMFM_BF16 && d.a_kcontig == 0), "mmfm_gemm: colsum needs dtype bf16 and a_kcontig == 0");
    return 0;
}

int mmfm_gemm_dw_pair_launch(const mmfm_gemm_desc* a, const mmfm_gemm_desc* b, hipStream_t st);   // gemm_dw.hip

extern "C" int mmfm_gemm(const mmfm_gemm_desc* dp, mmfm_stream stream) {
    mmfm_gemm_desc d;
    if (int rc = gemm_check(dp, d)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (d.dtype == MMFM_BF16) return mmfm_gemm_bf16_launch(&d, st);

    const int tiles = cdiv(d.M, BM) * cdiv(d.N, BN);
    const int vecA = (d.lda % 4 == 0) && ((uintptr_t)d.A % 16 == 0);
    const int vecB = (d.ldb % 4 == 0) && ((uintptr_t)d.B % 16 == 0);
    dim3 grid(tiles, d.splits), block(NTHREADS);
    if (d.a_kcontig && d.b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, st, d, vecA, vecB);
    else if (d.a_kcontig && !d.b_kcontig) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, st, d, vecA, vecB);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, st, d, vecA, vecB);
    MMFM_LAUNCH_CHECK("mmfm_gemm(f32)");
    return 0;
}

// Two independent GEMMs.  Two bf16 weight-gradient launches that the streaming kernel takes run as ONE launch, each on its share of the
// CUs; anything else is issued one after the other - same results either way.
extern "C" int mmfm_gemm_pair(const mmfm_gemm_desc* ap, const mmfm_gemm_desc* bp, mmfm_stream stream) {
    mmfm_gemm_desc a, b;
    if (int rc = gemm_check(ap, a)) return rc;
    if (int rc = gemm_check(bp, b)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a.dtype == MMFM_BF16 && b.dtype == MMFM_BF16) {
        const int rc = mmfm_gemm_dw_pair_launch(&a, &b, st);
        if (rc != -1000) return rc;
    }
    if (int rc = mmfm_gemm(ap, stream)) return rc;
    return mmfm_gemm(bp, stream);
}

// Several reductions in one launch: block b owns one 256-float chunk of one entry (found by bisection over the entries' chunk0
// prefix sums, <= 7 steps for the ~70 entries of a backward segment).
__global__ __launch_bounds__(256) void reduce_slabs_multi_kernel(const mmfm_reduce_entry* __restrict__ tab, int count) {
    int lo = 0, hi = count - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].chunk0 <= b) lo = mid; else hi = mid - 1;
    }
    const mmfm_reduce_entry e = tab[lo];
    const int64_t i = (int64_t)(b - e.chunk0) * 256 + threadIdx.x;
    if (i >= e.n) return;
    float s = e.accumulate ? e.dst[i] : 0.f;
    for (int k = 0; k < e.nslabs; ++k) s += e.src[(size_t)k * e.slab_stride + i];
    e.dst[i] = s;
}

extern "C" int mmfm_reduce_slabs_multi(const mmfm_reduce_entry* table, int count, int total_chunks, mmfm_stream stream) {
    MMFM_REQUIRE(table && count > 0 && total_chunks > 0, "mmfm_reduce_slabs_multi: bad arguments");
    hipLaunchKernelGGL(reduce_slabs_multi_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream, table, count);
    MMFM_LAUNCH_CHECK("mmfm_reduce_slabs_multi");
    return 0;
}

// NOTE: `src` is scratch and may be clobbered (the two-stage path sums each slab group into its first slab).
extern "C" int mmfm_reduce_slabs(float* dst, const float* src_c, int64_t n, int nslabs, int64_t slab_stride,
                                 int accumulate, mmfm_stream stream) {
    MMFM_REQUIRE(dst && src_c && n > 0 && nslabs > 0 && slab_stride >= n, "mmfm_reduce_slabs: bad arguments");
    float* src = const_cast<float*>(src_c);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (n % 4 == 0) && (slab_stride % 4 == 0) && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src % 16 == 0);
    if (!vec) {
        const int blocks = (int)std::min<int64_t>(2048, (n + 255) / 256);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, dst, src, n, nslabs, slab_stride, accumulate);
        MMFM_LAUNCH_CHECK("mmfm_reduce_slabs");
        return 0;
    }
    const int chunks = (int)((n + 255) / 256);
    // tall-skinny (few chunks, many slabs): split the slabs into groups first so that >= ~512 blocks stream
    int group = nslabs;
    if (nslabs >= 16 && chunks < 512) {
        const int want = std::min(nslabs / 4, std::max(1, 512 / chunks));     // number of groups
        group = (nslabs + want - 1) / want;
    }
    const int ngroups = (nslabs + group - 1) / group;
    if (ngroups > 1) {
        hipLaunchKernelGGL(reduce_slabs4_kernel<true>, dim3(chunks, ngroups), dim3(256), 0, st, dst, src, n, nslabs, slab_stride, group, 0);
        MMFM_LAUNCH_CHECK("mmfm_reduce_slabs(stage 1)");
        hipLaunchKernelGGL(reduce_slabs4_kernel<false>, dim3(chunks, 1), dim3(256), 0, st, dst, src, n, ngroups, slab_stride * group, ngroups,
                           accumulate);
    } else {
        hipLaunchKernelGGL(reduce_slabs4_kernel<false>, dim3(chunks, 1), dim3(256), 0, st, dst, src, n, nslabs, slab_stride, nslabs, accumulate);
    }
    MMFM_LAUNCH_CHECK("mmfm_reduce_slabs");
    return 0;
}

// short inputs (launch-bound regime): <= 15 partials so that the reduction is ONE small launch
static int colsum_splits(int64_t R) { return R <= 8192 ? (int)std::max<int64_t>(1, std::min<int64_t>(15, R / 256)) : (int)std::min<int64_t>(256, R / 64); }

extern "C" int64_t mmfm_colsum_workspace(int64_t R, int N) { return (int64_t)colsum_splits(R) * N * sizeof(float); }

extern "C" int mmfm_colsum(int dtype, const void* x, int64_t R, int N, int ld, float* out, int accumulate,
                           void* workspace, int64_t workspace_bytes, mmfm_stream stream) {
    MMFM_REQUIRE(x && out && R > 0 && N > 0 && ld >= N, "mmfm_colsum: bad arguments");
    const int S = colsum_splits(R);
    MMFM_REQUIRE(workspace && workspace_bytes >= mmfm_colsum_workspace(R, N), "mmfm_colsum: workspace too small");
    const int64_t rows_per = (R + S - 1) / S;
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = dtype == MMFM_F32 ? 4 : 2;
    const bool vec = (N % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)x % (4 * esz) == 0);
    MMFM_REQUIRE(dtype == MMFM_F32 || dtype == MMFM_BF16, "mmfm_colsum: bad dtype %d", dtype);
    if (vec) {
        dim3 grid(cdiv(N, 256), S);
        if (dtype == MMFM_F32)
            hipLaunchKernelGGL(colsum4_kernel<float>, grid, dim3(256), 0, st, (const float*)x, R, N, ld, (float*)workspace, rows_per);
        else
            hipLaunchKernelGGL(colsum4_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)x, R, N, ld, (float*)workspace, rows_per);
    } else {
        dim3 grid(cdiv(N, 64), S);
        if (dtype == MMFM_F32)
            hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)x, R, N, ld, (float*)workspace, rows_per);
        else
            hipLaunchKernelGGL(colsum_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)x, R, N, ld, (float*)workspace, rows_per);
    }
    MMFM_LAUNCH_CHECK("mmfm_colsum");
    return mmfm_reduce_slabs(out, (const float*)workspace, N, S, N, accumulate, stream);
}
