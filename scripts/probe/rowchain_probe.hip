// Diagnostic build (not part of the library): the plain row-owner GEMM loop with s_memtime stamps around its phases.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../multi_modal_foundation_model_amd/csrc rowchain_probe.hip -o rowchain_probe
#include "rowchain.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
using namespace rowchain;
int mmfm_set_error(int code, const char* fmt, ...) { return code; }

constexpr int NT = 256, NW = 4;
__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

template <int MODE>   // 0: full; 1: no stores; 2: no stager global loads (reuse regs); 3: no barrier+no LDS write (reads whatever is in LDS)
__global__ __launch_bounds__(NT) void probe_kernel(const uint16_t* X, const uint16_t* W, uint16_t* Yp, int64_t R, int N, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, h = lane >> 5;
    const int ntile = N >> 5, cpp = ntile;
    const int64_t npass = (R + 32 * NW - 1) / (32 * NW);
    const int my_passes = blockIdx.x < npass ? (int)((npass - 1 - blockIdx.x) / gridDim.x) + 1 : 0;
    if (my_passes == 0) return;
    auto src = [=](int g) { const int tt = g % cpp; WChunk c; c.base = W + (size_t)(32 * tt) * 256; c.ld = 256; c.kind = 0; return c; };
    const GBuf XB = gbuf(X, R * 512), YB = gbuf(Yp, R * N * 2);
    RING_DECL(NT);
    RING_START(smem, my_passes * cpp, src);
    unsigned long long tb = 0, tw = 0, tm = 0, te = 0, tx = 0;
    for (int pi = 0; pi < my_passes; ++pi) {
        const uint32_t row = (uint32_t)(((int64_t)(blockIdx.x + (int64_t)pi * gridDim.x) * NW + wave) * 32 + m);
        unsigned long long t0 = now();
        opnd x[16];
        load_rows<16>(x, XB, row * 512u, h);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tx += now() - t0;
        const uint32_t yoff = row * (uint32_t)(N * 2);
        for (int tt = 0; tt < ntile; ++tt) {
            unsigned long long a = now();
            if (MODE != 3) __syncthreads();
            unsigned long long b = now();
            if (MODE != 3) { const WChunk cw_ = src(min(ring_cc + 1, ring_last)); stage_write<NT>(ring_r, cw_.kind, ring_smem + ((ring_cc + 1) & 1) * CHUNK, t); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned long long c = now();
            if (MODE != 2 && MODE != 3) { const WChunk cf_ = src(min(ring_cc + 2, ring_last)); stage_load<NT>(ring_r, cf_, t); }
            const char* slot = ring_smem + (ring_cc & 1) * CHUNK; ++ring_cc;
            f32x16 acc = zero16();
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma(wfragA(slot, s, m, h), x[s], acc);
            asm volatile("s_nop 0" :: "v"(acc[0]));
            unsigned long long dd = now();
            if (MODE != 1) store_tile<true>(YB, yoff, tt, h, acc); else asm volatile("" :: "v"(acc[3]), "v"(acc[9]));
            unsigned long long e = now();
            tb += b - a; tw += c - b; tm += dd - c; te += e - dd;
        }
    }
    if (t == 0) { unsigned long long* o = stamps + blockIdx.x * 8; o[0] = tb; o[1] = tw; o[2] = tm; o[3] = te; o[4] = tx; o[5] = (unsigned long long)my_passes * ntile; }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024, N = argc > 2 ? atoi(argv[2]) : 768, percu = argc > 3 ? atoi(argv[3]) : 2;
    const int64_t R = (int64_t)B * 200;
    uint16_t *X, *W, *Y; unsigned long long* S;
    hipMalloc(&X, R * 512); hipMalloc(&W, (size_t)N * 512); hipMalloc(&Y, R * N * 2);
    std::vector<uint16_t> hx(R * 256), hw((size_t)N * 256);
    for (auto& v : hx) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    for (auto& v : hw) v = 0x3800 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    hipMemcpy(X, hx.data(), R * 512, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), (size_t)N * 512, hipMemcpyHostToDevice);
    const int64_t npass = (R + 127) / 128;
    const int grid = (int)std::min<int64_t>(npass, 256 * percu);
    hipMalloc(&S, grid * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(grid), dim3(NT), 0, 0, X, W, Y, R, N, S);
            if (mode == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(grid), dim3(NT), 0, 0, X, W, Y, R, N, S);
            if (mode == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(grid), dim3(NT), 0, 0, X, W, Y, R, N, S);
            if (mode == 3) hipLaunchKernelGGL(probe_kernel<3>, dim3(grid), dim3(NT), 0, 0, X, W, Y, R, N, S);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
        }
        std::vector<unsigned long long> hs(grid * 8);
        hipMemcpy(hs.data(), S, grid * 64, hipMemcpyDeviceToHost);
        double tb = 0, tw = 0, tm = 0, te = 0, tx = 0, n = 0, np = 0;
        for (int g = 0; g < grid; ++g) { tb += hs[g * 8]; tw += hs[g * 8 + 1]; tm += hs[g * 8 + 2]; te += hs[g * 8 + 3]; tx += hs[g * 8 + 4]; n += hs[g * 8 + 5]; }
        np = n / (N / 32);
        printf("mode %d: %.1f us | per chunk (s_memtime ticks = 100 MHz? see ratio): barrier %.0f  write %.0f  mfma+issue %.0f  store %.0f | x-load per pass %.0f | chunks/block %.0f grid %d\n",
               mode, best * 1e3, tb / n, tw / n, tm / n, te / n, tx / np, n / grid, grid);
    }
    return 0;
}
