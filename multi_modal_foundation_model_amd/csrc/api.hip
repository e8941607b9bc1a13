// Library-level entry points: version, thread-local error text, device check, RNG state.
#include "common.h"
#include <string.h>
#include <mutex>
#include <unordered_map>

static thread_local char g_err[512] = "";

int mmfm_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code == 0 ? -1 : code;
}

int mmfm_lds_opt_in(const void* kern, size_t bytes, const char* what) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, bool> done;
    if (bytes <= 65536) return 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t key = (uint64_t)(uintptr_t)kern ^ ((uint64_t)(dev + 1) << 56);
    std::lock_guard<std::mutex> g(mu);
    if (done.count(key)) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return mmfm_set_error((int)e, "%s: hipFuncSetAttribute(%zu B LDS): %s", what, bytes, hipGetErrorString(e));
    done[key] = true;
    return 0;
}

extern "C" int mmfm_version(void) { return MMFM_VERSION; }
extern "C" const char* mmfm_last_error(void) { return g_err; }

extern "C" int mmfm_device_check(int device) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return mmfm_set_error((int)e, "mmfm_device_check: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return mmfm_set_error(-1, "mmfm_device_check: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return 0;
}

namespace {
__global__ void rng_seed_kernel(uint32_t* st, uint32_t lo, uint32_t hi) {
    st[0] = mix32(lo ^ mix32(hi + 0x9E3779B9u));
    st[1] = 0;
}
__global__ void rng_advance_kernel(uint32_t* st) { st[1] += 1; }
}  // namespace

extern "C" int mmfm_rng_seed(void* state, uint64_t seed, mmfm_stream stream) {
    MMFM_REQUIRE(state, "mmfm_rng_seed: null state");
    hipLaunchKernelGGL(rng_seed_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint32_t*)state, (uint32_t)seed, (uint32_t)(seed >> 32));
    MMFM_LAUNCH_CHECK("mmfm_rng_seed");
    return 0;
}
extern "C" int mmfm_rng_advance(void* state, mmfm_stream stream) {
    MMFM_REQUIRE(state, "mmfm_rng_advance: null state");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint32_t*)state);
    MMFM_LAUNCH_CHECK("mmfm_rng_advance");
    return 0;
}
