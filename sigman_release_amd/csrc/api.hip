// api.hip -- C-ABI plumbing shared by every entry point of libsigman_gsplat.so: error string, version,
// and the optional per-kernel HIP-event profiler bench.py uses for its roofline numbers.
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void sgr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *sgr_last_error(void) { return g_err; }
extern "C" int sgr_abi_version(void) { return SGR_ABI_VERSION; }

static thread_local int g_debug = 0;
int sgr_debug_enabled() { return g_debug; }
extern "C" int sgr_set_debug(int enable) { const int old = g_debug; g_debug = enable ? 1 : 0; return old; }

// ---------------------------------------------------------------------------------------------
// profiler: hipEventRecord pairs on the launch stream around selected kernels (off by default; when off the
// cost is one predictable branch per launch).  Single-threaded use only (bench.py).
// ---------------------------------------------------------------------------------------------
namespace {
struct Slot { hipEvent_t a, b; int kid; };
uint32_t g_mask = 0;
std::vector<Slot> g_slots;
size_t g_used = 0;
}  // namespace

int sgr_prof_begin(int kid, hipStream_t s) {
    if (!((g_mask >> kid) & 1u)) return -1;
    if (g_used == g_slots.size()) {
        Slot sl;
        if (hipEventCreate(&sl.a) != hipSuccess || hipEventCreate(&sl.b) != hipSuccess) return -1;
        g_slots.push_back(sl);
    }
    Slot &sl = g_slots[g_used];
    sl.kid = kid;
    (void)hipEventRecord(sl.a, s);
    return (int)g_used++;
}
void sgr_prof_end(int slot, hipStream_t s) {
    if (slot >= 0) (void)hipEventRecord(g_slots[slot].b, s);
}

int sgr_prof_active() { return g_mask != 0; }

extern "C" int sgr_prof_configure(uint32_t kernel_mask) {
    g_mask = kernel_mask;
    g_used = 0;
    return 0;
}

// Sums the recorded durations per kernel id (SGR_K_*), clears the recordings.  Caller must have synchronised.
extern "C" int sgr_prof_collect(double *total_ms /*[SGR_K_COUNT]*/, uint32_t *counts /*[SGR_K_COUNT]*/) {
    for (int k = 0; k < SGR_K_COUNT; k++) { total_ms[k] = 0.0; counts[k] = 0; }
    for (size_t i = 0; i < g_used; i++) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(g_slots[i].b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, g_slots[i].a, g_slots[i].b);
        if (e != hipSuccess) { sgr_set_error("sgr_prof_collect: %s", hipGetErrorString(e)); return 1; }
        total_ms[g_slots[i].kid] += ms;
        counts[g_slots[i].kid] += 1;
    }
    g_used = 0;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// shader-clock probe (bench.py): one wave spins on a dependent FMA chain for ~0.5 ms and reads both of the chip's counters around it -- the
// cycle counter of its own clock domain (s_memtime) and the constant 100-MHz one (s_memrealtime); their ratio is the shader clock the wave
// actually ran at.  (The amdgpu sysfs node bench.py used to trust reported 157 MHz and 95 MHz under full load on some boxes of this pool.)
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void clock_probe_kernel(unsigned long long *out, int iters) {
    float x = (float)threadIdx.x * 1e-3f;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) x = fmaf(x, 0.999f, 1e-3f);
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (x == 123.456f) out[2] = 1ull;                        // (keeps the chain alive)
}
}  // namespace

extern "C" int sgr_clock_probe(double *mhz_host, void *stream_) {
    if (!mhz_host) { sgr_set_error("sgr_clock_probe: NULL argument"); return 1; }
    hipStream_t stream = (hipStream_t)stream_;
    unsigned long long *dev = nullptr, host[3] = {0, 0, 0};
    SGR_CHECK_HIP(hipMalloc((void **)&dev, sizeof(host)));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, dev, 250000);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(dev);
    if (e != hipSuccess) { sgr_set_error("sgr_clock_probe: %s", hipGetErrorString(e)); return 1; }
    *mhz_host = host[1] ? 100.0 * (double)host[0] / (double)host[1] : 0.0;
    return 0;
}
