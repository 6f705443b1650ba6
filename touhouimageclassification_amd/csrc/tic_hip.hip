// tic_hip.hip -- the single translation unit of libtic_hip.so (gfx950 device code + C ABI).
// Build:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared tic_hip.hip -o libtic_hip.so   (see build.py)
#include <hip/hip_runtime.h>
#include <math.h>
#include <atomic>

static inline const char* tic_rt_last_error() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}
#define TIC_RT_LAST_ERROR() tic_rt_last_error()
#define TIC_RT_MEMSET(p, v, n, s)                                                        \
    do {                                                                                  \
        if (hipMemsetAsync((p), (v), (n), (hipStream_t)(s)) != hipSuccess)                \
            return tic_fail(TIC_ELAUNCH, "hipMemsetAsync failed: %s", tic_rt_last_error()); \
    } while (0)
#define TIC_RT_MEMCPY(d, src, n, s)                                                                   \
    do {                                                                                               \
        if (hipMemcpyAsync((d), (src), (n), hipMemcpyDeviceToDevice, (hipStream_t)(s)) != hipSuccess)  \
            return tic_fail(TIC_ELAUNCH, "hipMemcpyAsync failed: %s", tic_rt_last_error());            \
    } while (0)
// kernels that use more than the default dynamic-LDS limit opt in once per call site
#define TIC_RT_MAX_LDS(kernel, bytes)                                                                            \
    do {                                                                                                          \
        static bool done_ = false;                                                                                \
        if (!done_) {                                                                                             \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)(bytes)) != hipSuccess)                                                  \
                return tic_fail(TIC_ELAUNCH, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", tic_rt_last_error()); \
            done_ = true;                                                                                         \
        }                                                                                                         \
    } while (0)

static inline int tic_rt_err_word(unsigned** host, unsigned** dev) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return -1;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) return -1;
    *host = (unsigned*)h;
    *dev = (unsigned*)d;
    **host = 0;
    return 0;
}
#define TIC_RT_ERR_WORD(host, dev) tic_rt_err_word((host), (dev))
static inline int tic_rt_is_capturing(void* stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) return 0;
    return st != hipStreamCaptureStatusNone;
}
#define TIC_RT_IS_CAPTURING(stream) tic_rt_is_capturing(stream)

// ---- live kernel timer (bench.py roofline): HIP events around every grouped-dW launch, on the launch stream ----------
#define TIC_TIMER_SLOTS 8192
static bool g_timer_on = false;
static int g_timer_n = 0;
static hipEvent_t g_timer_ev[TIC_TIMER_SLOTS][2];
static bool g_timer_made = false;
static inline void tic_rt_timer_mark(int which, void* stream) {
    if (!g_timer_on || g_timer_n >= TIC_TIMER_SLOTS) return;
    hipEventRecord(g_timer_ev[g_timer_n][which], (hipStream_t)stream);
    if (which == 1) ++g_timer_n;
}
static inline int tic_rt_timer_enable(int on) {
    if (on && !g_timer_made) {
        for (int i = 0; i < TIC_TIMER_SLOTS; ++i)
            if (hipEventCreate(&g_timer_ev[i][0]) != hipSuccess || hipEventCreate(&g_timer_ev[i][1]) != hipSuccess) return -1;
        g_timer_made = true;
    }
    g_timer_on = on != 0;
    g_timer_n = 0;
    return 0;
}
static inline int tic_rt_timer_read(int* launches, float* total_ms) {
    float tot = 0.f;
    for (int i = 0; i < g_timer_n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_timer_ev[i][1]) != hipSuccess || hipEventElapsedTime(&ms, g_timer_ev[i][0], g_timer_ev[i][1]) != hipSuccess) return -1;
        tot += ms;
    }
    *launches = g_timer_n;
    *total_ms = tot;
    return 0;
}
#define TIC_RT_TIMER_MARK(which, stream) tic_rt_timer_mark((which), (stream))
#define TIC_RT_TIMER_ENABLE(on) tic_rt_timer_enable(on)
#define TIC_RT_TIMER_READ(n, ms) tic_rt_timer_read((n), (ms))

#include "tic_api_impl.h"
