// tic_hip.hip -- the single translation unit of libtic_hip.so (gfx950 device code + C ABI).
// Build:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared tic_hip.hip -o libtic_hip.so   (see build.py)
#include <hip/hip_runtime.h>
#include <math.h>

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

#include "tic_api_impl.h"
