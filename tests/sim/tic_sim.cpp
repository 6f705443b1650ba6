// tic_sim.cpp -- TEST INFRASTRUCTURE: the C ABI of include/tic_hip.h compiled against the CPU
// wave-lockstep simulator (sim_runtime.h) instead of gfx950.  Pointers are host pointers.  Used by
// tests/test_sim_*.py to check lane maps, LDS swizzles and range-check handling of the kernel
// sources on tiny shapes without a GPU.  Never loaded by the product package.
#define TIC_SIM 1
#include <math.h>
#include <string.h>
#include <atomic>
#define TIC_RT_LAST_ERROR() ((const char*)nullptr)
#define TIC_RT_MEMSET(p, v, n, s) memset((p), (v), (n))
#define TIC_RT_MEMCPY(d, src, n, s) memcpy((d), (src), (n))
#define TIC_RT_MAX_LDS(kernel, bytes) do { } while (0)
#define TIC_RT_TIMER_MARK(which, stream) do { } while (0)
#define TIC_RT_TIMER_ENABLE(on) 0
#define TIC_RT_TIMER_READ(n, ms) (*(n) = 0, *(ms) = 0.f, 0)
static unsigned tic_sim_err_word = 0;
#define TIC_RT_ERR_WORD(host, dev) (*(host) = &tic_sim_err_word, *(dev) = &tic_sim_err_word, 0)
#define TIC_RT_IS_CAPTURING(stream) 0
#include "../../touhouimageclassification_amd/csrc/tic_api_impl.h"
