// tic_api_impl.h -- host side of the C ABI declared in include/tic_hip.h: argument checks, grid
// selection, kernel launches, and the whole-model phase drivers.  Compiled into libtic_hip.so by
// tic_hip.hip (device build) and, for CPU-side lane-map checks, into the test-only simulator
// library by tests/sim/tic_sim.cpp; the TIC_RT_* macros are the only difference.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tic_hip.h"
#include "attention.h"
#include "aug.h"
#include "conv.h"
#include "elementwise.h"
#include "gemm.h"
#include "gemm256.h"
#include "gemm_tn256.h"
#include "moe.h"
#include "norm.h"

static thread_local char g_tic_err[512] = "";
static int tic_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_tic_err, sizeof(g_tic_err), fmt, ap);
    va_end(ap);
    return code;
}
#define TIC_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return tic_fail(TIC_EINVAL, __VA_ARGS__); \
    } while (0)
#define TIC_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)
#define TIC_TRY(call)            \
    do {                         \
        const int rc_ = (call);  \
        if (rc_ != TIC_OK) return rc_; \
    } while (0)

// device -> host error word of the split-K hand-offs (tic_prims.h flag_wait): 64 bytes of pinned, device-mapped HOST memory, made once
// when a split-K scratch is first registered -- the one allocation this library makes, and not device memory.  Sticky: once set, every
// call fails with TIC_ELAUNCH until tic_gemm_nt_scratch() registers a scratch again (the caller re-zeroes the flags with it).
static unsigned* g_err_host = nullptr;
static unsigned* g_err_dev = nullptr;
static int tic_after_launch(const char* what) {
    const char* e = TIC_RT_LAST_ERROR();
    if (e) return tic_fail(TIC_ELAUNCH, "%s: %s", what, e);
    if (g_err_host) {
        const unsigned w = *(volatile unsigned*)g_err_host;
        if (w) return tic_fail(TIC_ELAUNCH, "%s: a split-K GEMM consumer timed out waiting for its producer's flag (code 0x%08x): the activations / "
                               "gradients written since are invalid; re-register the scratch (tic_gemm_nt_scratch) after zeroing its flag words", what, w);
    }
    return TIC_OK;
}

// tuning knobs (process-wide, for A/B measurements and tests): "gemm_tile" = 0 (auto) | 128 | 256 ;
// "tn_streamk" = 1 (default: grouped dW as 256 equal stream-K shares) | 0 (one workgroup per full-M tile)
static int g_opt_gemm_tile = 0;
static int g_opt_nt = 13;         // non-temporal loads / stores for once-touched streams (tools/ab_step.py stream_nt a b): 1 LayerNorm (-0.45 % step),
                                  // 2 AdamW (no effect: off), 4 staged GEMM epilogue stores (-0.8 %), 8 its residual / derivative operand loads (-0.1 %)
static int g_opt_ln_blocks = 4096;   // grid cap of the LayerNorm kernels (4 rows per block per pass)
static int g_opt_gemm_gm = 8;
static int g_opt_attn_fwd_waves = 8;   // attention forward: waves per (image, head) workgroup, 4 or 8 (two workgroups per CU either way; 8: 179 -> 150 us)
static int g_opt_tn_main_bias = 0;   // extra M steps given to the 'main' workgroups of the phase-aligned stream-K split (tail ones pay 3 prologues / epilogues)
#ifdef TIC_MEASURE
static int g_opt_gemm_dbg = 0;                   // measurement build (libtic_hip_dbg.so) only
static unsigned long long* g_dbg_stamps = nullptr;   // device buffer [grid][8] for the stage stamps of ONE 256x256 NT launch
static int g_dbg_stamp_at = -1, g_dbg_nt_launches = 0;
static int g_dbg_log[4096][4];                       // (epilogue, M, N, K) of the 256x256 NT launches since the last reset
// stamps go to dev_ptr for the launch_index-th 256x256 NT launch after this call (-1: every launch); also resets the launch log
extern "C" int tic_dbg_set_stamp_buffer(void* dev_ptr, int launch_index) {
    g_dbg_stamps = (unsigned long long*)dev_ptr;
    g_dbg_stamp_at = launch_index;
    g_dbg_nt_launches = 0;
    return TIC_OK;
}
extern "C" int tic_dbg_launch_log(int i, int* out4) {
    if (i < 0 || i >= g_dbg_nt_launches || i >= 4096) return TIC_EINVAL;
    for (int j = 0; j < 4; ++j) out4[j] = g_dbg_log[i][j];
    return TIC_OK;
}
#endif
static int g_opt_tn_slab = 1;             // few-tile weight gradients: row parts stored to the tic_gemm_tn_scratch slab + one reduce launch (1) | stream-K atomics (0)
static int g_opt_tn_split_wgs = 1024;    // split-M weight-gradient kernels (128x128 tiles): MOST workgroups per launch (every split adds its tile to C with fp32 atomics)
static int g_opt_tn_split_rows = 2048;   // ... rows of the reduction a workgroup is given, and ...
static int g_opt_tn_split_min_wgs = 384; // ... FEWEST workgroups (tn_split below)
static int g_opt_tn_slab_min_tiles = 8;  // single weight gradients: the 256x256 row-parts + slab route from this many 256x256 tiles on (below: split-M 128x128 kernel)
// Row splits of the 128x128 weight-gradient kernels (1x1 and implicit 3x3 / 7x7 convolutions, small Linear layers): every split costs a
// 64 KiB tile of fp32 atomics (1.3 TB/s chip-wide) and a prologue, a workgroup needs ~25-50 steps of 64 rows to amortise them, and the
// chip wants >= ~1.5 workgroups per CU.  Measured over every convolution of ResNet-50 at 256 images and ResNet-152 at 80
// (tools/resnet_conv_table.py with tn_split_wgs = 256 ... 2048, profiles/r03_conv_wgrad_routes.log): the best split keeps 1 600-3 600 rows
// per workgroup with 384-1 024 workgroups in all; a fixed 1 024 (round 2) was right for 802 816-row layers and cost 20-60 % on the
// 5 120 ... 81 920-row layers of ResNet-152 at the reference's batch.
static int tn_split(int M, int tiles) {
    int split = (M + g_opt_tn_split_rows - 1) / g_opt_tn_split_rows;
    const int lo = (g_opt_tn_split_min_wgs + tiles - 1) / tiles, hi = g_opt_tn_split_wgs / tiles;
    if (split < lo) split = lo;
    if (split > hi) split = hi;
    const int max_split = (M + 63) / 64;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    return split;
}
static int g_opt_tn_parts = -1;           // grouped dW without a phase-aligned split: -1 auto (256 / tiles equal row parts per tile), 0 never (flat stream-K), n forced
static int g_opt_tn_streamk_min_steps = 128;   // grouped dW: fewest 64-row steps for which the stream-K split is chosen when C is accumulated into ...
static int g_opt_tn_streamk_min_steps_store = 352;   // ... and when it is stored (overwrite): the one-workgroup-per-tile form then has no read of C and no
                                               // zeroing pass to pay for, and keeps winning up to ~110 images of ViT-L (tools/ab_step.py tn_streamk 1 0 in the
                                               // fused step: 48 images 25.93 -> 24.84 ms, 64: 31.17 -> 30.59, 83: 36.36 -> 35.91, 100: 46.57 -> 46.26, 128: 55.59 <- 56.12)
static int g_opt_tn_mfma16_min_steps = 256;    // stream-K launch: v_mfma_f32_16x16x32_bf16 from this many steps on (in the step: -0.5 ... -0.7 % at 64-128 images too)
static int g_opt_ln_bwd_blocks = 512;    // LayerNorm backward: most blocks per launch
static int g_opt_ln_bwd_rows = 2;        // LayerNorm backward: fewest rows per wave (bounds the number of dgamma / dbeta atomic rows)
static int g_opt_gemm_big_tiles = 128;   // fewest 256x256 tiles for which the 256x256 NT kernel is chosen (gemm_tile = 0)
static int g_opt_gemm_split = -1;       // split-K form of the 128x128 NT kernel (and, measurement build, the 256x256 one): -1 auto, 0 / 1 never, 2 / 4 forced where legal
static int g_opt_gemm_stagger_mask = 0x7f;   // bit e: apply "gemm_stagger" to epilogue e
static int g_opt_gemm_stagger_groups = 2; // groups of first-round workgroups; group g starts g x gemm_stagger sleep rounds late
static int g_opt_gemm_stagger = -1;   // -1: auto (see gemm_nt), 0: off, n: s_sleep rounds; gemm256.h    // measurement only: see gemm256.h DBG
static int g_opt_tn_streamk = 1;
static int g_opt_tn_block = -1;   // tile-walk block width of the grouped dW launch: -1 auto (one XCD share per block), 0 row-major, n fixed
static int g_opt_tn_mfma = 0;     // MFMA shape of the grouped dW stream-K launch: 0 auto (16x16x32 from g_opt_tn_mfma16_min_steps M steps per tile on: -1.7 % at
                                  // M = 65 404; stand-alone it measured +2..5 % at M = 12 608 (tools/dw_ab.py tn_mfma 16 32), inside the step -0.7 %), 16, 32
static int g_opt_tn_phase = 1;   // 1: phase-aligned stream-K split when the tile count allows; 0: always the flat split
#if defined(TIC_SIM) || defined(TIC_MEASURE)
static int g_opt_tn_waves = 8;   // experiment: 4 = the one-wave-per-SIMD form of the grouped dW tile (gemm_tn256.h tn256_tile_segment16_w4)
static int g_opt_attn_legacy = 0;   // test / measurement builds: attention backward with the round-1 tile-to-wave assignment
static int g_opt_nt_deep = -1;    // test / measurement builds: 4-stage form of the 128x128 NT kernel: -1 auto (<= 256 workgroups), 0 never, 1 always
static int g_opt_nt_fault = 0;   // test builds: part 0 of every split tile withholds its flag (exercises the timeout report of flag_wait)
#endif
// The SIX knobs of the product library select between numerically equivalent routes, so that the parity tests can force each one
// (include/tic_hip.h documents them).  They are plain process-wide ints read at launch time: set them while no other host thread is
// inside a tic_* call.  Everything else that was ever A/B-ed lives in the measurement build only (-DTIC_MEASURE, libtic_hip_dbg.so).
extern "C" int tic_set_option(const char* name, int value) {
    if (name && !strcmp(name, "gemm_tile") && (value == 0 || value == 128 || value == 256)) {
        g_opt_gemm_tile = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "gemm_split") && (value == -1 || value == 0 || value == 1 || value == 2 || value == 4)) {
        g_opt_gemm_split = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "tn_streamk") && value >= 0 && value <= 4096) {   // 0 off, 1 = 256 shares, n > 1 = n shares (tests)
        g_opt_tn_streamk = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "tn_parts") && value >= -1 && value <= 8) {
        g_opt_tn_parts = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "tn_mfma") && (value == 0 || value == 16 || value == 32)) {
        g_opt_tn_mfma = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "stream_nt") && value >= 0 && value < 16) {
        g_opt_nt = value;
        return TIC_OK;
    }
#if defined(TIC_SIM) || defined(TIC_MEASURE)
    if (name && !strcmp(name, "tn_waves") && (value == 4 || value == 8)) {
        g_opt_tn_waves = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "attn_legacy") && (value == 0 || value == 1)) {
        g_opt_attn_legacy = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "nt_deep") && value >= -1 && value <= 1) {
        g_opt_nt_deep = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "tn_slab_min_tiles") && value >= 1 && value <= 4096) {
        g_opt_tn_slab_min_tiles = value;
        return TIC_OK;
    }
    if (name && !strcmp(name, "nt_fault") && (value == 0 || value == 1)) {
        g_opt_nt_fault = value;
        return TIC_OK;
    }
#endif
#ifdef TIC_MEASURE
#define TIC_KNOB(NAME, VAR, LO, HI)                                      \
    if (name && !strcmp(name, NAME) && value >= (LO) && value <= (HI)) { \
        VAR = value;                                                     \
        return TIC_OK;                                                   \
    }
    TIC_KNOB("ln_blocks", g_opt_ln_blocks, 64, 65536)
    TIC_KNOB("attn_fwd_waves", g_opt_attn_fwd_waves, 4, 8)
    TIC_KNOB("tn_main_bias", g_opt_tn_main_bias, 0, 64)
    TIC_KNOB("gemm_gm", g_opt_gemm_gm, 1, 256)
    TIC_KNOB("tn_slab", g_opt_tn_slab, 0, 1)
    TIC_KNOB("tn_split_wgs", g_opt_tn_split_wgs, 64, 8192)
    TIC_KNOB("tn_split_rows", g_opt_tn_split_rows, 64, 1 << 20)
    TIC_KNOB("tn_split_min_wgs", g_opt_tn_split_min_wgs, 1, 8192)

    TIC_KNOB("tn_streamk_min_steps", g_opt_tn_streamk_min_steps, 1, 1 << 20)
    TIC_KNOB("tn_streamk_min_steps_store", g_opt_tn_streamk_min_steps_store, 1, 1 << 20)
    TIC_KNOB("tn_mfma16_min_steps", g_opt_tn_mfma16_min_steps, 1, 1 << 20)
    TIC_KNOB("ln_bwd_blocks", g_opt_ln_bwd_blocks, 64, 65536)
    TIC_KNOB("ln_bwd_rows", g_opt_ln_bwd_rows, 1, 64)
    TIC_KNOB("gemm_big_tiles", g_opt_gemm_big_tiles, 1, 65536)
    TIC_KNOB("gemm_stagger_mask", g_opt_gemm_stagger_mask, 0, 127)
    TIC_KNOB("gemm_stagger", g_opt_gemm_stagger, -1, 64)
    TIC_KNOB("gemm_stagger_groups", g_opt_gemm_stagger_groups, 2, 32)
    TIC_KNOB("gemm_dbg", g_opt_gemm_dbg, 0, 31)   // garbage results by construction
    TIC_KNOB("tn_block", g_opt_tn_block, -1, 64)
    TIC_KNOB("tn_phase", g_opt_tn_phase, 0, 1)
#undef TIC_KNOB
#endif
    return tic_fail(TIC_EINVAL, "set_option: unknown option/value %s=%d", name ? name : "(null)", value);
}

extern "C" int tic_version(void) { return TIC_ABI_VERSION; }
// live timing of the dominant kernel (the grouped stream-K dW launch): HIP events on the launch stream around every launch
// while enabled; read() synchronises those events and returns the launch count and their summed duration
extern "C" int tic_kernel_timer_enable(int on) {
    if (TIC_RT_TIMER_ENABLE(on) != 0) return tic_fail(TIC_ELAUNCH, "kernel_timer: hipEventCreate failed");
    return TIC_OK;
}
extern "C" int tic_kernel_timer_read(int* launches, float* total_ms) {
    TIC_REQUIRE(launches && total_ms, "kernel_timer_read: null pointer");
    if (TIC_RT_TIMER_READ(launches, total_ms) != 0) return tic_fail(TIC_ELAUNCH, "kernel_timer_read: event query failed");
    return TIC_OK;
}
extern "C" const char* tic_last_error_string(void) { return g_tic_err; }

// ---- GEMM ------------------------------------------------------------------------------------------
// Scratch for the split-K form of the big NT kernel: caller-owned, >= TIC_NT_SCRATCH_BYTES, the flag words (its last 4 KiB) zeroed
// ONCE by the caller; per host thread (one stream at a time).  Without it launches never split.
#define TIC_NT_SLAB_BYTES ((size_t)192 * 32 * 512 * 16)   /* 192 producer workgroups x 256 KiB */
static thread_local char* g_nt_scratch = nullptr;
static std::atomic<unsigned> g_nt_epoch{0};   // launch counter shared by every host thread (the flag words of two scratches may alias nothing,
                                              // but one counter keeps every epoch unique process-wide)
extern "C" int tic_gemm_nt_scratch(void* scratch, size_t bytes) {
    TIC_REQUIRE(!scratch || (bytes >= TIC_NT_SLAB_BYTES + 4096 && TIC_ALIGNED16(scratch)), "gemm_nt_scratch: need %zu bytes, 16-byte aligned", TIC_NT_SLAB_BYTES + 4096);
    if (scratch) {
        if (!g_err_host && TIC_RT_ERR_WORD(&g_err_host, &g_err_dev) != 0) return tic_fail(TIC_ELAUNCH, "gemm_nt_scratch: cannot map the host error word");
        *(volatile unsigned*)g_err_host = 0;
    }
    g_nt_scratch = (char*)scratch;
    return TIC_OK;
}
// Scratch for few-tile weight gradients (tic_gemm_tn_bf16 on its 256x256 route, gemm_tn256_parts_slab_kernel): caller-owned, per host
// thread.  Without it (or when parts x N x K x 4 bytes do not fit) those launches stay stream-K shares with fp32 atomics.
static thread_local float* g_tn_slab = nullptr;
static thread_local size_t g_tn_slab_bytes = 0;
extern "C" int tic_gemm_tn_scratch(void* scratch, size_t bytes) {
    TIC_REQUIRE(!scratch || TIC_ALIGNED16(scratch), "gemm_tn_scratch: need 16-byte alignment");
    g_tn_slab = (float*)scratch;
    g_tn_slab_bytes = scratch ? bytes : 0;
    return TIC_OK;
}
extern "C" int tic_gemm_nt_bf16_ex(const void* A, const void* B, int M, int N, int K, int epilogue, const float* bias,
                                   void* out_bf16, void* out2_bf16, float* out_f32, const float* resid,
                                   const void* aux_bf16, const float* rowtab, int patches, float* colsum, tic_stream_t stream);
extern "C" int tic_gemm_nt_bf16(const void* A, const void* B, int M, int N, int K, int epilogue, const float* bias,
                                void* out_bf16, void* out2_bf16, float* out_f32, const float* resid,
                                const void* aux_bf16, const float* rowtab, int patches, tic_stream_t stream) {
    return tic_gemm_nt_bf16_ex(A, B, M, N, K, epilogue, bias, out_bf16, out2_bf16, out_f32, resid, aux_bf16, rowtab, patches, nullptr, stream);
}
extern "C" int tic_gemm_nt_bf16_ex(const void* A, const void* B, int M, int N, int K, int epilogue, const float* bias,
                                   void* out_bf16, void* out2_bf16, float* out_f32, const float* resid,
                                   const void* aux_bf16, const float* rowtab, int patches, float* colsum, tic_stream_t stream) {
    TIC_REQUIRE(A && B, "gemm_nt: null operand");
    TIC_REQUIRE(!colsum || epilogue == TIC_EPI_BF16 || epilogue == TIC_EPI_DGELU || epilogue == TIC_EPI_MULAUX,
                "gemm_nt: colsum is available for EPI_BF16 / EPI_DGELU / EPI_MULAUX only");
    TIC_REQUIRE(M >= 1 && N >= 8 && K >= 64, "gemm_nt: bad shape M=%d N=%d K=%d", M, N, K);
    TIC_REQUIRE(N % 8 == 0 && K % 64 == 0, "gemm_nt: need N %% 8 == 0 and K %% 64 == 0 (N=%d K=%d)", N, K);
    TIC_REQUIRE(TIC_ALIGNED16(A) && TIC_ALIGNED16(B), "gemm_nt: operands must be 16-byte aligned");
    const long tiles_m = (M + 127) / 128;
    TIC_REQUIRE(((double)tiles_m * 128.0 + 128.0) * K * 2.0 < 4294967296.0 && (double)N * K * 2.0 < 4294967296.0,
                "gemm_nt: operand exceeds the 4 GiB buffer-resource range");
    GemmNtParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.N = N; p.K = K;
    p.bias = (epilogue == TIC_EPI_DGELU || epilogue == TIC_EPI_MULAUX || epilogue == TIC_EPI_ADDAUX) ? nullptr : bias;

    p.out = (bf16_t*)out_bf16; p.out2 = (bf16_t*)out2_bf16; p.out_f32 = out_f32; p.resid = resid;
    p.aux = (const bf16_t*)aux_bf16; p.rowtab = rowtab; p.patches = patches; p.colsum = colsum;
    // products with at least g_opt_gemm_big_tiles (128 = half the CUs) 256x256 tiles go to the deep-pipelined 256x256 kernel (one
    // workgroup per CU); below that the K loop's latency, not throughput, sets the time and the 128x128 kernel's four times as many
    // workgroups win (tools/small_batch_bench.py: 8-32 images per GPU, N = 1024: 16.7 vs 28.3, 46.9 vs 74.1, 23.9 vs 34.8 us)
    const bool big = (N % 256 == 0) && (g_opt_gemm_tile == 256 ||
                                        (g_opt_gemm_tile == 0 && (long)((M + 255) / 256) * (N / 256) >= (long)g_opt_gemm_big_tiles));
    const int grid = big ? (int)(((M + 255) / 256) * (N / 256)) : (int)(tiles_m * ((N + 127) / 128));
    // experiment knob (off by default): every other first-wave workgroup starts n x ~4 us late so that the two halves of the
    // chip alternate between epilogue and main loop.  Stand-alone back-to-back launches of fc1+GELU gain 11 %
    // (tools/stagger_probe.py); inside the training step the gain is zero (tools/ab_step.py gemm_stagger 0 2: 141.8 vs
    // 141.9 ms) -- there the previous kernel's tail already starts the CUs at different times.
    p.stagger = (g_opt_gemm_stagger > 0 && ((g_opt_gemm_stagger_mask >> epilogue) & 1)) ? (g_opt_gemm_stagger | (g_opt_gemm_stagger_groups << 8)) : 0;
    p.nt = (g_opt_nt >> 2) & 3;
    p.gm = g_opt_gemm_gm;
#ifdef TIC_MEASURE
    p.stamps = nullptr;
    if (big) {
        if (g_dbg_stamps && (g_dbg_stamp_at < 0 || g_dbg_stamp_at == g_dbg_nt_launches)) p.stamps = g_dbg_stamps;
        if (g_dbg_nt_launches < 4096) {
            g_dbg_log[g_dbg_nt_launches][0] = epilogue; g_dbg_log[g_dbg_nt_launches][1] = M;
            g_dbg_log[g_dbg_nt_launches][2] = N; g_dbg_log[g_dbg_nt_launches][3] = K;
        }
        ++g_dbg_nt_launches;
    }
#endif
    int split = 1;
#ifdef TIC_MEASURE
    // measurement build only -- the 256x256 kernel's split-K form (gemm256.h SPLITK): it lost to the 128x128 kernel's at every shape it
    // is legal for and is not part of the product library.  K tiles per part must be even, every workgroup resident at once
    if (big && g_nt_scratch && g_opt_gemm_split > 1 && (epilogue == TIC_EPI_BF16 || epilogue == TIC_EPI_RESID)) {
        const int nk_ = K / 64, sp = g_opt_gemm_split;
        if (nk_ % (2 * sp) == 0 && grid * sp <= 256) split = sp;
    }
#endif
    // few tiles, long reduction (ViT-L at 8-16 images per GPU, N = 1024, K >= 3072), scratch registered: split-K form of the 128x128 kernel
    // (two workgroups per CU = 512 slots): 2 or 4 parts while every part keeps >= 16 K tiles and the launch still fits the slots; any
    // epilogue (the consumer runs the ordinary one on the summed accumulators)
    if (!big && g_nt_scratch && g_opt_gemm_split != 0 && g_opt_gemm_split != 1 && epilogue != TIC_EPI_PATCH && N % 128 == 0) {
        const int nk_ = K / 64;
        for (int sp = 4; sp >= 2; sp >>= 1) {
            if (g_opt_gemm_split > 0 && sp != g_opt_gemm_split) continue;
            if (nk_ % sp != 0 || (long)grid * sp > 512 || grid > 256 || (size_t)grid * (sp - 1) * 65536 > TIC_NT_SLAB_BYTES) continue;
            // auto: two parts only (four measured slower at every shape: 38.0 vs 34.0, 42.7 vs 33.8 us), each >= 16 K tiles, and only while
            // both parts of every tile still get a CU of their own (<= 256 workgroups: the 4-stage ring below; 200 tiles at 16 images:
            // unsplit ring 41.7 / 38.9 / 30.9 us against 46.7 / 43.7 / 38.1 split on the 2-stage loop, profiles/r03_small_batch_ring.log)
            if (g_opt_gemm_split < 0 && (sp != 2 || nk_ / sp < 16 || (long)grid * sp > 256)) continue;
            split = sp;
            break;
        }
    }
    p.slab = (float*)g_nt_scratch;
    p.flags = (unsigned*)(g_nt_scratch ? g_nt_scratch + TIC_NT_SLAB_BYTES : nullptr);
    // a captured launch would replay with the epoch baked into its arguments: from the second replay the flags already hold it and the
    // consumer would read its slabs before they are written.  Under stream capture launches never split.
    if (split > 1 && TIC_RT_IS_CAPTURING(stream)) split = 1;
    p.split = split;
    p.epoch = split > 1 ? ++g_nt_epoch : 0;
    if (split > 1 && p.epoch == 0) p.epoch = ++g_nt_epoch;   // 0 is what freshly zeroed flags hold
    p.err = g_err_dev;
    p.fault = 0;
    // at most one workgroup per CU anyway: the 4-stage ring (128 KiB of LDS) covers the load latency that a second resident workgroup
    // would otherwise hide (gemm.h NST)
    bool deep = !big && (long)grid * split <= 256;
#if defined(TIC_SIM) || defined(TIC_MEASURE)
    p.fault = g_opt_nt_fault;
    if (g_opt_nt_deep >= 0) deep = !big && g_opt_nt_deep == 1;
#endif
#ifdef TIC_MEASURE
#define TIC_GEMM_NT_LAUNCH_B(E)                                                                      \
    do {                                                                                             \
        if (split > 1 && big) {                                                                      \
            TIC_RT_MAX_LDS((gemm_nt256_kernel<E, 0, 1>), G256_NT_LDS_BYTES);                         \
            TIC_LAUNCH((gemm_nt256_kernel<E, 0, 1>), grid * split, 512, G256_NT_LDS_BYTES, stream, p); \
        } else {                                                                                     \
            TIC_GEMM_NT_LAUNCH(E);                                                                   \
        }                                                                                            \
    } while (0)
#else
#define TIC_GEMM_NT_LAUNCH_B(E) TIC_GEMM_NT_LAUNCH(E)
#endif
#define TIC_GEMM_NT_LAUNCH(E)                                                                        \
    do {                                                                                             \
        if (big) {                                                                                   \
            TIC_RT_MAX_LDS(gemm_nt256_kernel<E>, G256_NT_LDS_BYTES);                                 \
            TIC_LAUNCH(gemm_nt256_kernel<E>, grid, 512, G256_NT_LDS_BYTES, stream, p);               \
        } else if (split > 1 && deep) {                                                              \
            TIC_RT_MAX_LDS((gemm_nt_kernel<E, false, 1, 4>), 4 * GEMM_STAGE_BYTES);                  \
            TIC_LAUNCH((gemm_nt_kernel<E, false, 1, 4>), grid * split, 256, 4 * GEMM_STAGE_BYTES, stream, p); \
        } else if (split > 1) {                                                                      \
            TIC_RT_MAX_LDS((gemm_nt_kernel<E, false, 1>), GEMM_LDS_BYTES);                           \
            TIC_LAUNCH((gemm_nt_kernel<E, false, 1>), grid * split, 256, GEMM_LDS_BYTES, stream, p); \
        } else if (deep) {                                                                           \
            TIC_RT_MAX_LDS((gemm_nt_kernel<E, false, 0, 4>), 4 * GEMM_STAGE_BYTES);                  \
            TIC_LAUNCH((gemm_nt_kernel<E, false, 0, 4>), grid, 256, 4 * GEMM_STAGE_BYTES, stream, p); \
        } else {                                                                                     \
            TIC_RT_MAX_LDS(gemm_nt_kernel<E>, GEMM_LDS_BYTES);                                       \
            TIC_LAUNCH(gemm_nt_kernel<E>, grid, 256, GEMM_LDS_BYTES, stream, p);                     \
        }                                                                                            \
    } while (0)
    switch (epilogue) {
        case TIC_EPI_BF16:
            TIC_REQUIRE(out_bf16, "gemm_nt: EPI_BF16 needs out_bf16");
#ifdef TIC_MEASURE
            if (big && g_opt_gemm_dbg) {
#define TIC_DBG_CASE(D)                                                                                   \
    case D:                                                                                               \
        TIC_RT_MAX_LDS((gemm_nt256_kernel<TIC_EPI_BF16, D>), G256_NT_LDS_BYTES);                          \
        TIC_LAUNCH((gemm_nt256_kernel<TIC_EPI_BF16, D>), grid, 512, G256_NT_LDS_BYTES, stream, p);        \
        break;
                switch (g_opt_gemm_dbg) {
                    TIC_DBG_CASE(1) TIC_DBG_CASE(2) TIC_DBG_CASE(3) TIC_DBG_CASE(4) TIC_DBG_CASE(5) TIC_DBG_CASE(6) TIC_DBG_CASE(7)
                    TIC_DBG_CASE(8) TIC_DBG_CASE(10) TIC_DBG_CASE(12) TIC_DBG_CASE(14) TIC_DBG_CASE(16) TIC_DBG_CASE(17) TIC_DBG_CASE(18) TIC_DBG_CASE(19) TIC_DBG_CASE(24)
                }
#undef TIC_DBG_CASE
                break;
            }
#endif
            TIC_GEMM_NT_LAUNCH_B(TIC_EPI_BF16);
            break;
        case TIC_EPI_GELU:
            TIC_REQUIRE(out_bf16 && out2_bf16, "gemm_nt: EPI_GELU needs out_bf16 and out2_bf16");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_GELU);
            break;
        case TIC_EPI_RESID:
            TIC_REQUIRE(out_f32 && resid, "gemm_nt: EPI_RESID needs out_f32 and resid");
            TIC_GEMM_NT_LAUNCH_B(TIC_EPI_RESID);
            break;
        case TIC_EPI_DGELU:
            TIC_REQUIRE(out_bf16 && aux_bf16, "gemm_nt: EPI_DGELU needs out_bf16 and aux_bf16");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_DGELU);
            break;
        case TIC_EPI_GELU_ONLY:
            TIC_REQUIRE(out2_bf16, "gemm_nt: EPI_GELU_ONLY needs out2_bf16");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_GELU_ONLY);
            break;
        case TIC_EPI_GELU_DG:
            TIC_REQUIRE(out_bf16 && out2_bf16, "gemm_nt: EPI_GELU_DG needs out_bf16 and out2_bf16");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_GELU_DG);
            break;
        case TIC_EPI_MULAUX:
            TIC_REQUIRE(out_bf16 && aux_bf16, "gemm_nt: EPI_MULAUX needs out_bf16 and aux_bf16");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_MULAUX);
            break;
        case TIC_EPI_ADDAUX:
            TIC_REQUIRE(out_bf16 && aux_bf16, "gemm_nt: EPI_ADDAUX needs out_bf16 and aux_bf16");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_ADDAUX);
            break;
        case TIC_EPI_PATCH:
            TIC_REQUIRE(out_f32 && rowtab && patches > 0 && M % patches == 0, "gemm_nt: EPI_PATCH needs out_f32, rowtab, M %% patches == 0");
            TIC_GEMM_NT_LAUNCH(TIC_EPI_PATCH);
            break;
        default:
            return tic_fail(TIC_EINVAL, "gemm_nt: unknown epilogue %d", epilogue);
    }
#undef TIC_GEMM_NT_LAUNCH
#undef TIC_GEMM_NT_LAUNCH_B
    return tic_after_launch("gemm_nt");
}

static int gemm_tn_group_impl(int nprob, const void* const* A, const void* const* B, float* const* C, const int* N, const int* K, int M,
                              bool force256, bool overwrite, tic_stream_t stream);
extern "C" int tic_gemm_tn_bf16(const void* A, const void* B, float* C, int M, int N, int K, tic_stream_t stream) {
    TIC_REQUIRE(A && B && C, "gemm_tn: null operand");
    // long reductions over 256-aligned outputs (the 1x1-convolution weight gradients of ResNet: a handful of tiles, M up to 10^5
    // rows) go to the deep-pipelined 256x256 kernel through its stream-K split, which needs no minimum tile count
    // (from 8 tiles on: with 4 -- a 256 <-> 1024 1x1 convolution -- 64 row parts of a 1 MiB output make 64 MiB of slab traffic and the
    // split-M 128x128 kernel is 20-25 % faster: 40.6 / 43.7 against 52 us at 20 480 rows, 71 / 69 against 78 / 80 us at 50 176)
    if (g_opt_gemm_tile != 128 && g_opt_tn_streamk && N % 256 == 0 && K % 256 == 0 && M >= 8192 && (long)(N / 256) * (K / 256) >= g_opt_tn_slab_min_tiles &&
        (long)(N / 256) * (K / 256) * ((M + 63) / 64) >= 256 && ((double)M + 320.0) * (N > K ? N : K) * 2.0 < 4294967296.0) {
        const void* a1[1] = {A};
        const void* b1[1] = {B};
        float* c1[1] = {C};
        const int n1[1] = {N}, k1[1] = {K};
        return gemm_tn_group_impl(1, a1, b1, c1, n1, k1, M, true, false, stream);
    }
    TIC_REQUIRE(M >= 1 && N % 8 == 0 && K % 8 == 0 && N >= 8 && K >= 8, "gemm_tn: need N, K multiples of 8 (M=%d N=%d K=%d)", M, N, K);
    TIC_REQUIRE(TIC_ALIGNED16(A) && TIC_ALIGNED16(B), "gemm_tn: operands must be 16-byte aligned");
    TIC_REQUIRE(((double)M + 64.0) * (N > K ? N : K) * 2.0 < 4294967296.0, "gemm_tn: operand exceeds the 4 GiB buffer-resource range");
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    // split the M reduction so that ~4 workgroups per CU are in flight (256 CUs)
    int split = tn_split(M, tiles);
    int m_per = ((M + split - 1) / split + 63) / 64 * 64;
    split = (M + m_per - 1) / m_per;
    GemmTnParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.M = M; p.N = N; p.K = K; p.m_per_slice = m_per;
    TIC_RT_MAX_LDS(gemm_tn_kernel<false>, GEMM_LDS_BYTES);
    TIC_LAUNCH(gemm_tn_kernel<false>, dim3(tiles, split), 256, GEMM_LDS_BYTES, stream, p);
    return tic_after_launch("gemm_tn");
}

// grouped dW: one launch of 256x256 full-M tiles when every problem allows it and there are enough tiles to
// occupy the chip; otherwise one split-M launch per problem
extern "C" int tic_gemm_tn_group_bf16(int nprob, const void* const* A, const void* const* B, float* const* C, const int* N,
                                      const int* K, int M, tic_stream_t stream) {
    return gemm_tn_group_impl(nprob, A, B, C, N, K, M, false, false, stream);
}
// overwrite != 0: C_g = dY_g^T X_g -- what C held is dropped (it need not be initialised).  One workgroup per full-M tile stores its
// tile; every other route (stream-K shares, row parts, the split-M 128x128 kernel) adds partial tiles, so C is zeroed first.
extern "C" int tic_gemm_tn_group_bf16_ex(int nprob, const void* const* A, const void* const* B, float* const* C, const int* N,
                                         const int* K, int M, int overwrite, tic_stream_t stream) {
    return gemm_tn_group_impl(nprob, A, B, C, N, K, M, false, overwrite != 0, stream);
}
static int gemm_tn_group_impl(int nprob, const void* const* A, const void* const* B, float* const* C, const int* N, const int* K, int M,
                              bool force256, bool overwrite, tic_stream_t stream) {
    TIC_REQUIRE(nprob >= 1 && nprob <= TN_MAX_GROUP && A && B && C && N && K && M >= 1, "gemm_tn_group: bad argument (nprob=%d)", nprob);
    bool ok256 = true;
    int tiles = 0;
    for (int g = 0; g < nprob; ++g) {
        TIC_REQUIRE(A[g] && B[g] && C[g] && TIC_ALIGNED16(A[g]) && TIC_ALIGNED16(B[g]), "gemm_tn_group: null / misaligned operand %d", g);
        TIC_REQUIRE(N[g] % 8 == 0 && K[g] % 8 == 0 && N[g] >= 8 && K[g] >= 8, "gemm_tn_group: need N, K multiples of 8 (problem %d: N=%d K=%d)", g, N[g], K[g]);
        TIC_REQUIRE(((double)M + 320.0) * (N[g] > K[g] ? N[g] : K[g]) * 2.0 < 4294967296.0, "gemm_tn_group: operand exceeds the 4 GiB buffer-resource range");
        if (N[g] % 256 || K[g] % 256) ok256 = false;
        tiles += (N[g] / 256) * (K[g] / 256);
    }
    auto zero_all = [&]() -> int {
        bool one_launch = nprob <= 4;
        for (int g = 0; g < nprob; ++g) one_launch = one_launch && ((size_t)N[g] * K[g]) % 4 == 0 && TIC_ALIGNED16(C[g]);
        if (!one_launch) {
            for (int g = 0; g < nprob; ++g) TIC_RT_MEMSET(C[g], 0, (size_t)N[g] * K[g] * 4, stream);
            return TIC_OK;
        }
        ZeroMany z;
        for (int g = 0; g < 4; ++g) {
            z.p[g] = g < nprob ? C[g] : nullptr;
            z.n4[g] = g < nprob ? (long)((size_t)N[g] * K[g] / 4) : 0;
        }
        TIC_LAUNCH(zero_many_kernel, dim3(512, (unsigned)nprob), 256, 0, stream, z);
        return TIC_OK;
    };
    if (ok256 && g_opt_gemm_tile != 128 && (force256 || g_opt_gemm_tile == 256 || (tiles >= 96 && M >= 512))) {
        GemmTnGroupParams gp;
        memset(&gp, 0, sizeof(gp));
        // launch order of the problems: those whose tile count is a whole number of XCD shares (tiles / 8 each) first, so that their
        // blocks line up with the XCD boundaries of the tile walk; then the rest
        int order[TN_MAX_GROUP], no = 0;
        const int per_xcd = tiles >= 8 ? tiles / 8 : tiles;
        for (int pass = 0; pass < 2; ++pass)
            for (int g = 0; g < nprob; ++g) {
                const bool whole = ((N[g] / 256) * (K[g] / 256)) % per_xcd == 0;
                if (whole == (pass == 0)) order[no++] = g;
            }
        int t = 0;
        for (int i = 0; i < nprob; ++i) {
            const int g = order[i];
            gp.prob[i].A = (const bf16_t*)A[g]; gp.prob[i].B = (const bf16_t*)B[g]; gp.prob[i].C = C[g];
            gp.prob[i].N = N[g]; gp.prob[i].K = K[g]; gp.prob[i].tile_start = t;
            const int tn_ = N[g] / 256, tk_ = K[g] / 256, ls = tn_ < tk_ ? tn_ : tk_;
            int bw = g_opt_tn_block >= 0 ? g_opt_tn_block : (per_xcd + ls / 2) / ls;   // (short dimension) x bw ~ one XCD's share
            if (bw < 1) bw = 1;
            if (g_opt_tn_block == 0) bw = tn_ < tk_ ? tk_ : tn_;                        // 0: the plain row-major walk (A/B)
            gp.prob[i].bw = bw;
            t += tn_ * tk_;
        }
        gp.nprob = nprob; gp.M = M; gp.total_tiles = t;
        const int nsteps = (M + 63) / 64;
        const long shares = g_opt_tn_streamk == 1 ? 256 : g_opt_tn_streamk;   // one share per CU by default
        // short reductions: one workgroup per full-M tile; the stream-K split's partial tiles (fp32 atomics) cost more than the idle quarter
        // of the CUs.  When C is accumulated into: below 128 steps of 64 rows (~42 images of ViT-L tokens; tools/small_batch_bench.py: 71 vs
        // 100 us at M = 1576, 166 vs 175 at M = 6304, 331 vs 288 at M = 12608); when C is stored (no read of C, no zeroing): below 352
        // steps (measured inside the step, see g_opt_tn_streamk_min_steps_store)
        const bool streamk = g_opt_tn_streamk && (force256 || g_opt_tn_streamk > 1 || nsteps >= (overwrite ? g_opt_tn_streamk_min_steps_store : g_opt_tn_streamk_min_steps));
        if (streamk && (long)t * nsteps >= shares) {
            if (overwrite) TIC_TRY(zero_all());   // partial tiles are ADDED
            // phase-aligned split when the tiles divide over the 8 XCDs and the tail workgroups get whole tiles
            int s_main = 0, tpx = 0, tail_each = 0;
            if (g_opt_tn_phase && shares % 8 == 0 && t % 8 == 0) {
                const int per_xcd = (int)(shares / 8), tp = t / 8, tails = per_xcd - tp;
                if (tails > 0 && tp % tails == 0) {
                    tpx = tp;
                    tail_each = tp / tails;
                    s_main = (int)((long)nsteps * tail_each / (tail_each + 1)) + g_opt_tn_main_bias;   // main: s_main steps; tail: tail_each x (S - s_main)
                    if (s_main < 1 || s_main >= nsteps) s_main = 0;
                }
            }
            // no phase-aligned split for this tile count: equal parts when 2..8 of them fit the chip and each keeps >= 64 steps
            int grid_wg = (int)shares;
            if (s_main == 0 && (g_opt_tn_parts > 0 || (g_opt_tn_parts < 0 && g_opt_tn_streamk == 1))) {
                const bool forced = g_opt_tn_parts > 0;   // tests
                int P = forced ? g_opt_tn_parts : (int)(256 / t);
                if (P > 8) P = 8;
                const int txm = (t >> 3) + ((t & 7) ? 1 : 0);
                // auto: only where each part keeps >= 64 steps and >= 3/4 of the CUs get one
                if (P >= 2 && 8 * txm * P <= 320 && (forced || (nsteps / P >= 64 && (long)P * t >= 192))) {
                    s_main = -P;
                    grid_wg = 8 * txm * P;
                }
            }
            // single problem of few tiles with a slab registered: equal row parts that STORE their partial tiles + one reduce launch
            if (force256 && nprob == 1 && g_opt_tn_slab && g_tn_slab && t <= 64 && t >= 1) {
                int P = (int)(256 / t);
                if (P > nsteps / 4) P = nsteps / 4;   // at least 4 steps per part
                const size_t nk = (size_t)N[0] * K[0];
                if (P >= 2 && (size_t)P * nk * 4 <= g_tn_slab_bytes) {
                    const int per = (nsteps + P - 1) / P, parts = (nsteps + per - 1) / per;   // every launched part is non-empty
                    TIC_RT_MAX_LDS(gemm_tn256_parts_slab_kernel, G256_LDS_BYTES);
                    TIC_LAUNCH(gemm_tn256_parts_slab_kernel, t * parts, 512, G256_LDS_BYTES, stream, gp, g_tn_slab, nsteps, per);
                    long blocks = ((long)(nk / 4) + 255) / 256;
                    if (blocks > 2048) blocks = 2048;
                    TIC_LAUNCH(tn_slab_reduce_kernel, (int)blocks, 256, 0, stream, C[0], (const float*)g_tn_slab, parts, (long)(nk / 4));
                    return tic_after_launch("gemm_tn(parts + slab)");
                }
            }
            if (force256) {   // single-problem route: same code, own kernel name, not part of the live timer
                TIC_RT_MAX_LDS(gemm_tn256_streamk_single_kernel, G256_LDS_BYTES);
                TIC_LAUNCH(gemm_tn256_streamk_single_kernel, grid_wg, 512, G256_LDS_BYTES, stream, gp, nsteps, s_main, tpx, tail_each);
                return tic_after_launch("gemm_tn(stream-K)");
            }
            TIC_RT_TIMER_MARK(0, stream);
#if defined(TIC_MEASURE) || defined(TIC_SIM)
            if (g_opt_tn_waves == 4) {
                TIC_RT_MAX_LDS(gemm_tn256_streamk_w4_kernel, G256_LDS_BYTES);
                TIC_LAUNCH(gemm_tn256_streamk_w4_kernel, grid_wg, 256, G256_LDS_BYTES, stream, gp, nsteps, s_main, tpx, tail_each);
            } else
#endif
            if (g_opt_tn_mfma == 32 || (g_opt_tn_mfma == 0 && nsteps < g_opt_tn_mfma16_min_steps)) {
                TIC_RT_MAX_LDS(gemm_tn256_streamk_mfma32_kernel, G256_LDS_BYTES);
                TIC_LAUNCH(gemm_tn256_streamk_mfma32_kernel, grid_wg, 512, G256_LDS_BYTES, stream, gp, nsteps, s_main, tpx, tail_each);
            } else {
                TIC_RT_MAX_LDS(gemm_tn256_streamk_kernel, G256_LDS_BYTES);
                TIC_LAUNCH(gemm_tn256_streamk_kernel, grid_wg, 512, G256_LDS_BYTES, stream, gp, nsteps, s_main, tpx, tail_each);
            }
            TIC_RT_TIMER_MARK(1, stream);
            return tic_after_launch("gemm_tn_group(stream-K)");
        }
#ifdef TIC_MEASURE
        if (g_opt_gemm_dbg) {
            if (overwrite) TIC_TRY(zero_all());
#define TIC_DBG_CASE(D)                                                                   \
    case D:                                                                               \
        TIC_RT_MAX_LDS(gemm_tn256_kernel<D>, G256_LDS_BYTES);                             \
        TIC_LAUNCH(gemm_tn256_kernel<D>, t, 512, G256_LDS_BYTES, stream, gp);             \
        break;
            switch (g_opt_gemm_dbg) {
                TIC_DBG_CASE(1) TIC_DBG_CASE(2) TIC_DBG_CASE(3) TIC_DBG_CASE(4) TIC_DBG_CASE(5) TIC_DBG_CASE(6) TIC_DBG_CASE(7)
                default: return tic_fail(TIC_EINVAL, "gemm_tn_group: gemm_dbg %d has no TN variant", g_opt_gemm_dbg);
            }
#undef TIC_DBG_CASE
            return tic_after_launch("gemm_tn_group(dbg)");
        }
#endif
        if (!force256) TIC_RT_TIMER_MARK(0, stream);
        if (overwrite) {
            TIC_RT_MAX_LDS(gemm_tn256_store_kernel, G256_LDS_BYTES);
            TIC_LAUNCH(gemm_tn256_store_kernel, t, 512, G256_LDS_BYTES, stream, gp);
        } else {
            TIC_RT_MAX_LDS(gemm_tn256_kernel<0>, G256_LDS_BYTES);
            TIC_LAUNCH(gemm_tn256_kernel<0>, t, 512, G256_LDS_BYTES, stream, gp);
        }
        if (!force256) TIC_RT_TIMER_MARK(1, stream);
        return tic_after_launch("gemm_tn_group");
    }
    if (overwrite) TIC_TRY(zero_all());
    for (int g = 0; g < nprob; ++g) TIC_TRY(tic_gemm_tn_bf16(A[g], B[g], C[g], M, N[g], K[g], stream));
    return TIC_OK;
}

// ---- LayerNorm -------------------------------------------------------------------------------------
// non-temporal accesses to the fp32 residual stream (stream_nt bit 0) only when that stream is larger than the 256 MB Infinity Cache:
// below, the next kernel finds it there and the hint costs 0.4-1.0 % of the step (tools/ab_step.py stream_nt 13 12: 64 images 31.02 vs
// 30.77 ms, 100: 45.72 vs 45.27, 128: 54.55 vs 54.29, 166: 65.93 vs 65.84; 332 images = 268 MB: 127.38 vs 127.43)
static bool ln_nt(int rows, int D) { return (g_opt_nt & 1) && (size_t)rows * D * 4 >= ((size_t)256 << 20); }
extern "C" int tic_layernorm_bwd_ex(const void* dy_bf16, const float* x, long stride, const float* gamma, const float* mean,
                                    const float* rstd, const float* dres, float* dx, void* dxb_bf16, float* dgamma,
                                    float* dbeta, float* colsum, int rows, int D, tic_stream_t stream);
static int ln_grid(int rows) {
    int g = (rows + 3) / 4;
    return g > g_opt_ln_blocks ? g_opt_ln_blocks : (g < 1 ? 1 : g);
}
extern "C" int tic_layernorm_fwd(const float* x, long in_stride, const float* gamma, const float* beta, void* y_bf16,
                                 float* mean, float* rstd, int rows, int D, float eps, tic_stream_t stream) {
    TIC_REQUIRE(x && gamma && beta && y_bf16 && mean && rstd, "layernorm_fwd: null pointer");
    TIC_REQUIRE(rows >= 1 && D >= 4 && D % 4 == 0 && D <= 1024 && in_stride % 4 == 0, "layernorm_fwd: need D %% 4 == 0, D <= 1024 (D=%d)", D);
    const int nv = (D + 255) / 256, grid = ln_grid(rows);
#define TIC_LN_FWD(NV)                                                                                                                   \
    do {                                                                                                                                 \
        if (ln_nt(rows, D)) TIC_LAUNCH((ln_fwd_kernel<NV, true>), grid, 256, 0, stream, x, in_stride, gamma, beta, (bf16_t*)y_bf16, mean, rstd, rows, D, eps); \
        else TIC_LAUNCH((ln_fwd_kernel<NV, false>), grid, 256, 0, stream, x, in_stride, gamma, beta, (bf16_t*)y_bf16, mean, rstd, rows, D, eps);          \
    } while (0)
    if (nv == 1) TIC_LN_FWD(1); else if (nv == 2) TIC_LN_FWD(2); else if (nv == 3) TIC_LN_FWD(3); else TIC_LN_FWD(4);
#undef TIC_LN_FWD
    return tic_after_launch("layernorm_fwd");
}
extern "C" int tic_layernorm_bwd(const void* dy_bf16, const float* x, long stride, const float* gamma, const float* mean,
                                 const float* rstd, const float* dres, float* dx, void* dxb_bf16, float* dgamma,
                                 float* dbeta, int rows, int D, tic_stream_t stream) {
    return tic_layernorm_bwd_ex(dy_bf16, x, stride, gamma, mean, rstd, dres, dx, dxb_bf16, dgamma, dbeta, nullptr, rows, D, stream);
}
extern "C" int tic_layernorm_bwd_ex(const void* dy_bf16, const float* x, long stride, const float* gamma, const float* mean,
                                    const float* rstd, const float* dres, float* dx, void* dxb_bf16, float* dgamma,
                                    float* dbeta, float* colsum, int rows, int D, tic_stream_t stream) {
    TIC_REQUIRE(dy_bf16 && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
    TIC_REQUIRE(rows >= 1 && D >= 4 && D % 4 == 0 && D <= 1024 && stride % 4 == 0, "layernorm_bwd: need D %% 4 == 0, D <= 1024 (D=%d)", D);
    const int nv = (D + 255) / 256;
    int grid = ln_grid(rows);
    if (grid > g_opt_ln_bwd_blocks) grid = g_opt_ln_bwd_blocks;   // fewer, longer blocks: one dgamma/dbeta atomic row per block
    // few rows: every block ends with 3 D atomics onto the same 3 D addresses, so at least g_opt_ln_bwd_rows rows per wave
    const int few = (rows + 4 * g_opt_ln_bwd_rows - 1) / (4 * g_opt_ln_bwd_rows);
    if (grid > few) grid = few;
    const size_t lds = (size_t)3 * 4 * D * 4;
#define TIC_LN_BWD(NV)                                                                                                                   \
    do {                                                                                                                                 \
        if (ln_nt(rows, D)) TIC_LAUNCH((ln_bwd_kernel<NV, true>), grid, 256, lds, stream, (const bf16_t*)dy_bf16, x, stride, gamma, mean, rstd, dres, dx, (bf16_t*)dxb_bf16, dgamma, dbeta, colsum, rows, D); \
        else TIC_LAUNCH((ln_bwd_kernel<NV, false>), grid, 256, lds, stream, (const bf16_t*)dy_bf16, x, stride, gamma, mean, rstd, dres, dx, (bf16_t*)dxb_bf16, dgamma, dbeta, colsum, rows, D);          \
    } while (0)
    if (nv == 1) TIC_LN_BWD(1); else if (nv == 2) TIC_LN_BWD(2); else if (nv == 3) TIC_LN_BWD(3); else TIC_LN_BWD(4);
#undef TIC_LN_BWD
    return tic_after_launch("layernorm_bwd");
}

// ---- attention -------------------------------------------------------------------------------------
extern "C" int tic_attention_bwd_ex(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias,
                                    int B, int H, int N, float scale, tic_stream_t stream);
extern "C" int tic_attention_fwd(const void* qkv, void* o, float* lse, int B, int H, int N, float scale, tic_stream_t stream) {
    TIC_REQUIRE(qkv && o && lse, "attention_fwd: null pointer");
    TIC_REQUIRE(B >= 1 && H >= 1 && N >= 1 && N <= 208, "attention_fwd: need 1 <= N <= 208 (N=%d)", N);
    TIC_REQUIRE((double)B * N * 3 * H * 64 * 2 < 4294967296.0, "attention_fwd: qkv exceeds the 4 GiB buffer-resource range");
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)o; p.lse = lse; p.B = B; p.H = H; p.N = N; p.D = H * 64; p.scale = scale;
    if (g_opt_attn_fwd_waves == 8) {
        TIC_RT_MAX_LDS(attn_fwd_kernel<8>, 2 * ATT_TILE_BYTES);
        TIC_LAUNCH(attn_fwd_kernel<8>, B * H, 512, 2 * ATT_TILE_BYTES, stream, p);
    } else {
        TIC_RT_MAX_LDS(attn_fwd_kernel<4>, 2 * ATT_TILE_BYTES);
        TIC_LAUNCH(attn_fwd_kernel<4>, B * H, 256, 2 * ATT_TILE_BYTES, stream, p);
    }
    return tic_after_launch("attention_fwd");
}
extern "C" int tic_attention_bwd(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, int B,
                                 int H, int N, float scale, tic_stream_t stream) {
    return tic_attention_bwd_ex(qkv, o, lse, d_o, dqkv, nullptr, B, H, N, scale, stream);
}
static int attention_bwd_launch(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias, float* dbias_part,
                                int skip_v_bias, int B, int H, int N, float scale, tic_stream_t stream);
extern "C" int tic_attention_bwd_ex(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias,
                                    int B, int H, int N, float scale, tic_stream_t stream) {
    return attention_bwd_launch(qkv, o, lse, d_o, dqkv, dbias, nullptr, 0, B, H, N, scale, stream);
}
// same, with a caller-owned fp32 scratch [B][3*H*64]: the q/k/v bias gradient is formed without global atomics
// skip_v_bias != 0: the v third of dbias is left untouched -- it equals the column sums of dO (the rows of P sum to 1), which the
// GEMM that produced dO can add for free (tic_gemm_nt_bf16_ex colsum)
extern "C" int tic_attention_bwd_ws(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias, float* scratch_b3d,
                                    int skip_v_bias, int B, int H, int N, float scale, tic_stream_t stream) {
    TIC_REQUIRE(!scratch_b3d || dbias, "attention_bwd_ws: scratch without dbias");
    return attention_bwd_launch(qkv, o, lse, d_o, dqkv, dbias, scratch_b3d, skip_v_bias, B, H, N, scale, stream);
}
static int attention_bwd_launch(const void* qkv, const void* o, const float* lse, const void* d_o, void* dqkv, float* dbias, float* dbias_part,
                                int skip_v_bias, int B, int H, int N, float scale, tic_stream_t stream) {
    TIC_REQUIRE(qkv && o && lse && d_o && dqkv, "attention_bwd: null pointer");
    TIC_REQUIRE(B >= 1 && H >= 1 && N >= 1 && N <= 208, "attention_bwd: need 1 <= N <= 208 (N=%d)", N);
    TIC_REQUIRE((double)B * N * 3 * H * 64 * 2 < 4294967296.0, "attention_bwd: qkv exceeds the 4 GiB buffer-resource range");
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = (const bf16_t*)qkv; p.o = (bf16_t*)o; p.lse = (float*)lse; p.d_o = (const bf16_t*)d_o; p.dqkv = (bf16_t*)dqkv; p.dbias = dbias;
    p.B = B; p.H = H; p.N = N; p.D = H * 64; p.scale = scale;
    p.dbias_part = dbias ? dbias_part : nullptr;
    p.skip_v_bias = (dbias && skip_v_bias) ? 1 : 0;
#if defined(TIC_SIM) || defined(TIC_MEASURE)
    p.legacy_assign = g_opt_attn_legacy;
#endif
    TIC_RT_MAX_LDS(attn_bwd_kernel, ATT_BWD_LDS);
    TIC_LAUNCH(attn_bwd_kernel, B * H, 1024, ATT_BWD_LDS, stream, p);
    if (p.dbias_part) {   // the scratch rows keep the [3D] stride; with skip_v_bias only the q and k thirds are summed
        const int used = (p.skip_v_bias ? 2 : 3) * H * 64;
        int splits = (B + 3) / 4;
        if (splits > 16) splits = 16;
        TIC_LAUNCH(attn_dbias_reduce_kernel, dim3((used + 63) / 64, splits), 256, 1024, stream, (const float*)dbias_part, dbias, B, 3 * H * 64, used);
    }
    return tic_after_launch("attention_bwd");
}

// ---- element-wise ----------------------------------------------------------------------------------
static int ew_grid(long work_items) {
    long g = (work_items + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
extern "C" int tic_patchify(const float* x, void* P_bf16, int B, int C, int img, int patch, tic_stream_t stream) {
    TIC_REQUIRE(x && P_bf16, "patchify: null pointer");
    TIC_REQUIRE(B >= 1 && C >= 1 && patch % 8 == 0 && img % patch == 0, "patchify: need patch %% 8 == 0 and img %% patch == 0");
    TIC_LAUNCH(patchify_kernel, ew_grid((long)B * C * img * (img / 8)), 256, 0, stream, x, (bf16_t*)P_bf16, B, C, img, patch);
    return tic_after_launch("patchify");
}
extern "C" int tic_embed_cls(const float* cls, const float* pos, float* h, int B, int N, int D, tic_stream_t stream) {
    TIC_REQUIRE(cls && pos && h && B >= 1, "embed_cls: bad argument");
    TIC_LAUNCH(embed_cls_kernel, (B * D + 255) / 256, 256, 0, stream, cls, pos, h, B, N, D);
    return tic_after_launch("embed_cls");
}
extern "C" int tic_embed_bwd(const float* dh, float* dcls, float* dpos, int B, int N, int D, tic_stream_t stream) {
    TIC_REQUIRE(dh && dcls && dpos && B >= 1, "embed_bwd: bad argument");
    TIC_LAUNCH(embed_bwd_kernel, (N * D + 255) / 256, 256, 0, stream, dh, dcls, dpos, B, N, D);
    return tic_after_launch("embed_bwd");
}
__global__ void __launch_bounds__(256) gather_patch_rows_kernel(const float* __restrict__ dh, bf16_t* __restrict__ out, int B, int N, int D) {
    const long total4 = (long)B * (N - 1) * D / 4;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total4; i += (long)TIC_NBLK_X * 256) {
        const long e = i * 4, row = e / D;
        const int d = (int)(e - row * D);
        const long b = row / (N - 1), pi = row - b * (N - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(dh + ((b * N + 1 + pi) * D + d));
        *reinterpret_cast<u32x2*>(out + e) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
}
extern "C" int tic_gather_patch_rows(const float* dh, void* out_bf16, int B, int N, int D, tic_stream_t stream) {
    TIC_REQUIRE(dh && out_bf16 && B >= 1 && N >= 2 && D % 4 == 0, "gather_patch_rows: bad argument");
    TIC_LAUNCH(gather_patch_rows_kernel, ew_grid((long)B * (N - 1) * D / 4), 256, 0, stream, dh, (bf16_t*)out_bf16, B, N, D);
    return tic_after_launch("gather_patch_rows");
}
extern "C" int tic_colsum_bf16(const void* in_bf16, float* out, int M, int N, tic_stream_t stream) {
    TIC_REQUIRE(in_bf16 && out && M >= 1 && N >= 8 && N % 8 == 0, "colsum: need N %% 8 == 0");
    int rs = (M + 63) / 64;
    if (rs > 64) rs = 64;
    TIC_LAUNCH(colsum_kernel, dim3((N + 255) / 256, rs), 256, 8 * 256 * 4, stream, (const bf16_t*)in_bf16, out, M, N);
    return tic_after_launch("colsum");
}
extern "C" int tic_cast_bf16(const float* in, void* out_bf16, long n, tic_stream_t stream) {
    TIC_REQUIRE(in && out_bf16 && n >= 4 && n % 4 == 0, "cast_bf16: need n %% 4 == 0");
    TIC_LAUNCH(cast_bf16_kernel, ew_grid(n / 4), 256, 0, stream, in, (bf16_t*)out_bf16, n / 4);
    return tic_after_launch("cast_bf16");
}
extern "C" int tic_cast_transpose_bf16(const float* in, void* out_bf16, int R, int C, tic_stream_t stream) {
    TIC_REQUIRE(in && out_bf16 && R % 64 == 0 && C % 64 == 0 && R >= 64 && C >= 64, "cast_transpose: need R, C multiples of 64");
    TIC_LAUNCH(cast_transpose_kernel, dim3(C / 64, R / 64), 256, 64 * CT_STRIDE * 2, stream, in, (bf16_t*)out_bf16, R, C);
    return tic_after_launch("cast_transpose");
}
extern "C" int tic_adamw(float* p, const float* g, float* m, float* v, void* w16, long n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, tic_stream_t stream) {
    TIC_REQUIRE(p && g && m && v && n >= 4 && n % 4 == 0 && step >= 1, "adamw: need n %% 4 == 0 and step >= 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float inv_bc1 = (float)(1.0 / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    if (g_opt_nt & 2)
        TIC_LAUNCH(adamw_kernel<true>, ew_grid(n / 4), 256, 0, stream, p, g, m, v, (bf16_t*)w16, n / 4, lr, beta1, beta2, eps, weight_decay, inv_bc1, inv_sqrt_bc2);
    else
        TIC_LAUNCH(adamw_kernel<false>, ew_grid(n / 4), 256, 0, stream, p, g, m, v, (bf16_t*)w16, n / 4, lr, beta1, beta2, eps, weight_decay, inv_bc1, inv_sqrt_bc2);
    return tic_after_launch("adamw");
}
// measurement only: stream n floats (n % 4 == 0) from src (and to dst when given) with `blocks` workgroups; mode bit 0 = non-temporal
extern "C" int tic_probe_stream(const float* src, float* dst, float* sink, long n, int blocks, int mode, tic_stream_t stream) {
    TIC_REQUIRE(src && sink && n >= 4 && n % 4 == 0 && blocks >= 1, "probe_stream: bad argument");
    if (mode & 1) TIC_LAUNCH(probe_stream_kernel<true>, blocks, 256, 0, stream, src, dst, sink, n / 4);
    else TIC_LAUNCH(probe_stream_kernel<false>, blocks, 256, 0, stream, src, dst, sink, n / 4);
    return tic_after_launch("probe_stream");
}
extern "C" int tic_head_fwd(const void* z_bf16, const float* W, const float* bias, float* logits, int B, int C, int D,
                            tic_stream_t stream) {
    TIC_REQUIRE(z_bf16 && W && bias && logits && B >= 1 && C >= 1 && D % 4 == 0, "head_fwd: bad argument");
    TIC_LAUNCH(head_fwd_kernel, (B * C + 3) / 4, 256, 0, stream, (const bf16_t*)z_bf16, W, bias, logits, B, C, D);
    return tic_after_launch("head_fwd");
}
extern "C" int tic_head_bwd(const float* dlogits, const void* z_bf16, const float* W, void* dz_bf16, float* dW, float* db,
                            int B, int C, int D, tic_stream_t stream) {
    TIC_REQUIRE(dlogits && z_bf16 && W && dz_bf16 && dW && db && B >= 1 && C >= 1, "head_bwd: bad argument");
    TIC_LAUNCH(head_bwd_dz_kernel, (B * D + 255) / 256, 256, 0, stream, dlogits, W, (bf16_t*)dz_bf16, B, C, D);
    TIC_LAUNCH(head_bwd_dw_kernel, (C * D + 255) / 256, 256, 0, stream, dlogits, (const bf16_t*)z_bf16, dW, db, B, C, D);
    return tic_after_launch("head_bwd");
}
extern "C" int tic_softmax_xent(const float* logits, const int64_t* labels, const float* soft, float* loss_sum,
                                float* dlogits, int B, int C, float gscale, tic_stream_t stream) {
    TIC_REQUIRE(logits && loss_sum && B >= 1 && C >= 1, "softmax_xent: bad argument");
    TIC_REQUIRE((labels != nullptr) != (soft != nullptr), "softmax_xent: exactly one of labels / soft must be given");
    TIC_LAUNCH(xent_kernel, (B + 3) / 4, 256, 0, stream, logits, (const long long*)labels, soft, loss_sum, dlogits, B, C, gscale);
    return tic_after_launch("softmax_xent");
}

// ---- mixture of experts: gate / combine / loss (BASELINE config 5) ---------------------------------------
extern "C" int tic_moe_gate(const float* logits, const float* noise, float noise_scale, float* gate_w, int64_t* topk_idx, float* topk_w,
                            int B, int E, int K, tic_stream_t stream) {
    TIC_REQUIRE(logits && gate_w && topk_idx && topk_w, "moe_gate: null pointer");
    TIC_REQUIRE(B >= 1 && E >= 1 && E <= 64 && K >= 1 && K <= TIC_MOE_MAX_K && K <= E, "moe_gate: need 1 <= K <= min(E, %d), E <= 64 (B=%d E=%d K=%d)", TIC_MOE_MAX_K, B, E, K);
    TIC_LAUNCH(moe_gate_kernel, (B + 3) / 4, 256, 0, stream, logits, noise, noise_scale, gate_w, (long long*)topk_idx, topk_w, B, E, K);
    return tic_after_launch("moe_gate");
}
extern "C" int tic_moe_gate_bwd(const float* gate_w, const float* d_gate_w, float* dlogits, int B, int E, tic_stream_t stream) {
    TIC_REQUIRE(gate_w && d_gate_w && dlogits && B >= 1 && E >= 1 && E <= 64, "moe_gate_bwd: bad argument");
    TIC_LAUNCH(moe_gate_bwd_kernel, (B + 3) / 4, 256, 0, stream, gate_w, d_gate_w, dlogits, B, E);
    return tic_after_launch("moe_gate_bwd");
}
extern "C" int tic_moe_combine(const float* expert_out, const float* gate_w, float* out, int B, int E, int C, tic_stream_t stream) {
    TIC_REQUIRE(expert_out && gate_w && out && B >= 1 && E >= 1 && C >= 1, "moe_combine: bad argument");
    TIC_LAUNCH(moe_combine_kernel, (int)(((long)B * C + 255) / 256), 256, 0, stream, expert_out, gate_w, out, B, E, C);
    return tic_after_launch("moe_combine");
}
extern "C" int tic_moe_combine_bwd(const float* expert_out, const float* gate_w, const float* dout, float* d_expert_out, float* d_gate_w,
                                   int B, int E, int C, tic_stream_t stream) {
    TIC_REQUIRE(expert_out && gate_w && dout && d_expert_out && d_gate_w && B >= 1 && E >= 1 && C >= 1, "moe_combine_bwd: bad argument");
    TIC_LAUNCH(moe_combine_bwd_kernel, (int)(((long)B * E + 3) / 4), 256, 0, stream, expert_out, gate_w, dout, d_expert_out, d_gate_w, B, E, C);
    return tic_after_launch("moe_combine_bwd");
}
extern "C" int tic_moe_loss(const float* logits, const float* targets, const float* gate_w, float* loss3, float* dlogits, float* d_gate_w,
                            int B, int C, int E, float a_ce, float b_rce, float a_balance, tic_stream_t stream) {
    TIC_REQUIRE(logits && targets && loss3 && B >= 1 && C >= 1, "moe_loss: bad argument");
    TIC_REQUIRE(!gate_w || (E >= 1 && E <= 64), "moe_loss: E must be 1..64");
    TIC_RT_MEMSET(loss3, 0, 3 * sizeof(float), stream);
    TIC_LAUNCH(moe_loss_rows_kernel, (B + 3) / 4, 256, 0, stream, logits, targets, loss3, dlogits, B, C, a_ce, b_rce);
    if (gate_w) TIC_LAUNCH(moe_balance_kernel, 1, 64, 0, stream, gate_w, loss3, d_gate_w, B, E, a_balance);
    return tic_after_launch("moe_loss");
}

// ---- augmentation ----------------------------------------------------------------------------------
extern "C" int tic_augment(const void* images_u8, int B, int Hs, int Ws, const float* params, float* out, int S,
                           const float* mean3, const float* std3, tic_stream_t stream) {
    TIC_REQUIRE(images_u8 && params && out && mean3 && std3, "augment: null pointer");
    TIC_REQUIRE(B >= 1 && Hs >= 1 && Ws >= 1 && S >= 1, "augment: bad shape");
    AugNorm nm;
    for (int c = 0; c < 3; ++c) {
        TIC_REQUIRE(std3[c] > 0.f, "augment: std must be positive");
        nm.mean[c] = mean3[c];
        nm.inv_std[c] = 1.0f / std3[c];
    }
    TIC_LAUNCH(augment_kernel, B, 256, 64, stream, (const unsigned char*)images_u8, Hs, Ws, params, out, S, nm);
    return tic_after_launch("augment");
}
extern "C" int tic_mix(const float* x, float* out, int B, int C, int H, int W, int mode, float lam, int x1, int y1, int x2,
                       int y2, tic_stream_t stream) {
    TIC_REQUIRE(x && out && x != out && B >= 1 && W % 4 == 0 && (mode == 0 || mode == 1), "mix: bad argument (out of place, W %% 4 == 0)");
    TIC_LAUNCH(mix_kernel, ew_grid((long)B * C * H * W / 4), 256, 0, stream, x, out, B, C, H, W, mode, lam, x1, y1, x2, y2);
    return tic_after_launch("mix");
}
extern "C" int tic_mix_labels(const int64_t* y, float* out, int B, int ncls, float lam, tic_stream_t stream) {
    TIC_REQUIRE(y && out && B >= 1 && ncls >= 1, "mix_labels: bad argument");
    TIC_LAUNCH(mix_labels_kernel, (B * ncls + 255) / 256, 256, 0, stream, (const long long*)y, out, B, ncls, lam);
    return tic_after_launch("mix_labels");
}

// ---- ResNet conv path ----------------------------------------------------------------------------------
static int conv_geom(ConvGeom& g, int B, int H, int W, int Ci, int kh, int kw, int stride, int pad) {
    TIC_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Ci >= 1 && kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0, "conv: bad geometry");
    g.B = B; g.H = H; g.W = W; g.Ci = Ci; g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad;
    g.Ho = (H + 2 * pad - kh) / stride + 1;
    g.Wo = (W + 2 * pad - kw) / stride + 1;
    TIC_REQUIRE(g.Ho >= 1 && g.Wo >= 1, "conv: empty output");
    g.K = kh * kw * Ci;
    g.Kp = (g.K + 63) / 64 * 64;
    return TIC_OK;
}
extern "C" int tic_conv_weight_pack(const float* w_oihw, void* w16, int Co, int Ci, int kh, int kw, int transposed, tic_stream_t stream) {
    TIC_REQUIRE(w_oihw && w16 && Co >= 1 && transposed >= 0 && transposed <= 7, "conv_weight_pack: bad argument");
    TIC_REQUIRE(transposed != 3 || (Ci == 3 && kh == 7 && kw == 7), "conv_weight_pack: layout 3 is the 3-channel 7x7 stem's");
    TIC_REQUIRE(transposed < 4 || (kh == 3 && kw == 3), "conv_weight_pack: layouts 4..7 are the parity classes of a 3x3 stride-2 input gradient");
    ConvGeom g;
    TIC_TRY(conv_geom(g, 1, kh, kw, Ci, kh, kw, 1, 0));
    TIC_LAUNCH(weight_ohwi_kernel, ew_grid((long)Co * (transposed == 3 ? TIC_STEM_KP : g.Kp)), 256, 0, stream, w_oihw, (bf16_t*)w16, Co, g, transposed);
    return tic_after_launch("conv_weight_pack");
}
// the same for a whole network: `descs` is a table of n entries in DEVICE memory (built once by the caller), one launch
extern "C" int tic_conv_weight_pack_many(const TicConvPackDesc* descs, int n, tic_stream_t stream) {
    TIC_REQUIRE(descs && n >= 1 && n <= 65535, "conv_weight_pack_many: bad argument");
    static_assert(sizeof(ConvPackDesc) == sizeof(TicConvPackDesc) && sizeof(ConvGradDesc) == sizeof(TicConvGradDesc), "descriptor layouts");
    TIC_LAUNCH(weight_ohwi_many_kernel, dim3(64, (unsigned)n), 256, 0, stream, (const ConvPackDesc*)descs);
    return tic_after_launch("conv_weight_pack_many");
}
extern "C" int tic_conv_weight_grad_many(const TicConvGradDesc* descs, int n, tic_stream_t stream) {
    TIC_REQUIRE(descs && n >= 1 && n <= 65535, "conv_weight_grad_many: bad argument");
    TIC_LAUNCH(weight_grad_oihw_many_kernel, dim3(64, (unsigned)n), 256, 0, stream, (const ConvGradDesc*)descs);
    return tic_after_launch("conv_weight_grad_many");
}
// The 7x7 / 2 stem (TIC/ResNet/model.py:148) as an implicit GEMM: the image is NHWC with its 3 channels zero-padded to 4 (8 bytes per pixel,
// tic_nchw_to_nhwc_pad_bf16), the filter is packed in layout 3 ([Cout, 256]: 7 rows x 8 pixels x 4 channels + a zero row), K = 256.
#define TIC_CONV_IS_STEM(Cin, kh, kw, stride, pad) ((Cin) == 4 && (kh) == 7 && (kw) == 7 && (stride) == 2 && (pad) == 3)
// implicit-GEMM convolution (3x3 and friends with Cin % 64 == 0): y[M = B*Ho*Wo, Cout] = gather(x) . Wpack^T
extern "C" int tic_conv_igemm_fwd(const void* x_nhwc, const void* w_pack, void* y, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                  int stride, int pad, tic_stream_t stream) {
    TIC_REQUIRE(x_nhwc && w_pack && y && B >= 1, "conv_igemm_fwd: bad argument");
    const bool stem = TIC_CONV_IS_STEM(Cin, kh, kw, stride, pad);
    TIC_REQUIRE((Cin % 64 == 0 || stem) && Cout % 8 == 0 && kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0, "conv_igemm_fwd: need Cin %% 64 == 0 (or the 4-channel-padded 7x7/2 stem), Cout %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    TIC_REQUIRE(!stem || W % 2 == 0, "conv_igemm_fwd: the stem form needs an even image width (16-byte pixel pairs)");
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    TIC_REQUIRE(Ho >= 1 && Wo >= 1, "conv_igemm_fwd: empty output");
    const long M = (long)B * Ho * Wo;
    const int K = stem ? TIC_STEM_KP : kh * kw * Cin;
    TIC_REQUIRE(M < (1L << 24) && (double)B * H * W * Cin * 2.0 < 4294967296.0 && (double)Cout * K * 2.0 < 4294967296.0,
                "conv_igemm_fwd: tensor exceeds the 32-bit offset / 2^24 row range");
    GemmNtParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)x_nhwc; p.B = (const bf16_t*)w_pack; p.M = (int)M; p.N = Cout; p.K = K; p.out = (bf16_t*)y;
    p.cg.H = H; p.cg.W = W; p.cg.Cin = Cin; p.cg.Ho = Ho; p.cg.Wo = Wo; p.cg.KW = kw; p.cg.stride = stride; p.cg.pad = pad;
    const int grid = (int)(((M + 127) / 128) * ((Cout + 127) / 128));
    TIC_RT_MAX_LDS((gemm_nt_kernel<TIC_EPI_BF16, true>), GEMM_LDS_BYTES);
    TIC_LAUNCH((gemm_nt_kernel<TIC_EPI_BF16, true>), grid, 256, GEMM_LDS_BYTES, stream, p);
    return tic_after_launch("conv_igemm_fwd");
}
// Input gradient of a STRIDE-2 convolution (3x3 pad 1: Bottleneck conv2 / BasicBlock conv1 of a stage's first block, model.py:87; 1x1 pad 0: the
// downsample projection, :193-197) without a column buffer or a col2im pass: the input pixels of parity class (py, px) receive contributions from
// (1 + py)(1 + px) of the 9 taps only (k = 3) or from the single tap when py = px = 0 (k = 1), so each class is a small stride-1 implicit GEMM over dY
// whose rows are stored at (2a + py, 2b + px).  One call = one class; w_class = tic_conv_weight_pack(..., transposed = 4 + 2 py + px) for k = 3,
// transposed = 2 for k = 1.  accumulate != 0: dx rows of the class += (TIC_EPI_ADDAUX semantics), the other classes are not touched.
extern "C" int tic_conv_igemm_dgrad_s2(const void* dy, const void* w_class, void* dx, int B, int H, int W, int Cin, int Cout, int k, int py, int px,
                                       int accumulate, tic_stream_t stream) {
    TIC_REQUIRE(dy && w_class && dx && B >= 1, "conv_igemm_dgrad_s2: bad argument");
    TIC_REQUIRE((k == 3 || (k == 1 && py == 0 && px == 0)) && (py | 1) == 1 && (px | 1) == 1, "conv_igemm_dgrad_s2: k = 3 with py, px in {0, 1}, or k = 1 with py = px = 0");
    TIC_REQUIRE(H % 2 == 0 && W % 2 == 0 && Cout % 64 == 0 && Cin % 8 == 0, "conv_igemm_dgrad_s2: need even H, W, Cout %% 64 == 0, Cin %% 8 == 0 (H=%d W=%d Cin=%d Cout=%d)", H, W, Cin, Cout);
    const int Hs = H / 2, Ws = W / 2, khp = (k == 3) ? 1 + py : 1, kwp = (k == 3) ? 1 + px : 1;
    const long M = (long)B * Hs * Ws;
    const int K = khp * kwp * Cout;
    TIC_REQUIRE(M < (1L << 24) && (double)B * H * W * Cin * 2.0 < 4294967296.0 && (double)Cin * K * 2.0 < 4294967296.0,
                "conv_igemm_dgrad_s2: tensor exceeds the 32-bit offset / 2^24 row range");
    GemmNtParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)dy; p.B = (const bf16_t*)w_class; p.M = (int)M; p.N = Cin; p.K = K; p.out = (bf16_t*)dx; p.aux = (const bf16_t*)dx;
    p.cg.H = Hs; p.cg.W = Ws; p.cg.Cin = Cout; p.cg.Ho = Hs; p.cg.Wo = Ws; p.cg.KW = kwp; p.cg.stride = 1; p.cg.pad = 0;   // dY is the "image": Ho x Wo = H/2 x W/2
    p.cg.rm_H = H; p.cg.rm_W = W; p.cg.rm_py = py; p.cg.rm_px = px;
    const int grid = (int)(((M + 127) / 128) * ((Cin + 127) / 128));
    if (accumulate) {
        TIC_RT_MAX_LDS((gemm_nt_kernel<TIC_EPI_ADDAUX, true>), GEMM_LDS_BYTES);
        TIC_LAUNCH((gemm_nt_kernel<TIC_EPI_ADDAUX, true>), grid, 256, GEMM_LDS_BYTES, stream, p);
    } else {
        TIC_RT_MAX_LDS((gemm_nt_kernel<TIC_EPI_BF16, true>), GEMM_LDS_BYTES);
        TIC_LAUNCH((gemm_nt_kernel<TIC_EPI_BF16, true>), grid, 256, GEMM_LDS_BYTES, stream, p);
    }
    return tic_after_launch("conv_igemm_dgrad_s2");
}
// dW[Cout, kh*kw*Cin] (fp32, tap-major) += dY[M, Cout]^T . gather(x)   -- the weight gradient without an im2col buffer
extern "C" int tic_conv_igemm_wgrad(const void* dy, const void* x_nhwc, float* dw, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                                    int stride, int pad, tic_stream_t stream) {
    TIC_REQUIRE(dy && x_nhwc && dw && B >= 1, "conv_igemm_wgrad: bad argument");
    const bool stem = TIC_CONV_IS_STEM(Cin, kh, kw, stride, pad);
    TIC_REQUIRE((Cin % 64 == 0 || stem) && Cout % 8 == 0 && kh >= 1 && kw >= 1 && stride >= 1 && pad >= 0, "conv_igemm_wgrad: need Cin %% 64 == 0 (or the 4-channel-padded 7x7/2 stem), Cout %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    TIC_REQUIRE(!stem || W % 2 == 0, "conv_igemm_wgrad: the stem form needs an even image width");
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    TIC_REQUIRE(Ho >= 1 && Wo >= 1, "conv_igemm_wgrad: empty output");
    const long Ml = (long)B * Ho * Wo;
    const int K = stem ? TIC_STEM_KP : kh * kw * Cin, N = Cout;
    TIC_REQUIRE(Ml < (1L << 24) - 64 && (double)B * H * W * Cin * 2.0 < 4294967296.0 && ((double)Ml + 64.0) * Cout * 2.0 < 4294967296.0,
                "conv_igemm_wgrad: tensor exceeds the 32-bit offset / 2^24 row range");
    const int M = (int)Ml;
    const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
    int split = tn_split(M, tiles);
    int m_per = ((M + split - 1) / split + 63) / 64 * 64;
    split = (M + m_per - 1) / m_per;
    GemmTnParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)dy; p.B = (const bf16_t*)x_nhwc; p.C = dw; p.M = M; p.N = N; p.K = K; p.m_per_slice = m_per;
    p.cg.H = H; p.cg.W = W; p.cg.Cin = Cin; p.cg.Ho = Ho; p.cg.Wo = Wo; p.cg.KW = kw; p.cg.stride = stride; p.cg.pad = pad;
    TIC_RT_MAX_LDS(gemm_tn_kernel<true>, GEMM_LDS_BYTES);
    TIC_LAUNCH(gemm_tn_kernel<true>, dim3(tiles, split), 256, GEMM_LDS_BYTES, stream, p);
    return tic_after_launch("conv_igemm_wgrad");
}
extern "C" int tic_conv_weight_grad(const float* dw, float* grad_oihw, int Co, int Ci, int kh, int kw, int layout, tic_stream_t stream) {
    TIC_REQUIRE(dw && grad_oihw && Co >= 1 && (layout == 0 || (layout == 3 && Ci == 3 && kh == 7 && kw == 7)), "conv_weight_grad: bad argument");
    ConvGeom g;
    TIC_TRY(conv_geom(g, 1, kh, kw, Ci, kh, kw, 1, 0));
    TIC_LAUNCH(weight_grad_oihw_kernel, ew_grid((long)Co * g.K), 256, 0, stream, dw, grad_oihw, Co, g, layout);
    return tic_after_launch("conv_weight_grad");
}
extern "C" int tic_nchw_to_nhwc_pad_bf16(const float* x, void* out, int B, int C, int Cpad, int H, int W, tic_stream_t stream) {
    TIC_REQUIRE(x && out && B >= 1 && C >= 1 && Cpad >= C, "nchw_to_nhwc: bad argument");
    TIC_LAUNCH(nchw_to_nhwc_kernel, ew_grid((long)B * Cpad * H * W), 256, 0, stream, x, (bf16_t*)out, B, C, Cpad, H, W);
    return tic_after_launch("nchw_to_nhwc");
}
extern "C" int tic_nchw_to_nhwc_bf16(const float* x, void* out, int B, int C, int H, int W, tic_stream_t stream) {
    return tic_nchw_to_nhwc_pad_bf16(x, out, B, C, C, H, W, stream);
}
extern "C" int tic_im2col_bf16(const void* x, void* col, int B, int H, int W, int Ci, int kh, int kw, int stride, int pad, tic_stream_t stream) {
    TIC_REQUIRE(x && col, "im2col: null pointer");
    ConvGeom g;
    TIC_TRY(conv_geom(g, B, H, W, Ci, kh, kw, stride, pad));
    const long M = (long)B * g.Ho * g.Wo;
    TIC_LAUNCH(im2col_kernel, ew_grid(M * (g.Kp / 8)), 256, 0, stream, (const bf16_t*)x, (bf16_t*)col, g);
    return tic_after_launch("im2col");
}
extern "C" int tic_col2im_bf16(const void* dcol, void* dx, int B, int H, int W, int Ci, int kh, int kw, int stride, int pad, int accumulate,
                               tic_stream_t stream) {
    TIC_REQUIRE(dcol && dx && Ci % 8 == 0, "col2im: need Ci %% 8 == 0");
    ConvGeom g;
    TIC_TRY(conv_geom(g, B, H, W, Ci, kh, kw, stride, pad));
    TIC_LAUNCH(col2im_kernel, ew_grid((long)B * H * W * (Ci / 8)), 256, 0, stream, (const bf16_t*)dcol, (bf16_t*)dx, g, accumulate);
    return tic_after_launch("col2im");
}
// row splits of the column-reduction kernels (grid.y): at most 512 (the kernels keep 4 / 2 rows per thread in flight), and no more than
// leave every thread 4 rows.  A block covers R = 256 / min(C/8, 32) rows per trip.  Every split stores 2 C partial sums (no atomics:
// conv.h explains why the sums must be order-independent), so the scratch is (1 + splits) x 2 C floats: [0, 2C) the final sums of the
// backward, then one 2 C row per split.
static int bn_rows(long M, int C) {
    const int cpb = (C / 8) < 32 ? (C / 8) : 32, R = 256 / cpb;
    long want = 512, most = (M + 4L * R - 1) / (4L * R);
    if (want > most) want = most;
    return (int)(want < 1 ? 1 : want);
}
extern "C" size_t tic_batchnorm_scratch_bytes(long M, int C) {
    if (M < 1 || C < 8) return 0;
    return (size_t)(1 + bn_rows(M, C)) * 2 * (size_t)C * sizeof(float);
}
#define TIC_BN_SCRATCH(name) \
    TIC_REQUIRE(scratch && scratch_bytes >= tic_batchnorm_scratch_bytes(M, C), name ": scratch smaller than tic_batchnorm_scratch_bytes(M, C)")
extern "C" int tic_batchnorm_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 int64_t* num_batches, float* mean, float* rstd, void* scratch, size_t scratch_bytes, const void* identity, void* y,
                                 long M, int C, float eps, float momentum, int train, int relu, tic_stream_t stream) {
    TIC_REQUIRE(x && gamma && beta && running_mean && running_var && mean && rstd && y, "batchnorm_fwd: null pointer");
    TIC_REQUIRE(M >= 1 && C >= 8 && C % 8 == 0 && 256 % (C / 8) == 0, "batchnorm_fwd: need C/8 to divide 256 (C = 64 .. 2048, powers of two)");
    TIC_BN_SCRATCH("batchnorm_fwd");
    float* part = (float*)scratch + 2 * C;
    const int ns = bn_rows(M, C);
    if (train) TIC_LAUNCH(bn_stats_kernel, dim3((C + 255) / 256, ns), 256, 2 * 8 * 256 * 4, stream, (const bf16_t*)x, part, M, C);
    TIC_LAUNCH(bn_finalize_kernel, (C + BN_SUM_CH - 1) / BN_SUM_CH, 256, 2 * 256 * 4, stream, part, ns, mean, rstd, running_mean, running_var, (long long*)num_batches, M, C, eps,
               momentum, train);
    TIC_LAUNCH(bn_apply_kernel, ew_grid(M * (C / 8)), 256, 0, stream, (const bf16_t*)x, mean, rstd, gamma, beta, (const bf16_t*)identity, (bf16_t*)y, M, C, relu);
    return tic_after_launch("batchnorm_fwd");
}
static int batchnorm_bwd_impl(const void* dy, const void* y_or_null, const void* x, const float* mean, const float* rstd, const float* gamma,
                              const float* beta_mask, void* scratch, size_t scratch_bytes, void* dx, void* dskip, int skip_accumulate, float* dgamma,
                              float* dbeta, long M, int C, tic_stream_t stream) {
    TIC_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta, "batchnorm_bwd: null pointer");
    TIC_REQUIRE(M >= 1 && C >= 8 && C % 8 == 0 && 256 % (C / 8) == 0, "batchnorm_bwd: need C/8 to divide 256 (C = 64 .. 2048, powers of two)");
    TIC_REQUIRE(!(y_or_null && beta_mask), "batchnorm_bwd: the ReLU mask comes from y OR from x, not both");
    TIC_BN_SCRATCH("batchnorm_bwd");
    float* red = (float*)scratch;
    float* part = red + 2 * C;
    const int ns = bn_rows(M, C);
    TIC_LAUNCH(bn_bwd_reduce_kernel, dim3((C + 255) / 256, ns), 256, 2 * 8 * 256 * 4, stream, (const bf16_t*)dy, (const bf16_t*)y_or_null,
               (const bf16_t*)x, mean, rstd, part, M, C, gamma, beta_mask ? beta_mask : gamma, beta_mask ? 1 : 0);
    TIC_LAUNCH(bn_param_grad_kernel, (C + BN_SUM_CH - 1) / BN_SUM_CH, 256, 2 * 256 * 4, stream, part, ns, red, dgamma, dbeta, C);
    TIC_LAUNCH(bn_bwd_apply_kernel, ew_grid(M * (C / 8)), 256, 0, stream, (const bf16_t*)dy, (const bf16_t*)y_or_null, (const bf16_t*)x, mean, rstd, gamma,
               red, (bf16_t*)dx, (bf16_t*)dskip, skip_accumulate, M, C, beta_mask ? beta_mask : gamma, beta_mask ? 1 : 0);
    return tic_after_launch("batchnorm_bwd");
}
extern "C" int tic_batchnorm_bwd(const void* dy, const void* y_or_null, const void* x, const float* mean, const float* rstd, const float* gamma,
                                 void* scratch, size_t scratch_bytes, void* dx, void* dskip, int skip_accumulate, float* dgamma, float* dbeta, long M,
                                 int C, tic_stream_t stream) {
    return batchnorm_bwd_impl(dy, y_or_null, x, mean, rstd, gamma, nullptr, scratch, scratch_bytes, dx, dskip, skip_accumulate, dgamma, dbeta, M, C, stream);
}
// backward of y = relu(bn(x)) WITHOUT a residual add: the ReLU mask is recomputed from x (bit-identical to the forward's), y is not read
extern "C" int tic_batchnorm_bwd_relu(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      void* scratch, size_t scratch_bytes, void* dx, float* dgamma, float* dbeta, long M, int C, tic_stream_t stream) {
    TIC_REQUIRE(beta, "batchnorm_bwd_relu: null beta");
    return batchnorm_bwd_impl(dy, nullptr, x, mean, rstd, gamma, beta, scratch, scratch_bytes, dx, nullptr, 0, dgamma, dbeta, M, C, stream);
}
// The ResNet stem's tail in one piece (TIC/ResNet/model.py:150-152: bn1 -> relu -> maxpool): y_pool = maxpool3x3/2(relu(bn(x))) and the
// argmax positions, WITHOUT storing relu(bn(x)).  Bit-identical to tic_batchnorm_fwd(relu) + tic_maxpool3x3s2_fwd_idx; the backward is
// tic_maxpool3x3s2_bwd_idx + tic_batchnorm_bwd_relu (fusing those too was measured slower: both BatchNorm passes would repeat the
// window gather, 462 + 343 us against 207 + 160 + 240).  scratch as in tic_batchnorm_fwd with M = B H W.
extern "C" int tic_bn_relu_maxpool_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* num_batches,
                                       float* mean, float* rstd, void* scratch, size_t scratch_bytes, void* y_pool, void* idx_u8_or_null, int B, int H, int W,
                                       int C, float eps, float momentum, int train, tic_stream_t stream) {
    TIC_REQUIRE(x && gamma && beta && running_mean && running_var && mean && rstd && y_pool, "bn_relu_maxpool_fwd: null pointer");
    TIC_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 8 && C % 8 == 0 && 256 % (C / 8) == 0, "bn_relu_maxpool_fwd: need C/8 to divide 256");
    const long M = (long)B * H * W;
    TIC_BN_SCRATCH("bn_relu_maxpool_fwd");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    float* part = (float*)scratch + 2 * C;
    const int ns = bn_rows(M, C);
    if (train) TIC_LAUNCH(bn_stats_kernel, dim3((C + 255) / 256, ns), 256, 2 * 8 * 256 * 4, stream, (const bf16_t*)x, part, M, C);
    TIC_LAUNCH(bn_finalize_kernel, (C + BN_SUM_CH - 1) / BN_SUM_CH, 256, 2 * 256 * 4, stream, part, ns, mean, rstd, running_mean, running_var, (long long*)num_batches, M, C, eps,
               momentum, train);
    TIC_LAUNCH(bn_relu_maxpool_fwd_kernel, ew_grid((long)B * Ho * Wo * (C / 8)), 256, 0, stream, (const bf16_t*)x, mean, rstd, gamma, beta, (bf16_t*)y_pool,
               (unsigned char*)idx_u8_or_null, B, H, W, C, Ho, Wo);
    return tic_after_launch("bn_relu_maxpool_fwd");
}
extern "C" int tic_maxpool3x3s2_fwd(const void* x, void* y, int B, int H, int W, int C, tic_stream_t stream) {
    TIC_REQUIRE(x && y && C % 8 == 0, "maxpool_fwd: need C %% 8 == 0");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    TIC_LAUNCH(maxpool_fwd_kernel, ew_grid((long)B * Ho * Wo * (C / 8)), 256, 0, stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo);
    return tic_after_launch("maxpool_fwd");
}
extern "C" int tic_maxpool3x3s2_bwd(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C, tic_stream_t stream) {
    TIC_REQUIRE(x && y && dy && dx && C % 8 == 0, "maxpool_bwd: null pointer or C %% 8 != 0");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    TIC_LAUNCH(maxpool_bwd_kernel, ew_grid((long)B * H * W * (C / 8)), 256, 0, stream, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C, Ho, Wo);
    return tic_after_launch("maxpool_bwd");
}
extern "C" int tic_maxpool3x3s2_fwd_idx(const void* x, void* y, void* idx_u8, int B, int H, int W, int C, tic_stream_t stream) {
    TIC_REQUIRE(x && y && idx_u8 && C % 8 == 0, "maxpool_fwd_idx: null pointer or C %% 8 != 0");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    TIC_LAUNCH(maxpool_fwd_idx_kernel, ew_grid((long)B * Ho * Wo * (C / 8)), 256, 0, stream, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)idx_u8, B, H, W, C, Ho, Wo);
    return tic_after_launch("maxpool_fwd_idx");
}
extern "C" int tic_maxpool3x3s2_bwd_idx(const void* idx_u8, const void* dy, void* dx, int B, int H, int W, int C, tic_stream_t stream) {
    TIC_REQUIRE(idx_u8 && dy && dx && C % 8 == 0, "maxpool_bwd_idx: null pointer or C %% 8 != 0");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    TIC_LAUNCH(maxpool_bwd_idx_kernel, ew_grid((long)B * H * W * (C / 8)), 256, 0, stream, (const unsigned char*)idx_u8, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C, Ho, Wo);
    return tic_after_launch("maxpool_bwd_idx");
}
extern "C" int tic_avgpool_fwd(const void* x, void* y, int B, int HW, int C, tic_stream_t stream) {
    TIC_REQUIRE(x && y && B >= 1 && HW >= 1, "avgpool_fwd: bad argument");
    TIC_LAUNCH(avgpool_fwd_kernel, (B * C + 255) / 256, 256, 0, stream, (const bf16_t*)x, (bf16_t*)y, B, HW, C);
    return tic_after_launch("avgpool_fwd");
}
extern "C" int tic_avgpool_bwd(const void* dy, void* dx, int B, int HW, int C, tic_stream_t stream) {
    TIC_REQUIRE(dy && dx && B >= 1 && HW >= 1, "avgpool_bwd: bad argument");
    TIC_LAUNCH(avgpool_bwd_kernel, ew_grid((long)B * HW * C), 256, 0, stream, (const bf16_t*)dy, (bf16_t*)dx, B, HW, C);
    return tic_after_launch("avgpool_bwd");
}
extern "C" int tic_add_bf16(void* a, const void* b, long n, tic_stream_t stream) {
    TIC_REQUIRE(a && b && n >= 8 && n % 8 == 0, "add_bf16: need n %% 8 == 0");
    TIC_LAUNCH(add_bf16_kernel, ew_grid(n / 8), 256, 0, stream, (bf16_t*)a, (const bf16_t*)b, n / 8);
    return tic_after_launch("add_bf16");
}

// ---- whole-model layout + phase drivers -------------------------------------------------------------
static long pad8(long n) { return (n + 7) / 8 * 8; }
static size_t pad256(size_t n) { return (n + 255) / 256 * 256; }

extern "C" int tic_vit_layout(const TicVitDims* d, TicVitLayout* o) {
    TIC_REQUIRE(d && o, "vit_layout: null pointer");
    TIC_REQUIRE(d->B >= 1 && d->L >= 1 && d->C >= 1 && d->H >= 1 && d->D == d->H * 64, "vit_layout: need D == H * 64 (D=%d H=%d)", d->D, d->H);
    TIC_REQUIRE(d->D % 128 == 0 && d->F % 128 == 0 && d->D <= 1024, "vit_layout: need D, F multiples of 128 and D <= 1024");
    TIC_REQUIRE(d->patch % 8 == 0 && d->img % d->patch == 0 && (d->chans * d->patch * d->patch) % 128 == 0, "vit_layout: bad image geometry");
    const long D = d->D, F = d->F, C = d->C, L = d->L, G = d->img / d->patch, N = G * G + 1, PK = (long)d->chans * d->patch * d->patch;
    TIC_REQUIRE(N <= 208, "vit_layout: tokens per image must be <= 208 (got %ld)", N);
    memset(o, 0, sizeof(*o));
    long off = 0;
    o->cls = off; off += pad8(D);
    o->pos = off; off += pad8(N * D);
    o->patch_w = off; off += pad8(D * PK);
    o->patch_b = off; off += pad8(D);
    o->layer0 = off;
    long lo = 0;
    o->ln1_g = lo; lo += pad8(D);
    o->ln1_b = lo; lo += pad8(D);
    o->wqkv = lo; lo += pad8(3 * D * D);
    o->bqkv = lo; lo += pad8(3 * D);
    o->wo = lo; lo += pad8(D * D);
    o->bo = lo; lo += pad8(D);
    o->ln2_g = lo; lo += pad8(D);
    o->ln2_b = lo; lo += pad8(D);
    o->w1 = lo; lo += pad8(F * D);
    o->b1 = lo; lo += pad8(F);
    o->w2 = lo; lo += pad8(D * F);
    o->b2 = lo; lo += pad8(D);
    o->layer_stride = lo;
    off += L * lo;
    o->lnf_g = off; off += pad8(D);
    o->lnf_b = off; off += pad8(D);
    o->cls_w = off; off += pad8(C * D);
    o->cls_b = off; off += pad8(C);
    o->n_params = off;
    long to = 0;
    o->t_wqkv = to; to += 3 * D * D;
    o->t_wo = to; to += D * D;
    o->t_w1 = to; to += F * D;
    o->t_w2 = to; to += D * F;
    o->t_layer_stride = to;
    o->t_total = L * to;
    // activations
    const size_t B = d->B, M = B * N;
    size_t w = 0;
    o->P = w; w += pad256(B * (N - 1) * PK * 2);
    o->hs = w; o->hs_stride = pad256(M * D * 4); w += (L + 1) * o->hs_stride;
    size_t lw = 0;
    o->a1 = lw; lw += pad256(M * D * 2);
    o->mean1 = lw; lw += pad256(M * 4);
    o->rstd1 = lw; lw += pad256(M * 4);
    o->qkv = lw; lw += pad256(M * 3 * D * 2);
    o->lse = lw; lw += pad256(B * d->H * N * 4);
    o->o = lw; lw += pad256(M * D * 2);
    o->hmid = lw; lw += pad256(M * D * 4);
    o->a2 = lw; lw += pad256(M * D * 2);
    o->mean2 = lw; lw += pad256(M * 4);
    o->rstd2 = lw; lw += pad256(M * 4);
    o->u = lw; lw += pad256(M * F * 2);
    o->g = lw; lw += pad256(M * F * 2);
    o->layer_ws = w; o->layer_ws_stride = lw; w += L * lw;
    o->zf = w; w += pad256(B * D * 2);
    o->meanf = w; w += pad256(B * 4);
    o->rstdf = w; w += pad256(B * 4);
    o->logits = w; w += pad256(B * C * 4);
    o->dlogits = w; w += pad256(B * C * 4);
    o->dzf = w; w += pad256(B * D * 2);
    o->dh = w; w += pad256(M * D * 4);
    o->dhb = w; w += pad256(M * D * 2);
    o->dhb2 = w; w += pad256(M * D * 2);
    o->du = w; w += pad256(M * F * 2);
    o->da = w; w += pad256(M * D * 2);
    o->dqkv = w; w += pad256(M * 3 * D * 2);
    o->dpatch = w; w += pad256(B * (N - 1) * D * 2);
    o->nt_scratch = w; w += pad256(TIC_NT_SCRATCH_BYTES);
    o->ws_bytes = w;
    return TIC_OK;
}


struct VitCtx {
    TicVitLayout lay;
    long D, F, C, L, N, M, PK, B, H;
    float* P;       // params
    float* G;       // grads
    bf16_t* W16;
    bf16_t* WT;
    char* ws;
    float eps;
    // while a tic_vit_* call runs, its GEMMs may split K through the scratch inside THIS state's workspace; the caller's own
    // registration (tic_gemm_nt_scratch) is put back when the call returns, on every path
    char* prev_scratch = nullptr;
    bool armed = false;
    ~VitCtx() {
        if (armed) g_nt_scratch = prev_scratch;
    }
};
static int vit_ctx(const TicVitState* st, VitCtx& c) {
    TIC_REQUIRE(st && st->params && st->grads && st->w16 && st->wT16 && st->workspace, "vit: null state pointer");
    TIC_TRY(tic_vit_layout(&st->dims, &c.lay));
    const TicVitDims& d = st->dims;
    const long G = d.img / d.patch;
    c.D = d.D; c.F = d.F; c.C = d.C; c.L = d.L; c.N = G * G + 1; c.B = d.B; c.M = c.B * c.N; c.H = d.H;
    c.PK = (long)d.chans * d.patch * d.patch;
    c.P = st->params; c.G = st->grads; c.W16 = (bf16_t*)st->w16; c.WT = (bf16_t*)st->wT16; c.ws = (char*)st->workspace;
    c.eps = d.eps;
    c.prev_scratch = g_nt_scratch;
    c.armed = true;
    g_nt_scratch = c.ws + c.lay.nt_scratch;
    return TIC_OK;
}

extern "C" int tic_vit_refresh_weights(const TicVitState* st, int transposes_only, tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    const TicVitLayout& y = c.lay;
    if (!transposes_only) TIC_TRY(tic_cast_bf16(c.P, c.W16, y.n_params, s));
    if (c.D % 64 == 0 && c.F % 64 == 0 && c.L <= 65535) {   // every Linear weight of the stack in one launch
        CastTransposeGroup gp;
        const long in_off[4] = {y.wqkv, y.wo, y.w1, y.w2}, out_off[4] = {y.t_wqkv, y.t_wo, y.t_w1, y.t_w2};
        const long R[4] = {3 * c.D, c.D, c.F, c.D}, C[4] = {c.D, c.D, c.D, c.F};
        int tiles = 0;
        for (int i = 0; i < 4; ++i) {
            gp.in_off[i] = in_off[i]; gp.out_off[i] = out_off[i]; gp.R[i] = (int)R[i]; gp.C[i] = (int)C[i];
            tiles += (int)((R[i] / 64) * (C[i] / 64));
            gp.tile_end[i] = tiles;
        }
        gp.in_stride = y.layer_stride; gp.out_stride = y.t_layer_stride;
        TIC_LAUNCH(cast_transpose_group_kernel, dim3(tiles, (unsigned)c.L), 256, 64 * CT_STRIDE * 2, s, c.P + y.layer0, c.WT, gp);
        return tic_after_launch("vit_refresh_weights");
    }
    for (long l = 0; l < c.L; ++l) {
        const float* lp = c.P + y.layer0 + l * y.layer_stride;
        bf16_t* lt = c.WT + l * y.t_layer_stride;
        TIC_TRY(tic_cast_transpose_bf16(lp + y.wqkv, lt + y.t_wqkv, (int)(3 * c.D), (int)c.D, s));
        TIC_TRY(tic_cast_transpose_bf16(lp + y.wo, lt + y.t_wo, (int)c.D, (int)c.D, s));
        TIC_TRY(tic_cast_transpose_bf16(lp + y.w1, lt + y.t_w1, (int)c.F, (int)c.D, s));
        TIC_TRY(tic_cast_transpose_bf16(lp + y.w2, lt + y.t_w2, (int)c.D, (int)c.F, s));
    }
    return TIC_OK;
}

// AdamW over the replica's flat buffers with BOTH bf16 operand copies written in the same pass (torch.optim.AdamW semantics of tic_adamw:
// TIC/ViT/ntrain.py:39-41, one parameter group, decoupled weight decay on every parameter): the four Linear matrices of every block tile by
// tile (adamw_tiles_kernel: fp32 update + w16 + wT16), everything else by the flat rule.  After it tic_vit_refresh_weights has nothing to do.
// the stretches of the parameter / gradient buffer outside the per-layer weight matrices (elementwise.h RestGaps)
static int vit_rest_gaps(const VitCtx& c, RestGaps& z) {
    const TicVitLayout& y = c.lay;
    const long R[4] = {3 * c.D, c.D, c.F, c.D}, C[4] = {c.D, c.D, c.D, c.F};
    long lo[4] = {y.wqkv, y.wo, y.w1, y.w2}, hi[4];
    for (int i = 0; i < 4; ++i) hi[i] = lo[i] + R[i] * C[i];
    for (int i = 0; i < 4; ++i)   // by start offset
        for (int j = i + 1; j < 4; ++j)
            if (lo[j] < lo[i]) {
                const long a = lo[i], b = hi[i];
                lo[i] = lo[j]; hi[i] = hi[j]; lo[j] = a; hi[j] = b;
            }
    z.l0 = y.layer0; z.stride = y.layer_stride; z.nlayers = c.L; z.n = y.n_params;
    long at = 0;
    for (int k = 0; k < 5; ++k) {
        const long end = k < 4 ? lo[k] : y.layer_stride;
        TIC_REQUIRE(end >= at && at % 4 == 0 && end % 4 == 0, "vit: unexpected parameter layout (matrix ranges overlap or are unaligned)");
        z.lo[k] = at; z.len[k] = end - at;
        if (k < 4) at = hi[k];
    }
    TIC_REQUIRE(y.layer0 % 4 == 0 && y.n_params % 4 == 0 && y.layer_stride % 4 == 0, "vit: unexpected parameter layout");
    return TIC_OK;
}

extern "C" int tic_vit_adamw(const TicVitState* st, float* m, float* v, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    TIC_REQUIRE(m && v && step >= 1, "vit_adamw: null moment buffers or step < 1");
    TIC_REQUIRE(c.D % 64 == 0 && c.F % 64 == 0 && c.L <= 65535, "vit_adamw: needs hidden / MLP widths that are multiples of 64 (use tic_adamw + tic_vit_refresh_weights)");
    const TicVitLayout& y = c.lay;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamwHyper h;
    h.lr = lr; h.b1 = beta1; h.b2 = beta2; h.eps = eps; h.wd = weight_decay; h.inv_bc1 = (float)(1.0 / bc1); h.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    CastTransposeGroup gp;
    RestGaps gaps;
    TIC_TRY(vit_rest_gaps(c, gaps));
    const long in_off[4] = {y.wqkv, y.wo, y.w1, y.w2}, out_off[4] = {y.t_wqkv, y.t_wo, y.t_w1, y.t_w2};
    const long R[4] = {3 * c.D, c.D, c.F, c.D}, C[4] = {c.D, c.D, c.D, c.F};
    int tiles = 0;
    for (int i = 0; i < 4; ++i) {
        gp.in_off[i] = y.layer0 + in_off[i]; gp.out_off[i] = out_off[i]; gp.R[i] = (int)R[i]; gp.C[i] = (int)C[i];
        tiles += (int)((R[i] / 64) * (C[i] / 64));
        gp.tile_end[i] = tiles;
    }
    gp.in_stride = y.layer_stride; gp.out_stride = y.t_layer_stride;
    TIC_LAUNCH(adamw_tiles_kernel, dim3(tiles, (unsigned)c.L), 256, 64 * CT_STRIDE * 2, s, c.P, c.G, m, v, c.W16, c.WT, gp, h);
    TIC_LAUNCH(adamw_rest_kernel, dim3((unsigned)c.L + 2, 5 * 8), 256, 0, s, c.P, c.G, m, v, c.W16, gaps, h);
    return tic_after_launch("vit_adamw");
}

// Clears the gradient buffer before a backward.  keep_matrices != 0: the ranges of the per-layer weight matrices (99.7 % of ViT-L) are
// left as they are -- for a backward that stores them (tic_vit_backward_layer_ex(..., 1, ...)); everything else is zeroed.
extern "C" int tic_vit_zero_grads(const TicVitState* st, int keep_matrices, tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    const TicVitLayout& y = c.lay;
    if (!keep_matrices) {
        TIC_RT_MEMSET(c.G, 0, (size_t)y.n_params * 4, s);
        return tic_after_launch("vit_zero_grads");
    }
    RestGaps z;
    TIC_TRY(vit_rest_gaps(c, z));
    TIC_LAUNCH(zero_gaps_kernel, dim3((unsigned)c.L + 2, 5 * 8), 256, 0, s, c.G, z);
    return tic_after_launch("vit_zero_grads");
}

static int vit_forward_impl(const TicVitState* st, const float* x, float* logits_out, bool infer, tic_stream_t s);
extern "C" int tic_vit_forward(const TicVitState* st, const float* x, float* logits_out, tic_stream_t s) {
    return vit_forward_impl(st, x, logits_out, false, s);
}
extern "C" int tic_vit_forward_infer(const TicVitState* st, const float* x, float* logits_out, tic_stream_t s) {
    return vit_forward_impl(st, x, logits_out, true, s);
}
static int vit_forward_impl(const TicVitState* st, const float* x, float* logits_out, bool infer, tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    TIC_REQUIRE(x, "vit_forward: null pixel_values");
    const TicVitLayout& y = c.lay;
    const int M = (int)c.M, D = (int)c.D, F = (int)c.F, B = (int)c.B, N = (int)c.N;
    bf16_t* Pm = (bf16_t*)(c.ws + y.P);
    float* h0 = (float*)(c.ws + y.hs);
    TIC_TRY(tic_patchify(x, Pm, B, st->dims.chans, st->dims.img, st->dims.patch, s));
    TIC_TRY(tic_gemm_nt_bf16(Pm, c.W16 + y.patch_w, B * (N - 1), D, (int)c.PK, TIC_EPI_PATCH, c.P + y.patch_b, nullptr, nullptr, h0,
                             nullptr, nullptr, c.P + y.pos, N - 1, s));
    TIC_TRY(tic_embed_cls(c.P + y.cls, c.P + y.pos, h0, B, N, D, s));
    for (long l = 0; l < c.L; ++l) {
        const float* lp = c.P + y.layer0 + l * y.layer_stride;
        const bf16_t* lw = c.W16 + y.layer0 + l * y.layer_stride;
        char* a = c.ws + y.layer_ws + l * y.layer_ws_stride;
        float* hin = (float*)(c.ws + y.hs + l * y.hs_stride);
        float* hout = (float*)(c.ws + y.hs + (l + 1) * y.hs_stride);
        float* hmid = (float*)(a + y.hmid);
        TIC_TRY(tic_layernorm_fwd(hin, D, lp + y.ln1_g, lp + y.ln1_b, a + y.a1, (float*)(a + y.mean1), (float*)(a + y.rstd1), M, D, c.eps, s));
        TIC_TRY(tic_gemm_nt_bf16(a + y.a1, lw + y.wqkv, M, 3 * D, D, TIC_EPI_BF16, lp + y.bqkv, a + y.qkv, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s));
        TIC_TRY(tic_attention_fwd(a + y.qkv, a + y.o, (float*)(a + y.lse), B, (int)c.H, N, 0.125f, s));
        TIC_TRY(tic_gemm_nt_bf16(a + y.o, lw + y.wo, M, D, D, TIC_EPI_RESID, lp + y.bo, nullptr, nullptr, hmid, hin, nullptr, nullptr, 0, s));
        TIC_TRY(tic_layernorm_fwd(hmid, D, lp + y.ln2_g, lp + y.ln2_b, a + y.a2, (float*)(a + y.mean2), (float*)(a + y.rstd2), M, D, c.eps, s));
        // fc1 + GELU: gelu'(u) (-> a.u) is stored for the backward only
        TIC_TRY(tic_gemm_nt_bf16(a + y.a2, lw + y.w1, M, F, D, infer ? TIC_EPI_GELU_ONLY : TIC_EPI_GELU_DG, lp + y.b1, infer ? nullptr : a + y.u, a + y.g, nullptr,
                                 nullptr, nullptr, nullptr, 0, s));
        TIC_TRY(tic_gemm_nt_bf16(a + y.g, lw + y.w2, M, D, F, TIC_EPI_RESID, lp + y.b2, nullptr, nullptr, hout, hmid, nullptr, nullptr, 0, s));
    }
    float* hL = (float*)(c.ws + y.hs + c.L * y.hs_stride);
    TIC_TRY(tic_layernorm_fwd(hL, (long)N * D, c.P + y.lnf_g, c.P + y.lnf_b, c.ws + y.zf, (float*)(c.ws + y.meanf), (float*)(c.ws + y.rstdf), B, D, c.eps, s));
    float* lg = (float*)(c.ws + y.logits);
    TIC_TRY(tic_head_fwd(c.ws + y.zf, c.P + y.cls_w, c.P + y.cls_b, lg, B, (int)c.C, D, s));
    if (logits_out && logits_out != lg) TIC_RT_MEMCPY(logits_out, lg, (size_t)B * c.C * 4, s);
    return TIC_OK;
}

extern "C" int tic_vit_backward_head(const TicVitState* st, const float* dlogits, tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    TIC_REQUIRE(dlogits, "vit_backward_head: null dlogits");
    const TicVitLayout& y = c.lay;
    const int D = (int)c.D, B = (int)c.B, N = (int)c.N;
    float* dh = (float*)(c.ws + y.dh);
    TIC_TRY(tic_head_bwd(dlogits, c.ws + y.zf, c.P + y.cls_w, c.ws + y.dzf, c.G + y.cls_w, c.G + y.cls_b, B, (int)c.C, D, s));
    TIC_RT_MEMSET(dh, 0, (size_t)c.M * D * 4, s);
    TIC_RT_MEMSET(c.ws + y.dhb, 0, (size_t)c.M * D * 2, s);
    float* hL = (float*)(c.ws + y.hs + c.L * y.hs_stride);
    // the bf16 gradient this emits is dY of the last layer's fc2: its column sums are that layer's b2 gradient
    TIC_TRY(tic_layernorm_bwd_ex(c.ws + y.dzf, hL, (long)N * D, c.P + y.lnf_g, (float*)(c.ws + y.meanf), (float*)(c.ws + y.rstdf), nullptr, dh,
                                 c.ws + y.dhb, c.G + y.lnf_g, c.G + y.lnf_b, c.G + y.layer0 + (c.L - 1) * y.layer_stride + y.b2, B, D, s));
    return TIC_OK;
}

extern "C" int tic_vit_backward_layer_ex(const TicVitState* st, int layer, int overwrite_dw, tic_stream_t s);
extern "C" int tic_vit_backward_layer(const TicVitState* st, int layer, tic_stream_t s) { return tic_vit_backward_layer_ex(st, layer, 0, s); }
// overwrite_dw != 0: the four weight-matrix gradients of the block are STORED, not accumulated (their ranges of the gradient buffer need
// not be zero: tic_vit_zero_grads(st, 1, ...) leaves them alone); everything else (biases, LayerNorm) still accumulates
extern "C" int tic_vit_backward_layer_ex(const TicVitState* st, int layer, int overwrite_dw, tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    TIC_REQUIRE(layer >= 0 && layer < c.L, "vit_backward_layer: layer %d out of range", layer);
    const TicVitLayout& y = c.lay;
    const int M = (int)c.M, D = (int)c.D, F = (int)c.F, B = (int)c.B, N = (int)c.N;
    const long l = layer;
    const float* lp = c.P + y.layer0 + l * y.layer_stride;
    float* lg = c.G + y.layer0 + l * y.layer_stride;
    const bf16_t* lt = c.WT + l * y.t_layer_stride;
    char* a = c.ws + y.layer_ws + l * y.layer_ws_stride;
    float* hin = (float*)(c.ws + y.hs + l * y.hs_stride);
    float* dh = (float*)(c.ws + y.dh);
    char* dhb = c.ws + y.dhb;
    char* du = c.ws + y.du;
    char* da = c.ws + y.da;
    char* dqkv = c.ws + y.dqkv;
    char* dhb2 = c.ws + y.dhb2;
    // dX chain first (MLP, then attention); the four weight gradients of the block go out as ONE grouped launch
    // bias gradients ride on the producers of each dY: b1 <- DGELU epilogue, bo <- LN2 backward, bqkv <- attention
    // backward, b2 (of the layer below) <- LN1 backward; b2 of THIS layer was added by the producer of dhb
    TIC_TRY(tic_gemm_nt_bf16_ex(dhb, lt + y.t_w2, M, F, D, TIC_EPI_MULAUX, nullptr, du, nullptr, nullptr, nullptr, a + y.u, nullptr, 0, lg + y.b1, s));
    TIC_TRY(tic_gemm_nt_bf16(du, lt + y.t_w1, M, D, F, TIC_EPI_BF16, nullptr, da, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s));
    TIC_TRY(tic_layernorm_bwd_ex(da, (float*)(a + y.hmid), D, lp + y.ln2_g, (float*)(a + y.mean2), (float*)(a + y.rstd2), dh, dh, dhb2, lg + y.ln2_g,
                                 lg + y.ln2_b, lg + y.bo, M, D, s));
    // dO = dh . Wo (no bias); its column sums ARE the gradient of the v bias (dV = P^T dO and the rows of P sum to 1)
    TIC_TRY(tic_gemm_nt_bf16_ex(dhb2, lt + y.t_wo, M, D, D, TIC_EPI_BF16, nullptr, da, nullptr, nullptr, nullptr, nullptr, nullptr, 0, lg + y.bqkv + 2 * D, s));
    // scratch for the per-image bias-gradient sums: the saved gelu' buffer of this layer ([M, F] bf16), dead since the MULAUX GEMM
    // above -- when it is large enough for [B][3D] floats (always at 197 tokens; not for a 5-token test model: atomics then)
    float* part = ((size_t)M * F * 2 >= (size_t)B * 3 * D * 4) ? (float*)(a + y.u) : nullptr;
    TIC_TRY(tic_attention_bwd_ws(a + y.qkv, a + y.o, (float*)(a + y.lse), da, dqkv, lg + y.bqkv, part, 1, B, (int)c.H, N, 0.125f, s));
    TIC_TRY(tic_gemm_nt_bf16(dqkv, lt + y.t_wqkv, M, D, 3 * D, TIC_EPI_BF16, nullptr, da, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s));
    {
        const void* gA[4] = {dhb, du, dhb2, dqkv};
        const void* gB[4] = {a + y.g, a + y.a2, a + y.o, a + y.a1};
        float* gC[4] = {lg + y.w2, lg + y.w1, lg + y.wo, lg + y.wqkv};
        const int gN[4] = {D, F, D, 3 * D}, gK[4] = {F, D, D, D};
        TIC_TRY(tic_gemm_tn_group_bf16_ex(4, gA, gB, gC, gN, gK, M, overwrite_dw, s));
    }
    TIC_TRY(tic_layernorm_bwd_ex(da, hin, D, lp + y.ln1_g, (float*)(a + y.mean1), (float*)(a + y.rstd1), dh, dh, dhb, lg + y.ln1_g, lg + y.ln1_b,
                                 l > 0 ? c.G + y.layer0 + (l - 1) * y.layer_stride + y.b2 : nullptr, M, D, s));
    return TIC_OK;
}

extern "C" int tic_vit_backward_embed(const TicVitState* st, tic_stream_t s) {
    VitCtx c;
    TIC_TRY(vit_ctx(st, c));
    const TicVitLayout& y = c.lay;
    const int D = (int)c.D, B = (int)c.B, N = (int)c.N;
    float* dh = (float*)(c.ws + y.dh);
    TIC_TRY(tic_embed_bwd(dh, c.G + y.cls, c.G + y.pos, B, N, D, s));
    TIC_TRY(tic_gather_patch_rows(dh, c.ws + y.dpatch, B, N, D, s));
    TIC_TRY(tic_gemm_tn_bf16(c.ws + y.dpatch, c.ws + y.P, c.G + y.patch_w, B * (N - 1), D, (int)c.PK, s));
    TIC_TRY(tic_colsum_bf16(c.ws + y.dpatch, c.G + y.patch_b, B * (N - 1), D, s));
    return TIC_OK;
}
