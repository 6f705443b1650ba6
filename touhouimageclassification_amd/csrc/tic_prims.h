// tic_prims.h -- the only place gfx950 builtins are spelled.
//
// Device build (hipcc --offload-arch=gfx950): thin always-inline wrappers over the CDNA4
// builtins (MFMA, LDS-DMA buffer loads, ds_read_b64_tr_b16, raw barriers, counted waits).
// Simulator build (-DTIC_SIM, clang++ -x c++, tests/sim/): the same names are provided by
// tests/sim/sim_runtime.h, which executes a workgroup as 64-lane lock-step fibers on the
// CPU so that lane maps / LDS swizzles / bounds handling can be checked without a GPU.
// The simulator is TEST INFRASTRUCTURE: the shipped library contains only the device build.
#pragma once
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPR)
typedef __attribute__((ext_vector_type(4))) short bf16x4;   // 4 bf16 (2 VGPR)
typedef __attribute__((ext_vector_type(2))) short bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_t;   // storage type of a bf16 element in memory

#ifdef TIC_SIM
#include "sim_runtime.h"
#else
#include <hip/hip_runtime.h>

#define TIC_DEV __device__ __forceinline__
#define TIC_KERNEL(name) extern "C" __global__ void name

// ---- dynamic LDS -----------------------------------------------------------------------------
// One dynamic array, 16-byte aligned, no static __shared__ beside it (guide G17 / "second
// __shared__ object" trap).  All LDS addressing is by BYTE OFFSET into this array.
extern __shared__ __attribute__((aligned(16))) char tic_smem[];

TIC_DEV bf16x8 lds_ld128(uint32_t off) { return *reinterpret_cast<const bf16x8*>(tic_smem + off); }
TIC_DEV bf16x4 lds_ld64(uint32_t off) { return *reinterpret_cast<const bf16x4*>(tic_smem + off); }
TIC_DEV float lds_ldf(uint32_t off) { return *reinterpret_cast<const float*>(tic_smem + off); }
TIC_DEV f32x4 lds_ldf4(uint32_t off) { return *reinterpret_cast<const f32x4*>(tic_smem + off); }
TIC_DEV void lds_st128(uint32_t off, bf16x8 v) { *reinterpret_cast<bf16x8*>(tic_smem + off) = v; }
TIC_DEV void lds_st64(uint32_t off, bf16x4 v) { *reinterpret_cast<bf16x4*>(tic_smem + off) = v; }
TIC_DEV void lds_stf(uint32_t off, float v) { *reinterpret_cast<float*>(tic_smem + off) = v; }
TIC_DEV void lds_stf4(uint32_t off, f32x4 v) { *reinterpret_cast<f32x4*>(tic_smem + off) = v; }
TIC_DEV void lds_addf(uint32_t off, float v) { atomicAdd(reinterpret_cast<float*>(tic_smem + off), v); }

// ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of block row q,
// columns 4p..4p+3 (8 bytes); lane i receives column i of the 4 rows (guide T10).
// EXEC must be all ones.
TIC_DEV bf16x4 lds_tr64(uint32_t off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(
            (__attribute__((address_space(3))) char*)tic_smem + off));
}

// The same read hidden from hipcc (inline asm): beside LDS-DMA in flight the compiler orders every LDS read it KNOWS of
// behind `s_waitcnt vmcnt(0)` (it sees buffer_load..lds as an LDS write that may alias), which drains the whole
// prefetch pipeline in every phase.  `addr` = LDS byte address in a VGPR (lds_base() + offset), `imm` = compile-time
// offset < 65536.  The caller waits with lds_wait(...) naming every destination before the first consumer (guide
// section 5.7 item 1, form (ii)) and orders the reads against the DMA by its own counted vmcnt + barrier.
TIC_DEV uint32_t lds_base() { return (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)tic_smem); }
TIC_DEV bf16x4 lds_tr64_hidden(uint32_t addr, uint32_t imm) {
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(imm));
    return r;
}
TIC_DEV bf16x8 lds_ld128_hidden(uint32_t addr, uint32_t imm) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(imm));
    return r;
}
// N = LDS operations that may still be outstanding (they return in order): lds_wait<8>(...) after 8 + 8 reads has the first 8 landed
template <int N = 0>
TIC_DEV void lds_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N = 0>
TIC_DEV void lds_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d, bf16x8& e, bf16x8& f, bf16x8& g, bf16x8& h) {
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}

// ---- buffer resources + LDS-DMA -----------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t tic_rsrc_t;
// bytes: hardware range check -- loads past it return 0, stores past it are dropped.
TIC_DEV tic_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
// 16 B per lane, global[voff + soff] -> LDS[lds_off + lane*16]; lds_off must be wave-uniform.
TIC_DEV void glds16(tic_rsrc_t r, uint32_t lds_off, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)tic_smem + lds_off),
        16, voff, soff, 0, 0);
}
// the same with the non-temporal cache policy (aux = 2): streams a workgroup reads once and nobody else re-reads (the attention
// tiles: -0.2 % step; NOT the GEMM operand panels, which the other tiles of the group re-read through L2: +1.4 .. 2.6 %)
TIC_DEV void glds16_nt(tic_rsrc_t r, uint32_t lds_off, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        r, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) char*)tic_smem + lds_off),
        16, voff, soff, 0, 2);
}
TIC_DEV u32x4 buf_ld128(tic_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
TIC_DEV u32x2 buf_ld64(tic_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}
TIC_DEV void buf_st128(tic_rsrc_t r, u32x4 v, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
TIC_DEV void buf_st128_nt(tic_rsrc_t r, u32x4 v, uint32_t voff, uint32_t soff) {   // non-temporal policy (aux = 2), as glds16_nt
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 2);
}
TIC_DEV void buf_st64(tic_rsrc_t r, u32x2 v, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// ---- waits / barriers ---------------------------------------------------------------------------
TIC_DEV void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> TIC_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
TIC_DEV void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
TIC_DEV void raw_barrier() { __builtin_amdgcn_s_barrier(); }
TIC_DEV void block_sync() { __syncthreads(); }
TIC_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
TIC_DEV void prio_hi() { __builtin_amdgcn_s_setprio(1); }
TIC_DEV void prio_lo() { __builtin_amdgcn_s_setprio(0); }

// ---- MFMA ---------------------------------------------------------------------------------------
// 16x16x32 bf16: lane l holds A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15];
//                D: col = l&15, row = 4(l>>4)+reg.
TIC_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// 32x32x16 bf16: lane l holds A[row l&31][k 8(l>>5)+j], B[k 8(l>>5)+j][col l&31];
//                D: col = l&31, row = (reg&3) + 8(reg>>2) + 4(l>>5).
TIC_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- cross-lane ---------------------------------------------------------------------------------
TIC_DEV float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }   // ds_bpermute_b32: LDS-path latency per step
// DPP forms (one VALU instruction per step, no LDS round trip): a butterfly of shfl_xor is a serial chain of ~120-cycle
// ds_bpermute round trips -- 32 of them per wave at the end of the attention backward cost 25 % of that kernel.
template <int CTRL, int ROW_MASK = 0xF>
TIC_DEV float dpp_mov(float v) {   // lanes the control does not reach read 0 (bound_ctrl)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
// sum over each 16-lane row, result in all 16 lanes: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
TIC_DEV float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// sum over the 64-lane wave, wave-uniform result: rows by DPP, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2-3,
// the total lands in lane 63
TIC_DEV float wave64_sum(float v) {
    v = row16_sum(v);
    v += dpp_mov<0x142, 0xA>(v);
    v += dpp_mov<0x143, 0xC>(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
TIC_DEV int lane_id() { return threadIdx.x & 63; }
// the lane index re-derived from the hardware (2 VALU instructions): used by code that runs AFTER a register-starved main loop, so
// that threadIdx-derived values need not stay live (hipcc spilled them to scratch rather than recomputing them)
TIC_DEV int lane_id_fresh() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
TIC_DEV int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
TIC_DEV uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

TIC_DEV void atomic_addf(float* p, float v) { atomicAdd(p, v); }
// Cross-workgroup hand-off inside one launch (the 2- / 4-way split-K form of the 256x256 NT kernel), the guide's counter-free
// flag form (cdna_hip_programming.md Guideline 16): the PRODUCER's waves have drained their plain stores (s_waitcnt vmcnt(0)) and
// met at a workgroup barrier; ONE lane then releases at agent scope (writes this XCD's dirty L2 lines back), waits for that, and
// stores the flag.  The CONSUMER polls the flag from ONE lane with relaxed agent-scope loads (bounded: a lost producer must not
// hang the device), acquires at agent scope (invalidates this CU's L1), waits for the invalidate; the workgroup barrier that
// follows orders every other wave's plain loads behind it.
TIC_DEV void flag_publish(unsigned* flag, unsigned value) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // always: hipcc can drop the fence's own wait (guide, compiler hazard)
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A poll that runs out (a producer that was never dispatched or died) must not pass silently: the consumer would add an unwritten or
// stale slab and store a normal-looking tile.  It reports through `err`, one word of HOST memory mapped into the device (system-scope
// store), which the library checks at the start and end of every later call (tic_after_launch: TIC_ELAUNCH, sticky).
TIC_DEV bool flag_wait(const unsigned* flag, unsigned value, unsigned* err) {
    bool ok = false;
    for (int spin = 0; spin < (1 << 22); ++spin) {   // ~ seconds at most, then give up and REPORT
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == value) {
            ok = true;
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (!ok && err) __hip_atomic_store(err, 0xDEAD0000u | (value & 0xFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return ok;
}
TIC_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32; exp2(-inf) = 0
TIC_DEV float fast_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
TIC_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }      // v_rcp_f32 (1 ulp), not the 10-instruction IEEE divide

#define TIC_TID ((int)threadIdx.x)
#define TIC_BID_X ((int)blockIdx.x)
#define TIC_BID_Y ((int)blockIdx.y)
#define TIC_BID_Z ((int)blockIdx.z)
#define TIC_NBLK_X ((int)gridDim.x)
#define TIC_NBLK_Y ((int)gridDim.y)
#define TIC_NTHR ((int)blockDim.x)

#define TIC_LAUNCH(kern, grid, block, lds, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, lds, (hipStream_t)(stream), __VA_ARGS__)
#endif  // !TIC_SIM

// ---- numerics shared by both builds -------------------------------------------------------------
TIC_DEV float bf2f(bf16_t u) {
    union { uint32_t i; float f; } x;
    x.i = ((uint32_t)u) << 16;
    return x.f;
}
// round-to-nearest-even via the compiler's cast (v_cvt_pk_bf16_f32 on gfx950: NaN stays NaN)
TIC_DEV bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
TIC_DEV float bfround(float f) { return bf2f(f2bf(f)); }
#ifdef TIC_SIM
TIC_DEV uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
#else
typedef __attribute__((ext_vector_type(2))) __bf16 tic_bf16x2_t;
TIC_DEV uint32_t pack2bf(float lo, float hi) {   // ONE v_cvt_pk_bf16_f32 (two scalar casts cost two converts and an SDWA or)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, tic_bf16x2_t));
}
#endif

// compile-time switches passed to generic lambdas (`[&](auto tag) { constexpr bool X = decltype(tag)::value; ... }`)
struct tic_false { static constexpr bool value = false; };
struct tic_true { static constexpr bool value = true; };
// a wave-uniform value the optimiser must treat as unknown (stays in an SGPR): keeps ONE scalar sweep offset from being
// strength-reduced into a vector induction variable per LDS read
TIC_DEV void opaque_s(uint32_t& v) {
#ifndef TIC_SIM
    asm volatile("" : "+s"(v));
#endif
}

// placed inside a wave-uniform `if` body: keeps the compiler from if-converting the branch into per-lane selects
TIC_DEV void keep_branch() {
#ifndef TIC_SIM
    asm volatile("");
#endif
}

// streaming accesses: data touched once per pass (the fp32 residual stream and its gradient); NT = non-temporal hint
template <bool NT> TIC_DEV f32x4 ld_f4(const float* p) {
#ifndef TIC_SIM
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#endif
    return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT> TIC_DEV void st_f4(float* p, f32x4 v) {
#ifndef TIC_SIM
    if (NT) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); return; }
#endif
    *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT> TIC_DEV void st_u2(bf16_t* p, u32x2 v) {
#ifndef TIC_SIM
    if (NT) { __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p)); return; }
#endif
    *reinterpret_cast<u32x2*>(p) = v;
}
template <bool NT> TIC_DEV u32x4 ld_u4(const bf16_t* p) {
#ifndef TIC_SIM
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#endif
    return *reinterpret_cast<const u32x4*>(p);
}
template <bool NT> TIC_DEV void st_u4(bf16_t* p, u32x4 v) {
#ifndef TIC_SIM
    if (NT) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); return; }
#endif
    *reinterpret_cast<u32x4*>(p) = v;
}

// exact-erf GELU (HF activations.py:83) and its derivative, with erfc evaluated by the
// Abramowitz-Stegun 7.1.26 rational form: erfc(z) = P(t) exp(-z^2), t = 1/(1 + 0.3275911 z), z >= 0,
// |abs error| <= 1.5e-7 -- three orders below the bf16 rounding applied to every value these feed.
// The small tail 1 + erf(z) (z < 0) is produced directly as P(t) exp(-z^2), without cancellation.
// ~17 VALU ops per element instead of libm erff's ~50: with 128 outputs per lane in the GEMM
// epilogue that is the difference between an epilogue longer than the MFMA main loop and one hidden in it.
struct GeluParts {
    float cdf;   // Phi(x) = 0.5 (1 + erf(x / sqrt 2))
    float e;     // exp(-x^2 / 2)
};
TIC_DEV GeluParts gelu_parts(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = fast_rcp(1.0f + 0.3275911f * z);
    const float e = fast_exp2(x * x * -0.72134752044448170f);   // exp(-x^2/2) = 2^(-x^2 * log2(e) / 2)
    float poly = 1.061405429f;
    poly = poly * t + -1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t + -0.284496736f;
    poly = poly * t + 0.254829592f;
    const float half_erfc = 0.5f * poly * t * e;                 // 0.5 erfc(|x| / sqrt 2)
    GeluParts r;
    r.cdf = (x < 0.f) ? half_erfc : 1.0f - half_erfc;
    r.e = e;
    return r;
}
TIC_DEV float gelu_erf(float x) { return x * gelu_parts(x).cdf; }
// gelu and gelu' of a PAIR of values from one shared evaluation, written on 2-vectors so the multiply-adds
// issue as v_pk_fma_f32 / v_pk_mul_f32 (two fp32 lanes per VALU slot): ~12 VALU slots + 2 transcendentals per element.
struct GeluPair {
    f32x2 g, dg;
};
TIC_DEV GeluPair gelu_pair(f32x2 x) {
    const f32x2 a = f32x2{fabsf(x[0]), fabsf(x[1])};
    const f32x2 den = a * 0.23164188f + 1.0f;                         // 1 + 0.3275911 |x| / sqrt 2
    const f32x2 t = f32x2{fast_rcp(den[0]), fast_rcp(den[1])};
    const f32x2 xe = x * x * -0.72134752044448170f;
    const f32x2 e = f32x2{fast_exp2(xe[0]), fast_exp2(xe[1])};      // exp(-x^2 / 2)
    f32x2 poly = t * 0.5307027145f + -0.7265760135f;                  // A&S 7.1.26 coefficients, halved
    poly = poly * t + 0.7107068705f;
    poly = poly * t + -0.142248368f;
    poly = poly * t + 0.127414796f;
    const f32x2 h = poly * t * e;                                     // 0.5 erfc(|x| / sqrt 2)
    const f32x2 up = 1.0f - h;
    const f32x2 cdf = f32x2{x[0] < 0.f ? h[0] : up[0], x[1] < 0.f ? h[1] : up[1]};
    GeluPair r;
    r.g = x * cdf;
    r.dg = x * 0.39894228040143268f * e + cdf;
    return r;
}
TIC_DEV float gelu_erf_grad(float x) {
    const GeluParts g = gelu_parts(x);
    return g.cdf + x * 0.39894228040143268f * g.e;
}
