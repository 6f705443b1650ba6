// sim_runtime.h -- CPU wave-lockstep simulator for the kernels in touhouimageclassification_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY (never shipped, never loaded by the product package).
// A workgroup runs as `blockDim` cooperative fibers (ucontext) on one OS thread; every
// wave-collective primitive of tic_prims.h (MFMA, ds_read_b64_tr_b16, shuffles, LDS-DMA)
// is a rendezvous of the wave's 64 lanes, executed by the last arriver with the lane maps
// documented in /opt/skills/guides/cdna_hip_programming.md section 3 / T10.  Barriers are block
// rendezvous.  LDS and buffer-resource accesses are bounds-checked (buffer loads past
// num_records return 0, as the hardware range check does).  A collective reached by only
// part of a wave, or a barrier missed by a lane, is reported as a deadlock instead of hanging.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

#define TIC_DEV static inline __attribute__((always_inline))
#define TIC_KERNEL(name) extern "C" void name
#define __global__
#define __launch_bounds__(...)
#define __expf expf
#define __logf logf

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace sim {

struct Wave {
    int arrived = 0;
    unsigned gen = 0;
    int nlanes = 64;
    alignas(16) char in[64][128];
    alignas(16) char out[64][64];
};

struct Block {
    int nthreads = 0;
    dim3 bid, grid;
    std::vector<char> lds;
    std::vector<Wave> waves;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
};

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;   // from a process-lifetime pool (no per-block allocation / zero-fill)
    int tid = 0;
    bool done = false;
};
static const size_t kStackBytes = 256 * 1024;
inline char* pooled_stack(int tid) {
    static std::vector<char*> pool;
    while ((int)pool.size() <= tid) pool.push_back((char*)malloc(kStackBytes));
    return pool[tid];
}

struct State {
    Block* blk = nullptr;
    Fiber* cur = nullptr;
    ucontext_t sched;
    unsigned long progress = 0;
    std::function<void()> body;
};
inline State& st() {
    static State s;
    return s;
}

inline void yield() { swapcontext(&st().cur->ctx, &st().sched); }

[[noreturn]] inline void die(const char* msg) {
    fprintf(stderr, "[tic-sim] FATAL: %s (block %u,%u tid %d)\n", msg, st().blk ? st().blk->bid.x : 0,
            st().blk ? st().blk->bid.y : 0, st().cur ? st().cur->tid : -1);
    abort();
}

inline void fiber_entry() {
    st().body();
    st().cur->done = true;
    st().progress++;
    swapcontext(&st().cur->ctx, &st().sched);
}

inline void run_block(Block& b) {
    State& s = st();
    s.blk = &b;
    std::vector<Fiber> fibers(b.nthreads);
    for (int t = 0; t < b.nthreads; ++t) {
        Fiber& f = fibers[t];
        f.tid = t;
        f.stack = pooled_stack(t);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = &s.sched;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int remaining = b.nthreads;
    while (remaining > 0) {
        unsigned long before = s.progress;
        remaining = 0;
        for (int t = 0; t < b.nthreads; ++t) {
            if (fibers[t].done) continue;
            s.cur = &fibers[t];
            swapcontext(&s.sched, &fibers[t].ctx);
            if (!fibers[t].done) remaining++;
        }
        if (remaining > 0 && s.progress == before) {
            s.cur = nullptr;
            die("deadlock: a wave collective or barrier was not reached by all of its lanes");
        }
    }
    s.cur = nullptr;
    s.blk = nullptr;
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t lds_bytes, F&& body) {
    if (block.x % 64 != 0) die("block size must be a multiple of 64");
    st().body = body;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                Block b;
                b.nthreads = block.x;
                b.bid = dim3(x, y, z);
                b.grid = grid;
                b.lds.assign(lds_bytes, (char)0xCD);   // poison: uninitialised LDS reads show up as garbage/NaN
                b.waves.resize(block.x / 64);
                run_block(b);
            }
}

inline Wave& cur_wave() { return st().blk->waves[st().cur->tid >> 6]; }
inline int cur_lane() { return st().cur->tid & 63; }

// rendezvous of the 64 lanes of a wave; fn(in[64], out[64]) runs once, in the last arriver
template <class In, class Out, class Fn>
inline Out wave_coll(const In& in, Fn&& fn) {
    static_assert(sizeof(In) <= 128 && sizeof(Out) <= 64, "collective slot too small");
    Wave& w = cur_wave();
    const int l = cur_lane();
    memcpy(w.in[l], &in, sizeof(In));
    if (++w.arrived == w.nlanes) {
        In ins[64];
        Out outs[64];
        for (int i = 0; i < 64; ++i) memcpy(&ins[i], w.in[i], sizeof(In));
        fn(ins, outs);
        for (int i = 0; i < 64; ++i) memcpy(w.out[i], &outs[i], sizeof(Out));
        w.arrived = 0;
        w.gen++;
        st().progress++;
    } else {
        const unsigned g = w.gen;
        while (w.gen == g) yield();
    }
    Out o;
    memcpy(&o, w.out[l], sizeof(Out));
    return o;
}

inline void block_barrier() {
    Block& b = *st().blk;
    if (++b.bar_arrived == b.nthreads) {
        b.bar_arrived = 0;
        b.bar_gen++;
        st().progress++;
    } else {
        const unsigned g = b.bar_gen;
        while (b.bar_gen == g) yield();
    }
}

inline char* lds_ptr(uint32_t off, uint32_t bytes, uint32_t align) {
    Block& b = *st().blk;
    if ((size_t)off + bytes > b.lds.size()) die("LDS access out of bounds");
    if (off % align) die("misaligned LDS access (guide G17)");
    return b.lds.data() + off;
}

}  // namespace sim

// ---- LDS ------------------------------------------------------------------------------------------
#define SIM_LDS_LD(T, off, al) ({ T v_; memcpy(&v_, sim::lds_ptr((off), sizeof(T), (al)), sizeof(T)); v_; })
TIC_DEV bf16x8 lds_ld128(uint32_t off) { return SIM_LDS_LD(bf16x8, off, 16); }
TIC_DEV bf16x4 lds_ld64(uint32_t off) { return SIM_LDS_LD(bf16x4, off, 8); }
TIC_DEV float lds_ldf(uint32_t off) { return SIM_LDS_LD(float, off, 4); }
TIC_DEV f32x4 lds_ldf4(uint32_t off) { return SIM_LDS_LD(f32x4, off, 16); }
TIC_DEV void lds_st128(uint32_t off, bf16x8 v) { memcpy(sim::lds_ptr(off, 16, 16), &v, 16); }
TIC_DEV void lds_st64(uint32_t off, bf16x4 v) { memcpy(sim::lds_ptr(off, 8, 8), &v, 8); }
TIC_DEV void lds_stf(uint32_t off, float v) { memcpy(sim::lds_ptr(off, 4, 4), &v, 4); }
TIC_DEV void lds_stf4(uint32_t off, f32x4 v) { memcpy(sim::lds_ptr(off, 16, 16), &v, 16); }
TIC_DEV void lds_addf(uint32_t off, float v) {
    float* p = reinterpret_cast<float*>(sim::lds_ptr(off, 4, 4));
    *p += v;
}

// ds_read_b64_tr_b16 (guide T10): per 16-lane group, lane 4q+p supplies the address of block
// row q, columns 4p..4p+3; lane i receives column i of the 4 rows, row q in its element q.
TIC_DEV bf16x4 lds_tr64(uint32_t off) {
    struct In { uint32_t off; };
    return sim::wave_coll<In, bf16x4>(In{off}, [](const In* in, bf16x4* out) {
        for (int g = 0; g < 4; ++g) {
            short blockv[4][16];
            for (int q = 0; q < 4; ++q)
                for (int p = 0; p < 4; ++p) {
                    const uint32_t a = in[16 * g + 4 * q + p].off;
                    memcpy(&blockv[q][4 * p], sim::lds_ptr(a, 8, 8), 8);
                }
            for (int i = 0; i < 16; ++i)
                for (int q = 0; q < 4; ++q) out[16 * g + i][q] = blockv[q][i];
        }
    });
}

// ---- buffer resources ---------------------------------------------------------------------------------
struct tic_rsrc_t {
    const char* base;
    uint32_t bytes;
};
TIC_DEV tic_rsrc_t make_rsrc(const void* base, uint32_t bytes) { return tic_rsrc_t{(const char*)base, bytes}; }
// Hardware range check (raw buffer, stride 0): only the VGPR offset is checked against
// num_records; the SGPR offset is added afterwards and is EXCLUDED from the check (LLVM
// IntrinsicsAMDGPU.td, raw.buffer.load: "soffset ... excluded from bounds checking").
TIC_DEV void sim_buf_read(tic_rsrc_t r, uint32_t voff, uint32_t soff, void* dst, uint32_t n) {
    if ((uint64_t)voff + n <= r.bytes) memcpy(dst, r.base + voff + soff, n);
    else memset(dst, 0, n);   // out-of-range loads return 0
}
TIC_DEV void glds16(tic_rsrc_t r, uint32_t lds_off, uint32_t voff, uint32_t soff) {
    struct In { uint32_t lds_off; };
    // the LDS destination base must be wave-uniform (M0): check it, then do this lane's copy
    sim::wave_coll<In, int>(In{lds_off}, [](const In* in, int* out) {
        for (int i = 1; i < 64; ++i)
            if (in[i].lds_off != in[0].lds_off) sim::die("glds16: LDS destination base is not wave-uniform");
        for (int i = 0; i < 64; ++i) out[i] = 0;
    });
    char tmp[16];
    sim_buf_read(r, voff, soff, tmp, 16);
    memcpy(sim::lds_ptr(lds_off + 16 * sim::cur_lane(), 16, 16), tmp, 16);
}
TIC_DEV u32x4 buf_ld128(tic_rsrc_t r, uint32_t voff, uint32_t soff) {
    u32x4 v;
    sim_buf_read(r, voff, soff, &v, 16);
    return v;
}
TIC_DEV u32x2 buf_ld64(tic_rsrc_t r, uint32_t voff, uint32_t soff) {
    u32x2 v;
    sim_buf_read(r, voff, soff, &v, 8);
    return v;
}
TIC_DEV void buf_st128(tic_rsrc_t r, u32x4 v, uint32_t voff, uint32_t soff) {
    if ((uint64_t)voff + 16 <= r.bytes) memcpy(const_cast<char*>(r.base) + voff + soff, &v, 16);
}
TIC_DEV void buf_st64(tic_rsrc_t r, u32x2 v, uint32_t voff, uint32_t soff) {
    if ((uint64_t)voff + 8 <= r.bytes) memcpy(const_cast<char*>(r.base) + voff + soff, &v, 8);
}

// ---- waits / barriers -----------------------------------------------------------------------------------
TIC_DEV void wait_vmcnt0() {}
template <int N> TIC_DEV void wait_vmcnt() {}
TIC_DEV void wait_lgkmcnt0() {}
TIC_DEV void raw_barrier() { sim::block_barrier(); }
TIC_DEV void block_sync() { sim::block_barrier(); }
TIC_DEV void sched_fence() {}
TIC_DEV void prio_hi() {}
TIC_DEV void prio_lo() {}

// ---- MFMA (lane maps: guide section 3) ----------------------------------------------------------------
TIC_DEV float sim_bf(short s) {
    union { uint32_t i; float f; } x;
    x.i = ((uint32_t)(uint16_t)s) << 16;
    return x.f;
}
TIC_DEV f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    struct In { bf16x8 a, b; f32x4 c; };
    return sim::wave_coll<In, f32x4>(In{a, b, c}, [](const In* in, f32x4* out) {
        float A[16][32], B[32][16];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                A[l & 15][8 * (l >> 4) + j] = sim_bf(in[l].a[j]);
                B[8 * (l >> 4) + j][l & 15] = sim_bf(in[l].b[j]);
            }
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                float acc = in[l].c[r];
                for (int k = 0; k < 32; ++k) acc += A[row][k] * B[k][col];
                out[l][r] = acc;
            }
    });
}
TIC_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    struct In { bf16x8 a, b; f32x16 c; };
    return sim::wave_coll<In, f32x16>(In{a, b, c}, [](const In* in, f32x16* out) {
        float A[32][16], B[16][32];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                A[l & 31][8 * (l >> 5) + j] = sim_bf(in[l].a[j]);
                B[8 * (l >> 5) + j][l & 31] = sim_bf(in[l].b[j]);
            }
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                float acc = in[l].c[r];
                for (int k = 0; k < 16; ++k) acc += A[row][k] * B[k][col];
                out[l][r] = acc;
            }
    });
}

// ---- cross-lane -----------------------------------------------------------------------------------------
TIC_DEV float shfl_xor(float v, int mask) {
    struct In { float v; int mask; };
    return sim::wave_coll<In, float>(In{v, mask}, [](const In* in, float* out) {
        for (int l = 0; l < 64; ++l) out[l] = in[(l ^ in[l].mask) & 63].v;
    });
}
TIC_DEV int lane_id() { return sim::cur_lane(); }
TIC_DEV int lane_id_fresh() { return lane_id(); }
TIC_DEV int wave_id() { return sim::st().cur->tid >> 6; }
TIC_DEV uint32_t uniform(uint32_t v) { return v; }
TIC_DEV void atomic_addf(float* p, float v) { *p += v; }
// workgroups run one after the other here (producers have the lower block indices): the hand-off is a plain store / compare
TIC_DEV void flag_publish(unsigned* flag, unsigned value) { *flag = value; }
// (a flag that does not hold the launch's value is what the GPU's bounded poll would time out on: reported through `err` as there)
TIC_DEV bool flag_wait(const unsigned* flag, unsigned value, unsigned* err) {
    if (*flag == value) return true;
    if (!err) sim::die("flag_wait: the producer workgroup has not run (block order)");
    *err = 0xDEAD0000u | (value & 0xFFFFu);
    return false;
}
TIC_DEV float fast_exp2(float x) { return exp2f(x); }
TIC_DEV float fast_rcp(float x) { return 1.0f / x; }
TIC_DEV void glds16_nt(tic_rsrc_t r, uint32_t lds_off, uint32_t voff, uint32_t soff) { glds16(r, lds_off, voff, soff); }
TIC_DEV void buf_st128_nt(tic_rsrc_t r, u32x4 v, uint32_t voff, uint32_t soff) { buf_st128(r, v, voff, soff); }
TIC_DEV float row16_sum(float v) {
    for (int m = 1; m < 16; m <<= 1) v += shfl_xor(v, m);
    return v;
}
TIC_DEV float wave64_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
    return v;
}
TIC_DEV uint32_t lds_base() { return 0u; }
TIC_DEV bf16x4 lds_tr64_hidden(uint32_t addr, uint32_t imm) { return lds_tr64(addr + imm); }
TIC_DEV bf16x8 lds_ld128_hidden(uint32_t addr, uint32_t imm) { return lds_ld128(addr + imm); }
template <int N = 0> TIC_DEV void lds_wait(bf16x8&, bf16x8&, bf16x8&, bf16x8&) {}
template <int N = 0> TIC_DEV void lds_wait(bf16x8&, bf16x8&, bf16x8&, bf16x8&, bf16x8&, bf16x8&, bf16x8&, bf16x8&) {}
TIC_DEV float fast_log2(float x) { return log2f(x); }

#define TIC_TID (sim::st().cur->tid)
#define TIC_BID_X ((int)sim::st().blk->bid.x)
#define TIC_BID_Y ((int)sim::st().blk->bid.y)
#define TIC_BID_Z ((int)sim::st().blk->bid.z)
#define TIC_NBLK_X ((int)sim::st().blk->grid.x)
#define TIC_NBLK_Y ((int)sim::st().blk->grid.y)
#define TIC_NTHR (sim::st().blk->nthreads)

#define TIC_LAUNCH(kern, grid, block, lds, stream, ...) \
    sim::launch(dim3(grid), dim3(block), (lds), [=]() { kern(__VA_ARGS__); })
