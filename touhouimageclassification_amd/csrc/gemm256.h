// gemm256.h -- the deep-pipelined bf16 GEMM for the big token-matrix products (gfx950).
//
//   C[M,N] = A[M,K] . B[N,K]^T (+ the epilogues of gemm.h),  N % 256 == 0, K % 64 == 0, any M.
//
// 256x256 block tile, 64-deep K tile, 512 threads = 8 waves (2 along M x 4 along N), one block per CU
// (128 KiB LDS = 2 K-tile buffers x {A 256x64, B 256x64} bf16).  Each wave owns a 128x64 output as four
// 64x32 quadrants Q(i,j): rows i*128 + wr*64 + [0,64), columns j*128 + wc*32 + [0,32) -- interleaved so
// that quadrant (i,j) needs only A half-tile i and B half-tile j (a half-tile = 128 rows x 64 k = 16 KiB =
// 2 LDS-DMA pieces per thread).  One K tile = 4 phases in the order Q00, Q01, Q11, Q10; a phase is
//     load segment : ds_read the newly needed operand sub-tile (A0+B0 | B1 | A1 | B0), issue the LDS-DMA of
//                    ONE half-tile of the NEXT K tile (A0 | B0 | B1 | A1), counted s_waitcnt vmcnt
//     s_barrier
//     MFMA segment : 16 x v_mfma_f32_16x16x32_bf16 (64x32 quadrant, K = 64)
//     s_barrier
// The two wave groups (wr = 0 / 1 -- the two waves that share each SIMD) run ONE barrier apart, so one
// group's MFMA segment overlaps the other's load segment (matrix beside memory on every SIMD).
// LDS-DMA stays in flight across barriers (raw s_barrier, never __syncthreads).  Prefetch is 5-6 phases
// deep inside the two K-tile buffers: the B0 fragments stay in registers for the whole tile, so every
// half-tile slot is read in exactly one phase and is re-filled as soon as that read has retired:
//     tile t phase 0: issue B1(t+1)   phase 1: A1(t+1)   phase 2: A0(t+2)   phase 3: B0(t+2)
// i.e. A0/B0 of tile t+2 go into the buffer tile t is still being computed from.  vmcnt(8) at phases
// 3, 0, 1 leaves four half-tiles (64 KiB per CU) in flight.  K tiles past the end are issued with an
// out-of-range offset (the hardware range check turns them into zero fills with no memory traffic) so
// the steady-state body and its vmcnt counts are the same for every tile.
// Hazard bookkeeping (interval = time between consecutive barriers; group 0 runs its load segment of
// global phase P in interval 2P, group 1 in 2P+1):
//   RAW  a half-tile read in phase P is waited for (by every wave that issued a piece of it) in the load
//        segment of phase P-1, i.e. no later than interval 2P-1, and read from interval 2P on;
//   WAR  a slot is re-filled >= 2 phases after its only read (A0: read ph 0 -> filled ph 2; B0: ph 0 -> ph 3;
//        B1: ph 1 -> next ph 0; A1: ph 2 -> next ph 1).
#pragma once
#include "gemm.h"

#define G256_BUF_BYTES 65536u   // A 32 KiB + B 32 KiB
#define G256_LDS_BYTES (2u * G256_BUF_BYTES)
#define G256_NT_LDS_BYTES (G256_LDS_BYTES + 8192u)   // + [8 waves][256] fp32 column-sum partials of the staged epilogue

TIC_DEV void g256_barrier() {
#ifndef TIC_SIM
    asm volatile("" ::: "memory");
#endif
    raw_barrier();
#ifndef TIC_SIM
    asm volatile("" ::: "memory");
#endif
}

// ---- staged epilogue ------------------------------------------------------------------------------------------
// Every epilogue starts from u = bf16(acc + bias).  The MFMA accumulator layout gives a lane 4 consecutive columns of
// 16 different rows per store -- 32-byte pieces of 16 rows, and extra operands (residual / saved derivative) fetched
// in the same shape, one latency-bound row at a time.  Here the block instead parks the bf16 tile in the LDS the main
// loop has just released (256 x 256 x 2 B = the whole 128 KiB), which frees the 128 accumulator registers, and walks
// it again row-contiguously: a wave touches whole 512 B (bf16) / 1 KiB (fp32) row segments per instruction, and the
// extra operands of PF rows are in flight while the previous PF rows are computed and stored.
// LDS layout: row stride 512 B; the 8-byte chunk index c8 (0..63) is XORed with (row & 15) << 2, so the 16 rows a
// staging write covers land in 16 different 32-byte groups and a row read back stays one permuted 512 B line.
TIC_DEV uint32_t g256_stage_off(int row, int c8) { return (uint32_t)row * 512u + (uint32_t)((c8 ^ ((row & 15) << 2)) * 8); }
TIC_DEV float bf_lo(uint32_t u) {
    union { uint32_t i; float f; } x;
    x.i = u << 16;
    return x.f;
}
TIC_DEV float bf_hi(uint32_t u) {
    union { uint32_t i; float f; } x;
    x.i = u & 0xffff0000u;
    return x.f;
}

// 8 columns per thread: the bf16-output epilogues.  c16 = tid & 31, rows (tid >> 5) + 16 k.
template <int EPI>
TIC_DEV void g256_finish_bf16(const GemmNtParams& p, int m0, int n0) {
    constexpr bool HAS_AUX = (EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX);
    constexpr bool HAS_COLSUM = (EPI == TIC_EPI_BF16 || EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX);
    constexpr int PF = 4;
    const int tid = TIC_TID, c16 = tid & 31, rsub = tid >> 5;
    const int n = n0 + c16 * 8;
    const uint32_t lds0 = (uint32_t)rsub * 512u + (uint32_t)(((c16 * 2) ^ (rsub << 2)) * 8);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 aux[2][PF];
    auto fetch = [&](int b) {
        if (HAS_AUX) {
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int m = m0 + (b * PF + i) * 16 + rsub;
                aux[b & 1][i] = (m < p.M) ? *reinterpret_cast<const u32x4*>(p.aux + (size_t)m * p.N + n) : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    fetch(0);
#pragma unroll
    for (int b = 0; b < 16 / PF; ++b) {
        if (b + 1 < 16 / PF) fetch(b + 1);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int k = b * PF + i;
            const int m = m0 + k * 16 + rsub;
            const u32x4 u = __builtin_bit_cast(u32x4, lds_ld128(lds0 + (uint32_t)k * 8192u));
            if (m >= p.M) continue;
            const size_t o = (size_t)m * p.N + n;
            if (EPI == TIC_EPI_BF16) {
                *reinterpret_cast<u32x4*>(p.out + o) = u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[2 * j] += bf_lo(u[j]);
                    cs[2 * j + 1] += bf_hi(u[j]);
                }
            } else if (EPI == TIC_EPI_GELU) {
                u32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = pack2bf(gelu_erf(bf_lo(u[j])), gelu_erf(bf_hi(u[j])));
                *reinterpret_cast<u32x4*>(p.out + o) = u;
                *reinterpret_cast<u32x4*>(p.out2 + o) = g;
            } else if (EPI == TIC_EPI_GELU_DG) {
                u32x4 g, dg;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const GeluPair r = gelu_pair(f32x2{bf_lo(u[j]), bf_hi(u[j])});
                    g[j] = pack2bf(r.g[0], r.g[1]);
                    dg[j] = pack2bf(r.dg[0], r.dg[1]);
                }
                *reinterpret_cast<u32x4*>(p.out + o) = dg;
                *reinterpret_cast<u32x4*>(p.out2 + o) = g;
            } else {   // DGELU / MULAUX
                const u32x4 a = aux[b & 1][i];
                u32x4 d;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float f0 = (EPI == TIC_EPI_DGELU) ? gelu_erf_grad(bf_lo(a[j])) : bf_lo(a[j]);
                    const float f1 = (EPI == TIC_EPI_DGELU) ? gelu_erf_grad(bf_hi(a[j])) : bf_hi(a[j]);
                    const float d0 = bf_lo(u[j]) * f0, d1 = bf_hi(u[j]) * f1;
                    d[j] = pack2bf(d0, d1);
                    cs[2 * j] += d0;
                    cs[2 * j + 1] += d1;
                }
                *reinterpret_cast<u32x4*>(p.out + o) = d;
            }
        }
    }
    // fused bias gradient: lanes l, l^32 hold the other rows of the same 8 columns; the 8 waves meet in LDS and 256
    // threads add one contiguous fp32 row to the global vector
    if (HAS_COLSUM && p.colsum) {   // kernel-argument condition: block-uniform
        const int l = tid & 63, w = tid >> 6;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cs[j] += shfl_xor(cs[j], 32);
            if (l < 32) lds_stf(G256_LDS_BYTES + (uint32_t)(w * 256 + c16 * 8 + j) * 4u, cs[j]);
        }
        block_sync();
        if (tid < 256) {
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) t += lds_ldf(G256_LDS_BYTES + (uint32_t)(ww * 256 + tid) * 4u);
            atomic_addf(p.colsum + n0 + tid, t);
        }
    }
}

// 4 columns per thread: the fp32-output epilogues (RESID, PATCH).  c8 = tid & 63, rows (tid >> 6) + 8 k.
template <int EPI>
TIC_DEV void g256_finish_f32(const GemmNtParams& p, int m0, int n0) {
    constexpr int PF = 4;
    const int tid = TIC_TID, c8 = tid & 63, rsub = tid >> 6;
    const int n = n0 + c8 * 4;
    f32x4 ex[2][PF];
    auto src_of = [&](int m) -> const float* {
        if (EPI == TIC_EPI_RESID) return p.resid + (size_t)m * p.N + n;
        return p.rowtab + (size_t)(1 + m % p.patches) * p.N + n;
    };
    auto fetch = [&](int b) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int m = m0 + (b * PF + i) * 8 + rsub;
            ex[b & 1][i] = (m < p.M) ? *reinterpret_cast<const f32x4*>(src_of(m)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
#pragma unroll
    for (int b = 0; b < 32 / PF; ++b) {
        if (b + 1 < 32 / PF) fetch(b + 1);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int k = b * PF + i;
            const int row = k * 8 + rsub, m = m0 + row;
            const u32x2 u = __builtin_bit_cast(u32x2, lds_ld64(g256_stage_off(row, c8)));
            if (m >= p.M) continue;
            const f32x4 e = ex[b & 1][i];
            const f32x4 y = f32x4{bf_lo(u[0]) + e[0], bf_hi(u[0]) + e[1], bf_lo(u[1]) + e[2], bf_hi(u[1]) + e[3]};
            size_t orow = (size_t)m;
            if (EPI == TIC_EPI_PATCH) {
                const int img = m / p.patches;
                orow = (size_t)img * (p.patches + 1) + 1 + (m - img * p.patches);
            }
            *reinterpret_cast<f32x4*>(p.out_f32 + orow * p.N + n) = y;
        }
    }
}

template <int EPI>
__global__ void __launch_bounds__(512, 2) gemm_nt256_kernel(GemmNtParams p) {
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wr = w >> 2, wc = w & 3;
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / 256;
    int tm, tn;
    tile_coords(TIC_BID_X, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const tic_rsrc_t ra = make_rsrc(p.A, (uint32_t)((size_t)p.M * p.K * 2));
    const tic_rsrc_t rb = make_rsrc(p.B, (uint32_t)((size_t)p.N * p.K * 2));

    // ---- LDS-DMA: half-tile h of operand X = 16 one-KiB pieces (8 rows each); this wave moves pieces 2w, 2w+1
    const uint32_t slot_log = (uint32_t)(l & 7) ^ ((((uint32_t)l >> 4) & 3u) << 1);
    uint32_t voa[2][2], vob[2][2];   // [half][piece]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = h * 128 + (2 * w + j) * 8 + (l >> 3);
            voa[h][j] = (uint32_t)(((size_t)(m0 + r) * p.K + slot_log * 8) * 2);
            vob[h][j] = (uint32_t)(((size_t)(n0 + r) * p.K + slot_log * 8) * 2);
        }
    // which: 0 = A0, 1 = B0, 2 = B1, 3 = A1.  K tiles >= nk become zero fills (see header).
    const int nk = p.K / 64;
    auto issue = [&](int buf, int kt, int which) {
        const uint32_t soff = (uint32_t)kt * 128u;
        const bool isA = (which == 0 || which == 3);
        const int h = (which >= 2) ? 1 : 0;
        const bool live = kt < nk;
        const uint32_t base = (uint32_t)buf * G256_BUF_BYTES + (isA ? 0u : 32768u) + (uint32_t)h * 16384u + (uint32_t)(2 * w) * 1024u;
        if (isA) {
            glds16(ra, base, live ? voa[h][0] : 0xFFFFFFF0u, live ? soff : 0u);
            glds16(ra, base + 1024u, live ? voa[h][1] : 0xFFFFFFF0u, live ? soff : 0u);
        } else {
            glds16(rb, base, live ? vob[h][0] : 0xFFFFFFF0u, live ? soff : 0u);
            glds16(rb, base + 1024u, live ? vob[h][1] : 0xFFFFFFF0u, live ? soff : 0u);
        }
    };

    // ---- fragment offsets inside a buffer ----------------------------------------------------------------
    const uint32_t sw = (((uint32_t)(l & 15) >> 1) & 3u) << 1;
    uint32_t fo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = (uint32_t)(l & 15) * 128u + ((((uint32_t)ks * 4 + ((uint32_t)l >> 4)) ^ sw) * 16u);
    const uint32_t a_row0 = (uint32_t)wr * 64, b_row0 = (uint32_t)wc * 32;

    f32x4 acc[2][2][4][2];   // [i][j][mt][nt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];   // [mt][ks], [nt][ks]; B0 fragments live for the whole K tile

    auto load_a = [&](uint32_t bufb, int i) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[mt][ks] = lds_ld128(bufb + ((uint32_t)i * 128 + a_row0 + (uint32_t)mt * 16) * 128u + fo[ks]);
    };
    auto load_b = [&](uint32_t bufb, int j, bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[nt][ks] = lds_ld128(bufb + 32768u + ((uint32_t)j * 128 + b_row0 + (uint32_t)nt * 16) * 128u + fo[ks]);
    };
    auto mma = [&](int i, int j, const bf16x8 (&fb)[2][2]) {
        prio_hi();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = mfma16(fb[nt][ks], fa[mt][ks], acc[i][j][mt][nt]);
        prio_lo();
    };

    // prologue: K tile 0 completely + A0, B0 of tile 1; wait for A0(0), B0(0); group 1 then falls one barrier behind
    issue(0, 0, 0);
    issue(0, 0, 1);
    issue(0, 0, 2);
    issue(0, 0, 3);
    issue(1, 1, 0);
    issue(1, 1, 1);
    wait_vmcnt<8>();
    g256_barrier();
    if (wr == 1) g256_barrier();

#pragma nounroll
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const uint32_t bufb = (uint32_t)cur * G256_BUF_BYTES;
        // ---- phase 0: Q00 (A0, B0 landed: waited for in phase 3 of the previous tile / prologue)
        load_a(bufb, 0);
        load_b(bufb, 0, fb0);
        issue(cur ^ 1, kt + 1, 2);   // B1(t+1)
        wait_vmcnt<8>();             // B1(t) has landed
        g256_barrier();
        mma(0, 0, fb0);
        g256_barrier();
        // ---- phase 1: Q01
        load_b(bufb, 1, fb1);
        issue(cur ^ 1, kt + 1, 3);   // A1(t+1)
        wait_vmcnt<8>();             // A1(t) has landed
        g256_barrier();
        mma(0, 1, fb1);
        g256_barrier();
        // ---- phase 2: Q11
        load_a(bufb, 1);
        issue(cur, kt + 2, 0);       // A0(t+2) into THIS buffer: its A0 slot was last read in phase 0
        g256_barrier();
        mma(1, 1, fb1);
        g256_barrier();
        // ---- phase 3: Q10 (B0 fragments still in registers: no LDS read)
        issue(cur, kt + 2, 1);       // B0(t+2)
        wait_vmcnt<8>();             // A0(t+1), B0(t+1) have landed
        g256_barrier();
        mma(1, 0, fb0);
        g256_barrier();
    }
    wait_vmcnt0();   // drain the zero fills issued for the tiles past the end before LDS is reused
    if (wr == 0) g256_barrier();   // re-balance the stagger
    g256_barrier();                // every wave's LDS reads and DMA writes have retired: the tile buffers are free

    // ---- stage u = bf16(acc + bias) into LDS: rows r = i*4 + mt, column groups g = j*2 + nt
    constexpr bool HAS_BIAS = (EPI != TIC_EPI_DGELU && EPI != TIC_EPI_MULAUX);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = (g >> 1) * 128 + wc * 32 + (g & 1) * 16 + 4 * (l >> 4);
        const f32x4 bias = (HAS_BIAS && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n0 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (l & 15);
            const f32x4 v = acc[r >> 2][g >> 1][r & 3][g & 1] + bias;
            lds_st64(g256_stage_off(row, col >> 2), __builtin_bit_cast(bf16x4, u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])}));
        }
    }
    block_sync();
    if (EPI == TIC_EPI_RESID || EPI == TIC_EPI_PATCH)
        g256_finish_f32<EPI>(p, m0, n0);
    else
        g256_finish_bf16<EPI>(p, m0, n0);
}
