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

TIC_DEV void g256_barrier() {
#ifndef TIC_SIM
    asm volatile("" ::: "memory");
#endif
    raw_barrier();
#ifndef TIC_SIM
    asm volatile("" ::: "memory");
#endif
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
    wait_vmcnt0();   // drain the zero fills issued for the tiles past the end before LDS is released
    if (wr == 0) g256_barrier();   // re-balance the stagger

    // rows r = i*4 + mt, column groups g = j*2 + nt
    gemm_epilogue<EPI, 8, 4>(
        p, [&](int r) { return m0 + (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (l & 15); },
        [&](int g) { return n0 + (g >> 1) * 128 + wc * 32 + (g & 1) * 16 + 4 * (l >> 4); },
        [&](int r, int g) { return acc[r >> 2][g >> 1][r & 3][g & 1]; });
}
