// gemm_tn256_w4.h -- EXPERIMENT, measurement (-DTIC_MEASURE) and simulator builds only: included from gemm_tn256.h after
// TN256_STREAMK_BODY; the product library does not contain it.  DESIGN.md 4c: 685 vs 1 231 TFLOP/s for the shipped 8-wave kernel.
#pragma once

// EXPERIMENT (VERDICT r2 task 4; measurement + simulator builds only): the same 256 x 256 tile segment on FOUR waves, one per SIMD, each
// owning a 64 x 64 block of every 128 x 128 quadrant (2 x 2 waves) = 256 accumulator registers, up to 512 registers per lane.  Per 64-row
// step the workgroup reads 4 x (16 + 16) fragments = 128 KiB from LDS instead of 8 x (16 + 8) = 192 KiB; the price: every wave issues 4
// LDS-DMA pieces per phase itself (no partner wave on its SIMD to issue MFMAs meanwhile) and runs alone through every barrier.
// 16x16x32 MFMAs, the LDS image, swizzle and slot schedule of tn256_tile_segment16; no wave-group stagger (nobody shares a SIMD).
template <int ATOMIC>
TIC_DEV void tn256_tile_segment16_w4(const bf16_t* Ap, const bf16_t* Bp, float* Cp, int N, int K, int M, int n0, int k0, int step0, int step1) {
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wr = w >> 1, wc = w & 1;
    const int row_end = (step1 * 64 < M) ? step1 * 64 : M;
    const tic_rsrc_t ra = make_rsrc(Ap, (uint32_t)((size_t)row_end * N * 2));
    const tic_rsrc_t rb = make_rsrc(Bp, (uint32_t)((size_t)row_end * K * 2));
    // LDS-DMA: half-tile = 16 pieces of 4 rows; this wave moves pieces 4w .. 4w+3 (rows 16w + 4j + rr: (row >> 3) & 1 == (j >> 1) & 1)
    const uint32_t rr = (uint32_t)l >> 4;
    uint32_t voa[2][4], vob[2][4];   // [half][piece], running
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ch_log = ((uint32_t)l & 15u) ^ (rr << 2) ^ ((((uint32_t)j >> 1) & 1u) << 1);
            const uint32_t row = (uint32_t)step0 * 64u + (uint32_t)(4 * w + j) * 4 + rr;
            voa[h][j] = (uint32_t)(((size_t)row * N + n0 + h * 128 + ch_log * 8) * 2);
            vob[h][j] = (uint32_t)(((size_t)row * K + k0 + h * 128 + ch_log * 8) * 2);
        }
    const uint32_t stepA = (uint32_t)N * 128u, stepB = (uint32_t)K * 128u;   // 64 rows in bytes
    auto issue = [&](int buf, int which) {   // which: 0 = A0, 1 = B0, 2 = B1, 3 = A1
        const bool isA = (which == 0 || which == 3);
        const int h = (which >= 2) ? 1 : 0;
        const uint32_t base = (uint32_t)buf * G256_BUF_BYTES + (isA ? 0u : 32768u) + (uint32_t)h * 16384u + (uint32_t)(4 * w) * 1024u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (isA) {
                glds16(ra, base + (uint32_t)j * 1024u, voa[h][j], 0);
                voa[h][j] += stepA;
            } else {
                glds16(rb, base + (uint32_t)j * 1024u, vob[h][j], 0);
                vob[h][j] += stepB;
            }
        }
    };
    const uint32_t g4 = (uint32_t)l >> 4, q4 = ((uint32_t)l >> 2) & 3u, p4 = (uint32_t)l & 3u;
    auto lane_addr = [&](uint32_t col0) -> uint32_t {   // col0: multiple of 16
        const uint32_t col = col0 + 4 * p4;
        const uint32_t row = 8 * g4 + q4;
        return lds_base() + row * 256u + (((col >> 3) ^ (q4 << 2) ^ ((g4 & 1u) << 1)) * 16u) + (col & 4u) * 2u;
    };
    uint32_t a_lane[4], b_lane[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a_lane[mt] = lane_addr((uint32_t)wr * 64 + (uint32_t)mt * 16);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b_lane[nt] = lane_addr((uint32_t)wc * 64 + (uint32_t)nt * 16);
    auto tr_frag = [&](uint32_t addr, uint32_t imm) -> bf16x8 {
        const bf16x4 lo = lds_tr64_hidden(addr, imm);
        const bf16x4 hi = lds_tr64_hidden(addr, imm + 1024u);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    f32x4 acc[2][2][4][4];   // [i][j][mt][nt]: 256 registers
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[i][j][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[4][2], fa1[4][2], fbx[4][2], fby[4][2];   // [mt][ks], [nt][ks]
    auto rd_a2 = [&](uint32_t bufb, int i, bf16x8 (&fa)[4][2], uint32_t ks, int m0) {   // two A blocks of one ks (4 reads)
#pragma unroll
        for (int mt = m0; mt < m0 + 2; ++mt) fa[mt][ks] = tr_frag(a_lane[mt] + bufb, (uint32_t)i * 16384u + ks * 8192u);
    };
    auto rd_b1 = [&](uint32_t bufb, int j, bf16x8 (&fb)[4][2], int nt) {                // one B block, both ks (4 reads)
#pragma unroll
        for (uint32_t ks = 0; ks < 2; ++ks) fb[nt][ks] = tr_frag(b_lane[nt] + bufb, 32768u + (uint32_t)j * 16384u + ks * 8192u);
    };
    auto mma8 = [&](int i, int j, const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[4][2], int ks, int m0) {   // 2 A blocks x 4 B blocks
#pragma unroll
        for (int mt = m0; mt < m0 + 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[i][j][mt][nt] = mfma16(fa[mt][ks], fb[nt][ks], acc[i][j][mt][nt]);
    };
    // one 64-row step; on entry B0(t) is in fbp and ks 0 of A0(t) in fa0[.][0]; on exit the same for t+1 with fbq
    auto step = [&](int st, bf16x8 (&fbp)[4][2], bf16x8 (&fbq)[4][2]) {
        const int cur = st & 1;
        const uint32_t bufb = (uint32_t)cur * G256_BUF_BYTES, bufn = (uint32_t)(cur ^ 1) * G256_BUF_BYTES;
        // ---- phase 0: Q00 = A0 x B0 ; reads A0(t) ks 1 and B1(t)
        issue(cur ^ 1, 3);   // A1(t+1)
        wait_vmcnt<16>();
        g256_barrier();
        prio_hi();
        rd_a2(bufb, 0, fa0, 1, 0);
        rd_a2(bufb, 0, fa0, 1, 2);
        sched_fence();
        mma8(0, 0, fa0, fbp, 0, 0);
        sched_fence();
        rd_b1(bufb, 1, fbq, 0);          // 12 LDS reads out (the counter holds 15)
        sched_fence();
        mma8(0, 0, fa0, fbp, 0, 2);
        sched_fence();
        lds_wait<4>(fa0[0][1], fa0[1][1], fa0[2][1], fa0[3][1]);   // A0 ks 1 landed; the 4 B1 reads may still be out
        rd_b1(bufb, 1, fbq, 1);
        sched_fence();
        mma8(0, 0, fa0, fbp, 1, 0);
        sched_fence();
        rd_b1(bufb, 1, fbq, 2);
        sched_fence();
        mma8(0, 0, fa0, fbp, 1, 2);
        sched_fence();
        rd_b1(bufb, 1, fbq, 3);
        lds_wait<0>(fbq[0][0], fbq[0][1], fbq[1][0], fbq[1][1], fbq[2][0], fbq[2][1], fbq[3][0], fbq[3][1]);
        prio_lo();
        g256_barrier();
        // ---- phase 1: Q01 = A0 x B1 ; reads A1(t) ks 0
        issue(cur, 1);       // B0(t+2)
        wait_vmcnt<16>();
        g256_barrier();
        prio_hi();
        mma8(0, 1, fa0, fbq, 0, 0);
        sched_fence();
        rd_a2(bufb, 1, fa1, 0, 0);
        rd_a2(bufb, 1, fa1, 0, 2);
        sched_fence();
        mma8(0, 1, fa0, fbq, 0, 2);
        mma8(0, 1, fa0, fbq, 1, 0);
        mma8(0, 1, fa0, fbq, 1, 2);
        sched_fence();
        lds_wait<0>(fa1[0][0], fa1[1][0], fa1[2][0], fa1[3][0]);
        prio_lo();
        g256_barrier();
        // ---- phase 2: Q11 = A1 x B1 ; reads A1(t) ks 1
        issue(cur, 0);       // A0(t+2)
        wait_vmcnt<16>();
        g256_barrier();
        prio_hi();
        rd_a2(bufb, 1, fa1, 1, 0);
        rd_a2(bufb, 1, fa1, 1, 2);
        sched_fence();
        mma8(1, 1, fa1, fbq, 0, 0);
        mma8(1, 1, fa1, fbq, 0, 2);
        sched_fence();
        lds_wait<0>(fa1[0][1], fa1[1][1], fa1[2][1], fa1[3][1]);
        sched_fence();
        mma8(1, 1, fa1, fbq, 1, 0);
        mma8(1, 1, fa1, fbq, 1, 2);
        prio_lo();
        g256_barrier();
        // ---- phase 3: Q10 = A1 x B0 ; reads A0(t+1) ks 0 and B0(t+1)
        issue(cur, 2);       // B1(t+2)
        wait_vmcnt<16>();
        g256_barrier();
        prio_hi();
        mma8(1, 0, fa1, fbp, 0, 0);
        sched_fence();
        rd_a2(bufn, 0, fa0, 0, 0);
        rd_a2(bufn, 0, fa0, 0, 2);
        sched_fence();
        mma8(1, 0, fa1, fbp, 0, 2);
        sched_fence();
        rd_b1(bufn, 0, fbq, 0);          // fbq held B1(t): last used in phase 2
        sched_fence();
        mma8(1, 0, fa1, fbp, 1, 0);
        sched_fence();
        lds_wait<4>(fa0[0][0], fa0[1][0], fa0[2][0], fa0[3][0]);
        rd_b1(bufn, 0, fbq, 1);
        rd_b1(bufn, 0, fbq, 2);
        sched_fence();
        mma8(1, 0, fa1, fbp, 1, 2);
        sched_fence();
        rd_b1(bufn, 0, fbq, 3);
        lds_wait<0>(fbq[0][0], fbq[0][1], fbq[1][0], fbq[1][1], fbq[2][0], fbq[2][1], fbq[3][0], fbq[3][1]);
        prio_lo();
        g256_barrier();
    };
    issue(0, 1);
    issue(0, 0);
    issue(0, 2);
    issue(0, 3);
    issue(1, 1);
    issue(1, 0);
    issue(1, 2);
    wait_vmcnt<16>();
    g256_barrier();
    rd_a2(0u, 0, fa0, 0, 0);
    rd_a2(0u, 0, fa0, 0, 2);
    rd_b1(0u, 0, fbx, 0);
    lds_wait<4>(fa0[0][0], fa0[1][0], fa0[2][0], fa0[3][0]);
    rd_b1(0u, 0, fbx, 1);
    rd_b1(0u, 0, fbx, 2);
    rd_b1(0u, 0, fbx, 3);
    lds_wait<0>(fbx[0][0], fbx[0][1], fbx[1][0], fbx[1][1], fbx[2][0], fbx[2][1], fbx[3][0], fbx[3][1]);
    const int nsteps = step1 - step0;
#pragma nounroll
    for (int st = 0; st < nsteps; st += 2) {
        step(st, fbx, fby);
        step(st + 1, fby, fbx);
    }
    wait_vmcnt0();
    g256_barrier();
    // C += acc : D column = l & 15 -> k (contiguous), row = 4 (l >> 4) + reg -> n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int kk = k0 + j * 128 + wc * 64 + nt * 16 + (l & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + i * 128 + wr * 64 + mt * 16 + 4 * (l >> 4) + r;
                        float* dst = Cp + (size_t)n * K + kk;
                        if (ATOMIC == 1) atomic_addf(dst, acc[i][j][mt][nt][r]);
                        else if (ATOMIC == 2) *dst = acc[i][j][mt][nt][r];
                        else *dst = *dst + acc[i][j][mt][nt][r];
                    }
                }
}

// EXPERIMENT: the 4-wave form (tic_set_option("tn_waves", 4), measurement / simulator builds)
__global__ void __launch_bounds__(256, 1) gemm_tn256_streamk_w4_kernel(GemmTnGroupParams gp, int nsteps, int s_main, int tpx, int tail_each) {
    TN256_STREAMK_BODY(tn256_tile_segment16_w4)
}
