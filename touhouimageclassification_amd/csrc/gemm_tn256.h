// gemm_tn256.h -- grouped, deep-pipelined weight-gradient GEMM (gfx950).
//
//   for each problem g of a group (the four Linear layers of one transformer block):
//       C_g[N_g, K_g] += A_g[M, N_g]^T . B_g[M, K_g]          (dW = dY^T . X, fp32)
//
// One launch covers the whole group: 256x256 output tiles, ONE workgroup (8 waves) per tile running the
// full reduction over M, so there is no split-K, no atomics and no reduction pass; a ViT-L block gives
// 48 + 16 + 64 + 64 = 192 tiles for 256 CUs.  Same 4-phase / counted-vmcnt / staggered-wave-group
// pipeline as gemm256.h (see the hazard bookkeeping there); what differs is the operand geometry:
// both operands are reduction-major in memory ([m][n] and [m][k]), so the LDS half-tiles are
// [64 m][128 columns] (256-B rows, swz256) and every fragment comes from ds_read_b64_tr_b16 feeding
// v_mfma_f32_32x32x16_bf16 in its natural orientation (D column = k = contiguous in C: each store
// instruction writes two full 128-B row segments).
#pragma once
#include "gemm256.h"

#define TN_MAX_GROUP 4
struct GemmTnProblem {
    const bf16_t* A;   // [M, N]
    const bf16_t* B;   // [M, K]
    float* C;          // [N, K]
    int N, K;
    int tile_start;    // first tile index of this problem inside the launch
    int bw;            // tile-walk block width along the LONGER tile dimension (see tn_tile_lookup), >= 1
};
struct GemmTnGroupParams {
    GemmTnProblem prob[TN_MAX_GROUP];
    int nprob, M, total_tiles;
};

TIC_DEV void tn_tile_lookup(const GemmTnGroupParams& gp, int tile, const bf16_t*& Ap, const bf16_t*& Bp, float*& Cp, int& N, int& K, int& n0, int& k0) {
    int pi = 0;
#pragma unroll
    for (int g = 1; g < TN_MAX_GROUP; ++g)
        if (g < gp.nprob && tile >= gp.prob[g].tile_start) pi = g;
    Ap = gp.prob[0].A; Bp = gp.prob[0].B; Cp = gp.prob[0].C; N = gp.prob[0].N; K = gp.prob[0].K;
    int t0 = 0;
#pragma unroll
    for (int g = 1; g < TN_MAX_GROUP; ++g)
        if (pi == g) {
            Ap = gp.prob[g].A; Bp = gp.prob[g].B; Cp = gp.prob[g].C; N = gp.prob[g].N; K = gp.prob[g].K; t0 = gp.prob[g].tile_start;
        }
    // Tile walk inside a problem: blocks of (short dimension) x bw tiles, so that the ~24 consecutive tiles an XCD takes share few
    // operand panels (a 4 x 6 block reads 4 + 6 panels; 24 tiles of a row-major walk over a 4 x 16 grid read 2 + 16).  Every A
    // panel (256 columns of dY) is streamed by all tiles of its tile row, every B panel (256 columns of X) by all tiles of its tile
    // column; the XCD's L2 serves the repeats only while the co-running tiles share them.  Any bijection is correct; this one is speed.
    const int tiles_k = K / 256, tiles_n = N / 256, lt = tile - t0;
    int bwv = gp.prob[0].bw;
#pragma unroll
    for (int g = 1; g < TN_MAX_GROUP; ++g)
        if (pi == g) bwv = gp.prob[g].bw;
    const bool long_k = tiles_k >= tiles_n;
    const int ls = long_k ? tiles_n : tiles_k, ll = long_k ? tiles_k : tiles_n;   // short / long tile counts
    const int blk = lt / (ls * bwv);                       // full blocks first; the last one may be narrower
    const int w_eff = (ll - blk * bwv < bwv) ? (ll - blk * bwv) : bwv;
    const int r = lt - blk * ls * bwv;
    const int s_idx = r / w_eff, l_idx = blk * bwv + (r - s_idx * w_eff);
    n0 = (long_k ? s_idx : l_idx) * 256;
    k0 = (long_k ? l_idx : s_idx) * 256;
}

// One 256x256 output tile (origin n0, k0 of problem A/B/C) over the M steps [step0, step1) of 64 rows.
// ATOMIC = 0: the workgroup owns the whole reduction, C += acc with plain read-modify-write;
// ATOMIC = 1: partial reduction (stream-K share), C += acc with fp32 atomics (two 128-B row segments / instruction);
// ATOMIC = 2: partial reduction STORED (C here is one slice of a slab that tn_slab_reduce_kernel adds up afterwards).
// DBG (measurement only, see gemm256.h): bit 0 = no LDS-DMA, bit 1 = no fragment reads, bit 2 = no MFMA.
template <int ATOMIC, int DBG = 0>
TIC_DEV void tn256_tile_segment(const bf16_t* Ap, const bf16_t* Bp, float* Cp, int N, int K, int M, int n0, int k0, int step0, int step1) {
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wr = w >> 2, wc = w & 3;
    // rows >= row_end (the end of this segment, or M) are outside the range check and read as 0
    const int row_end = (step1 * 64 < M) ? step1 * 64 : M;
    const tic_rsrc_t ra = make_rsrc(Ap, (uint32_t)((size_t)row_end * N * 2));
    const tic_rsrc_t rb = make_rsrc(Bp, (uint32_t)((size_t)row_end * K * 2));

    // ---- LDS-DMA: half-tile = [64 m][128 cols] = 16 pieces of 4 rows; this wave moves pieces 2w, 2w+1.
    // The row advance lives in the VGPR offset (only that offset is range-checked: rows >= M read 0).
    const uint32_t rr = (uint32_t)l >> 4, ch_log = ((uint32_t)l & 15u) ^ (rr << 2);
    uint32_t voa[2][2], vob[2][2];   // [half][piece], running
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t row = (uint32_t)step0 * 64u + (uint32_t)(2 * w + j) * 4 + rr;
            voa[h][j] = (uint32_t)(((size_t)row * N + n0 + h * 128 + ch_log * 8) * 2);
            vob[h][j] = (uint32_t)(((size_t)row * K + k0 + h * 128 + ch_log * 8) * 2);
        }
    const uint32_t stepA = (uint32_t)N * 128u, stepB = (uint32_t)K * 128u;   // 64 rows in bytes
    // which: 0 = A0, 1 = B0, 2 = B1, 3 = A1 ; each call also advances that half-tile's offsets to the next step
    auto issue = [&](int buf, int which) {
        if (DBG & 1) return;
        const bool isA = (which == 0 || which == 3);
        const int h = (which >= 2) ? 1 : 0;
        const uint32_t base = (uint32_t)buf * G256_BUF_BYTES + (isA ? 0u : 32768u) + (uint32_t)h * 16384u + (uint32_t)(2 * w) * 1024u;
        if (isA) {
            glds16(ra, base, voa[h][0], 0);
            glds16(ra, base + 1024u, voa[h][1], 0);
            voa[h][0] += stepA;
            voa[h][1] += stepA;
        } else {
            glds16(rb, base, vob[h][0], 0);
            glds16(rb, base + 1024u, vob[h][1], 0);
            vob[h][0] += stepB;
            vob[h][1] += stepB;
        }
    };

    // ---- transposed fragments (32x32x16): lane (h2 = l>>5, c16 = (l>>4)&1, q = (l>>2)&3, p4 = l&3).
    // Fragment address = lane part (VGPR, one per column group) + buffer base + compile-time offset (half-tile, 16 ks rows,
    // +4 rows for the second read): the reads are hidden from hipcc (lds_tr64_hidden) so that no vmcnt(0) is put in front
    // of them; they are ordered against the LDS-DMA by the counted vmcnt + barrier schedule below.
    const uint32_t h2 = (uint32_t)l >> 5, c16 = ((uint32_t)l >> 4) & 1u, q4 = ((uint32_t)l >> 2) & 3u, p4 = (uint32_t)l & 3u;
    auto lane_addr = [&](uint32_t col0) -> uint32_t {
        const uint32_t col = col0 + 16 * c16 + 4 * p4;
        const uint32_t row = 8 * h2 + q4;                        // (row + 16 ks) & 3 == q4 for both reads
        return lds_base() + row * 256u + (((col >> 3) ^ (q4 << 2)) * 16u) + (col & 4u) * 2u;
    };
    const uint32_t a_lane[2] = {lane_addr((uint32_t)wr * 64), lane_addr((uint32_t)wr * 64 + 32)};
    const uint32_t b_lane = lane_addr((uint32_t)wc * 32);
    auto tr_frag = [&](uint32_t addr, uint32_t imm) -> bf16x8 {
        const bf16x4 lo = lds_tr64_hidden(addr, imm);
        const bf16x4 hi = lds_tr64_hidden(addr, imm + 1024u);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x16 acc[2][2][2];   // [i][j][nt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][nt][r] = 0.f;
    // fragment registers: A0 / A1 of the current step ([nt][ks], 16 m rows per ks) and two B sets whose roles (B0 | B1)
    // swap every step.  Same schedule as gemm256.h (see the hazard table there): a load segment only issues one
    // half-tile and waits vmcnt(8); the fragment reads ride under the MFMAs of the previous phase, A in two ks halves so
    // that the fragment peak is 80 VGPRs.
    bf16x8 fa0[2][4], fa1[2][4], fbx[4], fby[4];
    if (DBG & 2) {
#pragma unroll
        for (int x = 0; x < 4; ++x) fa0[0][x] = fa0[1][x] = fa1[0][x] = fa1[1][x] = fbx[x] = fby[x] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
    }
    auto rd_a = [&](uint32_t bufb, int i, bf16x8 (&fa)[2][4], uint32_t ks) {   // 4 reads
        if (DBG & 2) return;
#pragma unroll
        for (uint32_t nt = 0; nt < 2; ++nt) fa[nt][ks] = tr_frag(a_lane[nt] + bufb, (uint32_t)i * 16384u + ks * 4096u);
    };
    auto rd_b = [&](uint32_t bufb, int j, bf16x8 (&fb)[4], uint32_t ks) {      // 2 reads
        if (DBG & 2) return;
        fb[ks] = tr_frag(b_lane + bufb, 32768u + (uint32_t)j * 16384u + ks * 4096u);
    };
    auto mma = [&](int i, int j, const bf16x8 (&fa)[2][4], const bf16x8 (&fb)[4], int ks) {   // 2 MFMAs
        if (DBG & 4) {
            acc[i][j][0][0] += (float)fa[0][ks][0] + (float)fa[1][ks][1] + (float)fb[ks][0];   // keep the reads alive
            return;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[i][j][nt] = mfma32(fa[nt][ks], fb[ks], acc[i][j][nt]);
    };
    // one 64-row step; on entry B0(t) is in fbp and ks 0,1 of A0(t) in fa0; on exit the same for t+1 with fbq.
    // sched_fence() pins the hand interleave (the reads are inline asm: hipcc neither counts nor places them).
    auto step = [&](int st, bf16x8 (&fbp)[4], bf16x8 (&fbq)[4]) {
        const int cur = st & 1;
        const uint32_t bufb = (uint32_t)cur * G256_BUF_BYTES, bufn = (uint32_t)(cur ^ 1) * G256_BUF_BYTES;
        // ---- phase 0: Q00 = A0 x B0 ; reads A0(t) ks 2,3 and B1(t)
        issue(cur ^ 1, 3);   // A1(t+1)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        rd_a(bufb, 0, fa0, 2);
        rd_a(bufb, 0, fa0, 3);
        sched_fence();
        mma(0, 0, fa0, fbp, 0);
        sched_fence();
        rd_b(bufb, 1, fbq, 0);
        rd_b(bufb, 1, fbq, 1);
        sched_fence();
        mma(0, 0, fa0, fbp, 1);
        sched_fence();
        rd_b(bufb, 1, fbq, 2);
        rd_b(bufb, 1, fbq, 3);
        if (!(DBG & 2)) lds_wait<8>(fa0[0][2], fa0[0][3], fa0[1][2], fa0[1][3]);   // the 8 B1 reads may still be out
        sched_fence();
        mma(0, 0, fa0, fbp, 2);
        mma(0, 0, fa0, fbp, 3);
        sched_fence();
        if (!(DBG & 2)) lds_wait<0>(fbq[0], fbq[1], fbq[2], fbq[3]);
        prio_lo();
        g256_barrier();
        // ---- phase 1: Q01 = A0 x B1 ; reads A1(t) ks 0,1
        issue(cur, 1);       // B0(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        mma(0, 1, fa0, fbq, 0);
        sched_fence();
        rd_a(bufb, 1, fa1, 0);
        sched_fence();
        mma(0, 1, fa0, fbq, 1);
        sched_fence();
        rd_a(bufb, 1, fa1, 1);
        sched_fence();
        mma(0, 1, fa0, fbq, 2);
        mma(0, 1, fa0, fbq, 3);
        sched_fence();
        if (!(DBG & 2)) lds_wait<0>(fa1[0][0], fa1[0][1], fa1[1][0], fa1[1][1]);
        prio_lo();
        g256_barrier();
        // ---- phase 2: Q11 = A1 x B1 ; reads A1(t) ks 2,3
        issue(cur, 0);       // A0(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        rd_a(bufb, 1, fa1, 2);
        rd_a(bufb, 1, fa1, 3);
        sched_fence();
        mma(1, 1, fa1, fbq, 0);
        mma(1, 1, fa1, fbq, 1);
        sched_fence();
        if (!(DBG & 2)) lds_wait<0>(fa1[0][2], fa1[0][3], fa1[1][2], fa1[1][3]);
        sched_fence();
        mma(1, 1, fa1, fbq, 2);
        mma(1, 1, fa1, fbq, 3);
        prio_lo();
        g256_barrier();
        // ---- phase 3: Q10 = A1 x B0 ; reads A0(t+1) ks 0,1 and B0(t+1)
        issue(cur, 2);       // B1(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        mma(1, 0, fa1, fbp, 0);
        sched_fence();
        rd_a(bufn, 0, fa0, 0);
        sched_fence();
        mma(1, 0, fa1, fbp, 1);
        sched_fence();
        rd_a(bufn, 0, fa0, 1);
        sched_fence();
        mma(1, 0, fa1, fbp, 2);
        sched_fence();
        rd_b(bufn, 0, fbq, 0);
        rd_b(bufn, 0, fbq, 1);
        sched_fence();
        mma(1, 0, fa1, fbp, 3);
        sched_fence();
        rd_b(bufn, 0, fbq, 2);
        rd_b(bufn, 0, fbq, 3);
        if (!(DBG & 2)) lds_wait<0>(fa0[0][0], fa0[0][1], fa0[1][0], fa0[1][1], fbq[0], fbq[1], fbq[2], fbq[3]);
        prio_lo();
        g256_barrier();
    };

    // prologue: all of step 0 and B0, A0, B1 of step 1 in the steady-state issue order B0, A0, B1, A1; steps past the end of
    // the segment read rows >= row_end, i.e. zero fills; B0(0), A0(0), B1(0) landed; first fragments; group 1 falls behind
    issue(0, 1);
    issue(0, 0);
    issue(0, 2);
    issue(0, 3);
    issue(1, 1);
    issue(1, 0);
    issue(1, 2);
    wait_vmcnt<8>();
    g256_barrier();
    rd_a(0u, 0, fa0, 0);
    rd_a(0u, 0, fa0, 1);
    rd_b(0u, 0, fbx, 0);
    rd_b(0u, 0, fbx, 1);
    rd_b(0u, 0, fbx, 2);
    rd_b(0u, 0, fbx, 3);
    if (!(DBG & 2)) lds_wait<0>(fa0[0][0], fa0[0][1], fa0[1][0], fa0[1][1], fbx[0], fbx[1], fbx[2], fbx[3]);
    if (wr == 1) g256_barrier();

    // two steps per trip (the B register sets swap roles); an odd count runs one extra all-zero step (rows >= row_end)
    const int nsteps = step1 - step0;
#pragma nounroll
    for (int st = 0; st < nsteps; st += 2) {
        step(st, fbx, fby);
        step(st + 1, fby, fbx);
    }
    wait_vmcnt0();
    if (wr == 0) g256_barrier();

    // C += acc : D column = l&31 -> k (contiguous), row = (r&3) + 8(r>>2) + 4(l>>5) -> n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int kk = k0 + j * 128 + wc * 32 + (l & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + i * 128 + wr * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    float* dst = Cp + (size_t)n * K + kk;
                    if (ATOMIC == 1) atomic_addf(dst, acc[i][j][nt][r]);
                    else if (ATOMIC == 2) *dst = acc[i][j][nt][r];
                    else *dst = *dst + acc[i][j][nt][r];
                }
            }
}

// The same tile segment on v_mfma_f32_16x16x32_bf16 (the NT kernel's instruction).  Cycles per FLOP equal those of the 32x32x16
// form, but the chip holds a higher clock on this shape under load (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15 x the
// FLOP/s in LDS-fed loops on random data), and the kernel is MFMA-clock-bound once its reads are hidden.
// What changes against the 32x32 form above:
//   fragments   a lane holds 8 consecutive m (reduction) values of ONE column of a 16-column block: lane (g = l>>4, q = (l>>2)&3,
//               p = l&3) addresses row 32 ks + 8 g + q (+4 for the second read), columns 4p..4p+3; the transposing read hands
//               lane i of each 16-lane group column i of those 4 rows.  A: 4 blocks (mt) of 16 n, B: 2 blocks (nt) of 16 k.
//   swizzle     one read instruction covers rows {8 g + q : g in a pair, q = 0..3} = 8 rows x 32 B; the 256-B bank row has 8
//               such slots, so the 16-B chunk index is XORed with (q << 2) ^ ((g & 1) << 1) on both sides (DMA source and read).
//   D layout    col = l & 15 -> k (16 contiguous fp32 = 64 B per row and instruction), row = 4 (l >> 4) + reg -> n.
template <int ATOMIC>
TIC_DEV void tn256_tile_segment16(const bf16_t* Ap, const bf16_t* Bp, float* Cp, int N, int K, int M, int n0, int k0, int step0, int step1) {
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wr = w >> 2, wc = w & 3;
    const int row_end = (step1 * 64 < M) ? step1 * 64 : M;
    const tic_rsrc_t ra = make_rsrc(Ap, (uint32_t)((size_t)row_end * N * 2));
    const tic_rsrc_t rb = make_rsrc(Bp, (uint32_t)((size_t)row_end * K * 2));

    // ---- LDS-DMA: half-tile = [64 m][128 cols] = 16 pieces of 4 rows; this wave moves pieces 2w, 2w+1 (rows 8w .. 8w+7:
    // (row >> 3) & 1 == w & 1 for both pieces)
    const uint32_t rr = (uint32_t)l >> 4, ch_log = ((uint32_t)l & 15u) ^ (rr << 2) ^ (((uint32_t)w & 1u) << 1);
    uint32_t voa[2][2], vob[2][2];   // [half][piece], running
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t row = (uint32_t)step0 * 64u + (uint32_t)(2 * w + j) * 4 + rr;
            voa[h][j] = (uint32_t)(((size_t)row * N + n0 + h * 128 + ch_log * 8) * 2);
            vob[h][j] = (uint32_t)(((size_t)row * K + k0 + h * 128 + ch_log * 8) * 2);
        }
    const uint32_t stepA = (uint32_t)N * 128u, stepB = (uint32_t)K * 128u;   // 64 rows in bytes
    auto issue = [&](int buf, int which) {   // which: 0 = A0, 1 = B0, 2 = B1, 3 = A1
        const bool isA = (which == 0 || which == 3);
        const int h = (which >= 2) ? 1 : 0;
        const uint32_t base = (uint32_t)buf * G256_BUF_BYTES + (isA ? 0u : 32768u) + (uint32_t)h * 16384u + (uint32_t)(2 * w) * 1024u;
        if (isA) {
            glds16(ra, base, voa[h][0], 0);
            glds16(ra, base + 1024u, voa[h][1], 0);
            voa[h][0] += stepA;
            voa[h][1] += stepA;
        } else {
            glds16(rb, base, vob[h][0], 0);
            glds16(rb, base + 1024u, vob[h][1], 0);
            vob[h][0] += stepB;
            vob[h][1] += stepB;
        }
    };

    const uint32_t g4 = (uint32_t)l >> 4, q4 = ((uint32_t)l >> 2) & 3u, p4 = (uint32_t)l & 3u;
    auto lane_addr = [&](uint32_t col0) -> uint32_t {   // col0: multiple of 16
        const uint32_t col = col0 + 4 * p4;
        const uint32_t row = 8 * g4 + q4;                 // (row & 3) == q4 and ((row >> 3) & 1) == (g4 & 1) for both reads and every ks
        return lds_base() + row * 256u + (((col >> 3) ^ (q4 << 2) ^ ((g4 & 1u) << 1)) * 16u) + (col & 4u) * 2u;
    };
    uint32_t a_lane[4], b_lane[2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a_lane[mt] = lane_addr((uint32_t)wr * 64 + (uint32_t)mt * 16);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) b_lane[nt] = lane_addr((uint32_t)wc * 32 + (uint32_t)nt * 16);
    auto tr_frag = [&](uint32_t addr, uint32_t imm) -> bf16x8 {
        const bf16x4 lo = lds_tr64_hidden(addr, imm);
        const bf16x4 hi = lds_tr64_hidden(addr, imm + 1024u);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    f32x4 acc[2][2][4][2];   // [i][j][mt][nt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[4][2], fa1[4][2], fbx[2][2], fby[2][2];   // [mt][ks], [nt][ks]
    // two of the four A blocks of one ks (4 reads)
    auto rd_a2 = [&](uint32_t bufb, int i, bf16x8 (&fa)[4][2], uint32_t ks, int m0) {
#pragma unroll
        for (int mt = m0; mt < m0 + 2; ++mt) fa[mt][ks] = tr_frag(a_lane[mt] + bufb, (uint32_t)i * 16384u + ks * 8192u);
    };
    // one B block, both ks (4 reads)
    auto rd_b1 = [&](uint32_t bufb, int j, bf16x8 (&fb)[2][2], int nt) {
#pragma unroll
        for (uint32_t ks = 0; ks < 2; ++ks) fb[nt][ks] = tr_frag(b_lane[nt] + bufb, 32768u + (uint32_t)j * 16384u + ks * 8192u);
    };
    // 4 MFMAs: A blocks m0, m0+1 x both B blocks, one ks
    auto mma4 = [&](int i, int j, const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2], int ks, int m0) {
#pragma unroll
        for (int mt = m0; mt < m0 + 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = mfma16(fa[mt][ks], fb[nt][ks], acc[i][j][mt][nt]);
    };
    // one 64-row step; on entry B0(t) is in fbp and ks 0 of A0(t) in fa0[.][0]; on exit the same for t+1 with fbq.  Same load /
    // wait / barrier skeleton as the 32x32 form; inside an MFMA segment groups of 4 MFMAs alternate with groups of 4 hidden reads.
    auto step = [&](int st, bf16x8 (&fbp)[2][2], bf16x8 (&fbq)[2][2]) {
        const int cur = st & 1;
        const uint32_t bufb = (uint32_t)cur * G256_BUF_BYTES, bufn = (uint32_t)(cur ^ 1) * G256_BUF_BYTES;
        // ---- phase 0: Q00 = A0 x B0 ; reads A0(t) ks 1 and B1(t)
        issue(cur ^ 1, 3);   // A1(t+1)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        rd_a2(bufb, 0, fa0, 1, 0);
        sched_fence();
        mma4(0, 0, fa0, fbp, 0, 0);
        sched_fence();
        rd_a2(bufb, 0, fa0, 1, 2);
        sched_fence();
        mma4(0, 0, fa0, fbp, 0, 2);
        sched_fence();
        rd_b1(bufb, 1, fbq, 0);
        lds_wait<4>(fa0[0][1], fa0[1][1], fa0[2][1], fa0[3][1]);   // the 4 B1 reads may still be out
        sched_fence();
        mma4(0, 0, fa0, fbp, 1, 0);
        sched_fence();
        rd_b1(bufb, 1, fbq, 1);
        sched_fence();
        mma4(0, 0, fa0, fbp, 1, 2);
        sched_fence();
        lds_wait<0>(fbq[0][0], fbq[0][1], fbq[1][0], fbq[1][1]);
        prio_lo();
        g256_barrier();
        // ---- phase 1: Q01 = A0 x B1 ; reads A1(t) ks 0
        issue(cur, 1);       // B0(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        mma4(0, 1, fa0, fbq, 0, 0);
        sched_fence();
        rd_a2(bufb, 1, fa1, 0, 0);
        sched_fence();
        mma4(0, 1, fa0, fbq, 0, 2);
        sched_fence();
        rd_a2(bufb, 1, fa1, 0, 2);
        sched_fence();
        mma4(0, 1, fa0, fbq, 1, 0);
        mma4(0, 1, fa0, fbq, 1, 2);
        sched_fence();
        lds_wait<0>(fa1[0][0], fa1[1][0], fa1[2][0], fa1[3][0]);
        prio_lo();
        g256_barrier();
        // ---- phase 2: Q11 = A1 x B1 ; reads A1(t) ks 1
        issue(cur, 0);       // A0(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        rd_a2(bufb, 1, fa1, 1, 0);
        sched_fence();
        mma4(1, 1, fa1, fbq, 0, 0);
        sched_fence();
        rd_a2(bufb, 1, fa1, 1, 2);
        sched_fence();
        mma4(1, 1, fa1, fbq, 0, 2);
        sched_fence();
        lds_wait<0>(fa1[0][1], fa1[1][1], fa1[2][1], fa1[3][1]);
        sched_fence();
        mma4(1, 1, fa1, fbq, 1, 0);
        mma4(1, 1, fa1, fbq, 1, 2);
        prio_lo();
        g256_barrier();
        // ---- phase 3: Q10 = A1 x B0 ; reads A0(t+1) ks 0 and B0(t+1)
        issue(cur, 2);       // B1(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        mma4(1, 0, fa1, fbp, 0, 0);
        sched_fence();
        rd_a2(bufn, 0, fa0, 0, 0);
        sched_fence();
        mma4(1, 0, fa1, fbp, 0, 2);
        sched_fence();
        rd_a2(bufn, 0, fa0, 0, 2);
        sched_fence();
        mma4(1, 0, fa1, fbp, 1, 0);
        sched_fence();
        rd_b1(bufn, 0, fbq, 0);
        sched_fence();
        mma4(1, 0, fa1, fbp, 1, 2);
        sched_fence();
        rd_b1(bufn, 0, fbq, 1);
        lds_wait<0>(fa0[0][0], fa0[1][0], fa0[2][0], fa0[3][0], fbq[0][0], fbq[0][1], fbq[1][0], fbq[1][1]);
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
    wait_vmcnt<8>();
    g256_barrier();
    rd_a2(0u, 0, fa0, 0, 0);
    rd_a2(0u, 0, fa0, 0, 2);
    rd_b1(0u, 0, fbx, 0);
    rd_b1(0u, 0, fbx, 1);
    lds_wait<0>(fa0[0][0], fa0[1][0], fa0[2][0], fa0[3][0], fbx[0][0], fbx[0][1], fbx[1][0], fbx[1][1]);
    if (wr == 1) g256_barrier();

    const int nsteps = step1 - step0;
#pragma nounroll
    for (int st = 0; st < nsteps; st += 2) {
        step(st, fbx, fby);
        step(st + 1, fby, fbx);
    }
    wait_vmcnt0();
    if (wr == 0) g256_barrier();

    // C += acc : D column = l & 15 -> k (contiguous), row = 4 (l >> 4) + reg -> n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int kk = k0 + j * 128 + wc * 32 + nt * 16 + (l & 15);
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


template <int DBG = 0>
__global__ void __launch_bounds__(512, 2) gemm_tn256_kernel(GemmTnGroupParams gp) {
    // XCD-contiguous tile order, then problem lookup (wave-uniform)
    int wg;
    {
        const int bid = TIC_BID_X, nwg = gp.total_tiles;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const bf16_t *Ap, *Bp;
    float* Cp;
    int N, K, n0, k0;
    tn_tile_lookup(gp, wg, Ap, Bp, Cp, N, K, n0, k0);
    tn256_tile_segment<0, DBG>(Ap, Bp, Cp, N, K, gp.M, n0, k0, 0, (gp.M + 63) / 64);
}

// The same launch STORING its tiles (C = dY^T X): the caller states that C holds nothing to keep -- the first backward after the
// gradients were dropped.  Saves the read of C (256 KiB per workgroup at one CU's ~25 GB/s = 10 us per launch) and the zeroing of it.
__global__ void __launch_bounds__(512, 2) gemm_tn256_store_kernel(GemmTnGroupParams gp) {
    int wg;
    {
        const int bid = TIC_BID_X, nwg = gp.total_tiles;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const bf16_t *Ap, *Bp;
    float* Cp;
    int N, K, n0, k0;
    tn_tile_lookup(gp, wg, Ap, Bp, Cp, N, K, n0, k0);
    tn256_tile_segment<2>(Ap, Bp, Cp, N, K, gp.M, n0, k0, 0, (gp.M + 63) / 64);
}

// Stream-K forms: every CU gets an equal share of the (tile, M step) work instead of 192 busy + 64 idle CUs.  A share is
// processed as the segments it has inside single tiles, each through the pipelined tile routine, and every partial
// tile is ADDED to C with fp32 atomics.
//
// s_main > 0 -- PHASE-ALIGNED split (used when the tile count allows it, e.g. ViT-L: 192 tiles = 8 XCDs x 24):
//   per XCD, `tpx` "main" workgroups take steps [0, s_main) of one tile each and the remaining 32 - tpx "tail"
//   workgroups take steps [s_main, S) of `tail_each` tiles of the SAME XCD.  All main CUs sweep the same rows at the same
//   time and so do the tail CUs (lock-step row slabs are what the L2 / Infinity Cache can share: the flat split below
//   fetched 2x the bytes of the lock-step kernel because its shares start at 256 different row offsets).
// s_main == 0 -- FLAT split: the flattened (tile, step) space cut into gridDim.x equal contiguous shares.
// s_main < 0  -- EQUAL PARTS (tile counts the phase-aligned split cannot take, e.g. ViT-B: 108 tiles): P = -s_main workgroups per tile,
//   each the same row range [part S/P, (part+1) S/P) of its tile; the tiles are dealt to the XCDs in contiguous runs of q or q + 1.
//   2 x 108 = 216 workgroups leave 40 CUs idle, but the workgroups of a part sweep the same rows at the same time (the flat split's
//   256 shares start at 256 different row offsets and fetch every panel twice).
// body shared by the two kernel names below (a macro, not a function: passing the by-value kernel argument struct on to a
// function makes hipcc copy it to scratch -- +136 B/lane, +21 VGPRs, 8 % slower)
#define TN256_STREAMK_BODY(SEG) \
    /* every mode is reduced to "nseg segments (tile, [a, b))" so that the pipelined tile routine is inlined at ONE call site: a second, \
       third and fourth site cost registers each, and the fourth (equal parts) tipped the kernel into 184 spilled registers */ \
    int nseg = 0, tile0 = 0, a0 = 0, b_last = nsteps, a_rest = 0; \
    if (s_main < 0) {   /* EQUAL PARTS: every tile is cut into P = -s_main row ranges, one workgroup each (P * tiles <= 256) */ \
        const int P = -s_main, xcd = TIC_BID_X & 7, idx = TIC_BID_X >> 3; \
        const int q = gp.total_tiles >> 3, r = gp.total_tiles & 7, txm = q + (r ? 1 : 0);   /* tiles of an XCD: q or q + 1 */ \
        const int tx = q + (xcd < r ? 1 : 0), part = idx / txm, ti = idx - part * txm; \
        const int per = (nsteps + P - 1) / P; \
        tile0 = xcd * q + (xcd < r ? xcd : r) + ti; \
        a0 = part * per; \
        b_last = (a0 + per < nsteps) ? a0 + per : nsteps; \
        nseg = (ti < tx && part < P && a0 < b_last) ? 1 : 0; \
    } else if (s_main > 0) { \
        const int xcd = TIC_BID_X & 7, idx = TIC_BID_X >> 3;   /* blocks b, b+8, ... share an XCD (speed only, never correctness) */ \
        if (idx < tpx) { \
            tile0 = xcd * tpx + idx; \
            a0 = 0; \
            b_last = s_main; \
            nseg = 1; \
        } else { \
            tile0 = xcd * tpx + (idx - tpx) * tail_each; \
            a0 = a_rest = s_main; \
            b_last = nsteps; \
            nseg = tail_each; \
        } \
    } else { \
        const long total_units = (long)gp.total_tiles * nsteps; \
        const int u0 = (int)(total_units * TIC_BID_X / TIC_NBLK_X), u1 = (int)(total_units * (TIC_BID_X + 1) / TIC_NBLK_X); \
        if (u1 > u0) { \
            const int tile_l = (u1 - 1) / nsteps; \
            tile0 = u0 / nsteps; \
            a0 = u0 - tile0 * nsteps; \
            b_last = u1 - tile_l * nsteps; \
            nseg = tile_l - tile0 + 1; \
        } \
    } \
    for (int it = 0; it < nseg; ++it) {   /* wave-uniform */ \
        const bf16_t *Ap, *Bp; \
        float* Cp; \
        int N, K, n0, k0; \
        tn_tile_lookup(gp, tile0 + it, Ap, Bp, Cp, N, K, n0, k0); \
        SEG<1>(Ap, Bp, Cp, N, K, gp.M, n0, k0, it == 0 ? a0 : a_rest, it == nseg - 1 ? b_last : nsteps); \
    }

// the grouped launch of one transformer block (the step's dominant kernel: bench.py times exactly these launches) ...
__global__ void __launch_bounds__(512, 2) gemm_tn256_streamk_kernel(GemmTnGroupParams gp, int nsteps, int s_main, int tpx, int tail_each) {
    TN256_STREAMK_BODY(tn256_tile_segment16)
}
// the 32x32x16-MFMA form of the same launch (tic_set_option("tn_mfma", 32): A/B measurements)
__global__ void __launch_bounds__(512, 2) gemm_tn256_streamk_mfma32_kernel(GemmTnGroupParams gp, int nsteps, int s_main, int tpx, int tail_each) {
    TN256_STREAMK_BODY(tn256_tile_segment)
}
#if defined(TIC_MEASURE) || defined(TIC_SIM)
#include "gemm_tn256_w4.h"   // the 4-wave experiment (measurement / simulator builds): its own file, so that the sha256 of THIS file, which
                              // profiles/r03_traffic.json is keyed to, changes only with the shipped kernels
#endif
// ... and the same code under its own name for single weight-gradient problems routed here by tic_gemm_tn_bf16 (patch embedding,
// ResNet 1x1 convolutions), so that per-kernel profiler averages of the block launch stay clean
__global__ void __launch_bounds__(512, 2) gemm_tn256_streamk_single_kernel(GemmTnGroupParams gp, int nsteps, int s_main, int tpx, int tail_each) {
    TN256_STREAMK_BODY(tn256_tile_segment16)
}

// Few tiles, long reduction (the 1x1-convolution weight gradients of a deep ResNet stage: C_out x C_in = 1024 x 256 -> 4 tiles over
// 40 960 rows).  As stream-K shares every workgroup ends by adding a whole 256 KiB tile to C with fp32 atomics, and one CU issues those
// at ~5 GB/s: 51 us of an 88 us launch, with nothing else resident on the CU to hide it.  Here every tile is cut into P equal row
// parts (P x tiles <= 256), part p of every tile STORES its partial tile into slice p of a caller-owned slab [P][N][K] (plain stores:
// 10 us per CU) and tn_slab_reduce_kernel adds the slices to C.  Workgroup -> (tile = bid % tiles, part = bid / tiles): the workgroups
// of one part sweep the same rows of both operands at the same time.
__global__ void __launch_bounds__(512, 2) gemm_tn256_parts_slab_kernel(GemmTnGroupParams gp, float* slab, int nsteps, int per) {
    const int tile = TIC_BID_X % gp.total_tiles, part = TIC_BID_X / gp.total_tiles;
    const bf16_t *Ap, *Bp;
    float* Cp;
    int N, K, n0, k0;
    tn_tile_lookup(gp, tile, Ap, Bp, Cp, N, K, n0, k0);
    const int a = part * per, b = (a + per < nsteps) ? a + per : nsteps;   // the host launches only parts with a < nsteps
    tn256_tile_segment16<2>(Ap, Bp, slab + (size_t)part * N * K, N, K, gp.M, n0, k0, a, b);
}
// C[i] += sum over the `parts` slices of slab[part][i]; n4 = N*K / 4
__global__ void __launch_bounds__(256) tn_slab_reduce_kernel(float* __restrict__ C, const float* __restrict__ slab, int parts, long n4) {
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < n4; i += (long)TIC_NBLK_X * 256) {
        f32x4 a = *reinterpret_cast<const f32x4*>(C + i * 4);
        for (int p = 0; p < parts; ++p) a += *reinterpret_cast<const f32x4*>(slab + ((long)p * n4 + i) * 4);
        *reinterpret_cast<f32x4*>(C + i * 4) = a;
    }
}
