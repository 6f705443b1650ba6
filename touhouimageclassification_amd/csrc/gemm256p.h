// gemm256p.h -- the PERSISTENT form of the 256x256 NT kernel of gemm256.h (same tile, same 4-phase main loop, same staged epilogue).
//
// Why: tools/tile_timeline.py (stage stamps inside the real training step, profiles/r02_tile_timeline_*.log) shows every workgroup
// of a launch running in lock-step, and of a K = 1024 tile's 28.6 us only 20.7 us are the K loop: 2.0 us are the prologue (bias
// load, then the first LDS-DMA round trip), 2.0 us the drain + staging pass, 2.2 us the second (store) pass and 1.2-1.6 us lie
// BETWEEN one workgroup's s_endpgm (which waits for its stores) and the first instruction of the next workgroup on that CU.
// One workgroup per CU that loops over its tiles removes the last item and hides most of the first:
//
//   * the K loop's eight 16 KiB half-tile slots are laid out so that the staged bf16 tile ([32, 160 KiB): the top 32 KiB are LDS
//     the non-persistent kernel never used) covers six of them and leaves buffer 0's A0 / B0 slots ([0, 32 KiB)) alone;
//   * right after the staging pass the NEXT tile's bias is in the accumulators and its B0(0), A0(0) are streaming into those two
//     slots; once the second pass has consumed the staged rows 0..127 (one extra barrier) B1(0), A1(0), B0(1), A0(1) stream into
//     [32, 96 KiB); B1(1) follows when the pass is over.  The next K loop starts with its first K tile already in LDS;
//   * the second pass stores through a range-checked buffer resource, so every thread issues the SAME number of stores whatever the
//     row (rows >= M are dropped by the hardware): LDS-DMA and stores share the in-order vmcnt counter, and the counted waits of
//     the first K tile must step over exactly the N2 stores of rows 128..255 that were issued between the prefetches
//     (op order per thread:  B0 A0 | stores of rows 0..127 | B1 A1 B0' A0' | N2 stores | B1' | K tile 0 ...; the wait that needs
//     B1(0), and the waits of phases 0 / 1 / 2 of K tile 0, leave 8 + N2 operations pending; from phase 3 on it is 8 again).
//
// What it does NOT do is run a tile's epilogue UNDER the next tile's main loop: the finished tile is 128 KiB as bf16, the K loop
// needs 128 of the CU's 160 KiB, and 128 accumulators + 80 fragment registers leave no room to park it in the 256-register half
// of a 2-waves-per-SIMD kernel either.  Epilogues that load during the second pass (RESID, PATCH) stay on
// the non-persistent kernel: a load the compiler can see, next to LDS-DMA in flight, is waited for with vmcnt(0), which would
// serialise that pass behind its own stores.
#pragma once
#include "gemm256.h"

#define G256P_LDS_BYTES 163840u
#define G256P_AUX_EARLY 12        // operand-tile rows fetched before the staging pass
#define G256P_STAGE_OFF 32768u    // the staged bf16 tile: [32, 160 KiB)
#define G256P_COLSUM_OFF 131072u  // [128, 136 KiB): outside every K-loop slot; the staged tile has been consumed when it is written
template <int N> struct tic_ic { static constexpr int value = N; };
// K-loop half-tile slots.  which: 0 = A0, 1 = B0, 2 = B1, 3 = A1
// buffer 0's A0 / B0 sit in [0, 32 KiB), below the staged tile: they take the next tile's first two half-tiles while the second
// pass still reads the staged rows; the other six lie inside [32, 128 KiB) in the order the second pass frees them (rows 0..127 =
// [32, 96 KiB) first).  Four slots per 64 KiB window: the ds_read offsets of a window fit the 16-bit immediate of one base VGPR.
TIC_DEV constexpr uint32_t g256p_slot(int buf, int which) {
    return buf == 0 ? (which == 0 ? 0u : which == 1 ? 16384u : which == 2 ? 32768u : 49152u)
                    : (which == 1 ? 65536u : which == 0 ? 81920u : which == 2 ? 98304u : 114688u);
}

// rows [16 K0, 16 K1) of the second pass: 8 columns per thread, bf16 outputs through range-checked buffer stores
template <int EPI, bool NTS, int K0, int K1>
TIC_DEV void g256p_finish_rows(const GemmNtParams& p, tic_rsrc_t rout, tic_rsrc_t rout2, int tid, int m0, int n0, const u32x4 (&aux)[16], float (&cs)[8]) {
    const int c16 = tid & 31, rsub = tid >> 5;
    const int n = n0 + c16 * 8;
    const uint32_t lds0 = G256P_STAGE_OFF + (uint32_t)rsub * 512u + (uint32_t)(((c16 * 2) ^ (rsub << 2)) * 8);
    auto store = [&](tic_rsrc_t r, u32x4 v, uint32_t off) {
        if (NTS) buf_st128_nt(r, v, off, 0);
        else buf_st128(r, v, off, 0);
    };
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const int m = m0 + k * 16 + rsub;
        const u32x4 u = __builtin_bit_cast(u32x4, lds_ld128(lds0 + (uint32_t)k * 8192u));
        const uint32_t o = (uint32_t)(((size_t)m * p.N + n) * 2);   // rows >= M lie past the resource's range: the store is dropped
        const bool live = m < p.M;                                   // (their column sums must not count)
        if (EPI == TIC_EPI_BF16) {
            store(rout, u, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cs[2 * j] += live ? bf_lo(u[j]) : 0.f;
                cs[2 * j + 1] += live ? bf_hi(u[j]) : 0.f;
            }
        } else if (EPI == TIC_EPI_GELU) {
            u32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = pack2bf(gelu_erf(bf_lo(u[j])), gelu_erf(bf_hi(u[j])));
            store(rout, u, o);
            store(rout2, g, o);
        } else if (EPI == TIC_EPI_GELU_DG) {
            u32x4 g, dg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const GeluPair r = gelu_pair(f32x2{bf_lo(u[j]), bf_hi(u[j])});
                g[j] = pack2bf(r.g[0], r.g[1]);
                dg[j] = pack2bf(r.dg[0], r.dg[1]);
            }
            store(rout, dg, o);
            store(rout2, g, o);
        } else {   // DGELU / MULAUX (aux of rows >= M was fetched as zero)
            const u32x4 a = aux[k];
            u32x4 d;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f0 = (EPI == TIC_EPI_DGELU) ? gelu_erf_grad(bf_lo(a[j])) : bf_lo(a[j]);
                const float f1 = (EPI == TIC_EPI_DGELU) ? gelu_erf_grad(bf_hi(a[j])) : bf_hi(a[j]);
                const float d0 = bf_lo(u[j]) * f0, d1 = bf_hi(u[j]) * f1;
                d[j] = pack2bf(d0, d1);
                cs[2 * j] += live ? d0 : 0.f;
                cs[2 * j + 1] += live ? d1 : 0.f;
            }
            store(rout, d, o);
        }
    }
}

template <int EPI>
__global__ void __launch_bounds__(512, 2) gemm_nt256p_kernel(GemmNtParams p) {
    static_assert(EPI == TIC_EPI_BF16 || EPI == TIC_EPI_GELU || EPI == TIC_EPI_GELU_DG || EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX,
                  "epilogues that load during the second pass stay on gemm_nt256_kernel");
    constexpr bool HAS_AUX = (EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX);
    constexpr bool HAS_COLSUM = (EPI == TIC_EPI_BF16 || EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX);
    constexpr int NST = (EPI == TIC_EPI_GELU || EPI == TIC_EPI_GELU_DG) ? 2 : 1;   // stores per row of the second pass
    constexpr int N2 = 8 * NST;                                                     // stores of rows 128..255 per thread
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wr = w >> 2, wc = w & 3;
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / 256, ntiles = tiles_m * tiles_n;
    const tic_rsrc_t ra = make_rsrc(p.A, (uint32_t)((size_t)p.M * p.K * 2));
    const tic_rsrc_t rb = make_rsrc(p.B, (uint32_t)((size_t)p.N * p.K * 2));
    // (the output resources are rebuilt in every epilogue: 8 SGPRs that need not live through the K loop -- the kernel is at the
    // SGPR limit too, and a spilled SGPR costs a VGPR lane)
    const uint32_t slot_log = (uint32_t)(l & 7) ^ ((((uint32_t)l >> 4) & 3u) << 1);
    const uint32_t row8 = (uint32_t)p.K * 16u;   // 8 rows in bytes
    const int nk = p.K / 64;
    const int gm = p.gm > 0 ? p.gm : 8;

    auto dma_base = [&](int m0, int n0, uint32_t& voa0, uint32_t& vob0) {
        voa0 = (uint32_t)(((size_t)(m0 + 2 * w * 8 + (l >> 3)) * p.K + slot_log * 8) * 2);
        vob0 = (uint32_t)(((size_t)(n0 + 2 * w * 8 + (l >> 3)) * p.K + slot_log * 8) * 2);
    };
    // one half-tile (2 pieces of this wave) of K tile kt into slot (BUF, WHICH); K tiles >= nk become zero fills
    auto issue = [&](auto bufc, auto whichc, int kt, uint32_t voa0, uint32_t vob0) {
        constexpr int BUF = decltype(bufc)::value, WHICH = decltype(whichc)::value;
        constexpr bool isA = (WHICH == 0 || WHICH == 3);
        constexpr int h = (WHICH >= 2) ? 1 : 0;
        const bool live = kt < nk;
        const uint32_t soff = live ? (uint32_t)kt * 128u : 0u;
        const uint32_t base = g256p_slot(BUF, WHICH) + (uint32_t)(2 * w) * 1024u;
        const uint32_t v0 = (isA ? voa0 : vob0) + (uint32_t)h * 16u * row8;
        glds16(isA ? ra : rb, base, live ? v0 : 0xFFFFFFF0u, soff);
        glds16(isA ? ra : rb, base + 1024u, live ? v0 + row8 : 0xFFFFFFF0u, soff);
    };

    const uint32_t sw = (((uint32_t)(l & 15) >> 1) & 3u) << 1;
    uint32_t fo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = (uint32_t)(l & 15) * 128u + ((((uint32_t)ks * 4 + ((uint32_t)l >> 4)) ^ sw) * 16u);
    const uint32_t a_row0 = (uint32_t)wr * 64, b_row0 = (uint32_t)wc * 32;

    f32x4 acc[2][2][4][2];   // [i][j][mt][nt]
    bf16x8 fa0[4][2], fa1[4][2], fbx[2][2], fby[2][2];
    auto load_a = [&](uint32_t slot, bf16x8 (&fa)[4][2], int ks) {   // one k-half of an A half-tile
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) fa[mt][ks] = lds_ld128(slot + (a_row0 + (uint32_t)mt * 16) * 128u + fo[ks]);
    };
    auto load_b = [&](uint32_t slot, bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[nt][ks] = lds_ld128(slot + (b_row0 + (uint32_t)nt * 16) * 128u + fo[ks]);
    };
    auto mma = [&](int i, int j, const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = mfma16(fb[nt][ks], fa[mt][ks], acc[i][j][mt][nt]);
    };
    // accumulators start at the bias of their columns (nullptr for the epilogues that have none)
    auto load_bias = [&](int n0, f32x4 (&b4)[2][2]) {
        float zero_v = 0.f;
#ifndef TIC_SIM
        asm volatile("" : "+v"(zero_v));   // opaque zero (see gemm256.h)
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = n0 + j * 128 + wc * 32 + nt * 16 + 4 * (l >> 4);
                b4[j][nt] = f32x4{zero_v, zero_v, zero_v, zero_v};
                if (!HAS_AUX && p.bias) b4[j][nt] = *reinterpret_cast<const f32x4*>(p.bias + col);   // (DGELU / MULAUX have no bias)
            }
    };
    auto init_acc = [&](const f32x4 (&b4)[2][2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[i][j][mt][nt] = b4[j][nt];
    };

    // one K tile in buffer CUR (see gemm256.h for the phase / hazard table).  The counted waits of phases 0..2 leave 8 operations
    // pending in the steady state and 8 + N2 in the FIRST K tile of a tile (header); s_waitcnt takes an immediate, so that is a
    // wave-uniform branch around two waits (next to a barrier, outside the MFMA / ds_read interleave)
    auto wait_phase = [&](bool first) {
        if (first) wait_vmcnt<8 + N2>();
        else wait_vmcnt<8>();
    };
    auto tile = [&](auto curc, bool first, int kt, uint32_t voa0, uint32_t vob0, bf16x8 (&fbp)[2][2], bf16x8 (&fbq)[2][2]) {
        constexpr int CUR = decltype(curc)::value, NXT = CUR ^ 1;
        // ---- phase 0: Q00 = A0 x B0 ; reads A0(t) k-half 1, B1(t)
        issue(tic_ic<NXT>{}, tic_ic<3>{}, kt + 1, voa0, vob0);   // A1(t+1)
        wait_phase(first);
        g256_barrier();
        prio_hi();
        load_a(g256p_slot(CUR, 0), fa0, 1);
        load_b(g256p_slot(CUR, 2), fbq);
        mma(0, 0, fa0, fbp);
        G256_INTERLEAVE(8);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
        // ---- phase 1: Q01 = A0 x B1 ; reads A1(t) k-half 0
        issue(tic_ic<CUR>{}, tic_ic<1>{}, kt + 2, voa0, vob0);   // B0(t+2)
        wait_phase(first);
        g256_barrier();
        prio_hi();
        load_a(g256p_slot(CUR, 3), fa1, 0);
        mma(0, 1, fa0, fbq);
        G256_INTERLEAVE(4);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
        // ---- phase 2: Q11 = A1 x B1 ; reads A1(t) k-half 1
        issue(tic_ic<CUR>{}, tic_ic<0>{}, kt + 2, voa0, vob0);   // A0(t+2)
        wait_phase(first);
        g256_barrier();
        prio_hi();
        load_a(g256p_slot(CUR, 3), fa1, 1);
        mma(1, 1, fa1, fbq);
        G256_INTERLEAVE(4);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
        // ---- phase 3: Q10 = A1 x B0 ; reads A0(t+1) k-half 0, B0(t+1)
        issue(tic_ic<CUR>{}, tic_ic<2>{}, kt + 2, voa0, vob0);   // B1(t+2)
        wait_vmcnt<8>();
        g256_barrier();
        prio_hi();
        load_a(g256p_slot(NXT, 0), fa0, 0);
        load_b(g256p_slot(NXT, 1), fbq);
        mma(1, 0, fa1, fbp);
        G256_INTERLEAVE(8);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
    };

    int t = TIC_BID_X;
    if (t >= ntiles) return;   // block-uniform
    int tm, tn;
    tile_coords(t, ntiles, tiles_m, tiles_n, tm, tn, gm);
    int m0 = tm * 256, n0 = tn * 256;
    uint32_t voa0, vob0;
    dma_base(m0, n0, voa0, vob0);
    {
        f32x4 b4[2][2];
        load_bias(n0, b4);
        init_acc(b4);
    }
    // first tile of this workgroup: the whole prologue, in the steady-state issue order B0, A0, B1, A1, with N2 out-of-range
    // (dropped, but counted) stores where the later tiles have their second-pass stores -- ONE op sequence, one set of wait counts
    issue(tic_ic<0>{}, tic_ic<1>{}, 0, voa0, vob0);
    issue(tic_ic<0>{}, tic_ic<0>{}, 0, voa0, vob0);
    issue(tic_ic<0>{}, tic_ic<2>{}, 0, voa0, vob0);
    issue(tic_ic<0>{}, tic_ic<3>{}, 0, voa0, vob0);
    issue(tic_ic<1>{}, tic_ic<1>{}, 1, voa0, vob0);
    issue(tic_ic<1>{}, tic_ic<0>{}, 1, voa0, vob0);
#pragma unroll
    for (int i = 0; i < N2; ++i) buf_st128(ra, u32x4{0u, 0u, 0u, 0u}, 0xFFFFF000u + 16u * (uint32_t)i, 0);   // DISTINCT offsets: identical stores are merged into one
    issue(tic_ic<1>{}, tic_ic<2>{}, 1, voa0, vob0);
    wait_vmcnt<8 + N2>();
#if defined(TIC_MEASURE) && !defined(TIC_SIM)
#define G256P_STAMP(i)                                                                                       \
    do {                                                                                                     \
        if (p.stamps && TIC_TID == 0) p.stamps[(size_t)t * 8 + (i)] = __builtin_amdgcn_s_memrealtime();      \
    } while (0)
#else
#define G256P_STAMP(i) do { } while (0)
#endif
    for (;;) {
        G256P_STAMP(0);
        g256_barrier();
        load_a(g256p_slot(0, 0), fa0, 0);
        load_b(g256p_slot(0, 1), fbx);
        wait_lgkmcnt0();
        if (wr == 1) g256_barrier();   // the two wave groups run one barrier apart from here on
        G256P_STAMP(1);

#pragma nounroll
        for (int kt = 0; kt < nk; kt += 2) {
            tile(tic_ic<0>{}, kt == 0, kt, voa0, vob0, fbx, fby);
            tile(tic_ic<1>{}, false, kt + 1, voa0, vob0, fby, fbx);
        }
        G256P_STAMP(2);
        wait_vmcnt0();   // only zero fills are pending here

        // ---- epilogue of tile t, prologue of tile t + gridDim.x ---------------------------------------------------------------
        const int le = lane_id_fresh(), tide = w * 64 + le;   // re-derived: threadIdx-derived registers need not survive the K loop
        const int t_next = t + TIC_NBLK_X;
        const bool has_next = t_next < ntiles;   // block-uniform
        int m0n = 0, n0n = 0;
        if (has_next) {
            tile_coords(t_next, ntiles, tiles_m, tiles_n, tm, tn, gm);
            m0n = tm * 256;
            n0n = tn * 256;
        }
        u32x4 auxr[16];
        if (HAS_AUX) {   // G256P_AUX_EARLY rows now (32 registers beside the accumulators), the rest after the staging pass
            if (p.nt & 2) g256_fetch_aux<true, 0, G256P_AUX_EARLY>(p, tide, m0, n0, auxr);
            else g256_fetch_aux<false, 0, G256P_AUX_EARLY>(p, tide, m0, n0, auxr);
        }
        // the next tile's bias: requested before the staging pass (its latency hides under that pass)
        f32x4 bnext[2][2];
        if (has_next) load_bias(HAS_AUX ? 0 : n0n, bnext);   // (the operand-tile epilogues have no bias: the host passes nullptr, this is the opaque zero)
        sched_fence();
        if (wr == 0) g256_barrier();   // re-balance the stagger
        g256_barrier();                // every wave's LDS reads and DMA writes have retired: the tile buffers are free
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = (g >> 1) * 128 + wc * 32 + (g & 1) * 16 + 4 * (le >> 4);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int row = (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (le & 15);
                const f32x4 v = acc[r >> 2][g >> 1][r & 3][g & 1];
                lds_st64(G256P_STAGE_OFF + g256_stage_off(row, col >> 2), __builtin_bit_cast(bf16x4, u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])}));
            }
        }
        wait_lgkmcnt0();
        g256_barrier();   // the staged tile is visible; the accumulators are dead
        G256P_STAMP(3);
        sched_fence();
        if (HAS_AUX) {
            if (p.nt & 2) g256_fetch_aux<true, G256P_AUX_EARLY, 16>(p, tide, m0, n0, auxr);
            else g256_fetch_aux<false, G256P_AUX_EARLY, 16>(p, tide, m0, n0, auxr);
        }
        uint32_t voa0n = 0, vob0n = 0;
        auto consume_bias = [&]() {
#ifndef TIC_SIM
            // the bias loads are consumed HERE (the wait hipcc puts in front of this statement finds only them and the operand-tile
            // loads in flight); left to the compiler, the copy into the accumulators sank to the top of the next K loop together
            // with an s_waitcnt vmcnt(0) that drained the prefetches issued below
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) asm volatile("" : "+v"(bnext[j][nt]));
#endif
            init_acc(bnext);
        };
        if (has_next) {
            if (!HAS_AUX) consume_bias();   // (the operand-tile epilogues need the registers in the second pass; their "bias" is a zero)
            dma_base(m0n, n0n, voa0n, vob0n);
            sched_fence();
            issue(tic_ic<0>{}, tic_ic<1>{}, 0, voa0n, vob0n);   // B0(0), A0(0) of the next tile -> [0, 32 KiB)
            issue(tic_ic<0>{}, tic_ic<0>{}, 0, voa0n, vob0n);
        }
        const tic_rsrc_t rout = make_rsrc(p.out, (uint32_t)((size_t)p.M * p.N * 2));
        const tic_rsrc_t rout2 = make_rsrc(NST == 2 ? (const void*)p.out2 : (const void*)p.out, (uint32_t)((size_t)p.M * p.N * 2));
        float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.nt & 1) g256p_finish_rows<EPI, true, 0, 8>(p, rout, rout2, tide, m0, n0, auxr, cs);
        else g256p_finish_rows<EPI, false, 0, 8>(p, rout, rout2, tide, m0, n0, auxr, cs);
        wait_lgkmcnt0();
        g256_barrier();   // staged rows 0..127 have been read by every wave: [32, 96 KiB) is free
        if (has_next) {
            issue(tic_ic<0>{}, tic_ic<2>{}, 0, voa0n, vob0n);   // B1(0), A1(0), B0(1), A0(1)
            issue(tic_ic<0>{}, tic_ic<3>{}, 0, voa0n, vob0n);
            issue(tic_ic<1>{}, tic_ic<1>{}, 1, voa0n, vob0n);
            issue(tic_ic<1>{}, tic_ic<0>{}, 1, voa0n, vob0n);
        }
        if (p.nt & 1) g256p_finish_rows<EPI, true, 8, 16>(p, rout, rout2, tide, m0, n0, auxr, cs);
        else g256p_finish_rows<EPI, false, 8, 16>(p, rout, rout2, tide, m0, n0, auxr, cs);
        wait_lgkmcnt0();
        g256_barrier();   // the whole staged tile has been read: [96, 160 KiB) is free
        if (HAS_COLSUM && p.colsum) {   // kernel-argument condition: block-uniform
            const int c16 = tide & 31;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                cs[j] += shfl_xor(cs[j], 32);
                if (le < 32) lds_stf(G256P_COLSUM_OFF + (uint32_t)(w * 256 + c16 * 8 + j) * 4u, cs[j]);
            }
            wait_lgkmcnt0();
            g256_barrier();
            if (tide < 256) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) s += lds_ldf(G256P_COLSUM_OFF + (uint32_t)(ww * 256 + tide) * 4u);
                atomic_addf(p.colsum + n0 + tide, s);
            }
        }
#if defined(TIC_MEASURE) && !defined(TIC_SIM)
        if (p.stamps && TIC_TID == 0) {
            p.stamps[(size_t)t * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
            p.stamps[(size_t)t * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        }
#endif
        G256P_STAMP(4);
        if (!has_next) break;
        if (HAS_AUX) consume_bias();
        issue(tic_ic<1>{}, tic_ic<2>{}, 1, voa0n, vob0n);   // B1(1)
        wait_vmcnt<8 + N2>();   // B0(0), A0(0), B1(0) of the next tile have landed (header: the N2 stores of rows 128..255 are younger)
        G256P_STAMP(5);   // (index t: the tile that just ended) -> stamp 0 of the next tile follows immediately
        t = t_next;
        m0 = m0n;
        n0 = n0n;
        voa0 = voa0n;
        vob0 = vob0n;
    }
}
