// gemm256.h -- the deep-pipelined bf16 GEMM for the big token-matrix products (gfx950).
//
//   C[M,N] = A[M,K] . B[N,K]^T (+ the epilogues of gemm.h),  N % 256 == 0, K % 64 == 0, any M.
//
// 256x256 block tile, 64-deep K tile, 512 threads = 8 waves (2 along M x 4 along N), one block per CU
// (128 KiB LDS = 2 K-tile buffers x {A 256x64, B 256x64} bf16).  Each wave owns a 128x64 output as four
// 64x32 quadrants Q(i,j): rows i*128 + wr*64 + [0,64), columns j*128 + wc*32 + [0,32) -- interleaved so
// that quadrant (i,j) needs only A half-tile i and B half-tile j (a half-tile = 128 rows x 64 k = 16 KiB =
// 2 LDS-DMA pieces per thread).  One K tile = 4 phases in the order Q00, Q01, Q11, Q10; a phase is
//     load segment : issue the LDS-DMA of ONE half-tile (2 pieces per wave), s_waitcnt vmcnt(8)
//     s_barrier
//     MFMA segment : 16 x v_mfma_f32_16x16x32_bf16 (64x32 quadrant, K = 64) with the ds_reads of the NEXT phase's
//                    fragments slotted into the MFMA gaps, s_waitcnt lgkmcnt(0)
//     s_barrier
// The two wave groups (wr = 0 / 1 -- the two waves that share each SIMD) run ONE barrier apart, so one
// group's MFMA segment overlaps the other's load segment.  Measured with parts of the loop compiled out
// (tools/gemm_dbg.py, M 32702 N K 4096): MFMA alone 506 us, LDS-DMA alone 553 us, fragment reads alone 265 us -- but
// with the reads in the LOAD segment (next to the DMA issue, 100-185 cycles per piece) that segment outlasted
// the other group's 256-cycle MFMA segment and the three together took 814 us; moving the reads under the MFMAs
// leaves the load segment with the DMA issue and the counted wait only.
// LDS-DMA stays in flight across barriers (raw s_barrier, never __syncthreads).  Every load segment issues one
// half-tile and waits for the one issued four phases earlier (4 half-tiles = 64 KiB per CU always in flight):
//     tile t phase 0: issue A1(t+1)   phase 1: B0(t+2)   phase 2: A0(t+2)   phase 3: B1(t+2)
//     MFMA segment of phase 0 reads A0(t) k-half 1 + B1(t), phase 1: A1(t) k-half 0, phase 2: A1(t) k-half 1,
//     phase 3: A0(t+1) k-half 0 + B0(t+1)   (B0 stays in registers from phase 0 to phase 3; fragment peak 80 VGPRs)
// K tiles past the end are issued with an out-of-range offset (the hardware range check turns them into zero
// fills with no memory traffic) so the steady-state body and its vmcnt counts are the same for every tile.
// Hazard bookkeeping (interval = time between consecutive barriers; group 0 runs the load segment LS(P) of global
// phase P in interval 2P and its MFMA segment MS(P) in 2P+1, group 1 one interval later):
//   RAW  a half-tile read in MS(q) (group 0: interval 2q+1) must have been waited for by EVERY wave that issued a
//        piece of it no later than interval 2q, i.e. in LS(q-1) (group 1's LS(q-1) is interval 2q-1).  With
//        vmcnt(8) the half-tile issued in LS(p-4) is complete in LS(p):
//          B1(t): issued LS(4t-5), complete LS(4t-1), read MS(4t)      A1(t): LS(4t-4) / LS(4t) / MS(4t+1), MS(4t+2)
//          B0(t+1): LS(4t-3) / LS(4t+1) / MS(4t+3)                      A0(t+1): LS(4t-2) / LS(4t+2) / MS(4t+3), MS(4t+4)
//   WAR  reads issued in MS(q) are retired (lgkmcnt(0)) before that segment's closing barrier, by both groups no
//        later than interval 2q+2; the slot may be re-filled from LS(q+2) on (group 0: interval 2q+4):
//          A0 slot: last read MS(4t), refilled LS(4t+2)     B0: MS(4t-1) -> LS(4t+1)
//          B1 slot: read MS(4t),      refilled LS(4t+3)     A1: last read MS(4t+2) -> LS(4t+4)
#pragma once
#include "gemm.h"

#define G256_BUF_BYTES 65536u   // A 32 KiB + B 32 KiB
#define G256_LDS_BYTES (2u * G256_BUF_BYTES)
#define G256_NT_LDS_BYTES (G256_LDS_BYTES + 8192u)   // + [8 waves][256] fp32 column-sum partials of the staged epilogue

// measurement build (-DTIC_MEASURE, libtic_hip_dbg.so): lane 0 of every workgroup stamps the constant 100 MHz clock at the stage
// boundaries of its tile into p.stamps (tools/tile_timeline.py).  The product build compiles none of it.
#if defined(TIC_MEASURE) && !defined(TIC_SIM)
#define G256_STAMP(i)                                                                                            \
    do {                                                                                                         \
        if (p.stamps && TIC_TID == 0) p.stamps[(size_t)TIC_BID_X * 8 + (i)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#define G256_STAMP_ID()                                                                                          \
    do {                                                                                                         \
        if (p.stamps && TIC_TID == 0) {                                                                          \
            p.stamps[(size_t)TIC_BID_X * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   /* HW_ID */ \
            p.stamps[(size_t)TIC_BID_X * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  /* XCC_ID */ \
        }                                                                                                        \
    } while (0)
#else
#define G256_STAMP(i) do { } while (0)
#define G256_STAMP_ID() do { } while (0)
#endif

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
// The extra operand of the DGELU / MULAUX epilogues (the saved pre-activation / derivative tile, 128 KiB per workgroup) is fetched
// in ONE go BEFORE the staging pass: once the K loop has ended the 80 fragment registers are free, so all 16 rows of a thread
// (64 registers) can be in flight while the accumulators are packed and parked in LDS.  Fetched 4 rows deep inside the second pass
// (the round-1 form) the tile's operand arrived latency-bound: the second pass of fc2^T * gelu' took 9.0 us against 2.2 us for a
// plain bf16 tile (tools/tile_timeline.py --in-step).
#define G256_AUX_EARLY 12   // rows fetched before the staging pass (48 registers beside the 128 accumulators); the last 4 follow it
template <bool NTL, int K0, int K1>
TIC_DEV void g256_fetch_aux(const GemmNtParams& p, int tid, int m0, int n0, u32x4 (&aux)[16]) {
    const int c16 = tid & 31, rsub = tid >> 5;
    const int n = n0 + c16 * 8;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const int m = m0 + k * 16 + rsub;
        aux[k] = (m < p.M) ? ld_u4<NTL>(p.aux + (size_t)m * p.N + n) : u32x4{0u, 0u, 0u, 0u};
    }
}

template <int EPI, bool NTS, bool NTL>
TIC_DEV void g256_finish_bf16(const GemmNtParams& p, int tid, int m0, int n0, u32x4 (&aux)[16]) {
    constexpr bool HAS_COLSUM = (EPI == TIC_EPI_BF16 || EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX);
    if (EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX || EPI == TIC_EPI_ADDAUX) g256_fetch_aux<NTL, G256_AUX_EARLY, 16>(p, tid, m0, n0, aux);
    constexpr int PF = 4;
    const int c16 = tid & 31, rsub = tid >> 5;
    const int n = n0 + c16 * 8;
    const uint32_t lds0 = (uint32_t)rsub * 512u + (uint32_t)(((c16 * 2) ^ (rsub << 2)) * 8);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 16 / PF; ++b) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int k = b * PF + i;
            const int m = m0 + k * 16 + rsub;
            const u32x4 u = __builtin_bit_cast(u32x4, lds_ld128(lds0 + (uint32_t)k * 8192u));
            if (m >= p.M) continue;
            const size_t o = (size_t)m * p.N + n;
            if (EPI == TIC_EPI_BF16) {
                st_u4<NTS>(p.out + o, u);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cs[2 * j] += bf_lo(u[j]);
                    cs[2 * j + 1] += bf_hi(u[j]);
                }
            } else if (EPI == TIC_EPI_GELU) {
                u32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = pack2bf(gelu_erf(bf_lo(u[j])), gelu_erf(bf_hi(u[j])));
                st_u4<NTS>(p.out + o, u);
                st_u4<NTS>(p.out2 + o, g);
            } else if (EPI == TIC_EPI_GELU_ONLY) {
                u32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = pack2bf(gelu_erf(bf_lo(u[j])), gelu_erf(bf_hi(u[j])));
                st_u4<NTS>(p.out2 + o, g);
            } else if (EPI == TIC_EPI_GELU_DG) {
                u32x4 g, dg;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const GeluPair r = gelu_pair(f32x2{bf_lo(u[j]), bf_hi(u[j])});
                    g[j] = pack2bf(r.g[0], r.g[1]);
                    dg[j] = pack2bf(r.dg[0], r.dg[1]);
                }
                st_u4<NTS>(p.out + o, dg);
                st_u4<NTS>(p.out2 + o, g);
            } else {   // DGELU / MULAUX / ADDAUX
                const u32x4 a = aux[k];
                u32x4 d;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float f0 = (EPI == TIC_EPI_DGELU) ? gelu_erf_grad(bf_lo(a[j])) : bf_lo(a[j]);
                    const float f1 = (EPI == TIC_EPI_DGELU) ? gelu_erf_grad(bf_hi(a[j])) : bf_hi(a[j]);
                    const float d0 = (EPI == TIC_EPI_ADDAUX) ? bf_lo(u[j]) + f0 : bf_lo(u[j]) * f0;
                    const float d1 = (EPI == TIC_EPI_ADDAUX) ? bf_hi(u[j]) + f1 : bf_hi(u[j]) * f1;
                    d[j] = pack2bf(d0, d1);
                    cs[2 * j] += d0;
                    cs[2 * j + 1] += d1;
                }
                st_u4<NTS>(p.out + o, d);
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
// The extra operand (the fp32 residual tile, 256 KiB per workgroup; the position rows for PATCH) is requested like the bf16
// operand tile above: G256_EX_EARLY of a thread's 32 rows before the staging pass, the rest right after it -- the whole tile is
// in flight when the second pass starts.  Fetched 4 rows deep inside the pass (round 1) the pass ran at the latency-bound rate
// of 32 KiB in flight per CU: 18.2 us for 512 KiB (tools/tile_timeline.py --in-step).
#define G256_EX_EARLY 16
template <int EPI, bool NTL, int K0, int K1>
TIC_DEV void g256_fetch_ex(const GemmNtParams& p, int tid, int m0, int n0, f32x4 (&ex)[32]) {
    const int c8 = tid & 63, rsub = tid >> 6;
    const int n = n0 + c8 * 4;
#pragma unroll
    for (int k = K0; k < K1; ++k) {
        const int m = m0 + k * 8 + rsub;
        const float* src = (EPI == TIC_EPI_RESID) ? p.resid + (size_t)m * p.N + n : p.rowtab + (size_t)(1 + m % p.patches) * p.N + n;
        ex[k] = (m < p.M) ? ld_f4<NTL && EPI == TIC_EPI_RESID>(src) : f32x4{0.f, 0.f, 0.f, 0.f};   // the position table is re-read: never NT
    }
}

template <int EPI, bool NTS, bool NTL>
TIC_DEV void g256_finish_f32(const GemmNtParams& p, int tid, int m0, int n0, f32x4 (&ex)[32]) {
    const int c8 = tid & 63, rsub = tid >> 6;
    const int n = n0 + c8 * 4;
    g256_fetch_ex<EPI, NTL, G256_EX_EARLY, 32>(p, tid, m0, n0, ex);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int row = k * 8 + rsub, m = m0 + row;
        const u32x2 u = __builtin_bit_cast(u32x2, lds_ld64(g256_stage_off(row, c8)));
        if (m >= p.M) continue;
        const f32x4 e = ex[k];
        const f32x4 y = f32x4{bf_lo(u[0]) + e[0], bf_hi(u[0]) + e[1], bf_lo(u[1]) + e[2], bf_hi(u[1]) + e[3]};
        size_t orow = (size_t)m;
        if (EPI == TIC_EPI_PATCH) {
            const int img = m / p.patches;
            orow = (size_t)img * (p.patches + 1) + 1 + (m - img * p.patches);
        }
        st_f4<NTS>(p.out_f32 + orow * p.N + n, y);
    }
}

// DBG (measurement builds only, EPI_BF16): bit 0 = no LDS-DMA, bit 1 = no fragment ds_reads, bit 2 = no MFMA, bit 3 = vmcnt(12)
// in the loop (6 half-tiles in flight: a RACE, timing only), bit 4 = ONE barrier per phase and no wave-group stagger (a RACE too) -- isolates which
// of the three pipes bounds the main loop (tic_set_option("gemm_dbg")); results are garbage by construction.
// SPLITK = 1: the split-K form for launches with FEW tiles and a LONG reduction (ViT-L at the batch sizes the reference fine-tunes
// at: 32 images = 25 row tiles x 4 column tiles = 100 workgroups for the N = 1024, K = 4096 / 3072 GEMMs, a third of the chip, 86
// us of K loop each).  p.split (2 or 4) workgroups share a tile; workgroup (part q, tile t) = block q * tiles + t reduces K tiles
// [q nk / split, (q + 1) nk / split).  Parts 0 .. split - 2 are PRODUCERS: accumulators from zero, written as they stand (each
// thread its own 32 float4, fully coalesced) into slab[t][q], then flag[t][q] = epoch (tic_prims.h flag_publish).  The last part is
// the CONSUMER: accumulators from the bias, after its K loop it waits for each flag, adds the slab (same thread, same registers:
// no layout question) and runs the ordinary epilogue.  All split x tiles <= 256 workgroups are resident at once (one per CU), so
// the consumer never waits for a workgroup that cannot start; its poll is bounded anyway.  One rounding to bf16, as without the
// split; only the fp32 summation order differs.
template <int EPI, int DBG = 0, int SPLITK = 0>
__global__ void __launch_bounds__(512, 2) gemm_nt256_kernel(GemmNtParams p) {
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wr = w >> 2, wc = w & 3;
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / 256;
    int tm, tn;
    int tile_id = TIC_BID_X, part = 0, nparts = 1;
    if (SPLITK) {
        nparts = p.split;
        part = TIC_BID_X / (tiles_m * tiles_n);
        tile_id = TIC_BID_X - part * (tiles_m * tiles_n);
    }
    tile_coords(tile_id, tiles_m * tiles_n, tiles_m, tiles_n, tm, tn, p.gm > 0 ? p.gm : 8);
    const int m0 = tm * 256, n0 = tn * 256;
    G256_STAMP(0);
    G256_STAMP_ID();
#ifndef TIC_SIM
    // experiment: delay every other workgroup of the first wave of tiles so that epilogues (HBM bursts) of one half of the
    // CUs fall into the main loops of the other half
    // (bits 8..15 of p.stagger: number of groups G, default 2: group g = (index within the XCD) % G sleeps g x rounds)
    if ((p.stagger & 255) > 0 && TIC_BID_X < 256) {   // the host zeroes it for epilogues outside "gemm_stagger_mask"
        const int G = (p.stagger >> 8) > 1 ? (p.stagger >> 8) : 2;
        const int rounds = (p.stagger & 255) * ((TIC_BID_X >> 3) % G);
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const tic_rsrc_t ra = make_rsrc(p.A, (uint32_t)((size_t)p.M * p.K * 2));
    const tic_rsrc_t rb = make_rsrc(p.B, (uint32_t)((size_t)p.N * p.K * 2));

    // ---- LDS-DMA: half-tile h of operand X = 16 one-KiB pieces (8 rows each); this wave moves pieces 2w, 2w+1
    const uint32_t slot_log = (uint32_t)(l & 7) ^ ((((uint32_t)l >> 4) & 3u) << 1);
    // ONE long-lived offset VGPR per operand (piece 0 of half 0); the other pieces add a wave-uniform row distance when
    // they are issued.  The row stays in the VGPR offset because only that offset is range-checked (rows >= M read 0).
    const uint32_t voa0 = (uint32_t)(((size_t)(m0 + 2 * w * 8 + (l >> 3)) * p.K + slot_log * 8) * 2);
    const uint32_t vob0 = (uint32_t)(((size_t)(n0 + 2 * w * 8 + (l >> 3)) * p.K + slot_log * 8) * 2);
    const uint32_t row8 = (uint32_t)p.K * 16u;   // 8 rows in bytes
    // which: 0 = A0, 1 = B0, 2 = B1, 3 = A1.  K tiles >= nk become zero fills (see header).
    const int nk = SPLITK ? (p.K / 64) / nparts : p.K / 64;   // K tiles of THIS workgroup, starting at kt0
    const int kt0 = SPLITK ? part * nk : 0;
    auto issue = [&](int buf, int kt, int which) {
        if (DBG & 1) return;
        const bool isA = (which == 0 || which == 3);
        const int h = (which >= 2) ? 1 : 0;
        const bool live = kt < nk;
        const uint32_t soff = live ? (uint32_t)(kt0 + kt) * 128u : 0u;
        const uint32_t base = (uint32_t)buf * G256_BUF_BYTES + (isA ? 0u : 32768u) + (uint32_t)h * 16384u + (uint32_t)(2 * w) * 1024u;
        const uint32_t v0 = (isA ? voa0 : vob0) + (uint32_t)h * 16u * row8;
        glds16(isA ? ra : rb, base, live ? v0 : 0xFFFFFFF0u, soff);
        glds16(isA ? ra : rb, base + 1024u, live ? v0 + row8 : 0xFFFFFFF0u, soff);
    };

    // ---- fragment offsets inside a buffer ----------------------------------------------------------------
    const uint32_t sw = (((uint32_t)(l & 15) >> 1) & 3u) << 1;
    uint32_t fo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = (uint32_t)(l & 15) * 128u + ((((uint32_t)ks * 4 + ((uint32_t)l >> 4)) ^ sw) * 16u);
    const uint32_t a_row0 = (uint32_t)wr * 64, b_row0 = (uint32_t)wc * 32;

    // accumulators start at the bias of their columns (u = bf16(bias + sum): the add costs nothing and no bias
    // registers live through the loop or the staging pass)
    // (the host passes bias = nullptr for the epilogues that have none: DGELU / MULAUX)
    f32x4 acc[2][2][4][2];   // [i][j][mt][nt]
    float zero_v = 0.f;
#ifndef TIC_SIM
    asm volatile("" : "+v"(zero_v));   // opaque zero: a literal 0 makes hipcc peel the first K tile (C = 0 MFMA forms) and spill
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int col = n0 + j * 128 + wc * 32 + nt * 16 + 4 * (l >> 4);
            f32x4 b4 = f32x4{zero_v, zero_v, zero_v, zero_v};
            if (p.bias && (!SPLITK || part == nparts - 1)) b4 = *reinterpret_cast<const f32x4*>(p.bias + col);   // split-K: the consumer only
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[i][j][mt][nt] = b4;
        }
    // fragment registers: A0 / A1 of the current tile and two B sets whose roles (B0 | B1) swap every K tile
    bf16x8 fa0[4][2], fa1[4][2], fbx[2][2], fby[2][2];   // [mt][ks], [nt][ks]
    if (DBG & 2) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            fa0[x][0] = fa0[x][1] = fa1[x][0] = fa1[x][1] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
            fbx[x & 1][x >> 1] = fby[x & 1][x >> 1] = bf16x8{1, 2, 3, 4, 5, 6, 7, 8};
        }
    }

    auto load_a = [&](uint32_t bufb, int i, bf16x8 (&fa)[4][2], int ks) {   // one k-half (32 of the 64 k) of an A half-tile
        if (DBG & 2) return;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) fa[mt][ks] = lds_ld128(bufb + ((uint32_t)i * 128 + a_row0 + (uint32_t)mt * 16) * 128u + fo[ks]);
    };
    auto load_b = [&](uint32_t bufb, int j, bf16x8 (&fb)[2][2]) {
        if (DBG & 2) return;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[nt][ks] = lds_ld128(bufb + 32768u + ((uint32_t)j * 128 + b_row0 + (uint32_t)nt * 16) * 128u + fo[ks]);
    };
    auto mma = [&](int i, int j, const bf16x8 (&fa)[4][2], const bf16x8 (&fb)[2][2]) {
        if (DBG & 4) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[i][j][mt][0][0] += (float)fa[mt][0][0] + (float)fb[0][0][1];   // keep the reads alive
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][j][mt][nt] = mfma16(fb[nt][ks], fa[mt][ks], acc[i][j][mt][nt]);
    };
    // MFMA segment = 16 MFMAs with the NEXT phase's NR fragment reads slotted one per MFMA gap (the reads ride in the
    // shadow of the matrix pipe instead of lengthening the load segment), retired before the closing barrier.
#ifdef TIC_SIM
#define G256_INTERLEAVE(NR) do { } while (0)
#else
#define G256_INTERLEAVE(NR)                                              \
    do {                                                                 \
        _Pragma("unroll") for (int q_ = 0; q_ < (NR); ++q_) {            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           \
        }                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - (NR), 0);       \
    } while (0)
#endif
    // one K tile; on entry B0(t) is in fbp and the first k-half of A0(t) in fa0[.][0]; on exit the same for t+1 with fbq.
    // Phase p: load segment {issue one half-tile, vmcnt(8): the half-tile issued 4 phases ago has landed} | barrier |
    // MFMA segment {16 MFMA (k-half 0 first) + the fragment reads listed} | lgkmcnt(0), barrier.  A fragments arrive one
    // k-half at a time (the second half at the start of the segment that consumes it, under its first 8 MFMAs): the
    // peak is 80 fragment registers, which is what fits beside 128 accumulators in the 256 registers of a wave.
    auto tile = [&](int kt, bf16x8 (&fbp)[2][2], bf16x8 (&fbq)[2][2]) {
        const int cur = kt & 1;
        const uint32_t bufb = (uint32_t)cur * G256_BUF_BYTES, bufn = (uint32_t)(cur ^ 1) * G256_BUF_BYTES;
        // ---- phase 0: Q00 = A0 x B0 ; reads A0(t) k-half 1, B1(t)
        issue(cur ^ 1, kt + 1, 3);   // A1(t+1): that slot was last read in phase 2 of tile t-1
        wait_vmcnt<(DBG & 8) ? 12 : 8>();   // A1(t) has landed (first read: phase 1)
        if (!(DBG & 16)) g256_barrier();
        prio_hi();
        load_a(bufb, 0, fa0, 1);
        load_b(bufb, 1, fbq);
        mma(0, 0, fa0, fbp);
        G256_INTERLEAVE(8);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
        // ---- phase 1: Q01 = A0 x B1 ; reads A1(t) k-half 0
        issue(cur, kt + 2, 1);       // B0(t+2): B0(t) was read in phase 3 of tile t-1 (and lives in fbp)
        wait_vmcnt<(DBG & 8) ? 12 : 8>();   // B0(t+1) has landed (read: phase 3)
        if (!(DBG & 16)) g256_barrier();
        prio_hi();
        load_a(bufb, 1, fa1, 0);
        mma(0, 1, fa0, fbq);
        G256_INTERLEAVE(4);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
        // ---- phase 2: Q11 = A1 x B1 ; reads A1(t) k-half 1
        issue(cur, kt + 2, 0);       // A0(t+2): A0(t) was last read in phase 0
        wait_vmcnt<(DBG & 8) ? 12 : 8>();   // A0(t+1) has landed (first read: phase 3)
        if (!(DBG & 16)) g256_barrier();
        prio_hi();
        load_a(bufb, 1, fa1, 1);
        mma(1, 1, fa1, fbq);
        G256_INTERLEAVE(4);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
        // ---- phase 3: Q10 = A1 x B0 ; reads A0(t+1) k-half 0, B0(t+1)
        issue(cur, kt + 2, 2);       // B1(t+2): B1(t) was read in phase 0
        wait_vmcnt<(DBG & 8) ? 12 : 8>();   // B1(t+1) has landed (read: phase 0 of tile t+1)
        if (!(DBG & 16)) g256_barrier();
        prio_hi();
        load_a(bufn, 0, fa0, 0);
        load_b(bufn, 0, fbq);
        mma(1, 0, fa1, fbp);
        G256_INTERLEAVE(8);
        prio_lo();
        wait_lgkmcnt0();
        g256_barrier();
    };

    // prologue: all of K tile 0 and B0, A0, B1 of tile 1 (7 of the 8 half-tile slots) in the steady-state issue order
    // B0, A0, B1, A1; B0(0), A0(0), B1(0) landed; the first fragments are read here, then group 1 falls one barrier behind
    issue(0, 0, 1);
    issue(0, 0, 0);
    issue(0, 0, 2);
    issue(0, 0, 3);
    issue(1, 1, 1);
    issue(1, 1, 0);
    issue(1, 1, 2);
    wait_vmcnt<8>();
    g256_barrier();
    load_a(0u, 0, fa0, 0);
    load_b(0u, 0, fbx);
    wait_lgkmcnt0();
    if (wr == 1 && !(DBG & 16)) g256_barrier();
    G256_STAMP(1);

    // two tiles per trip (the B register sets swap roles every tile); an odd tile count runs one extra all-zero tile
    // (its DMAs are the zero fills above) rather than a second loop exit, which made hipcc copy all 128 accumulators
#pragma nounroll
    for (int kt = 0; kt < nk; kt += 2) {
        tile(kt, fbx, fby);
        tile(kt + 1, fby, fbx);
    }
    G256_STAMP(2);
    wait_vmcnt0();   // drain the zero fills issued for the tiles past the end before LDS is reused
    // the extra operand tile of the DGELU / MULAUX epilogues: every row of it in flight from here (see g256_fetch_aux)
    // from here on the thread index is re-derived from the hardware: threadIdx-derived registers need not survive the K loop
    const int le = lane_id_fresh(), tide = w * 64 + le;
    u32x4 auxr[16];
    if ((EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX || EPI == TIC_EPI_ADDAUX) && !SPLITK) {
        if (p.nt & 2) g256_fetch_aux<true, 0, G256_AUX_EARLY>(p, tide, m0, n0, auxr);
        else g256_fetch_aux<false, 0, G256_AUX_EARLY>(p, tide, m0, n0, auxr);
        sched_fence();
    }
    f32x4 exr[32];
    if ((EPI == TIC_EPI_RESID || EPI == TIC_EPI_PATCH) && !SPLITK) {   // (split-K: after the hand-off below -- the slab loads need the registers)
        if (p.nt & 2) g256_fetch_ex<EPI, true, 0, G256_EX_EARLY>(p, tide, m0, n0, exr);
        else g256_fetch_ex<EPI, false, 0, G256_EX_EARLY>(p, tide, m0, n0, exr);
        sched_fence();
    }
    if (wr == 0 && !(DBG & 16)) g256_barrier();   // re-balance the stagger
    g256_barrier();                // every wave's LDS reads and DMA writes have retired: the tile buffers are free

    if (SPLITK) {   // hand-off of the partial accumulators (header); thread t owns float4 slots t, 512 + t, ... of a slab
        const size_t slab_f4 = (size_t)32 * 512;
        if (part < nparts - 1) {
            f32x4* slab = reinterpret_cast<f32x4*>(p.slab) + ((size_t)tile_id * (nparts - 1) + part) * slab_f4 + tide;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 8; ++r) slab[(size_t)(g * 8 + r) * 512] = acc[r >> 2][g >> 1][r & 3][g & 1];
            wait_vmcnt0();
            g256_barrier();
            if (tide == 0 && !(p.fault && part == 0)) flag_publish(p.flags + tile_id * 4 + part, p.epoch);
            return;
        }
        for (int q = 0; q < nparts - 1; ++q) {   // block-uniform trip count
            if (tide == 0) flag_wait(p.flags + tile_id * 4 + q, p.epoch, p.err);
            g256_barrier();
            const f32x4* slab = reinterpret_cast<const f32x4*>(p.slab) + ((size_t)tile_id * (nparts - 1) + q) * slab_f4 + tide;
#pragma unroll
            for (int g = 0; g < 4; ++g) {   // 8 loads in flight at a time: 32 registers beside the 128 accumulators
                f32x4 v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = slab[(size_t)(g * 8 + r) * 512];
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r >> 2][g >> 1][r & 3][g & 1] += v[r];
                sched_fence();
            }
        }
        if (EPI == TIC_EPI_RESID || EPI == TIC_EPI_PATCH) {
            if (p.nt & 2) g256_fetch_ex<EPI, true, 0, G256_EX_EARLY>(p, tide, m0, n0, exr);
            else g256_fetch_ex<EPI, false, 0, G256_EX_EARLY>(p, tide, m0, n0, exr);
            sched_fence();
        }
    }

    // ---- stage u = bf16(acc) into LDS: rows r = i*4 + mt, column groups g = j*2 + nt
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = (g >> 1) * 128 + wc * 32 + (g & 1) * 16 + 4 * (le >> 4);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = (r >> 2) * 128 + wr * 64 + (r & 3) * 16 + (le & 15);
            const f32x4 v = acc[r >> 2][g >> 1][r & 3][g & 1];
            lds_st64(g256_stage_off(row, col >> 2), __builtin_bit_cast(bf16x4, u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])}));
        }
    }
    block_sync();
    G256_STAMP(3);
    sched_fence();   // keep the second pass (and its operand prefetch) below the staging pass: the accumulators are dead from here
    // kernel-argument (block-uniform) choice of the cache policy of the second pass
#define G256_FINISH(FN, ...)                                              \
    do {                                                                  \
        if (p.nt == 0) FN<EPI, false, false>(p, tide, m0, n0, ##__VA_ARGS__);   \
        else if (p.nt == 1) FN<EPI, true, false>(p, tide, m0, n0, ##__VA_ARGS__);  \
        else if (p.nt == 2) FN<EPI, false, true>(p, tide, m0, n0, ##__VA_ARGS__);  \
        else FN<EPI, true, true>(p, tide, m0, n0, ##__VA_ARGS__);               \
    } while (0)
    if (EPI == TIC_EPI_RESID || EPI == TIC_EPI_PATCH)
        G256_FINISH(g256_finish_f32, exr);
    else
        G256_FINISH(g256_finish_bf16, auxr);
#undef G256_FINISH
    G256_STAMP(4);
}
