// gemm.h -- bf16 MFMA GEMMs for the ViT token stream (gfx950).
//
//   gemm_nt : C[M,N]  = A[M,K] . B[N,K]^T   (+ fused epilogue)     forward Linear and dX (with W^T copies)
//   gemm_tn : C[N,K] += A[M,N]^T . B[M,K]                            dW = dY^T . X, fp32, split over M
//
// Replaces the cuBLAS calls behind nn.Linear in HF modeling_vit.py:202-205,216-218,235-236,
// 246-247 (reached from TIC/ViT/model.py:27-45) and their autograd backward (TIC/ViT/finetune.py:62).
//
// Both: 128x128 block tile, 64-deep K step, 256 threads = 2x2 waves of 64x64, operands staged
// HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B/lane) through a hardware-range-checked buffer
// resource (ragged M needs no padding: rows past M read as 0), 2 LDS stages (64 KiB -> 2 blocks
// per CU), XOR-swizzled 16-B chunks applied on the DMA *source* address and on the LDS read
// (guide rule 21) so every fragment read is bank-conflict-free.
#pragma once
#include "tic_prims.h"

// epilogue codes: TIC_EPI_* in include/tic_hip.h
#include "../../include/tic_hip.h"

// Implicit-GEMM convolution: the A operand is never materialised -- row m = output pixel (n, oy, ox) and K tile kt
// (64 channels of ONE filter tap, Cin % 64 == 0) is gathered straight from the NHWC activation:
//   A[m][k = (ky*KW + kx)*Cin + c] = x[n][oy*stride - pad + ky][ox*stride - pad + kx][c]   (0 outside the image)
// Replaces F.conv2d forward for 3x3 convolutions (TIC/ResNet/model.py:6-9) and, fed with dY and the flipped/transposed
// filter, its stride-1 input gradient.
struct ConvGather {
    int H, W, Cin, Ho, Wo, KW, stride, pad;
    // output-row remap (rm_W > 0): GEMM row m = (image, a, b) of the Ho x Wo grid is stored at row (image, 2a + rm_py, 2b + rm_px) of an
    // rm_H x rm_W image -- one parity class of the input gradient of a stride-2 convolution (tic_conv_igemm_dgrad_s2)
    int rm_H, rm_W, rm_py, rm_px;
};
struct GemmNtParams {
    const bf16_t* A;   // [M,K]  (CONV: the NHWC activation [B,H,W,Cin])
    const bf16_t* B;   // [N,K]
    int M, N, K;
    const float* bias;       // [N] or nullptr
    bf16_t* out;             // [M,N] bf16
    bf16_t* out2;            // [M,N] bf16 (GELU)
    float* out_f32;          // [M,N] fp32 (RESID) / remapped rows (PATCH)
    const float* resid;      // [M,N] fp32
    const bf16_t* aux;       // [M,N] bf16 (DGELU: pre-activation u)
    const float* rowtab;     // [(P+1),N] fp32 (PATCH: position embeddings)
    int patches;             // P (PATCH)
    float* colsum;           // optional [N]: += column sums of the stored output (bias gradient of the consumer), BF16 / DGELU
    ConvGather cg;           // gemm_nt_kernel<EPI, true> only
    int nt;                  // 256x256 staged epilogue: bit 0 non-temporal output stores, bit 1 non-temporal operand loads
    int gm;                  // m-tiles per group of the XCD-local tile walk (tile_coords), 0 = 8
    int stagger;             // experiment (tic_set_option "gemm_stagger"): s_sleep rounds for every other first-wave workgroup
    // split-K form of the 256x256 kernel (few tiles, long K): `split` workgroups share a tile, each reduces 1/split of K; the first
    // split - 1 hand their fp32 accumulators to the last one through `slab` ([tile][part][32][512] float4) and `flags` ([tile][4])
    int split;               // 1 (off), 2 or 4
    float* slab;
    unsigned* flags;
    unsigned epoch;          // value a flag takes in THIS launch (the host counts launches: flags never need clearing)
    unsigned* err;           // host-mapped error word: a consumer whose flag poll ran out stores 0xDEADxxxx there (tic_prims.h flag_wait)
    int fault;               // test builds only (tic_set_option "nt_fault"): part 0 of every tile never publishes its flag
#ifdef TIC_MEASURE
    unsigned long long* stamps;   // measurement build only: [grid][8] s_memrealtime stamps of the 256x256 kernel's stages, or nullptr
#endif
};

// 16-byte-chunk XOR swizzle for 128-byte LDS rows: conflict-free for the 16x16x32 row-fragment
// ds_read_b128 pattern AND for ds_read_b64_tr_b16 (see DESIGN.md "LDS images").
TIC_DEV uint32_t swz128(uint32_t row) { return ((row >> 1) & 3u) << 1; }

// XCD-aware, grouped tile order: blocks that share an XCD (bid % 8, guide T1) get a contiguous
// range of tiles, walked GM m-tiles at a time so A and B panels are re-read from that XCD's L2.
TIC_DEV void tile_coords(int bid, int nwg, int tiles_m, int tiles_n, int& tm, int& tn, int GM = 8) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = GM * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * GM;
    const int gsize = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
    const int in_group = wg - group * per_group;
    tm = first_m + in_group % gsize;
    tn = in_group / gsize;
}

// ---- epilogue ---------------------------------------------------------------------------------------
// A lane owns, per output row m, NG groups of 4 consecutive columns.  The bias is already inside the accumulators
// (they start at it); the per-element extra operand (residual / pre-activation / position
// row) is fetched one row AHEAD of its use, so the loads of row r+1 fly while row r is computed and stored.
struct EpiExtra {
    f32x4 f;   // RESID: residual, PATCH: position-embedding row
    u32x2 u;   // DGELU: 4 bf16 pre-activations, MULAUX: 4 bf16 multipliers
};
template <int EPI>
TIC_DEV EpiExtra epi_fetch(const GemmNtParams& p, int m, int n) {
    EpiExtra e;
    e.f = f32x4{0.f, 0.f, 0.f, 0.f};
    e.u = u32x2{0u, 0u};
    if (m < p.M && n < p.N) {
        if (EPI == TIC_EPI_RESID) e.f = *reinterpret_cast<const f32x4*>(p.resid + (size_t)m * p.N + n);
        if (EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX || EPI == TIC_EPI_ADDAUX) e.u = *reinterpret_cast<const u32x2*>(p.aux + (size_t)m * p.N + n);
        if (EPI == TIC_EPI_PATCH) e.f = *reinterpret_cast<const f32x4*>(p.rowtab + (size_t)(1 + m % p.patches) * p.N + n);
    }
    return e;
}
// returns the 4 values written (0 for masked rows / columns) so that callers can form column sums
template <int EPI>
TIC_DEV f32x4 epi_store(const GemmNtParams& p, int m, int n, f32x4 v, EpiExtra e) {   // v = bias + sum (the accumulators start at the bias)
    if (m >= p.M || n >= p.N) return f32x4{0.f, 0.f, 0.f, 0.f};   // ragged M; N not a multiple of the tile (conv channels 64, C*k*k ...)
    const size_t o = (size_t)m * p.N + n;
    if (EPI == TIC_EPI_BF16) {
        *reinterpret_cast<u32x2*>(p.out + o) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        return v;
    } else if (EPI == TIC_EPI_GELU) {
        float u[4], g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u[r] = bfround(v[r]);
            g[r] = gelu_erf(u[r]);
        }
        *reinterpret_cast<u32x2*>(p.out + o) = u32x2{pack2bf(u[0], u[1]), pack2bf(u[2], u[3])};
        *reinterpret_cast<u32x2*>(p.out2 + o) = u32x2{pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
    } else if (EPI == TIC_EPI_RESID) {
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = e.f[r] + bfround(v[r]);
        *reinterpret_cast<f32x4*>(p.out_f32 + o) = y;
    } else if (EPI == TIC_EPI_DGELU) {
        const float u0 = bf2f((bf16_t)(e.u[0] & 0xffff)), u1 = bf2f((bf16_t)(e.u[0] >> 16));
        const float u2 = bf2f((bf16_t)(e.u[1] & 0xffff)), u3 = bf2f((bf16_t)(e.u[1] >> 16));
        const float d0 = bfround(v[0]) * gelu_erf_grad(u0), d1 = bfround(v[1]) * gelu_erf_grad(u1);
        const float d2 = bfround(v[2]) * gelu_erf_grad(u2), d3 = bfround(v[3]) * gelu_erf_grad(u3);
        *reinterpret_cast<u32x2*>(p.out + o) = u32x2{pack2bf(d0, d1), pack2bf(d2, d3)};
        return f32x4{d0, d1, d2, d3};
    } else if (EPI == TIC_EPI_GELU_ONLY) {
        float g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = gelu_erf(bfround(v[r]));
        *reinterpret_cast<u32x2*>(p.out2 + o) = u32x2{pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
    } else if (EPI == TIC_EPI_GELU_DG) {
        const GeluPair lo = gelu_pair(f32x2{bfround(v[0]), bfround(v[1])}), hi = gelu_pair(f32x2{bfround(v[2]), bfround(v[3])});
        *reinterpret_cast<u32x2*>(p.out + o) = u32x2{pack2bf(lo.dg[0], lo.dg[1]), pack2bf(hi.dg[0], hi.dg[1])};
        *reinterpret_cast<u32x2*>(p.out2 + o) = u32x2{pack2bf(lo.g[0], lo.g[1]), pack2bf(hi.g[0], hi.g[1])};
    } else if (EPI == TIC_EPI_MULAUX) {
        const float d0 = bfround(v[0]) * bf2f((bf16_t)(e.u[0] & 0xffff)), d1 = bfround(v[1]) * bf2f((bf16_t)(e.u[0] >> 16));
        const float d2 = bfround(v[2]) * bf2f((bf16_t)(e.u[1] & 0xffff)), d3 = bfround(v[3]) * bf2f((bf16_t)(e.u[1] >> 16));
        *reinterpret_cast<u32x2*>(p.out + o) = u32x2{pack2bf(d0, d1), pack2bf(d2, d3)};
        return f32x4{d0, d1, d2, d3};
    } else if (EPI == TIC_EPI_ADDAUX) {
        const float d0 = bfround(v[0]) + bf2f((bf16_t)(e.u[0] & 0xffff)), d1 = bfround(v[1]) + bf2f((bf16_t)(e.u[0] >> 16));
        const float d2 = bfround(v[2]) + bf2f((bf16_t)(e.u[1] & 0xffff)), d3 = bfround(v[3]) + bf2f((bf16_t)(e.u[1] >> 16));
        *reinterpret_cast<u32x2*>(p.out + o) = u32x2{pack2bf(d0, d1), pack2bf(d2, d3)};
        return f32x4{d0, d1, d2, d3};
    } else if (EPI == TIC_EPI_PATCH) {
        const int img = m / p.patches, pi = m - img * p.patches;
        const size_t orow = (size_t)img * (p.patches + 1) + 1 + pi;
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = bfround(v[r]) + e.f[r];
        *reinterpret_cast<f32x4*>(p.out_f32 + orow * p.N + n) = y;
    }
    return v;
}
// NR rows x NG column groups per lane; row_of(r) / col_of(g) give the global coordinates, acc_of(r, g) the value
template <int EPI, int NR, int NG, class RowF, class ColF, class AccF>
TIC_DEV void gemm_epilogue(const GemmNtParams& p, RowF row_of, ColF col_of, AccF acc_of) {
    constexpr bool HAS_EXTRA = (EPI == TIC_EPI_RESID || EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX || EPI == TIC_EPI_ADDAUX || EPI == TIC_EPI_PATCH);
    constexpr bool HAS_COLSUM = (EPI == TIC_EPI_BF16 || EPI == TIC_EPI_DGELU || EPI == TIC_EPI_MULAUX);
    EpiExtra ex[2][NG];
    f32x4 cs[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) cs[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (HAS_EXTRA) {
#pragma unroll
        for (int g = 0; g < NG; ++g) ex[0][g] = epi_fetch<EPI>(p, row_of(0), col_of(g));
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (HAS_EXTRA && r + 1 < NR) {
#pragma unroll
            for (int g = 0; g < NG; ++g) ex[(r + 1) & 1][g] = epi_fetch<EPI>(p, row_of(r + 1), col_of(g));
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const f32x4 w4 = epi_store<EPI>(p, row_of(r), col_of(g), acc_of(r, g), ex[r & 1][g]);
            if (HAS_COLSUM) cs[g] += w4;
        }
    }
    // fused bias gradient: this lane's rows are summed above; the 16 lanes that share (l>>4) hold the other rows of the
    // same 4 columns -> butterfly over lane bits 0..3, then one lane per column group adds to the global vector
    if (HAS_COLSUM && p.colsum) {   // kernel-argument condition: wave-uniform
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = row16_sum(cs[g][c]);
                if ((lane_id() & 15) == 0 && col_of(g) + c < p.N) atomic_addf(p.colsum + col_of(g) + c, v);
            }
        }
    }
}

// a bare s_barrier (no s_waitcnt vmcnt(0) in front of it, unlike __syncthreads): LDS-DMA stays in flight across it
TIC_DEV void raw_sync() {
#ifndef TIC_SIM
    asm volatile("" ::: "memory");
#endif
    raw_barrier();
#ifndef TIC_SIM
    asm volatile("" ::: "memory");
#endif
}

#define GEMM_BM 128
#define GEMM_BN 128
#define GEMM_BK 64
#define GEMM_STAGE_BYTES 32768   // A tile 16 KiB + B tile 16 KiB
#define GEMM_LDS_BYTES (2 * GEMM_STAGE_BYTES)

// SPLITK = 1 (few tiles, long reduction: ViT-L at 8-16 images per GPU, N = 1024, K >= 3072 -- 104 / 200 tiles on 512 workgroup slots, each
// a 64-tile K loop whose LATENCY is the launch time): p.split = 2 or 4 workgroups per tile, block b -> (tile b % tiles, part b / tiles).
// Parts before the last store their fp32 accumulators (64 KiB, coalesced float4 per lane) to p.slab and publish p.flags[tile * 4 + part]
// = p.epoch; the last part (highest block ids: its producers are always dispatched first) waits, adds the slabs in part order and
// runs the ordinary epilogue.  Same hand-off as gemm256.h's split-K form.
// NST = 4 (few workgroups: every one alone on its CU, so the second workgroup that would hide a 2-stage loop's load latency is not
// there -- ViT-L at 8-16 images per GPU, N = 1024: 104-208 workgroups whose 16-64 K steps each waited ~1 us for ONE 32 KiB stage):
// a ring of four stages (128 KiB), three in flight (96 KiB per CU > the ~60 KB that cover the CU's LDS-DMA intake x latency), counted
// vmcnt, one raw barrier per step, fragment reads hidden from hipcc (it would drain the ring in front of every one).  Same products in
// the same order as NST = 2: the two forms agree bit for bit.
template <int EPI, bool CONV = false, int SPLITK = 0, int NST = 2>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(GemmNtParams p) {
    static_assert(NST == 2 || NST == 4, "gemm_nt_kernel: 2 or 4 stages");
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wm = w >> 1, wn = w & 1;
    const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM, tiles_n = (p.N + GEMM_BN - 1) / GEMM_BN;   // B rows >= N read 0
    const int ntiles = tiles_m * tiles_n, nparts = SPLITK ? p.split : 1;
    const int part = SPLITK ? TIC_BID_X / ntiles : 0, tile_id = SPLITK ? TIC_BID_X - part * ntiles : TIC_BID_X;
    int tm, tn;
    tile_coords(tile_id, ntiles, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * GEMM_BM, n0 = tn * GEMM_BN;

    const size_t a_bytes = CONV ? (size_t)(p.M / (p.cg.Ho * p.cg.Wo)) * p.cg.H * p.cg.W * p.cg.Cin * 2 : (size_t)p.M * p.K * 2;
    const tic_rsrc_t ra = make_rsrc(p.A, (uint32_t)a_bytes);
    const tic_rsrc_t rb = make_rsrc(p.B, (uint32_t)((size_t)p.N * p.K * 2));

    // ---- LDS-DMA source offsets (bytes) for this lane's 4 A chunks and 4 B chunks --------------
    // chunk c = i*4 + w covers tile rows 8c..8c+7 (1 KiB); lane -> (row l>>3, physical 16-B slot l&7)
    uint32_t voa[4], vob[4];
    int iy0[4], ix0[4];   // CONV: top-left input coordinate of this row's window (row >= M: far outside the image)
    {
        const uint32_t slot_log = (uint32_t)(l & 7) ^ ((((uint32_t)l >> 4) & 3u) << 1);   // = phys ^ swz128(row)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (i * 4 + w) * 8 + (l >> 3);
            if (CONV) {
                const int m = m0 + r, hw = p.cg.Ho * p.cg.Wo;
                const int n = m / hw, rem = m - n * hw, oy = rem / p.cg.Wo, ox = rem - oy * p.cg.Wo;
                iy0[i] = (m < p.M) ? oy * p.cg.stride - p.cg.pad : -(1 << 20);
                ix0[i] = ox * p.cg.stride - p.cg.pad;
                if (p.cg.Cin == 4) {
                    // STEM (TIC/ResNet/model.py:148: 7x7 / 2, pad 3, on the 3-channel image zero-padded to 4 channels = 8 B per pixel).
                    // A K tile of 64 = 2 filter rows x 8 pixels x 4 channels; the 8-pixel window starts at the EVEN pixel 2 ox - 4 (one
                    // pixel left of the filter: weight 0), so every 16-byte chunk = 2 pixels is aligned and -- image widths being even --
                    // entirely inside or entirely outside the image.  This lane's chunk: filter row 2 kt + (slot >> 2), pixels
                    // 2 ox - 4 + 2 (slot & 3) (+1).  iy0 / ix0 hold THAT row / pixel for kt = 0.
                    iy0[i] += (int)(slot_log >> 2);
                    ix0[i] += -1 + 2 * (int)(slot_log & 3u);
                    voa[i] = (uint32_t)((((long)n * p.cg.H + iy0[i]) * p.cg.W + ix0[i]) * 8);
                } else
                // 32-bit wrap is fine: the tap offset added in stage() brings every in-image address back into range
                voa[i] = (uint32_t)((((long)n * p.cg.H + iy0[i]) * p.cg.W + ix0[i]) * p.cg.Cin * 2 + (long)slot_log * 16);
            } else {
                voa[i] = (uint32_t)(((size_t)(m0 + r) * p.K + slot_log * 8) * 2);
            }
            vob[i] = (uint32_t)(((size_t)(n0 + r) * p.K + slot_log * 8) * 2);
        }
    }
    auto stage = [&](int buf, int kt) {
        const uint32_t soff = (uint32_t)kt * (GEMM_BK * 2);
        const uint32_t base = (uint32_t)buf * GEMM_STAGE_BYTES;
        if (CONV && p.cg.Cin == 4) {   // stem: filter rows 2 kt and 2 kt + 1 (row 7 does not exist: zero fill)
            const uint32_t tapoff = (uint32_t)(2 * kt * p.cg.W * 8);
            const int krow = 2 * kt + (int)((((uint32_t)(l & 7) ^ ((((uint32_t)l >> 4) & 3u) << 1))) >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = krow < 7 && (unsigned)(iy0[i] + 2 * kt) < (unsigned)p.cg.H && (unsigned)ix0[i] < (unsigned)p.cg.W;
                glds16(ra, base + (uint32_t)(i * 4 + w) * 1024u, in ? voa[i] + tapoff : 0xFFFFFFF0u, 0);
                glds16(rb, base + 16384u + (uint32_t)(i * 4 + w) * 1024u, vob[i], soff);
            }
            return;
        }
        if (CONV) {
            const int kk = kt * GEMM_BK, tap = kk / p.cg.Cin, c0 = kk - tap * p.cg.Cin;   // wave-uniform
            const int ky = tap / p.cg.KW, kx = tap - ky * p.cg.KW;
            const uint32_t tapoff = (uint32_t)(((ky * p.cg.W + kx) * p.cg.Cin + c0) * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = (unsigned)(iy0[i] + ky) < (unsigned)p.cg.H && (unsigned)(ix0[i] + kx) < (unsigned)p.cg.W;
                glds16(ra, base + (uint32_t)(i * 4 + w) * 1024u, in ? voa[i] + tapoff : 0xFFFFFFF0u, 0);
                glds16(rb, base + 16384u + (uint32_t)(i * 4 + w) * 1024u, vob[i], soff);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(ra, base + (uint32_t)(i * 4 + w) * 1024u, voa[i], soff);
            glds16(rb, base + 16384u + (uint32_t)(i * 4 + w) * 1024u, vob[i], soff);
        }
    };

    // ---- fragment read offsets: row (l&15), logical chunk 4ks + (l>>4) -------------------------------
    const uint32_t sw = (((uint32_t)(l & 15) >> 1) & 3u) << 1;
    uint32_t fo[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fo[ks] = (uint32_t)(l & 15) * 128u + ((((uint32_t)ks * 4 + ((uint32_t)l >> 4)) ^ sw) * 16u);
    const uint32_t a_base = (uint32_t)wm * 64 * 128, b_base = 16384u + (uint32_t)wn * 64 * 128;

    // accumulators start at the bias of their columns (same summation order as gemm256.h: the two kernels agree bit for bit)
    f32x4 acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + 4 * (l >> 4);
        const f32x4 b4 = (p.bias && col < p.N && part == nparts - 1) ? *reinterpret_cast<const f32x4*>(p.bias + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = b4;
    }

    const int nk = (p.K / GEMM_BK) / nparts, kt0 = part * nk;   // K tiles of THIS workgroup
    if constexpr (NST == 4) {
        const uint32_t la[2] = {lds_base() + a_base + fo[0], lds_base() + a_base + fo[1]};
        const uint32_t lb[2] = {lds_base() + b_base + fo[0], lds_base() + b_base + fo[1]};
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < nk) stage(s, kt0 + s);
        for (int kt = 0; kt < nk; ++kt) {
            // stage kt has landed once at most min(2, nk - 1 - kt) younger stages (8 loads per lane each) are outstanding
            const int younger = nk - 1 - kt;
            if (younger >= 2) wait_vmcnt<16>();
            else if (younger == 1) wait_vmcnt<8>();
            else wait_vmcnt0();
            raw_sync();   // every wave's pieces of stage kt are in LDS, and every wave has finished reading stage kt - 1 ...
            if (kt + 3 < nk) stage((kt + 3) & 3, kt0 + kt + 3);   // ... whose buffer this refills
            const uint32_t sb = (uint32_t)(kt & 3) * GEMM_STAGE_BYTES;
            bf16x8 fa[2][4], fb[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[0][t] = lds_ld128_hidden(la[0] + sb, (uint32_t)t * 2048u);
                fb[0][t] = lds_ld128_hidden(lb[0] + sb, (uint32_t)t * 2048u);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[1][t] = lds_ld128_hidden(la[1] + sb, (uint32_t)t * 2048u);
                fb[1][t] = lds_ld128_hidden(lb[1] + sb, (uint32_t)t * 2048u);
            }
            lds_wait<8>(fa[0][0], fa[0][1], fa[0][2], fa[0][3], fb[0][0], fb[0][1], fb[0][2], fb[0][3]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(fb[0][nt], fa[0][mt], acc[mt][nt]);
            sched_fence();   // else hipcc hoists the second wait above these 16 MFMAs and the k-half-1 reads are not overlapped
            lds_wait<0>(fa[1][0], fa[1][1], fa[1][2], fa[1][3], fb[1][0], fb[1][1], fb[1][2], fb[1][3]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(fb[1][nt], fa[1][mt], acc[mt][nt]);
        }
    } else {
    stage(0, kt0);
    wait_vmcnt0();
    block_sync();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt0 + kt + 1);
        const uint32_t sb = (uint32_t)cur * GEMM_STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = lds_ld128(sb + a_base + (uint32_t)t * 2048u + fo[ks]);
                fb[t] = lds_ld128(sb + b_base + (uint32_t)t * 2048u + fo[ks]);
            }
            // swapped operands: D rows = n (4 consecutive per lane), D cols = m (lane&15)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16(fb[nt], fa[mt], acc[mt][nt]);
        }
        wait_vmcnt0();
        block_sync();
    }
    }

    if (SPLITK) {   // thread t owns float4 slots t, 256 + t, ... of a 64-KiB slab
        if (part < nparts - 1) {
            float* dst = p.slab + ((size_t)(tile_id * (nparts - 1) + part) * 16 * 256 + tid) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + (size_t)(i * 4 + j) * 256 * 4) = acc[i][j];
            wait_vmcnt0();
            block_sync();
            if (tid == 0 && !(p.fault && part == 0)) flag_publish(p.flags + tile_id * 4 + part, p.epoch);
            return;
        }
        for (int q = 0; q < nparts - 1; ++q) {
            if (tid == 0) flag_wait(p.flags + tile_id * 4 + q, p.epoch, p.err);
            block_sync();
            const float* src = p.slab + ((size_t)(tile_id * (nparts - 1) + q) * 16 * 256 + tid) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(src + (size_t)(i * 4 + j) * 256 * 4);
        }
    }
    // ---- epilogue: lane holds rows m = ..+(l&15), 4 consecutive n = ..+4(l>>4) ------------------------
    int drow[4];   // destination row of this lane's 4 accumulator rows
#pragma unroll
    for (int r = 0; r < 4; ++r) drow[r] = m0 + wm * 64 + r * 16 + (l & 15);
    int m_lim = p.M;
    if (CONV && p.cg.rm_W > 0) {   // rows land in one parity class of a larger image (kernel argument: uniform)
        const int hw = p.cg.Ho * p.cg.Wo;
        const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)p.cg.Wo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = drow[r];
            const int img = (int)((float)m * inv_hw + 0.5f * inv_hw), rem = m - img * hw;   // exact for m < 2^24 up to the corrections below
            int im = img, rm = rem;
            if (rm < 0) { --im; rm += hw; }
            if (rm >= hw) { ++im; rm -= hw; }
            int a = (int)((float)rm * inv_w);
            if (a * p.cg.Wo > rm) --a;
            if ((a + 1) * p.cg.Wo <= rm) ++a;
            const int b = rm - a * p.cg.Wo;
            drow[r] = m < p.M ? (im * p.cg.rm_H + 2 * a + p.cg.rm_py) * p.cg.rm_W + 2 * b + p.cg.rm_px : 0x7fffffff;
        }
        m_lim = (p.M / hw) * p.cg.rm_H * p.cg.rm_W;
    }
    if (CONV) {
        GemmNtParams q = p;
        q.M = m_lim;
        gemm_epilogue<EPI, 4, 4>(
            q, [&](int r) { return drow[r]; }, [&](int g) { return n0 + wn * 64 + g * 16 + 4 * (l >> 4); }, [&](int r, int g) { return acc[r][g]; });
        return;
    }
    gemm_epilogue<EPI, 4, 4>(
        p, [&](int r) { return m0 + wm * 64 + r * 16 + (l & 15); }, [&](int g) { return n0 + wn * 64 + g * 16 + 4 * (l >> 4); },
        [&](int r, int g) { return acc[r][g]; });
}

// ------------------------------------------------------------------------------------------------------
// gemm_tn: C[N,K] (fp32, row-major, ldc = K) += sum_m A[m,n] * B[m,k]   -- dW = dY^T . X
//   A = dY [M,N] bf16, B = X [M,K] bf16.  The reduction runs over M (ragged; rows past M read as 0
//   through the buffer range check), split over gridDim.y slices that add their partial tile with
//   fp32 atomics (one 32x32 accumulator register = two 128-B row segments per wave-instruction, the
//   full-rate atomic shape of MI355X_MICROARCH "Global float atomics").  C must be zero (or hold the
//   gradient being accumulated into) before the launch.
//   Both operands are reduction-major in memory, so fragments come from ds_read_b64_tr_b16.
//   LDS tiles: [64 m][128 n or k] bf16 = 256-B rows, 16 KiB each.
struct GemmTnParams {
    const bf16_t* A;   // [M,N]
    const bf16_t* B;   // [M,K]  (CONV: the NHWC activation [B,H,W,Cin]; B[m][k] gathered as in ConvGather)
    float* C;          // [N,K]
    int M, N, K;
    int m_per_slice;   // multiple of 64
    ConvGather cg;     // gemm_tn_kernel<true> only
};
// floor(m / d) for 0 <= m < 2^24 from a float reciprocal (the hardware has no integer divide; this is 5 VALU ops)
TIC_DEV int div_small(int m, int d, float inv) {
    int q = (int)((float)m * inv);
    if (q * d > m) --q;
    if ((q + 1) * d <= m) ++q;
    return q;
}

// 16-B-chunk swizzle for 256-byte rows read by the 32x32x16 transposed-fragment pattern
TIC_DEV uint32_t swz256(uint32_t row) { return (row & 3u) << 2; }

template <bool CONV = false>
__global__ void __launch_bounds__(256, 2) gemm_tn_kernel(GemmTnParams p) {
    const int tid = TIC_TID, l = tid & 63, w = wave_id();
    const int wn = w >> 1, wk = w & 1;
    const int tiles_n = (p.N + 127) / 128, tiles_k = (p.K + 127) / 128;
    int tnn, tkk;
    tile_coords(TIC_BID_X, tiles_n * tiles_k, tiles_n, tiles_k, tnn, tkk);
    const int n0 = tnn * 128, k0 = tkk * 128;
    const int m_begin = TIC_BID_Y * p.m_per_slice;
    int m_end = m_begin + p.m_per_slice;
    if (m_end > p.M) m_end = p.M;
    if (m_begin >= p.M) return;
    const int nsteps = (m_end - m_begin + 63) / 64;

    // range check at the END OF THIS SLICE so a partial last step reads zeros, not the next slice
    const tic_rsrc_t ra = make_rsrc(p.A, (uint32_t)((size_t)m_end * p.N * 2));
    const size_t b_bytes = CONV ? (size_t)(p.M / (p.cg.Ho * p.cg.Wo)) * p.cg.H * p.cg.W * p.cg.Cin * 2 : (size_t)m_end * p.K * 2;
    const tic_rsrc_t rb = make_rsrc(p.B, (uint32_t)b_bytes);

    // staging: a 1-KiB DMA piece = 4 rows x 256 B; piece c = i*4 + w covers tile rows 4c..4c+3;
    // lane -> (row l>>4, physical chunk l&15); source chunk = phys ^ swz256(row)
    uint32_t voa[4], vob[4];
    bool a_ok, b_ok;
    {
        const uint32_t rr = (uint32_t)l >> 4;                    // row within piece == row & 3
        const uint32_t ch_log = ((uint32_t)l & 15u) ^ (rr << 2);
        a_ok = (n0 + (int)ch_log * 8) < p.N;                     // columns past N / K (ragged tiles) are zero-filled
        b_ok = (k0 + (int)ch_log * 8) < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (i * 4 + w) * 4 + (int)rr;
            voa[i] = (uint32_t)(((size_t)(m_begin + r) * p.N + n0 + ch_log * 8) * 2);
            vob[i] = (uint32_t)(((size_t)(m_begin + r) * p.K + k0 + ch_log * 8) * 2);
        }
    }
    // CONV: this lane's 8 columns of the B tile belong to ONE filter tap (Cin % 64 == 0): (ky, kx, channel offset) are fixed,
    // the pixel of row m changes every step
    int c_ky = 0, c_kx = 0, c_row = 0, c_m = 0;
    uint32_t c_choff = 0;
    float inv_hw = 0.f, inv_w = 0.f;
    if (CONV) {
        const uint32_t rr = (uint32_t)l >> 4, ch_log = ((uint32_t)l & 15u) ^ (rr << 2);
        const int kcol = k0 + (int)ch_log * 8, tap = kcol / p.cg.Cin;
        c_ky = tap / p.cg.KW;
        c_kx = tap - c_ky * p.cg.KW;
        c_choff = (uint32_t)((kcol - tap * p.cg.Cin) * 2);
        if (p.cg.Cin == 4) {   // stem layout (see gemm_nt_kernel): k = ky * 32 + (pixel pair) * 8 + ...; pixel = 2 ox - 4 + 2 pair = ix0 - 1 + 2 pair
            c_ky = kcol >> 5;
            c_kx = -1 + 2 * ((kcol & 31) >> 3);
            c_choff = 0;
        }
        c_row = w * 4 + (int)rr;     // piece i adds 16 rows
        c_m = m_begin;
        inv_hw = 1.0f / (float)(p.cg.Ho * p.cg.Wo);
        inv_w = 1.0f / (float)p.cg.Wo;
    }
    // the row advance lives in the VGPR offset: only that offset is range-checked by the hardware
    const uint32_t strideA = (uint32_t)p.N * 2 * 64, strideB = (uint32_t)p.K * 2 * 64;   // bytes per 64-row step
    auto stage = [&](int buf) {
        const uint32_t base = (uint32_t)buf * GEMM_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(ra, base + (uint32_t)(i * 4 + w) * 1024u, a_ok ? voa[i] : 0xFFFFFFF0u, 0);
            if (CONV) {
                const int m = c_m + c_row + i * 16, hw = p.cg.Ho * p.cg.Wo;
                const int n = div_small(m, hw, inv_hw), rem = m - n * hw, oy = div_small(rem, p.cg.Wo, inv_w), ox = rem - oy * p.cg.Wo;
                const int iy = oy * p.cg.stride - p.cg.pad + c_ky, ix = ox * p.cg.stride - p.cg.pad + c_kx;
                const bool in = b_ok && m < m_end && (unsigned)iy < (unsigned)p.cg.H && (unsigned)ix < (unsigned)p.cg.W && (p.cg.Cin != 4 || c_ky < 7);   // (stem: filter row 7 is padding)
                const uint32_t off = (uint32_t)((((long)n * p.cg.H + iy) * p.cg.W + ix) * p.cg.Cin * 2) + c_choff;
                glds16(rb, base + 16384u + (uint32_t)(i * 4 + w) * 1024u, in ? off : 0xFFFFFFF0u, 0);
            } else {
                glds16(rb, base + 16384u + (uint32_t)(i * 4 + w) * 1024u, b_ok ? vob[i] : 0xFFFFFFF0u, 0);
                vob[i] += strideB;
            }
            voa[i] += strideA;
        }
        if (CONV) c_m += 64;
    };

    // transposed-fragment addresses (32x32x16): lane l: h = l>>5, c16 = (l>>4)&1, q = (l>>2)&3, p4 = l&3
    // supplies &tile[row = 16*ks + 8h + q (+4)][col = col0 + 16*c16 + 4*p4]
    const uint32_t h = (uint32_t)l >> 5, c16 = ((uint32_t)l >> 4) & 1u, q = ((uint32_t)l >> 2) & 3u, p4 = (uint32_t)l & 3u;
    auto tr_off = [&](uint32_t tile_base, uint32_t col0, uint32_t ks, uint32_t second) -> uint32_t {
        const uint32_t row = 16 * ks + 8 * h + 4 * second + q;
        const uint32_t col = col0 + 16 * c16 + 4 * p4;            // element column, multiple of 4
        const uint32_t chunk = (col >> 3) ^ swz256(row);
        return tile_base + row * 256u + chunk * 16u + (col & 4u) * 2u;
    };

    uint32_t tr_lane[2][2];   // [operand][t]: lane part of the fragment address (ks and the +4-row read are immediates)
#pragma unroll
    for (uint32_t t = 0; t < 2; ++t) {
        tr_lane[0][t] = lds_base() + tr_off(0u, (uint32_t)wn * 64 + t * 32, 0, 0);
        tr_lane[1][t] = lds_base() + tr_off(0u, (uint32_t)wk * 64 + t * 32, 0, 0);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0);
    wait_vmcnt0();
    block_sync();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        if (st + 1 < nsteps) stage(cur ^ 1);
        const uint32_t sb = (uint32_t)cur * GEMM_STAGE_BYTES;
        // fragment reads hidden from hipcc (it would put vmcnt(0) in front of them and serialise the prefetch of the
        // next stage with this stage's MFMAs); ks+1 is read while ks is multiplied
        bf16x8 fa[2][2], fb[2][2];   // [ks & 1][t]
        auto rd = [&](uint32_t ks) {
#pragma unroll
            for (uint32_t t = 0; t < 2; ++t) {
                const bf16x4 a0 = lds_tr64_hidden(tr_lane[0][t] + sb, ks * 4096u);
                const bf16x4 a1 = lds_tr64_hidden(tr_lane[0][t] + sb, ks * 4096u + 1024u);
                const bf16x4 b0 = lds_tr64_hidden(tr_lane[1][t] + sb, 16384u + ks * 4096u);
                const bf16x4 b1 = lds_tr64_hidden(tr_lane[1][t] + sb, 16384u + ks * 4096u + 1024u);
                fa[ks & 1][t] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                fb[ks & 1][t] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
        };
        rd(0);
#pragma unroll
        for (uint32_t ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
                rd(ks + 1);
                lds_wait<8>(fa[ks & 1][0], fa[ks & 1][1], fb[ks & 1][0], fb[ks & 1][1]);   // the 8 reads of ks+1 may still be out
            } else {
                lds_wait<0>(fa[ks & 1][0], fa[ks & 1][1], fb[ks & 1][0], fb[ks & 1][1]);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) acc[nt][kt] = mfma32(fa[ks & 1][nt], fb[ks & 1][kt], acc[nt][kt]);
        }
        wait_vmcnt0();
        block_sync();
    }

    // D: col = l&31 -> k (contiguous in C), row = (r&3) + 8(r>>2) + 4(l>>5) -> n
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int kk = k0 + wk * 64 + kt * 32 + (l & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                if (n < p.N && kk < p.K) atomic_addf(p.C + (size_t)n * p.K + kk, acc[nt][kt][r]);
            }
        }
}
