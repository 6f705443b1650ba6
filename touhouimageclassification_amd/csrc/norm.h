// norm.h -- LayerNorm forward / backward on the fp32 residual stream (HBM-bound, one wave per row).
//
// Replaces nn.LayerNorm(eps=1e-12) in HF modeling_vit.py:261-262,274,281,348,385 and its autograd
// backward.  Semantics under the reference's bf16 autocast: statistics and normalisation in fp32 on
// the fp32 stream, output consumed by a bf16 GEMM -> we emit bf16 directly (one 4 B read + 2 B write
// per element) and keep (mean, rstd) per row for the backward.
// The backward fuses the residual-branch add (dh = dres + dLN) and emits both the fp32 stream
// gradient and the bf16 copy the next dX/dW GEMMs read; dgamma/dbeta are reduced per block in
// registers -> LDS -> one contiguous fp32 atomic row per block.
#pragma once
#include "tic_prims.h"

TIC_DEV float wave_sum(float v) { return wave64_sum(v); }

// NV = ceil(D / 256): float4 groups per lane.  in_stride: elements between consecutive rows of x.
template <int NV, bool NT = false>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, long in_stride,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, int rows, int D, float eps) {
    const int l = lane_id(), w = wave_id();
    f32x4 gm[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + l) * 4;
        if (c < D) {
            gm[i] = *reinterpret_cast<const f32x4*>(gamma + c);
            bt[i] = *reinterpret_cast<const f32x4*>(beta + c);
        } else {
            gm[i] = f32x4{0, 0, 0, 0};
            bt[i] = f32x4{0, 0, 0, 0};
        }
    }
    const float invD = 1.0f / (float)D;
    // the loads of the wave's NEXT row are issued before the current row is reduced and stored
    f32x4 v[NV];
    auto load_row = [&](int row, f32x4 (&vo)[NV]) {
        const float* xr = x + (long)row * in_stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + l) * 4;
            vo[i] = (c < D) ? ld_f4<NT>(xr + c) : f32x4{0, 0, 0, 0};
        }
    };
    const int step = TIC_NBLK_X * 4;
    int row = TIC_BID_X * 4 + w;
    if (row < rows) load_row(row, v);
    for (; row < rows; row += step) {
        f32x4 vn[NV];
        const bool more = row + step < rows;   // wave-uniform
        if (more) load_row(row + step, vn);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);   // columns past D hold zeros
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + l) * 4;
            if (c < D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = v[i][r] - mu;
                    q += d * d;
                }
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * invD + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + l) * 4;
            if (c < D) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[i][r] - mu) * rs * gm[i][r] + bt[i][r];
                st_u2<NT>(y + (long)row * D + c, u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])});
            }
        }
        if (l == 0) {
            mean_out[row] = mu;
            rstd_out[row] = rs;
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = vn[i];
        }
    }
}

// dy: bf16 [rows, D] dense.  x / dres / dx / dxb use `stride` elements between rows (dxb: bf16).
// dx = (dres ? dres : 0) + LN'(dy).  dgamma / dbeta: fp32 [D], accumulated with atomics.
// colsum (optional, fp32 [D]) += column sums of dx = the bias gradient of the Linear whose output gradient dx is.
// LDS: 3 * 4 * D floats.
template <int NV, bool NT = false>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x, long stride,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                      const float* __restrict__ rstd_in, const float* dres, float* dx,
                                                      bf16_t* dxb, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      float* colsum, int rows, int D) {
    const int l = lane_id(), w = wave_id();
    f32x4 gm[NV], dg[NV], db[NV], dc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + l) * 4;
        gm[i] = (c < D) ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0, 0, 0, 0};
        dg[i] = f32x4{0, 0, 0, 0};
        db[i] = f32x4{0, 0, 0, 0};
        dc[i] = f32x4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    // the loads of the wave's NEXT row are issued before the current row is reduced and stored: two rows of loads in flight per wave
    f32x4 xv[NV], dv[NV];
    u32x2 dr[NV];
    float mu = 0.f, rs = 0.f;
    auto load_row = [&](int row, f32x4 (&xo)[NV], u32x2 (&dro)[NV], f32x4 (&dvo)[NV], float& muo, float& rso) {
        muo = mean_in[row];
        rso = rstd_in[row];
        const float* xr = x + (long)row * stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + l) * 4;
            if (c < D) {
                xo[i] = ld_f4<NT>(xr + c);
                dro[i] = *reinterpret_cast<const u32x2*>(dy + (long)row * D + c);
                dvo[i] = dres ? ld_f4<NT>(dres + (long)row * stride + c) : f32x4{0, 0, 0, 0};
            } else {
                xo[i] = f32x4{0, 0, 0, 0};
                dro[i] = u32x2{0, 0};
                dvo[i] = f32x4{0, 0, 0, 0};
            }
        }
    };
    int row = TIC_BID_X * 4 + w;
    const int step = TIC_NBLK_X * 4;
    if (row < rows) load_row(row, xv, dr, dv, mu, rs);
    for (; row < rows; row += step) {
        f32x4 xn[NV], dn[NV];
        u32x2 drn[NV];
        float mun = 0.f, rsn = 0.f;
        const bool more = row + step < rows;   // wave-uniform
        if (more) load_row(row + step, xn, drn, dn, mun, rsn);
        f32x4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + l) * 4;
            if (c < D) {
                const float d[4] = {bf2f((bf16_t)(dr[i][0] & 0xffff)), bf2f((bf16_t)(dr[i][0] >> 16)),
                                    bf2f((bf16_t)(dr[i][1] & 0xffff)), bf2f((bf16_t)(dr[i][1] >> 16))};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xh[i][r] = (xv[i][r] - mu) * rs;
                    g[i][r] = d[r] * gm[i][r];
                    s1 += g[i][r];
                    s2 += g[i][r] * xh[i][r];
                    dg[i][r] += d[r] * xh[i][r];
                    db[i][r] += d[r];
                }
            } else {
                xh[i] = f32x4{0, 0, 0, 0};
                g[i] = f32x4{0, 0, 0, 0};
            }
        }
        const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + l) * 4;
            if (c < D) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rs * (g[i][r] - c1 - xh[i][r] * c2);
                const long off = (long)row * stride + c;
                o += dv[i];
                st_f4<NT>(dx + off, o);
                dc[i] += o;
                if (dxb) st_u2<NT>(dxb + off, u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])});
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                xv[i] = xn[i];
                dr[i] = drn[i];
                dv[i] = dn[i];
            }
            mu = mun;
            rs = rsn;
        }
    }
    // block reduction of dgamma / dbeta: waves -> LDS [2][4][D] -> contiguous atomics
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + l) * 4;
        if (c < D) {
            lds_stf4((uint32_t)((0 * 4 + w) * D + c) * 4u, dg[i]);
            lds_stf4((uint32_t)((1 * 4 + w) * D + c) * 4u, db[i]);
            lds_stf4((uint32_t)((2 * 4 + w) * D + c) * 4u, dc[i]);
        }
    }
    block_sync();
    for (int c = TIC_TID; c < D; c += 256) {
        float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            a += lds_ldf((uint32_t)((0 * 4 + ww) * D + c) * 4u);
            b += lds_ldf((uint32_t)((1 * 4 + ww) * D + c) * 4u);
            cc += lds_ldf((uint32_t)((2 * 4 + ww) * D + c) * 4u);
        }
        atomic_addf(dgamma + c, a);
        atomic_addf(dbeta + c, b);
        if (colsum) atomic_addf(colsum + c, cc);
    }
}
