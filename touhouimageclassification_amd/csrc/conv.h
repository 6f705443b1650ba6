// conv.h -- the ResNet conv stack of TIC/ResNet/model.py on gfx950 (NHWC bf16 activations).
//
// Convolutions are GEMMs over the MFMA kernels of gemm.h / gemm256.h / gemm_tn256.h:
//   forward   Y[M, Co]   = col(X)[M, Kp] . W[Co, Kp]^T        M = B*Ho*Wo, K = kh*kw*Ci (tap-major, padded to 64)
//   dgrad     dcol[M,Kp] = dY[M, Co] . (W^T)[Kp, Co]^T ; dX = col2im(dcol)   (gather form: no atomics)
//   wgrad     dW[Co, Kp] += dY^T . col(X)
// 1x1 stride-1 convs (most of a Bottleneck) need no im2col at all: the NHWC activation IS the GEMM operand.
// Round 1 materialises col(X) for k > 1 / strided convs (explicit im2col: 9x the activation bytes for a 3x3);
// folding the gather into the LDS-DMA source addresses (implicit GEMM) is the next step for this path.
// Replaces conv3x3 / conv1x1 / the 7x7 stem (TIC/ResNet/model.py:6-14,148), train-mode BatchNorm2d + ReLU
// (+ residual add) (:51-61,99-113,150-151), MaxPool2d(3,2,1) (:152), AdaptiveAvgPool2d(1) (:164) and their backward.
#pragma once
#include "norm.h"

struct ConvGeom {
    int B, H, W, Ci, Ho, Wo, kh, kw, stride, pad, K, Kp;   // K = kh*kw*Ci, Kp = K rounded up to 64
};

// OIHW fp32 -> [Co, Kp] bf16 with k = (ky*kw + kx)*Ci + c, zero padded; transposed == 1 writes [Kp, Co] (operand of the
// explicit dgrad GEMM); transposed == 2 writes the implicit-GEMM dgrad filter [Ci][(ky', kx') * Co + o] = w[o][c][kh-1-ky'][kw-1-kx']
// transposed == 3: the STEM layout of the implicit 7x7 / 2 convolution on a 4-channel-padded image (gemm.h): [Co][256] with
// k = ky * 32 + px * 4 + c, px = kx + 1 in 1..7 (px 0 is the extra left pixel of the aligned 8-pixel window: weight 0), c < 3 (c = 3: the
// padding channel), ky < 7 (row 7: zero)
#define TIC_STEM_KP 256
// transposed == 4 + 2 py + px (3x3 filters): the filter of ONE parity class of the stride-2 input gradient (tic_conv_igemm_dgrad_s2):
// dx[2a + py][2b + px] = sum over the taps ky = py ? {2, 0} : {1}, kx likewise, of dy[a + ky'][b + kx'] . w[:, :, ky, kx], with ky' = 0 for
// ky = 2 or 1 and ky' = 1 for ky = 0.  Layout [Ci][(ky' * kw' + kx') * Co + o], kh' = 1 + py, kw' = 1 + px: 1, 2, 2 and 4 of the 9 taps.
TIC_DEV void weight_ohwi_body(const float* __restrict__ w, bf16_t* __restrict__ out, int Co, const ConvGeom& g, int transposed) {
    const int taps = g.kh * g.kw;
    if (transposed >= 4 && transposed <= 7) {
        const int py = (transposed - 4) >> 1, px = (transposed - 4) & 1, kwp = 1 + px, tp = (1 + py) * kwp;
        const long total4 = (long)g.Ci * tp * Co;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total4; i += (long)TIC_NBLK_X * 256) {
            const int o = (int)(i % Co);
            long t = i / Co;
            const int tap = (int)(t % tp), c = (int)(t / tp);
            const int kyp = tap / kwp, kxp = tap - kyp * kwp;
            const int ky = py ? (kyp == 0 ? 2 : 0) : 1, kx = px ? (kxp == 0 ? 2 : 0) : 1;
            out[i] = f2bf(w[(((long)o * g.Ci + c) * 3 + ky) * 3 + kx]);
        }
        return;
    }
    if (transposed == 3) {
        const long total3 = (long)Co * TIC_STEM_KP;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total3; i += (long)TIC_NBLK_X * 256) {
            const int o = (int)(i / TIC_STEM_KP), k = (int)(i % TIC_STEM_KP), ky = k >> 5, px = (k >> 2) & 7, c = k & 3;
            float v = 0.f;
            if (ky < 7 && px >= 1 && c < 3) v = w[(((long)o * 3 + c) * 7 + ky) * 7 + (px - 1)];
            out[i] = f2bf(v);
        }
        return;
    }
    if (transposed == 2 && Co % 8 == 0) {   // [Ci][tap' * Co + o]: 8 consecutive o per thread, one 16-byte store
        const int oc = Co / 8;
        const long total2 = (long)g.Ci * taps * oc;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total2; i += (long)TIC_NBLK_X * 256) {
            const int o0 = (int)(i % oc) * 8;
            long t = i / oc;
            const int tap = (int)(t % taps), c = (int)(t / taps);
            const int ky = g.kh - 1 - tap / g.kw, kx = g.kw - 1 - tap % g.kw;
            const float* src = w + (((long)o0 * g.Ci + c) * g.kh + ky) * g.kw + kx;
            const long os = (long)g.Ci * taps;   // elements between consecutive output channels
            *reinterpret_cast<u32x4*>(out + i * 8) = u32x4{pack2bf(src[0], src[os]), pack2bf(src[2 * os], src[3 * os]),
                                                           pack2bf(src[4 * os], src[5 * os]), pack2bf(src[6 * os], src[7 * os])};
        }
        return;
    }
    if (transposed == 2) {
        const long total2 = (long)g.Ci * taps * Co;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total2; i += (long)TIC_NBLK_X * 256) {
            const int o = (int)(i % Co);
            long t = i / Co;
            const int tap = (int)(t % taps), c = (int)(t / taps);
            const int ky = g.kh - 1 - tap / g.kw, kx = g.kw - 1 - tap % g.kw;
            out[i] = f2bf(w[(((long)o * g.Ci + c) * g.kh + ky) * g.kw + kx]);
        }
        return;
    }
    if (transposed == 1 && Co % 8 == 0) {   // [Kp, Co]: 8 consecutive o per thread for one k
        const int oc = Co / 8;
        const long total1 = (long)g.Kp * oc;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total1; i += (long)TIC_NBLK_X * 256) {
            const int o0 = (int)(i % oc) * 8, k = (int)(i / oc);
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (k < g.K) {
                const int tap = k / g.Ci, c = k - tap * g.Ci, ky = tap / g.kw, kx = tap - ky * g.kw;
                const float* src = w + (((long)o0 * g.Ci + c) * g.kh + ky) * g.kw + kx;
                const long os = (long)g.Ci * taps;
                v = u32x4{pack2bf(src[0], src[os]), pack2bf(src[2 * os], src[3 * os]), pack2bf(src[4 * os], src[5 * os]), pack2bf(src[6 * os], src[7 * os])};
            }
            *reinterpret_cast<u32x4*>(out + i * 8) = v;
        }
        return;
    }
    if (transposed == 0 && g.Ci % 8 == 0) {   // [Co, Kp]: 8 consecutive channels of one tap per thread (Kp == K when Ci % 8 == 0 and K % 64 == 0, else zero tail)
        const int kc = g.Kp / 8;
        const long total0 = (long)Co * kc;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total0; i += (long)TIC_NBLK_X * 256) {
            const int o = (int)(i / kc), k0 = (int)(i - (long)o * kc) * 8;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (k0 < g.K) {
                const int tap = k0 / g.Ci, c0 = k0 - tap * g.Ci, ky = tap / g.kw, kx = tap - ky * g.kw;
                const float* src = w + (((long)o * g.Ci + c0) * g.kh + ky) * g.kw + kx;
                v = u32x4{pack2bf(src[0], src[taps]), pack2bf(src[2 * taps], src[3 * taps]), pack2bf(src[4 * taps], src[5 * taps]),
                          pack2bf(src[6 * taps], src[7 * taps])};
            }
            *reinterpret_cast<u32x4*>(out + i * 8) = v;
        }
        return;
    }
    const long total = (long)Co * g.Kp;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int o = (int)(i / g.Kp), k = (int)(i - (long)o * g.Kp);
        float v = 0.f;
        if (k < g.K) {
            const int tap = k / g.Ci, c = k - tap * g.Ci, ky = tap / g.kw, kx = tap - ky * g.kw;
            v = w[(((long)o * g.Ci + c) * g.kh + ky) * g.kw + kx];
        }
        out[transposed ? ((long)k * Co + o) : i] = f2bf(v);
    }
}
__global__ void __launch_bounds__(256) weight_ohwi_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Co, ConvGeom g, int transposed) {
    weight_ohwi_body(w, out, Co, g, transposed);
}
// every packed operand of a network in ONE launch: blockIdx.y = entry of a descriptor table in device memory (the per-conv launches
// are ~6 us each and there are two per convolution and optimizer step: 105 launches = 0.62 ms of a 29 ms ResNet-50 step)
struct ConvPackDesc {   // = TicConvPackDesc (include/tic_hip.h)
    const float* w;
    void* out;
    int Co, Ci, kh, kw, transposed, pad_;
};
struct ConvGradDesc {   // = TicConvGradDesc
    const float* dw;
    float* grad;
    int Co, Ci, kh, kw, layout, pad_;   // layout 0: dw is [Co, Kp] tap-major; 3: the stem layout above ([Co, 256])
};
TIC_DEV ConvGeom weight_geom(int Ci, int kh, int kw) {
    ConvGeom g;
    g.B = 1; g.H = kh; g.W = kw; g.Ci = Ci; g.Ho = 1; g.Wo = 1; g.kh = kh; g.kw = kw; g.stride = 1; g.pad = 0;
    g.K = kh * kw * Ci;
    g.Kp = (g.K + 63) / 64 * 64;
    return g;
}
__global__ void __launch_bounds__(256) weight_ohwi_many_kernel(const ConvPackDesc* __restrict__ descs) {
    const ConvPackDesc d = descs[TIC_BID_Y];
    weight_ohwi_body(d.w, (bf16_t*)d.out, d.Co, weight_geom(d.Ci, d.kh, d.kw), d.transposed);
}
// grad OIHW fp32 += dW [Co, Kp] (tap-major)
TIC_DEV void weight_grad_oihw_body(const float* __restrict__ dw, float* __restrict__ grad, int Co, const ConvGeom& g, int layout) {
    const long total = (long)Co * g.K;
    if (layout == 3) {
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
            const int kx = (int)(i % 7);
            long t = i / 7;
            const int ky = (int)(t % 7);
            t /= 7;
            const int c = (int)(t % 3), o = (int)(t / 3);
            grad[i] += dw[(long)o * TIC_STEM_KP + ky * 32 + (kx + 1) * 4 + c];
        }
        return;
    }
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        // i indexes the OIHW gradient
        const int kx = (int)(i % g.kw);
        long t = i / g.kw;
        const int ky = (int)(t % g.kh);
        t /= g.kh;
        const int c = (int)(t % g.Ci), o = (int)(t / g.Ci);
        grad[i] += dw[(long)o * g.Kp + (ky * g.kw + kx) * g.Ci + c];
    }
}
__global__ void __launch_bounds__(256) weight_grad_oihw_kernel(const float* __restrict__ dw, float* __restrict__ grad, int Co, ConvGeom g, int layout) {
    weight_grad_oihw_body(dw, grad, Co, g, layout);
}
__global__ void __launch_bounds__(256) weight_grad_oihw_many_kernel(const ConvGradDesc* __restrict__ descs) {
    const ConvGradDesc d = descs[TIC_BID_Y];
    weight_grad_oihw_body(d.dw, d.grad, d.Co, weight_geom(d.Ci, d.kh, d.kw), d.layout);
}

// x fp32 NCHW -> bf16 NHWC with Cp >= C channels per pixel (the stem's input: Cp = 4, channel 3 = 0, so a pixel is 8 bytes)
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int B, int C, int Cp, int H, int W) {
    const long total = (long)B * H * W * Cp;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c = (int)(i % Cp);
        long t = i / Cp;
        const int xx = (int)(t % W);
        t /= W;
        const int yy = (int)(t % H), b = (int)(t / H);
        out[i] = c < C ? f2bf(x[(((long)b * C + c) * H + yy) * W + xx]) : f2bf(0.f);
    }
}

// col[m, (ky*kw+kx)*Ci + c] = x[b, oy*s+ky-p, ox*s+kx-p, c] (0 outside the image; columns K..Kp-1 are 0).
// One thread per (m, tap, 8-channel chunk) when Ci % 8 == 0, per (m, k) otherwise (stem, Ci = 3).
__global__ void __launch_bounds__(256) im2col_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, ConvGeom g) {
    const long M = (long)g.B * g.Ho * g.Wo;
    if (g.Ci % 8 == 0) {
        const int cpt = g.Ci / 8, chunks = g.Kp / 8;
        const long total = M * chunks;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
            const long m = i / chunks;
            const int ch = (int)(i - m * chunks);
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (ch * 8 < g.K) {
                const int tap = ch / cpt, c0 = (ch - tap * cpt) * 8, ky = tap / g.kw, kx = tap - ky * g.kw;
                const int ox = (int)(m % g.Wo);
                const long t = m / g.Wo;
                const int oy = (int)(t % g.Ho), b = (int)(t / g.Ho);
                const int iy = oy * g.stride + ky - g.pad, ix = ox * g.stride + kx - g.pad;
                if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    v = *reinterpret_cast<const u32x4*>(x + (((long)b * g.H + iy) * g.W + ix) * g.Ci + c0);
            }
            *reinterpret_cast<u32x4*>(col + m * g.Kp + (long)ch * 8) = v;
        }
    } else {
        // stem (Ci = 3): one thread per (m, 8 consecutive k) -> one 16-byte store instead of eight 2-byte ones; the 8 elements
        // walk (ky, kx, c) incrementally (no division per element).  1.2 GB of col at B = 256: 2.03 ms -> see DESIGN
        const int chunks = g.Kp / 8;
        const long total = M * chunks;
        for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
            const long m = i / chunks;
            const int k0 = (int)(i - m * chunks) * 8;
            const int ox = (int)(m % g.Wo);
            const long t = m / g.Wo;
            const int oy = (int)(t % g.Ho), b = (int)(t / g.Ho);
            int tap = k0 / g.Ci, c = k0 - tap * g.Ci, ky = tap / g.kw, kx = tap - ky * g.kw;
            const bf16_t* img = x + (long)b * g.H * g.W * g.Ci;
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16_t v = 0;
                if (k0 + j < g.K) {
                    const int iy = oy * g.stride + ky - g.pad, ix = ox * g.stride + kx - g.pad;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) v = img[((long)iy * g.W + ix) * g.Ci + c];
                }
                e[j] = (unsigned short)v;
                if (++c == g.Ci) {
                    c = 0;
                    if (++kx == g.kw) {
                        kx = 0;
                        ++ky;
                    }
                }
            }
            *reinterpret_cast<u32x4*>(col + m * g.Kp + k0) = u32x4{e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16),
                                                                   e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16)};
        }
    }
}

// dx[b,iy,ix,c] (+)= sum over taps with (iy + p - ky) % s == 0 of dcol[(b,oy,ox), tap*Ci + c]   (Ci % 8 == 0)
__global__ void __launch_bounds__(256) col2im_kernel(const bf16_t* __restrict__ dcol, bf16_t* dx, ConvGeom g, int accumulate) {
    const int cpt = g.Ci / 8;
    const long total = (long)g.B * g.H * g.W * cpt;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c0 = (int)(i % cpt) * 8;
        long t = i / cpt;
        const int ix = (int)(t % g.W);
        t /= g.W;
        const int iy = (int)(t % g.H), b = (int)(t / g.H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int ky = 0; ky < g.kh; ++ky) {
            const int ny = iy + g.pad - ky;
            if (ny < 0 || ny % g.stride) continue;
            const int oy = ny / g.stride;
            if (oy >= g.Ho) continue;
            for (int kx = 0; kx < g.kw; ++kx) {
                const int nx = ix + g.pad - kx;
                if (nx < 0 || nx % g.stride) continue;
                const int ox = nx / g.stride;
                if (ox >= g.Wo) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcol + (((long)b * g.Ho + oy) * g.Wo + ox) * g.Kp + (ky * g.kw + kx) * g.Ci + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += bf2f((bf16_t)v[j]);
            }
        }
        bf16_t* dst = dx + (((long)b * g.H + iy) * g.W + ix) * g.Ci + c0;
        if (accumulate) {
            const bf16x8 o = *reinterpret_cast<const bf16x8*>(dst);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f((bf16_t)o[j]);
        }
        *reinterpret_cast<u32x4*>(dst) = u32x4{pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7])};
    }
}

// ---- BatchNorm (train mode) over [M, C] bf16, C % 8 == 0 ------------------------------------------------
// Column reductions WITHOUT atomics, so that they are bit-reproducible: row split k (blockIdx.y) STORES its partial sums
// part[k][0..C) = sum_m x, part[k][C..2C) = sum_m x^2 over its rows, and the consuming kernel adds the splits up in a fixed order
// (bn_sum_parts).  Why it matters: a BatchNorm statistic that moves by one fp32 ulp with the order of the adds moves a few bf16
// roundings of the normalised activation, and three or four layers later the whole activation carries an independent realisation
// of the bf16 storage noise -- two runs of the SAME step then differ by as much as either differs from exact arithmetic (measured on
// the atomics form: 13 % relative L2 between the gradients of two identical ResNet-18 steps).  With fixed-order sums the forward and
// the activation gradients are bit-identical run to run; only the weight-gradient accumulations (outputs, nothing downstream) keep
// an order dependence of a few fp32 ulps.  A block covers cpb = min(C/8, 32) 8-channel chunks with R = 256/cpb row lanes (so narrow
// layers -- C = 64 -- still use every thread); grid (ceil(C/256), row_splits); LDS 2*2048 floats.
__global__ void __launch_bounds__(256) bn_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long M, int C) {
    const int cpb = (C / 8) < 32 ? (C / 8) : 32, R = 256 / cpb;
    const int tx = TIC_TID % cpb, ty = TIC_TID / cpb;
    const int c0 = TIC_BID_X * 256 + tx * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
        // four rows per trip: four independent 16-byte loads in flight per thread (one per trip left the kernel latency-bound:
        // 30 us for a 13 MB tensor)
        const long st = (long)TIC_NBLK_Y * R;
        long m = (long)TIC_BID_Y * R + ty;
        for (; m + 3 * st < M; m += 4 * st) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(x + (m + u * st) * C + c0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f((bf16_t)v[u][j]);
                    s[j] += f;
                    q[j] += f * f;
                }
        }
        for (; m < M; m += st) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + m * C + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = bf2f((bf16_t)v[j]);
                s[j] += f;
                q[j] += f * f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lds_stf((uint32_t)(TIC_TID * 8 + j) * 4u, s[j]);            // [ty][tx][j]
        lds_stf((uint32_t)(2048 + TIC_TID * 8 + j) * 4u, q[j]);
    }
    block_sync();
    const int cw = cpb * 8;                                           // channels covered by this block
    if (TIC_TID < cw && TIC_BID_X * 256 + TIC_TID < C) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < R; ++k) {
            a += lds_ldf((uint32_t)(k * cw + TIC_TID) * 4u);
            b += lds_ldf((uint32_t)(2048 + k * cw + TIC_TID) * 4u);
        }
        float* dst = part + (size_t)TIC_BID_Y * 2 * C;
        dst[TIC_BID_X * 256 + TIC_TID] = a;
        dst[C + TIC_BID_X * 256 + TIC_TID] = b;
    }
}
// fixed-order sum of the row splits' partial sums for the block's BN_SUM_CH = 16 channels: thread (l, ch) = (tid / 16, tid % 16) adds
// splits l, l + 16, l + 32, ... -- eight independent running sums (eight loads in flight: the loop is a chain of L2 round trips, 512
// splits took 64 of them with one sum per thread), combined in a fixed order -- then lane 0 adds the sixteen lane sums in lane order.
// Valid in threads tid < 16 (channel c = blockIdx.x * 16 + tid, caller checks c < C).  LDS: 2 * 256 floats.
#define BN_SUM_CH 16
TIC_DEV void bn_sum_parts(const float* __restrict__ part, int nsplit, int C, float& a, float& b) {
    const int ch = TIC_TID % BN_SUM_CH, l = TIC_TID / BN_SUM_CH, c = TIC_BID_X * BN_SUM_CH + ch;
    constexpr int L = 256 / BN_SUM_CH;
    float sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
        int k = l;
        for (; k + 7 * L < nsplit; k += 8 * L) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sa[u] += part[(size_t)(k + u * L) * 2 * C + c];
                sb[u] += part[(size_t)(k + u * L) * 2 * C + C + c];
            }
        }
        for (int u = 0; k < nsplit; k += L, ++u) {
            sa[u] += part[(size_t)k * 2 * C + c];
            sb[u] += part[(size_t)k * 2 * C + C + c];
        }
    }
    a = ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]));
    b = ((sb[0] + sb[1]) + (sb[2] + sb[3])) + ((sb[4] + sb[5]) + (sb[6] + sb[7]));
    lds_stf((uint32_t)TIC_TID * 4u, a);
    lds_stf((uint32_t)(256 + TIC_TID) * 4u, b);
    block_sync();
    if (l == 0)
        for (int k = 1; k < L; ++k) {
            a += lds_ldf((uint32_t)(k * BN_SUM_CH + ch) * 4u);
            b += lds_ldf((uint32_t)(256 + k * BN_SUM_CH + ch) * 4u);
        }
}
// train: mean/rstd from the batch sums (the row splits' partials added up in a fixed order), running stats updated (unbiased var),
// counter += 1; eval (nsplit = 0): from running stats.  Grid ceil(C / BN_SUM_CH).
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* running_mean, float* running_var, long long* num_batches, long M, int C,
                                                           float eps, float momentum, int train) {
    float sa, sb;
    bn_sum_parts(part, train ? nsplit : 0, C, sa, sb);
    const int c = TIC_BID_X * BN_SUM_CH + TIC_TID;
    if (TIC_TID < BN_SUM_CH && c < C) {
        if (train) {
            const float mu = sa / (float)M;
            float var = sb / (float)M - mu * mu;
            if (var < 0.f) var = 0.f;
            mean[c] = mu;
            rstd[c] = 1.0f / sqrtf(var + eps);
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)M / (float)(M > 1 ? M - 1 : 1));
        } else {
            mean[c] = running_mean[c];
            rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
        }
    }
    if (train && num_batches && TIC_BID_X == 0 && TIC_TID == 0) *num_batches += 1;
}
// y = [relu]( (x - mean) rstd gamma + beta [+ identity] )
__global__ void __launch_bounds__(256) bn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, const bf16_t* identity,
                                                        bf16_t* __restrict__ y, long M, int C, int relu) {
    const int cpr = C / 8;
    const long total = M * cpr;
    const long first = (long)TIC_BID_X * 256 + TIC_TID, step = (long)TIC_NBLK_X * 256;
    // the launcher makes `step` a multiple of cpr, so a thread keeps ONE 8-channel chunk: per-channel scale / shift in registers
    const int c0 = (int)(first % cpr) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = rstd[c0 + j] * gamma[c0 + j];
        sh[j] = beta[c0 + j] - mean[c0 + j] * sc[j];
    }
    for (long i = first; i < total; i += step) {
        const long off = (i / cpr) * C + c0;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + off);
        bf16x8 idv = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (identity) idv = *reinterpret_cast<const bf16x8*>(identity + off);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf2f((bf16_t)v[j]) * sc[j] + sh[j];
            if (identity) f += bf2f((bf16_t)idv[j]);
            o[j] = (relu && f < 0.f) ? 0.f : f;
        }
        *reinterpret_cast<u32x4*>(y + off) = u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
}
// dz = dy * [y > 0] (if y given);  part[k][0..C) = sum dz ; part[k][C..2C) = sum dz * xhat over row split k (same thread mapping and
// the same atomics-free partial sums as bn_stats)
// mask_from_x (instead of y): the layer was y = relu(bn(x)) WITHOUT a residual add, so the mask is recomputed from x exactly as
// bn_apply_kernel formed it (same fp32 expression, same bf16 rounding) and y is not read at all.  gamma_m / beta_m are ALWAYS valid
// pointers (the host passes gamma for both when the mask is not wanted): a null test per channel in the prologue turned its
// vector loads into 8 dependent load-wait-branch rounds (+60 us per launch)
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const bf16_t* __restrict__ dy, const bf16_t* y, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ part, long M, int C,
                                                             const float* __restrict__ gamma_m, const float* __restrict__ beta_m, int mask_from_x) {
    const int cpb = (C / 8) < 32 ? (C / 8) : 32, R = 256 / cpb;
    const int tx = TIC_TID % cpb, ty = TIC_TID / cpb;
    const int c0 = TIC_BID_X * 256 + tx * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < C) {
        float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mu[j] = mean[c0 + j];
            rs[j] = rstd[c0 + j];
            sc[j] = rs[j] * gamma_m[c0 + j];
            sh[j] = beta_m[c0 + j] - mu[j] * sc[j];
        }
        const long st = (long)TIC_NBLK_Y * R;
        auto row = [&](const bf16x8& d, const bf16x8& xv, const bf16x8& yv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float dz = bf2f((bf16_t)d[j]);
                const float xf = bf2f((bf16_t)xv[j]);
                if (y && !(bf2f((bf16_t)yv[j]) > 0.f)) dz = 0.f;
                if (mask_from_x && !(bfround(xf * sc[j] + sh[j]) > 0.f)) dz = 0.f;
                s[j] += dz;
                q[j] += dz * (xf - mu[j]) * rs[j];
            }
        };
        long m = (long)TIC_BID_Y * R + ty;
        for (; m + st < M; m += 2 * st) {   // two rows per trip: 4-6 independent 16-byte loads in flight per thread
            bf16x8 d[2], xv[2], yv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                d[u] = *reinterpret_cast<const bf16x8*>(dy + (m + u * st) * C + c0);
                xv[u] = *reinterpret_cast<const bf16x8*>(x + (m + u * st) * C + c0);
                yv[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (y) yv[u] = *reinterpret_cast<const bf16x8*>(y + (m + u * st) * C + c0);
            }
            row(d[0], xv[0], yv[0]);
            row(d[1], xv[1], yv[1]);
        }
        for (; m < M; m += st) {
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + m * C + c0);
            const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + m * C + c0);
            bf16x8 yv = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (y) yv = *reinterpret_cast<const bf16x8*>(y + m * C + c0);
            row(d, xv, yv);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        lds_stf((uint32_t)(TIC_TID * 8 + j) * 4u, s[j]);
        lds_stf((uint32_t)(2048 + TIC_TID * 8 + j) * 4u, q[j]);
    }
    block_sync();
    const int cw = cpb * 8;
    if (TIC_TID < cw && TIC_BID_X * 256 + TIC_TID < C) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < R; ++k) {
            a += lds_ldf((uint32_t)(k * cw + TIC_TID) * 4u);
            b += lds_ldf((uint32_t)(2048 + k * cw + TIC_TID) * 4u);
        }
        float* dst = part + (size_t)TIC_BID_Y * 2 * C;
        dst[TIC_BID_X * 256 + TIC_TID] = a;
        dst[C + TIC_BID_X * 256 + TIC_TID] = b;
    }
}
// red[0..2C) = the splits' partial sums added up in a fixed order (what bn_bwd_apply reads);  dbeta += red[0..C), dgamma += red[C..2C).
// Grid ceil(C / BN_SUM_CH).
__global__ void __launch_bounds__(256) bn_param_grad_kernel(const float* __restrict__ part, int nsplit, float* __restrict__ red, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int C) {
    float sa, sb;
    bn_sum_parts(part, nsplit, C, sa, sb);
    const int c = TIC_BID_X * BN_SUM_CH + TIC_TID;
    if (TIC_TID < BN_SUM_CH && c < C) {
        red[c] = sa;
        red[C + c] = sb;
        dbeta[c] += sa;
        dgamma[c] += sb;
    }
}
// dx = gamma rstd (dz - dbeta/M - xhat dgamma/M);  dskip (optional) (+)= dz  (the identity-path gradient)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* y, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ red, bf16_t* __restrict__ dx, bf16_t* dskip, int skip_accumulate, long M, int C,
                                                            const float* __restrict__ beta_m, int mask_from_x) {
    const int cpr = C / 8;
    const long total = M * cpr;
    const float invM = 1.0f / (float)M;
    const long first = (long)TIC_BID_X * 256 + TIC_TID, step = (long)TIC_NBLK_X * 256;
    const int c0 = (int)(first % cpr) * 8;   // invariant per thread (step % cpr == 0): per-channel terms live in registers
    float mu[8], rs[8], k0[8], k1[8], k2[8], sc[8], sh[8];  // dx = k0 dz - k1 - xhat k2;  beta_m / mask_from_x: see the reduce kernel
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = mean[c0 + j];
        rs[j] = rstd[c0 + j];
        sc[j] = rs[j] * gamma[c0 + j];
        sh[j] = beta_m[c0 + j] - mu[j] * sc[j];
        k0[j] = gamma[c0 + j] * rs[j];
        k1[j] = k0[j] * red[c0 + j] * invM;
        k2[j] = k0[j] * red[C + c0 + j] * invM;
    }
    for (long i = first; i < total; i += step) {
        const long off = (i / cpr) * C + c0;
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + off);
        bf16x8 yv = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (y) yv = *reinterpret_cast<const bf16x8*>(y + off);
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + off);
        float o[8], z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float dz = bf2f((bf16_t)d[j]);
            const float xf = bf2f((bf16_t)xv[j]);
            if (y && !(bf2f((bf16_t)yv[j]) > 0.f)) dz = 0.f;
            if (mask_from_x && !(bfround(xf * sc[j] + sh[j]) > 0.f)) dz = 0.f;
            const float xh = (xf - mu[j]) * rs[j];
            o[j] = k0[j] * dz - k1[j] - xh * k2[j];
            z[j] = dz;
        }
        *reinterpret_cast<u32x4*>(dx + off) = u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
        if (dskip) {
            if (skip_accumulate) {
                const bf16x8 p = *reinterpret_cast<const bf16x8*>(dskip + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] += bf2f((bf16_t)p[j]);
            }
            *reinterpret_cast<u32x4*>(dskip + off) = u32x4{pack2bf(z[0], z[1]), pack2bf(z[2], z[3]), pack2bf(z[4], z[5]), pack2bf(z[6], z[7])};
        }
    }
}

// ---- pools (NHWC bf16, C % 8 == 0) -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo) {
    const int cpr = C / 8;
    const long total = (long)B * Ho * Wo * cpr;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c0 = (int)(i % cpr) * 8;
        long t = i / cpr;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho), b = (int)(t / Ho);
        float mx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mx[j] = -__builtin_huge_valf();
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((long)b * H + iy) * W + ix) * C + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) mx[j] = fmaxf(mx[j], bf2f((bf16_t)v[j]));
            }
        }
        *reinterpret_cast<u32x4*>(y + (((long)b * Ho + oy) * Wo + ox) * C + c0) =
            u32x4{pack2bf(mx[0], mx[1]), pack2bf(mx[2], mx[3]), pack2bf(mx[4], mx[5]), pack2bf(mx[6], mx[7])};
    }
}
// gather form: an input pixel receives dy of every window whose FIRST maximum (scan order) it is.
// One thread per (pixel, 8 channels); per window only the positions scanned before this pixel are re-read.
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const bf16_t* __restrict__ dy,
                                                           bf16_t* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
    const int cpr = C / 8;
    const long total = (long)B * H * W * cpr;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c0 = (int)(i % cpr) * 8;
        long t = i / cpr;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H), b = (int)(t / H);
        const bf16x8 xr = *reinterpret_cast<const bf16x8*>(x + i * 8);
        float xv[8], acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xv[j] = bf2f((bf16_t)xr[j]);
            acc[j] = 0.f;
        }
        const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;   // windows holding row iy: 2 oy - 1 <= iy <= 2 oy + 1
        for (int oy = oy0; oy <= oy1 && oy < Ho; ++oy) {
            for (int ox = ox0; ox <= ox1 && ox < Wo; ++ox) {
                const long oo = (((long)b * Ho + oy) * Wo + ox) * C + c0;
                const bf16x8 yr = *reinterpret_cast<const bf16x8*>(y + oo);
                bool m[8], any = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    m[j] = bf2f((bf16_t)yr[j]) == xv[j];
                    any |= m[j];
                }
                if (!any) continue;
                const int py = iy - (oy * 2 - 1), px = ix - (ox * 2 - 1);   // my position inside the window
                for (int ky = 0; ky <= py; ++ky) {
                    const int yy = oy * 2 + ky - 1;
                    if (yy < 0) continue;
                    const int kend = ky < py ? 3 : px;
                    for (int kx = 0; kx < kend; ++kx) {
                        const int xx = ox * 2 + kx - 1;
                        if (xx < 0 || xx >= W) continue;
                        const bf16x8 e = *reinterpret_cast<const bf16x8*>(x + (((long)b * H + yy) * W + xx) * C + c0);
#pragma unroll
                        for (int j = 0; j < 8; ++j) m[j] = m[j] && !(bf2f((bf16_t)e[j]) == xv[j]);
                    }
                }
                const bf16x8 dr = *reinterpret_cast<const bf16x8*>(dy + oo);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += m[j] ? bf2f((bf16_t)dr[j]) : 0.f;
            }
        }
        *reinterpret_cast<u32x4*>(dx + i * 8) =
            u32x4{pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7])};
    }
}
// The same pool with the window position (ky * 3 + kx, 0..8) of the FIRST maximum written beside y: the backward then needs neither x
// nor y -- an input pixel asks the <= 4 windows that hold it whether it was their argmax (24 bytes per window from tensors a quarter of
// the input's size, shared by up to 9 pixels) instead of re-scanning every window (maxpool_bwd_kernel: 0.9 ms at B = 256, 4.5 x the
// time of its bytes).
__global__ void __launch_bounds__(256) maxpool_fwd_idx_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, unsigned char* __restrict__ idx,
                                                               int B, int H, int W, int C, int Ho, int Wo) {
    const int cpr = C / 8;
    const long total = (long)B * Ho * Wo * cpr;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c0 = (int)(i % cpr) * 8;
        long t = i / cpr;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho), b = (int)(t / Ho);
        float mx[8];
        uint32_t k8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mx[j] = -__builtin_huge_valf();
            k8[j] = 0;
        }
        bool first = true;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((long)b * H + iy) * W + ix) * C + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = bf2f((bf16_t)v[j]);
                    if (first || f > mx[j]) {   // strictly greater: ties keep the earlier position
                        mx[j] = f;
                        k8[j] = (uint32_t)(ky * 3 + kx);
                    }
                }
                first = false;
            }
        }
        const long o = (((long)b * Ho + oy) * Wo + ox) * C + c0;
        *reinterpret_cast<u32x4*>(y + o) = u32x4{pack2bf(mx[0], mx[1]), pack2bf(mx[2], mx[3]), pack2bf(mx[4], mx[5]), pack2bf(mx[6], mx[7])};
        *reinterpret_cast<u32x2*>(idx + o) = u32x2{k8[0] | (k8[1] << 8) | (k8[2] << 16) | (k8[3] << 24), k8[4] | (k8[5] << 8) | (k8[6] << 16) | (k8[7] << 24)};
    }
}
// The stem's bn1 -> relu -> maxpool in one pass: y = maxpool3x3/2(relu(bn(x))) with the argmax position, relu(bn(x)) rounded to bf16 per
// element exactly as bn_apply_kernel stores it but never stored (411 MB written + read again at B = 256)
__global__ void __launch_bounds__(256) bn_relu_maxpool_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                                   unsigned char* idx, int B, int H, int W, int C, int Ho, int Wo) {
    const int cpr = C / 8;
    const long total = (long)B * Ho * Wo * cpr;
    const long first = (long)TIC_BID_X * 256 + TIC_TID, step = (long)TIC_NBLK_X * 256;   // step % cpr == 0: one channel chunk per thread
    const int c0 = (int)(first % cpr) * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = rstd[c0 + j] * gamma[c0 + j];
        sh[j] = beta[c0 + j] - mean[c0 + j] * sc[j];
    }
    for (long i = first; i < total; i += step) {
        long t = i / cpr;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho), b = (int)(t / Ho);
        float mx[8];
        uint32_t k8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mx[j] = -__builtin_huge_valf();
            k8[j] = 0;
        }
        bool firstpos = true;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky - 1;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((long)b * H + iy) * W + ix) * C + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = bf2f((bf16_t)v[j]) * sc[j] + sh[j];
                    f = bfround(f < 0.f ? 0.f : f);
                    if (firstpos || f > mx[j]) {
                        mx[j] = f;
                        k8[j] = (uint32_t)(ky * 3 + kx);
                    }
                }
                firstpos = false;
            }
        }
        const long o = (((long)b * Ho + oy) * Wo + ox) * C + c0;
        *reinterpret_cast<u32x4*>(y + o) = u32x4{pack2bf(mx[0], mx[1]), pack2bf(mx[2], mx[3]), pack2bf(mx[4], mx[5]), pack2bf(mx[6], mx[7])};
        if (idx) *reinterpret_cast<u32x2*>(idx + o) = u32x2{k8[0] | (k8[1] << 8) | (k8[2] << 16) | (k8[3] << 24), k8[4] | (k8[5] << 8) | (k8[6] << 16) | (k8[7] << 24)};
    }
}
__global__ void __launch_bounds__(256) maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx,
                                                               int B, int H, int W, int C, int Ho, int Wo) {
    const int cpr = C / 8;
    const long total = (long)B * H * W * cpr;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c0 = (int)(i % cpr) * 8;
        long t = i / cpr;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H), b = (int)(t / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;   // windows holding row iy: 2 oy - 1 <= iy <= 2 oy + 1
        for (int oy = oy0; oy <= oy1 && oy < Ho; ++oy)
            for (int ox = ox0; ox <= ox1 && ox < Wo; ++ox) {
                const long oo = (((long)b * Ho + oy) * Wo + ox) * C + c0;
                const uint32_t me = (uint32_t)((iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1)));   // my position inside the window
                const u32x2 k = *reinterpret_cast<const u32x2*>(idx + oo);
                const bf16x8 dr = *reinterpret_cast<const bf16x8*>(dy + oo);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (((k[j >> 2] >> (8 * (j & 3))) & 0xffu) == me) ? bf2f((bf16_t)dr[j]) : 0.f;
            }
        *reinterpret_cast<u32x4*>(dx + i * 8) =
            u32x4{pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7])};
    }
}
// [B, HW, C] -> [B, C] mean ; backward broadcast / HW
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int HW, int C) {
    const int i = TIC_BID_X * 256 + TIC_TID;
    if (i < B * C) {
        const int b = i / C, c = i - b * C;
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += bf2f(x[((long)b * HW + p) * C + c]);
        y[i] = f2bf(s / (float)HW);
    }
}
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int HW, int C) {
    const long total = (long)B * HW * C;
    const float inv = 1.0f / (float)HW;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int c = (int)(i % C), b = (int)(i / ((long)HW * C));
        dx[i] = f2bf(bf2f(dy[(long)b * C + c]) * inv);
    }
}
__global__ void __launch_bounds__(256) add_bf16_kernel(bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long n8) {
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < n8; i += (long)TIC_NBLK_X * 256) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + i * 8), y = *reinterpret_cast<const bf16x8*>(b + i * 8);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)x[j]) + bf2f((bf16_t)y[j]);
        *reinterpret_cast<u32x4*>(a + i * 8) = u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
}
