// aug.h -- the ntrain.py augmentation pipeline as HIP image kernels (gfx950), so that the host DataLoader
// only has to deliver raw uint8 thumbnails.
//
// Replaces the per-sample CPU transforms of TIC/ViT/ntrain.py:93-136 (torchvision v2: RandomResizedCrop,
// RandomHorizontalFlip, ColorJitter, RandomGrayscale, RandomErasing, ToTensor, Normalize; the val/test
// Resize path is the same kernel with a full-image box and everything else off) and the batch-level
// MixUp / CutMix of ntrain.py:30-33,45-46.  Random PARAMETERS are sampled on the host side with the
// documented torchvision distributions (touhouimageclassification_amd/aug.py) and handed over as one
// [B, AUG_NPARAM] fp32 table; the pixel work -- HBM-trivial: 196 KB read + 602 KB written per image --
// happens here, one workgroup per image, fp32 between stages (no uint8 re-quantisation).
//
// ColorJitter's contrast needs the mean grey level of the image AS IT IS at that point of the random op
// order: pass 1 resizes/flips, applies the ops that precede contrast, stores the intermediate image in
// the output buffer and block-reduces the grey sum; pass 2 finishes in place.
#pragma once
#include "norm.h"

#define AUG_NPARAM 20
// params: 0 top 1 left 2 h 3 w 4 flip 5..8 op order 9 brightness 10 contrast 11 saturation 12 hue
//         13 jitter? 14 gray? 15 erase? 16 ei 17 ej 18 eh 19 ew
struct AugNorm { float mean[3], inv_std[3]; };

TIC_DEV float aug_gray(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
TIC_DEV float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

TIC_DEV void aug_hue(float& r, float& g, float& b, float dh) {
    const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
    const bool eq = (maxc == minc);
    const float cr = maxc - minc;
    const float s = cr / (eq ? 1.f : maxc);
    const float crd = eq ? 1.f : cr;
    const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
    float h;
    if (maxc == r) h = bc - gc;
    else if (maxc == g) h = 2.f + rc - bc;
    else h = 4.f + gc - rc;
    h = fmodf(h / 6.f + 1.f, 1.f);
    h = fmodf(h + dh + 1.f, 1.f);
    const float v = maxc;
    const float i = floorf(h * 6.f), f = h * 6.f - i;
    const int ii = ((int)i) % 6;
    const float p = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - f * s)), t = clamp01(v * (1.f - (1.f - f) * s));
    switch (ii) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// anti-aliased triangle filter taps of output index i over a source interval of `len` pixels starting at 0
TIC_DEV void aug_taps(int i, int len, int out, int& x0, int& n, float& center, float& support, float& invscale) {
    const float scale = (float)len / (float)out;
    support = scale >= 1.f ? scale : 1.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * ((float)i + 0.5f);
    x0 = (int)(center - support + 0.5f);
    if (x0 < 0) x0 = 0;
    int x1 = (int)(center + support + 0.5f);
    if (x1 > len) x1 = len;
    n = x1 - x0;
}

TIC_DEV void aug_apply_op(int op, float& r, float& g, float& b, const float* P, float mean_gray) {
    if (op == 0) {
        r = clamp01(r * P[9]); g = clamp01(g * P[9]); b = clamp01(b * P[9]);
    } else if (op == 1) {
        const float c = P[10], m = (1.f - c) * mean_gray;
        r = clamp01(c * r + m); g = clamp01(c * g + m); b = clamp01(c * b + m);
    } else if (op == 2) {
        const float s = P[11], gy = (1.f - s) * aug_gray(r, g, b);
        r = clamp01(s * r + gy); g = clamp01(s * g + gy); b = clamp01(s * b + gy);
    } else {
        aug_hue(r, g, b, P[12]);
    }
}

// images: [B, Hs, Ws, 3] uint8; out: [B, 3, S, S] fp32.  One workgroup (256 threads) per image.  LDS: 8 floats.
__global__ void __launch_bounds__(256) augment_kernel(const unsigned char* __restrict__ images, int Hs, int Ws,
                                                      const float* __restrict__ params, float* __restrict__ out, int S, AugNorm nm) {
    const int b = TIC_BID_X, tid = TIC_TID;
    const float* P = params + (long)b * AUG_NPARAM;
    const unsigned char* img = images + (long)b * Hs * Ws * 3;
    float* o = out + (long)b * 3 * S * S;
    const int top = (int)P[0], left = (int)P[1], ch = (int)P[2], cw = (int)P[3];
    const bool flip = P[4] != 0.f, jitter = P[13] != 0.f, gray = P[14] != 0.f, erase = P[15] != 0.f;
    int order[4] = {(int)P[5], (int)P[6], (int)P[7], (int)P[8]};
    int cpos = 4;   // position of the contrast op in the order (4 = none / jitter off)
    if (jitter) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (order[k] == 1) cpos = k;
    }
    const int ei = (int)P[16], ej = (int)P[17], eh = (int)P[18], ew = (int)P[19];
    float gsum = 0.f;
    // ---- pass 1: crop + antialiased resize + flip + jitter ops that precede contrast ----------------------
    for (int px = tid; px < S * S; px += 256) {
        const int oy = px / S, ox = px - oy * S;
        const int sx = flip ? (S - 1 - ox) : ox;
        int y0, ny, x0, nx;
        float cy, sy, iy, cx, sxp, ix;
        aug_taps(oy, ch, S, y0, ny, cy, sy, iy);
        aug_taps(sx, cw, S, x0, nx, cx, sxp, ix);
        float wxs = 0.f, wys = 0.f;
        for (int k = 0; k < nx; ++k) wxs += fmaxf(0.f, 1.f - fabsf(((float)(x0 + k) - cx + 0.5f) * ix));
        for (int k = 0; k < ny; ++k) wys += fmaxf(0.f, 1.f - fabsf(((float)(y0 + k) - cy + 0.5f) * iy));
        float r = 0.f, g = 0.f, bl = 0.f;
        for (int ky = 0; ky < ny; ++ky) {
            const float wy = fmaxf(0.f, 1.f - fabsf(((float)(y0 + ky) - cy + 0.5f) * iy)) / wys;
            const unsigned char* row = img + ((long)(top + y0 + ky) * Ws + left) * 3;
            float rr = 0.f, gg = 0.f, bb = 0.f;
            for (int kx = 0; kx < nx; ++kx) {
                const float wx = fmaxf(0.f, 1.f - fabsf(((float)(x0 + kx) - cx + 0.5f) * ix)) / wxs;
                const unsigned char* q = row + (x0 + kx) * 3;
                rr += wx * (float)q[0]; gg += wx * (float)q[1]; bb += wx * (float)q[2];
            }
            r += wy * rr; g += wy * gg; bl += wy * bb;
        }
        r *= (1.f / 255.f); g *= (1.f / 255.f); bl *= (1.f / 255.f);
        if (jitter) {
            for (int k = 0; k < cpos && k < 4; ++k) aug_apply_op(order[k], r, g, bl, P, 0.f);
            gsum += aug_gray(r, g, bl);
        }
        o[px] = r; o[S * S + px] = g; o[2 * S * S + px] = bl;
    }
    // ---- mean grey level of the intermediate image (block reduction) -----------------------------------------
    float mean_gray = 0.f;
    if (jitter) {   // wave-uniform: P is per block
        gsum = wave_sum(gsum);
        if (lane_id() == 0) lds_stf((uint32_t)wave_id() * 4u, gsum);
        block_sync();
        mean_gray = (lds_ldf(0) + lds_ldf(4) + lds_ldf(8) + lds_ldf(12)) / (float)(S * S);
    }
    // ---- pass 2: contrast + remaining ops + grayscale + erase + normalize (in place; same thread, same pixels)
    for (int px = tid; px < S * S; px += 256) {
        const int oy = px / S, ox = px - oy * S;
        float r = o[px], g = o[S * S + px], bl = o[2 * S * S + px];
        if (jitter)
            for (int k = cpos; k < 4; ++k) aug_apply_op(order[k], r, g, bl, P, mean_gray);
        if (gray) {
            const float gy = aug_gray(r, g, bl);
            r = gy; g = gy; bl = gy;
        }
        if (erase && oy >= ei && oy < ei + eh && ox >= ej && ox < ej + ew) {
            r = 0.f; g = 0.f; bl = 0.f;
        }
        o[px] = (r - nm.mean[0]) * nm.inv_std[0];
        o[S * S + px] = (g - nm.mean[1]) * nm.inv_std[1];
        o[2 * S * S + px] = (bl - nm.mean[2]) * nm.inv_std[2];
    }
}

// MixUp (mode 0): out[b] = lam x[b] + (1-lam) x[b-1];  CutMix (mode 1): box [y1,y2) x [x1,x2) pasted from x[b-1]
__global__ void __launch_bounds__(256) mix_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int H, int W, int mode,
                                                  float lam, int x1, int y1, int x2, int y2) {
    const long per = (long)C * H * W, total4 = (long)B * per / 4;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total4; i += (long)TIC_NBLK_X * 256) {
        const long e = i * 4, b = e / per, r = e - b * per;
        const long pb = (b == 0 ? (long)B - 1 : b - 1) * per + r;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + e), p = *reinterpret_cast<const f32x4*>(x + pb);
        f32x4 y;
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = lam * a[k] + (1.f - lam) * p[k];
        } else {
            const int hw = (int)(r % ((long)H * W)), yy = hw / W, xx = hw - yy * W;
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = (yy >= y1 && yy < y2 && xx + k >= x1 && xx + k < x2) ? p[k] : a[k];
        }
        *reinterpret_cast<f32x4*>(out + e) = y;
    }
}
// soft labels: out[b, c] = lam * [y[b] == c] + (1 - lam) * [y[b-1] == c]
__global__ void __launch_bounds__(256) mix_labels_kernel(const long long* __restrict__ y, float* __restrict__ out, int B, int ncls, float lam) {
    const int i = TIC_BID_X * 256 + TIC_TID;
    if (i < B * ncls) {
        const int b = i / ncls, c = i - b * ncls;
        const int pb = b == 0 ? B - 1 : b - 1;
        out[i] = (y[b] == c ? lam : 0.f) + (y[pb] == c ? 1.f - lam : 0.f);
    }
}
