// elementwise.h -- the HBM-bound side kernels of the ViT fine-tune step (gfx950).
//
//   patchify        pixel_values fp32 NCHW -> bf16 patch matrix [B*196, 768]  (HF modeling_vit.py:60,69:
//                   Conv2d(k = s = 16) is a pure re-index + GEMM; feature order (c, ky, kx))
//   embed_cls       h[b,0,:] = cls + pos[0,:]                                 (HF modeling_vit.py:146-157)
//   colsum          bias gradients: out[n] += sum_m dY[m,n]
//   cast / cast_transpose   fp32 master weights -> bf16 GEMM operands W and W^T (autocast's per-step
//                   weight cast, TIC/ViT/ntrain.py:241 precision="bf16-mixed")
//   adamw           torch.optim.AdamW semantics, every parameter decayed (TIC/ViT/ntrain.py:39-41)
//   head_fwd/bwd    classifier Linear(D -> C) on the CLS rows               (HF modeling_vit.py:559-561)
//   xent            mean cross-entropy, hard or soft targets, + dlogits      (TIC/ViT/finetune.py:61;
//                   TIC/ViT/ntrain.py:48)
#pragma once
#include "norm.h"

// x: [B,3,224,224] fp32 -> P: [B*G*G, 3*16*16] bf16.  One thread = 8 consecutive kx of one (b,c,y).
__global__ void __launch_bounds__(256) patchify_kernel(const float* __restrict__ x, bf16_t* __restrict__ P, int B, int C, int img, int patch) {
    const int G = img / patch, per_row = img / 8;
    const long total = (long)B * C * img * per_row;
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < total; i += (long)TIC_NBLK_X * 256) {
        const int x8 = (int)(i % per_row);
        long t = i / per_row;
        const int y = (int)(t % img);
        t /= img;
        const int c = (int)(t % C), b = (int)(t / C);
        const float* src = x + (((long)b * C + c) * img + y) * img + x8 * 8;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
        const int gy = y / patch, ky = y - gy * patch, xx = x8 * 8, gx = xx / patch, kx = xx - gx * patch;
        bf16_t* dst = P + ((long)b * G * G + gy * G + gx) * (C * patch * patch) + (c * patch + ky) * patch + kx;
        *reinterpret_cast<u32x4*>(dst) = u32x4{pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]), pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3])};
    }
}

__global__ void __launch_bounds__(256) embed_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ h, int B, int N, int D) {
    const int i = TIC_BID_X * 256 + TIC_TID;
    if (i < B * D) {
        const int b = i / D, d = i - b * D;
        h[(long)b * N * D + d] = cls[d] + pos[d];
    }
}

// d(cls) = sum_b dh[b,0,:],  d(pos)[t,:] = sum_b dh[b,t,:]   (fp32 stream gradient, [B,N,D])
__global__ void __launch_bounds__(256) embed_bwd_kernel(const float* __restrict__ dh, float* __restrict__ dcls, float* __restrict__ dpos, int B, int N, int D) {
    const int i = TIC_BID_X * 256 + TIC_TID;
    if (i < N * D) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dh[(long)b * N * D + i];
        dpos[i] += s;
        if (i < D) dcls[i] += s;
    }
}

// out[n] += sum_m in[m,n];  grid (ceil(N/256), row_splits); LDS 8*256 floats
__global__ void __launch_bounds__(256) colsum_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, int M, int N) {
    const int tx = TIC_TID & 31, ty = TIC_TID >> 5;
    const int c0 = TIC_BID_X * 256 + tx * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c0 < N) {
        for (int m = TIC_BID_Y * 8 + ty; m < M; m += TIC_NBLK_Y * 8) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(in + (long)m * N + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += bf2f((bf16_t)v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) lds_stf((uint32_t)(ty * 256 + tx * 8 + j) * 4u, acc[j]);
    block_sync();
    const int c = TIC_BID_X * 256 + TIC_TID;
    if (c < N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += lds_ldf((uint32_t)(k * 256 + TIC_TID) * 4u);
        atomic_addf(out + c, s);
    }
}

__global__ void __launch_bounds__(256) cast_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n4) {
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < n4; i += (long)TIC_NBLK_X * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i * 4);
        *reinterpret_cast<u32x2*>(out + i * 4) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    }
}

// in fp32 [R,C] -> out bf16 [C,R]; 64x64 tiles through LDS (row stride 68 elements keeps 8-B alignment)
#define CT_STRIDE 68
__global__ void __launch_bounds__(256) cast_transpose_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int R, int C) {
    const int r0 = TIC_BID_Y * 64, c0 = TIC_BID_X * 64, t = TIC_TID;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // a thread owns a PAIR of rows so every transposed LDS store is one whole 32-bit word
        const int row = 2 * (i * 16 + (t >> 4)), col = (t & 15) * 4;
        const f32x4 va = *reinterpret_cast<const f32x4*>(in + (long)(r0 + row) * C + c0 + col);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(in + (long)(r0 + row + 1) * C + c0 + col);
#pragma unroll
        for (int j = 0; j < 4; ++j)   // tile[col + j][row .. row+1]
            lds_stf((uint32_t)((col + j) * CT_STRIDE + row) * 2u, __builtin_bit_cast(float, pack2bf(va[j], vb[j])));
    }
    block_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int crow = i * 16 + (t >> 4), rr = (t & 15) * 4;
        const bf16x4 v = lds_ld64((uint32_t)(crow * CT_STRIDE + rr) * 2u);
        *reinterpret_cast<bf16x4*>(out + (long)(c0 + crow) * R + r0 + rr) = v;
    }
}

// the same for every Linear weight of a transformer stack in ONE launch: blockIdx.y = layer, blockIdx.x = 64x64 tile of the layer's
// four matrices (tile_end[i]: running tile count).  96 launches of ~6 us -> one.
struct CastTransposeGroup {
    long in_off[4], out_off[4];   // element offsets inside a layer (fp32 parameters / bf16 transposes)
    int R[4], C[4], tile_end[4];
    long in_stride, out_stride;   // elements between layers
};
__global__ void __launch_bounds__(256) cast_transpose_group_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, CastTransposeGroup gp) {
    int tile = TIC_BID_X, which = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (TIC_BID_X >= gp.tile_end[i]) {
            which = i + 1;
            tile = TIC_BID_X - gp.tile_end[i];
        }
    const int R = gp.R[which], C = gp.C[which], tc = C / 64;
    const float* src = in + (long)TIC_BID_Y * gp.in_stride + gp.in_off[which];
    bf16_t* dst = out + (long)TIC_BID_Y * gp.out_stride + gp.out_off[which];
    const int r0 = (tile / tc) * 64, c0 = (tile % tc) * 64, t = TIC_TID;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * (i * 16 + (t >> 4)), col = (t & 15) * 4;
        const f32x4 va = *reinterpret_cast<const f32x4*>(src + (long)(r0 + row) * C + c0 + col);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(src + (long)(r0 + row + 1) * C + c0 + col);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_stf((uint32_t)((col + j) * CT_STRIDE + row) * 2u, __builtin_bit_cast(float, pack2bf(va[j], vb[j])));
    }
    block_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int crow = i * 16 + (t >> 4), rr = (t & 15) * 4;
        const bf16x4 v = lds_ld64((uint32_t)(crow * CT_STRIDE + rr) * 2u);
        *reinterpret_cast<bf16x4*>(dst + (long)(c0 + crow) * R + r0 + rr) = v;
    }
}

// flat AdamW over n4*4 elements; optional bf16 shadow of the updated parameters
template <bool NT = false>
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, bf16_t* w16, long n4, float lr, float b1, float b2,
                                                     float eps, float wd, float inv_bc1, float inv_sqrt_bc2) {
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < n4; i += (long)TIC_NBLK_X * 256) {
        f32x4 pv = ld_f4<NT>(p + i * 4);
        const f32x4 gv = ld_f4<NT>(g + i * 4);
        f32x4 mv = ld_f4<NT>(m + i * 4);
        f32x4 vv = ld_f4<NT>(v + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] *= (1.0f - lr * wd);
            mv[r] = b1 * mv[r] + (1.0f - b1) * gv[r];
            vv[r] = b2 * vv[r] + (1.0f - b2) * gv[r] * gv[r];
            const float denom = sqrtf(vv[r]) * inv_sqrt_bc2 + eps;
            pv[r] -= (lr * inv_bc1) * (mv[r] / denom);
        }
        st_f4<NT>(p + i * 4, pv);
        st_f4<NT>(m + i * 4, mv);
        st_f4<NT>(v + i * 4, vv);
        if (w16) *reinterpret_cast<u32x2*>(w16 + i * 4) = u32x2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};   // re-read by the next forward
    }
}

// AdamW of the four Linear weight matrices of every transformer block, tile by tile: the update is the flat kernel's, and the tile that is
// in registers anyway also leaves as BOTH bf16 operand forms the next step's GEMMs read -- w16 (row-major, forward / dW) and wT16 (transposed,
// the dX GEMMs) -- so the per-step cast + transpose pass over the fp32 weights (1.2 GB read + 0.6 GB written for ViT-L) disappears.
// Grid (tiles per layer, layers) as cast_transpose_group_kernel; a 64 x 64 tile = 256 threads x 2 trips x (2 rows x 4 columns).
struct AdamwHyper {
    float lr, b1, b2, eps, wd, inv_bc1, inv_sqrt_bc2;
};
TIC_DEV void adamw4(f32x4& pv, const f32x4& gv, f32x4& mv, f32x4& vv, const AdamwHyper& h) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // the same expression order as adamw_kernel: bit-identical parameters
        pv[r] *= (1.0f - h.lr * h.wd);
        mv[r] = h.b1 * mv[r] + (1.0f - h.b1) * gv[r];
        vv[r] = h.b2 * vv[r] + (1.0f - h.b2) * gv[r] * gv[r];
        const float denom = sqrtf(vv[r]) * h.inv_sqrt_bc2 + h.eps;
        pv[r] -= (h.lr * h.inv_bc1) * (mv[r] / denom);
    }
}
__global__ void __launch_bounds__(256) adamw_tiles_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                           bf16_t* __restrict__ w16, bf16_t* __restrict__ wT, CastTransposeGroup gp, AdamwHyper h) {
    int tile = TIC_BID_X, which = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (TIC_BID_X >= gp.tile_end[i]) {
            which = i + 1;
            tile = TIC_BID_X - gp.tile_end[i];
        }
    const int R = gp.R[which], C = gp.C[which], tc = C / 64;
    const long base = (long)TIC_BID_Y * gp.in_stride + gp.in_off[which];   // the flat fp32 buffers and w16 share one layout
    bf16_t* dst = wT + (long)TIC_BID_Y * gp.out_stride + gp.out_off[which];
    const int r0 = (tile / tc) * 64, c0 = (tile % tc) * 64, t = TIC_TID;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 2 * (i * 16 + (t >> 4)), col = (t & 15) * 4;
        f32x4 pv[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long o = base + (long)(r0 + row + k) * C + c0 + col;
            pv[k] = *reinterpret_cast<const f32x4*>(p + o);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + o);
            f32x4 mv = *reinterpret_cast<const f32x4*>(m + o), vv = *reinterpret_cast<const f32x4*>(v + o);
            adamw4(pv[k], gv, mv, vv, h);
            *reinterpret_cast<f32x4*>(p + o) = pv[k];
            *reinterpret_cast<f32x4*>(m + o) = mv;
            *reinterpret_cast<f32x4*>(v + o) = vv;
            *reinterpret_cast<u32x2*>(w16 + o) = u32x2{pack2bf(pv[k][0], pv[k][1]), pack2bf(pv[k][2], pv[k][3])};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_stf((uint32_t)((col + j) * CT_STRIDE + row) * 2u, __builtin_bit_cast(float, pack2bf(pv[0][j], pv[1][j])));
    }
    block_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int crow = i * 16 + (t >> 4), rr = (t & 15) * 4;
        const bf16x4 tv = lds_ld64((uint32_t)(crow * CT_STRIDE + rr) * 2u);
        *reinterpret_cast<bf16x4*>(dst + (long)(c0 + crow) * R + r0 + rr) = tv;
    }
}
// The element ranges OUTSIDE the per-layer weight matrices (biases, LayerNorm, embeddings, head: 0.3 % of ViT-L), as a launch grid:
// grid (L + 2, 5 x parts): blocks (l, k + 5 j) walk gap k of layer l (the <= 5 stretches between / around its four matrices); x = L walks
// [0, l0), x = L + 1 walks [l0 + L * stride, n) (there all of y strides).  Every boundary is a multiple of 8 elements (tic_vit_layout pads).
// (The first version tested every float4 of the whole buffer against the ranges with a 64-bit modulo: 82 us for 0.3 % of the bytes.)
struct RestGaps {
    long l0, stride, nlayers, n;
    long lo[5], len[5];   // relative to the layer's start
};
TIC_DEV void rest_range(const RestGaps& z, long& start, long& n4, long& first, long& step) {
    const long x = TIC_BID_X;
    first = TIC_TID;
    step = 256;
    if (x < z.nlayers) {
        const int k = TIC_BID_Y % 5, part = TIC_BID_Y / 5, parts = TIC_NBLK_Y / 5;   // grid.y = 5 x parts
        start = z.l0 + x * z.stride + z.lo[k];
        n4 = z.len[k] / 4;
        first += (long)part * 256;
        step *= parts;
    } else {
        start = (x == z.nlayers) ? 0 : z.l0 + z.nlayers * z.stride;
        n4 = ((x == z.nlayers) ? z.l0 : z.n - start) / 4;
        first += (long)TIC_BID_Y * 256;
        step *= TIC_NBLK_Y;
    }
}
// ... and everything else by the flat rule
__global__ void __launch_bounds__(256) adamw_rest_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                          bf16_t* __restrict__ w16, RestGaps z, AdamwHyper h) {
    long start, n4, first, step;
    rest_range(z, start, n4, first, step);
    for (long j = first; j < n4; j += step) {
        const long i = start + j * 4;
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 mv = *reinterpret_cast<const f32x4*>(m + i), vv = *reinterpret_cast<const f32x4*>(v + i);
        adamw4(pv, gv, mv, vv, h);
        *reinterpret_cast<f32x4*>(p + i) = pv;
        *reinterpret_cast<f32x4*>(m + i) = mv;
        *reinterpret_cast<f32x4*>(v + i) = vv;
        *reinterpret_cast<u32x2*>(w16 + i) = u32x2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
    }
}

// up to 4 fp32 arrays cleared by ONE launch (the C matrices of a grouped weight-gradient launch whose route adds partial tiles: four
// hipMemsetAsync calls cost ~6 us each on the stream, this ~8 us for the 50 MB of a ViT-L block).  grid (blocks, n arrays); n4 = float4 count
struct ZeroMany {
    float* p[4];
    long n4[4];
};
__global__ void __launch_bounds__(256) zero_many_kernel(ZeroMany z) {
    float* p = z.p[0];
    long n4 = z.n4[0];
#pragma unroll
    for (int g = 1; g < 4; ++g)
        if (TIC_BID_Y == g) {
            p = z.p[g];
            n4 = z.n4[g];
        }
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < n4; i += (long)TIC_NBLK_X * 256) *reinterpret_cast<f32x4*>(p + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// g = 0 outside the weight-matrix ranges: the gradients that ACCUMULATE (bias / LayerNorm column sums, embeddings, head) start from
// zero while the matrix gradients, which the backward stores whole, are not touched
__global__ void __launch_bounds__(256) zero_gaps_kernel(float* __restrict__ g, RestGaps z) {
    long start, n4, first, step;
    rest_range(z, start, n4, first, step);
    for (long j = first; j < n4; j += step) *reinterpret_cast<f32x4*>(g + start + j * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// logits[b,c] = bf16( sum_d bf16(z[b,d]) * bf16(W[c,d]) + bf16(bias[c]) ) as fp32; one wave per (b,c)
__global__ void __launch_bounds__(256) head_fwd_kernel(const bf16_t* __restrict__ z, const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ logits, int B, int C, int D) {
    const int l = lane_id(), w = wave_id();
    const int pair = TIC_BID_X * 4 + w;
    if (pair >= B * C) return;   // wave-uniform
    const int b = pair / C, c = pair - b * C;
    float s = 0.f;
    for (int d = l * 4; d < D; d += 256) {
        const bf16x4 zv = *reinterpret_cast<const bf16x4*>(z + (long)b * D + d);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(W + (long)c * D + d);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += bf2f((bf16_t)zv[j]) * bfround(wv[j]);
    }
    s = wave_sum(s);
    if (l == 0) logits[pair] = bfround(s + bfround(bias[c]));
}

// dz[b,d] = bf16( sum_c bf16(dl[b,c]) * bf16(W[c,d]) );  thread per (b,d)
__global__ void __launch_bounds__(256) head_bwd_dz_kernel(const float* __restrict__ dl, const float* __restrict__ W, bf16_t* __restrict__ dz, int B, int C, int D) {
    const int i = TIC_BID_X * 256 + TIC_TID;
    if (i < B * D) {
        const int b = i / D, d = i - b * D;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += bfround(dl[b * C + c]) * bfround(W[(long)c * D + d]);
        dz[i] = f2bf(s);
    }
}
// dW[c,d] += sum_b bf16(dl[b,c]) * z[b,d];  db[c] += sum_b bf16(dl[b,c]);  thread per (c,d)
__global__ void __launch_bounds__(256) head_bwd_dw_kernel(const float* __restrict__ dl, const bf16_t* __restrict__ z, float* __restrict__ dW, float* __restrict__ db, int B, int C, int D) {
    const int i = TIC_BID_X * 256 + TIC_TID;
    if (i < C * D) {
        const int c = i / D, d = i - c * D;
        float s = 0.f, sb = 0.f;
        for (int b = 0; b < B; ++b) {
            const float g = bfround(dl[b * C + c]);
            s += g * bf2f(z[(long)b * D + d]);
            sb += g;
        }
        dW[i] += s;
        if (d == 0) db[c] += sb;
    }
}

// one wave per row: loss_sum += -(sum_c t log_softmax)/B ; dlogits = gscale * (softmax * sum_c t - t) / B
// labels (int64, hard) XOR soft ([B,C] fp32) is non-null.
__global__ void __launch_bounds__(256) xent_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, const float* __restrict__ soft,
                                                    float* __restrict__ loss_sum, float* __restrict__ dlogits, int B, int C, float gscale) {
    const int l = lane_id(), w = wave_id();
    const int b = TIC_BID_X * 4 + w;
    if (b >= B) return;   // wave-uniform
    const float* z = logits + (long)b * C;
    float mx = -__builtin_huge_valf();
    for (int c = l; c < C; c += 64) mx = fmaxf(mx, z[c]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, shfl_xor(mx, m));
    float se = 0.f;
    for (int c = l; c < C; c += 64) se += __expf(z[c] - mx);
    se = wave_sum(se);
    const float lz = mx + __logf(se);
    float tl = 0.f, ts = 0.f;
    const int lab = labels ? (int)labels[b] : -1;
    for (int c = l; c < C; c += 64) {
        const float t = labels ? (c == lab ? 1.f : 0.f) : soft[(long)b * C + c];
        tl += t * (z[c] - lz);
        ts += t;
    }
    tl = wave_sum(tl);
    ts = wave_sum(ts);
    const float invB = 1.0f / (float)B;
    if (l == 0) atomic_addf(loss_sum, -tl * invB);
    if (dlogits)
        for (int c = l; c < C; c += 64) {
            const float t = labels ? (c == lab ? 1.f : 0.f) : soft[(long)b * C + c];
            dlogits[(long)b * C + c] = gscale * (__expf(z[c] - lz) * ts - t) * invB;
        }
}

// measurement only (tools/hbm_probe.py): streaming read of n4 float4 (mode bit 0: non-temporal loads), optional copy to dst
template <bool NT>
__global__ void __launch_bounds__(256) probe_stream_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ sink, long n4) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long i = (long)TIC_BID_X * 256 + TIC_TID; i < n4; i += (long)TIC_NBLK_X * 256) {
        const f32x4 v = ld_f4<NT>(src + i * 4);
        if (dst) st_f4<NT>(dst + i * 4, v);
        else acc += v;
    }
    if (!dst && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;   // keeps the loads alive
}
