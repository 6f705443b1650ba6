// moe.h -- gate / combine / loss arithmetic of the mixture-of-ViT-experts classifier (BASELINE config 5).
//
// The reference evaluates these on [B, E] and [B, E, C] tensors with a chain of small torch ops
// (TIC/ResMoE/model.py:33-38 gate, :53-57 combine; TIC/ResMoE/train.py:21-36 losses); here each is one launch,
// forward and backward, so a step's gate/combine/loss work is 6 launches instead of ~40 framework kernels.
// Shapes are small (B <= a few hundred, E <= 64, C ~ 120): one wave per sample row, no LDS, no atomics except the scalar loss.
//
//   moe_gate          z = logits + noise_scale * noise ; top-k by value (ties: lowest expert index) ; softmax over the k
//                     selected values ; scattered into a dense [B, E] weight row (zeros elsewhere)       model.py:33-38,53-54
//   moe_gate_bwd      d logits = w (.) (d w - <w, d w>)  on the selected experts, 0 elsewhere (softmax Jacobian; the noise is
//                     additive, the top-k selection piecewise constant)
//   moe_combine       out[b, :] = sum_e w[b, e] X[e, b, :]      ( bmm(gate_weights[B,1,E], expert_outputs[B,E,C]) )  model.py:56-57
//   moe_combine_bwd   d X[e, b, :] = w[b, e] d out[b, :] ;  d w[b, e] = <X[e, b, :], d out[b, :]>
//   moe_loss_rows     0.1 CE(z, t) + 1.0 RCE(z, t) with RCE = -mean_b sum_c softmax(z)_c log_softmax(t)_c  (the reference applies
//                     log_softmax to the one-hot TARGETS, train.py:23 -- kept as written) + d/dz                   train.py:21-25
//   moe_balance       alpha * mean_b <w[b, :], mean_b' w[b', :]>  =  alpha * sum_e avg_e^2 , d/dw[b, e] = 2 alpha avg_e / B   train.py:27-36
//
// Expert outputs are EXPERT-major, X[E, B, C]: every expert (or, expert-parallel, every peer rank) writes one contiguous slab.
#pragma once
#include "norm.h"

TIC_DEV float wave_maxf(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, shfl_xor(v, m));
    return v;
}
TIC_DEV float wave_minf(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, shfl_xor(v, m));
    return v;
}

#define TIC_MOE_MAX_K 8

// one wave per row, 4 rows per block.  gate_w [B,E] fp32, topk_idx [B,K] int64, topk_w [B,K] fp32.
__global__ void __launch_bounds__(256) moe_gate_kernel(const float* __restrict__ logits, const float* __restrict__ noise, float noise_scale,
                                                       float* __restrict__ gate_w, long long* __restrict__ topk_idx, float* __restrict__ topk_w,
                                                       int B, int E, int K) {
    const int l = lane_id(), b = TIC_BID_X * 4 + wave_id();
    if (b >= B) return;   // wave-uniform
    const float ninf = -__builtin_huge_valf();
    float v = ninf;
    if (l < E) v = logits[(long)b * E + l] + (noise ? noise_scale * noise[(long)b * E + l] : 0.f);
    float tv[TIC_MOE_MAX_K];
    int ti[TIC_MOE_MAX_K];
#pragma unroll
    for (int k = 0; k < TIC_MOE_MAX_K; ++k) {
        if (k < K) {
            const float mx = wave_maxf(v);
            const int idx = (int)wave_minf((v == mx && l < E) ? (float)l : 64.f);   // lowest index among equal values
            tv[k] = mx;
            ti[k] = idx;
            if (l == idx) v = ninf;
        } else {
            tv[k] = ninf;
            ti[k] = -1;
        }
    }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < TIC_MOE_MAX_K; ++k) den += (k < K) ? __expf(tv[k] - tv[0]) : 0.f;
    const float inv = 1.0f / den;
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < TIC_MOE_MAX_K; ++k) {
        const float w = (k < K) ? __expf(tv[k] - tv[0]) * inv : 0.f;
        if (l == ti[k]) mine = w;
        if (l == k && k < K) {
            topk_idx[(long)b * K + k] = ti[k];
            topk_w[(long)b * K + k] = w;
        }
    }
    if (l < E) gate_w[(long)b * E + l] = mine;
}

__global__ void __launch_bounds__(256) moe_gate_bwd_kernel(const float* __restrict__ gate_w, const float* __restrict__ d_gate_w,
                                                           float* __restrict__ dlogits, int B, int E) {
    const int l = lane_id(), b = TIC_BID_X * 4 + wave_id();
    if (b >= B) return;
    const float w = l < E ? gate_w[(long)b * E + l] : 0.f;
    const float d = l < E ? d_gate_w[(long)b * E + l] : 0.f;
    const float dot = wave_sum(w * d);
    if (l < E) dlogits[(long)b * E + l] = w * (d - dot);
}

// X [E,B,C], w [B,E] -> out [B,C]; one thread per (b, c)
__global__ void __launch_bounds__(256) moe_combine_kernel(const float* __restrict__ X, const float* __restrict__ w, float* __restrict__ out,
                                                          int B, int E, int C) {
    const long i = (long)TIC_BID_X * 256 + TIC_TID;
    if (i >= (long)B * C) return;
    const int b = (int)(i / C);
    float s = 0.f;
    for (int e = 0; e < E; ++e) s += w[(long)b * E + e] * X[(long)e * B * C + i];
    out[i] = s;
}

// one wave per (b, e): dX[e,b,:] = w[b,e] dout[b,:] ; dw[b,e] = <X[e,b,:], dout[b,:]>
__global__ void __launch_bounds__(256) moe_combine_bwd_kernel(const float* __restrict__ X, const float* __restrict__ w, const float* __restrict__ dout,
                                                              float* __restrict__ dX, float* __restrict__ dw, int B, int E, int C) {
    const int l = lane_id();
    const long r = (long)TIC_BID_X * 4 + wave_id();
    if (r >= (long)B * E) return;
    const int b = (int)(r / E), e = (int)(r - (long)b * E);
    const float we = w[r];
    const float* x = X + ((long)e * B + b) * C;
    float* dx = dX + ((long)e * B + b) * C;
    const float* d = dout + (long)b * C;
    float dot = 0.f;
    for (int c = l; c < C; c += 64) {
        const float dc = d[c];
        dot += x[c] * dc;
        dx[c] = we * dc;
    }
    dot = wave_sum(dot);
    if (l == 0) dw[r] = dot;
}

// one wave per row.  loss3[0] += (a CE_b + bt RCE_b) / B (total), loss3[1] += the same (classification part); dlogits may be null.
__global__ void __launch_bounds__(256) moe_loss_rows_kernel(const float* __restrict__ logits, const float* __restrict__ targets, float* __restrict__ loss3,
                                                            float* __restrict__ dlogits, int B, int C, float a_ce, float b_rce) {
    const int l = lane_id(), b = TIC_BID_X * 4 + wave_id();
    if (b >= B) return;
    const float* z = logits + (long)b * C;
    const float* t = targets + (long)b * C;
    float mz = -__builtin_huge_valf(), mt = mz;
    for (int c = l; c < C; c += 64) {
        mz = fmaxf(mz, z[c]);
        mt = fmaxf(mt, t[c]);
    }
    mz = wave_maxf(mz);
    mt = wave_maxf(mt);
    float sz = 0.f, st = 0.f;
    for (int c = l; c < C; c += 64) {
        sz += __expf(z[c] - mz);
        st += __expf(t[c] - mt);
    }
    const float lz = mz + __logf(wave_sum(sz)), lt = mt + __logf(wave_sum(st));
    float ce = 0.f, rce = 0.f, ts = 0.f;
    for (int c = l; c < C; c += 64) {
        const float p = __expf(z[c] - lz);
        ce -= t[c] * (z[c] - lz);
        rce -= p * (t[c] - lt);
        ts += t[c];
    }
    ce = wave_sum(ce);
    rce = wave_sum(rce);
    ts = wave_sum(ts);
    const float invB = 1.0f / (float)B;
    if (l == 0) {
        const float v = (a_ce * ce + b_rce * rce) * invB;
        atomic_addf(loss3, v);
        atomic_addf(loss3 + 1, v);
    }
    if (dlogits)
        for (int c = l; c < C; c += 64) {
            const float p = __expf(z[c] - lz);
            dlogits[(long)b * C + c] = (a_ce * (p * ts - t[c]) - b_rce * p * ((t[c] - lt) + rce)) * invB;
        }
}

// ONE block of 64 threads: thread e owns expert column e.  loss3[0] += alpha sum_e avg_e^2, loss3[2] += sum_e avg_e^2 (unweighted).
__global__ void __launch_bounds__(64) moe_balance_kernel(const float* __restrict__ gate_w, float* __restrict__ loss3, float* __restrict__ d_gate_w,
                                                         int B, int E, float alpha) {
    const int e = lane_id();
    float s = 0.f;
    if (e < E)
        for (int b = 0; b < B; ++b) s += gate_w[(long)b * E + e];
    const float avg = s / (float)B;
    const float bal = wave_sum(e < E ? avg * avg : 0.f);
    if (e == 0) {
        atomic_addf(loss3, alpha * bal);
        atomic_addf(loss3 + 2, bal);
    }
    if (d_gate_w && e < E) {
        const float g = 2.0f * alpha * avg / (float)B;
        for (int b = 0; b < B; ++b) d_gate_w[(long)b * E + e] = g;
    }
}
