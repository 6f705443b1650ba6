"""Per-operator Python entry points over the C ABI (used by tests and by callers that want one op).
Arguments are torch CUDA tensors; every call enqueues on torch's current stream."""
from __future__ import annotations

import torch

from ._lib import call, current_stream, require_gpu

EPI_BF16, EPI_GELU, EPI_RESID, EPI_DGELU, EPI_PATCH = range(5)


def _p(t):
    if t is None:
        return None
    require_gpu(t)
    if not t.is_contiguous():
        raise ValueError("TIC ops need contiguous tensors")
    return t.data_ptr()


def gemm_nt(A, B, epilogue=EPI_BF16, bias=None, resid=None, aux=None, rowtab=None, patches=0, out_rows=None):
    """C = A[M,K] . B[N,K]^T with a fused epilogue; returns the epilogue's output(s)."""
    M, K = A.shape
    N = B.shape[0]
    dev = A.device
    out = out2 = out_f32 = None
    if epilogue in (EPI_BF16, EPI_DGELU, EPI_GELU):
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epilogue == EPI_GELU:
        out2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epilogue == EPI_RESID:
        out_f32 = torch.empty(M, N, dtype=torch.float32, device=dev)
    if epilogue == EPI_PATCH:
        out_f32 = torch.zeros(out_rows, N, dtype=torch.float32, device=dev)
    call("tic_gemm_nt_bf16", _p(A), _p(B), M, N, K, epilogue, _p(bias), _p(out), _p(out2), _p(out_f32), _p(resid),
         _p(aux), _p(rowtab), patches, current_stream())
    if epilogue == EPI_GELU:
        return out, out2
    return out if out is not None else out_f32


def gemm_tn_accumulate(A, B, C):
    """C[N,K] += A[M,N]^T . B[M,K] (fp32)."""
    M, N = A.shape
    K = B.shape[1]
    call("tic_gemm_tn_bf16", _p(A), _p(B), _p(C), M, N, K, current_stream())
    return C


def layernorm_fwd(x, gamma, beta, eps=1e-12, rows=None, in_stride=None):
    D = gamma.numel()
    rows = x.numel() // D if rows is None else rows
    in_stride = D if in_stride is None else in_stride
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    call("tic_layernorm_fwd", _p(x), in_stride, _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, D, eps, current_stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, want_bf16=True):
    rows, D = dy.shape
    dx = torch.empty(rows, D, dtype=torch.float32, device=x.device)
    dxb = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    dg = torch.zeros(D, dtype=torch.float32, device=x.device)
    db = torch.zeros_like(dg)
    call("tic_layernorm_bwd", _p(dy), _p(x), D, _p(gamma), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dxb), _p(dg), _p(db), rows, D, current_stream())
    return dx, dxb, dg, db


def attention_fwd(qkv, B, H, N, scale=0.125):
    o = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B * H, N, dtype=torch.float32, device=qkv.device)
    call("tic_attention_fwd", _p(qkv), _p(o), _p(lse), B, H, N, scale, current_stream())
    return o, lse


def attention_bwd(qkv, o, lse, d_o, B, H, N, scale=0.125):
    dqkv = torch.empty_like(qkv)
    call("tic_attention_bwd", _p(qkv), _p(o), _p(lse), _p(d_o), _p(dqkv), B, H, N, scale, current_stream())
    return dqkv


def adamw_(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, w16=None):
    call("tic_adamw", _p(p), _p(g), _p(m), _p(v), _p(w16), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, current_stream())


def softmax_xent(logits, target, want_grad=True, gscale=1.0):
    """mean CE (hard int64 [B] or soft fp32 [B,C] targets) -> (loss scalar tensor, dlogits)."""
    B, C = logits.shape
    loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    hard = target.dtype in (torch.int64,)
    if not hard:
        target = target.float().contiguous()
    call("tic_softmax_xent", _p(logits), _p(target) if hard else None, None if hard else _p(target), _p(loss), _p(dl), B, C, gscale, current_stream())
    return loss[0], dl
