"""Pure-torch CPU restatement of the ViT fine-tune step (TEST INFRASTRUCTURE ONLY).

Follows, line by line, the arithmetic the reference reaches through
``TIC/ViT/model.py:27-45`` (HuggingFace ``ViTForImageClassification``):

  * patch embedding       HF models/vit/modeling_vit.py:42-69
  * CLS + position embed  HF models/vit/modeling_vit.py:129-161
  * attention (SDPA)      HF models/vit/modeling_vit.py:164-238
  * MLP (exact-erf GELU)  HF models/vit/modeling_vit.py:241-254, HF activations.py:83
  * layer (pre-LN, res.)  HF models/vit/modeling_vit.py:257-286
  * final LN + CLS + head HF models/vit/modeling_vit.py:385, 559-561
  * loss                  TIC/ViT/finetune.py:61,315 (hard CE); TIC/ViT/ntrain.py:48 (soft CE)
  * optimizer             TIC/ViT/ntrain.py:39-41 / finetune.py:314 (torch AdamW defaults,
                          weight decay on every parameter)

No transformers / torchvision / reference import: this file travels to the GPU box.
Parameter names are the transformers-5.x ``state_dict`` keys.

``emulate_autocast=True`` rounds to bf16 at exactly the points where
``torch.autocast(bfloat16)`` does on a GPU (GEMM inputs/outputs bf16, fp32
residual stream, fp32 LayerNorm/softmax/CE) -- used for tight GPU comparisons;
the default (False) is the reference's fp32 CPU path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class ViTSpec:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    mlp: int = 3072
    num_labels: int = 120
    image: int = 224
    patch: int = 16
    channels: int = 3
    eps: float = 1e-12          # HF configuration_vit.py:58
    init_range: float = 0.02    # HF configuration_vit.py:57

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


VIT_BASE = dict(hidden=768, layers=12, heads=12, mlp=3072)
VIT_LARGE = dict(hidden=1024, layers=24, heads=16, mlp=4096)
VIT_TINY = dict(hidden=128, layers=2, heads=2, mlp=512)   # test fixture size


def param_shapes(spec: ViTSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """Ordered (name, shape) list, transformers-5.x key names."""
    D, Fm, C = spec.hidden, spec.mlp, spec.num_labels
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("vit.embeddings.cls_token", (1, 1, D)),
        ("vit.embeddings.position_embeddings", (1, spec.tokens, D)),
        ("vit.embeddings.patch_embeddings.projection.weight", (D, spec.channels, spec.patch, spec.patch)),
        ("vit.embeddings.patch_embeddings.projection.bias", (D,)),
    ]
    for i in range(spec.layers):
        p = f"vit.layers.{i}."
        out += [
            (p + "attention.q_proj.weight", (D, D)), (p + "attention.q_proj.bias", (D,)),
            (p + "attention.k_proj.weight", (D, D)), (p + "attention.k_proj.bias", (D,)),
            (p + "attention.v_proj.weight", (D, D)), (p + "attention.v_proj.bias", (D,)),
            (p + "attention.o_proj.weight", (D, D)), (p + "attention.o_proj.bias", (D,)),
            (p + "layernorm_before.weight", (D,)), (p + "layernorm_before.bias", (D,)),
            (p + "layernorm_after.weight", (D,)), (p + "layernorm_after.bias", (D,)),
            (p + "mlp.fc1.weight", (Fm, D)), (p + "mlp.fc1.bias", (Fm,)),
            (p + "mlp.fc2.weight", (D, Fm)), (p + "mlp.fc2.bias", (D,)),
        ]
    out += [
        ("vit.layernorm.weight", (D,)), ("vit.layernorm.bias", (D,)),
        ("classifier.weight", (C, D)), ("classifier.bias", (C,)),
    ]
    return out


def init_params(spec: ViTSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded ``pretrained=False`` style init (TIC/ViT/model.py:41-45 -> HF _init_weights,
    modeling_vit.py:323-332): Linear/Conv weight ~ N(0, 0.02), biases 0, LN (1, 0),
    cls/pos ~ trunc_normal(0, 0.02).  Seeded per tensor in name order so any consumer
    regenerates identical weights."""
    g = torch.Generator().manual_seed(seed)
    params: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(spec):
        if name.endswith("cls_token") or name.endswith("position_embeddings"):
            t = torch.empty(shape).normal_(0.0, spec.init_range, generator=g).clamp_(-2 * spec.init_range, 2 * spec.init_range)
        elif "layernorm" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith("bias"):
            t = torch.zeros(shape)
        else:
            t = torch.empty(shape).normal_(0.0, spec.init_range, generator=g)
        params[name] = t
    return params


def randomize_small_params(params: Dict[str, torch.Tensor], seed: int = 1, scale: float = 0.05) -> Dict[str, torch.Tensor]:
    """Fine-tuned checkpoints have non-trivial biases / LN affine; tests use this so that
    bias and gamma/beta paths are exercised (zero biases hide indexing bugs)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in params.items():
        if "layernorm" in k and k.endswith("weight"):
            out[k] = v + torch.empty_like(v).normal_(0, scale, generator=g)
        elif k.endswith("bias"):
            out[k] = v + torch.empty_like(v).normal_(0, scale, generator=g)
        else:
            out[k] = v.clone()
    return out


def _r(t: torch.Tensor, on: bool) -> torch.Tensor:
    """bf16 round-trip (value stays fp32) when emulating autocast."""
    return t.to(torch.bfloat16).to(torch.float32) if on else t


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    # HF activations.py:83 ("gelu" -> exact erf form)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    # biased variance over the last dim, fp32 (HF modeling_vit.py:261-262)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def linear(x, w, b, ac: bool):
    y = _r(x, ac) @ _r(w, ac).t()
    if b is not None:
        y = y + _r(b, ac)
    return _r(y, ac)


def patchify(x: torch.Tensor, spec: ViTSpec) -> torch.Tensor:
    """[B,C,H,W] -> [B, grid*grid, C*P*P]; feature order (c, ky, kx) == Conv2d weight.flatten(1)."""
    B = x.shape[0]
    P, G, C = spec.patch, spec.grid, spec.channels
    x = x.reshape(B, C, G, P, G, P).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, G * G, C * P * P)


def forward_hidden(params: Dict[str, torch.Tensor], x: torch.Tensor, spec: ViTSpec,
                   emulate_autocast: bool = False, return_all: bool = False):
    ac = emulate_autocast
    if x.shape[1] != spec.channels:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the configuration.")
    if x.shape[2] != spec.image or x.shape[3] != spec.image:
        raise ValueError(f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model ({spec.image}*{spec.image}).")
    B, D, H, hd, N = x.shape[0], spec.hidden, spec.heads, spec.head_dim, spec.tokens
    P = params
    # patch-embed conv (k = s = 16) == GEMM over re-indexed patches
    wp = P["vit.embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
    pe = linear(patchify(x.float(), spec), wp, P["vit.embeddings.patch_embeddings.projection.bias"], ac)
    h = torch.cat([P["vit.embeddings.cls_token"].expand(B, -1, -1), pe], dim=1) + P["vit.embeddings.position_embeddings"]
    hs = [h]
    scale = hd ** -0.5
    for i in range(spec.layers):
        p = f"vit.layers.{i}."
        a = layer_norm(h, P[p + "layernorm_before.weight"], P[p + "layernorm_before.bias"], spec.eps)
        q = linear(a, P[p + "attention.q_proj.weight"], P[p + "attention.q_proj.bias"], ac)
        k = linear(a, P[p + "attention.k_proj.weight"], P[p + "attention.k_proj.bias"], ac)
        v = linear(a, P[p + "attention.v_proj.weight"], P[p + "attention.v_proj.bias"], ac)
        q = q.view(B, N, H, hd).transpose(1, 2)
        k = k.view(B, N, H, hd).transpose(1, 2)
        v = v.view(B, N, H, hd).transpose(1, 2)
        s = torch.softmax((q @ k.transpose(2, 3)) * scale, dim=-1)       # fp32 softmax
        o = _r(_r(s, ac) @ v, ac).transpose(1, 2).reshape(B, N, D)
        h = h + linear(o, P[p + "attention.o_proj.weight"], P[p + "attention.o_proj.bias"], ac)
        m = layer_norm(h, P[p + "layernorm_after.weight"], P[p + "layernorm_after.bias"], spec.eps)
        u = linear(m, P[p + "mlp.fc1.weight"], P[p + "mlp.fc1.bias"], ac)
        h = h + linear(_r(gelu_erf(u), ac), P[p + "mlp.fc2.weight"], P[p + "mlp.fc2.bias"], ac)
        hs.append(h)
    return (h, hs) if return_all else h


def forward(params: Dict[str, torch.Tensor], x: torch.Tensor, spec: ViTSpec,
            emulate_autocast: bool = False) -> torch.Tensor:
    """pixel_values [B,3,224,224] fp32 -> logits [B,C] fp32."""
    h = forward_hidden(params, x, spec, emulate_autocast)
    z = layer_norm(h, params["vit.layernorm.weight"], params["vit.layernorm.bias"], spec.eps)[:, 0, :]
    return linear(z, params["classifier.weight"], params["classifier.bias"], emulate_autocast)


def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """mean_b( -sum_c t[b,c] * log_softmax(z)[b,c] ); ``target`` int64 [B] (hard) or fp32 [B,C] (soft)."""
    logp = torch.log_softmax(logits.float(), dim=-1)
    if target.dtype in (torch.int64, torch.int32):
        return -logp.gather(1, target.long().unsqueeze(1)).squeeze(1).mean()
    return -(target.float() * logp).sum(-1).mean()


def loss_and_grads(params: Dict[str, torch.Tensor], x: torch.Tensor, target: torch.Tensor, spec: ViTSpec,
                   emulate_autocast: bool = False):
    """One fwd+bwd: returns (logits, loss, {name: grad}).  Autograd is the differentiator
    (the reference's backward *is* torch autograd, TIC/ViT/finetune.py:62)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits = forward(leaves, x, spec, emulate_autocast)
    loss = cross_entropy(logits, target)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return logits.detach(), loss.detach(), {k: g for k, g in zip(leaves.keys(), grads)}


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
               lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, wd: float = 0.01):
    """torch.optim.AdamW single-tensor update (decoupled decay on EVERY param, as
    TIC/ViT/ntrain.py:40 builds no param groups).  ``step`` is 1-based.  In-place on p,m,v."""
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p, m, v


def linear_warmup_lambda(step: int, warmup: int, total: int) -> float:
    """HF optimization.py:101-104 (get_linear_schedule_with_warmup), used at TIC/ViT/finetune.py:324."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def topk_indices(logits: torch.Tensor, k: int) -> torch.Tensor:
    return logits.topk(k, dim=-1).indices


def train_flops_per_image(spec: ViTSpec) -> float:
    """Algorithmic FLOPs: fwd = L(24 N D^2 + 4 N^2 D) + 2*196*768*D + 2 D C ; train = 3 fwd."""
    N, D, L, C = spec.tokens, spec.hidden, spec.layers, spec.num_labels
    fwd = L * (24 * N * D * D + 4 * N * N * D) + 2 * (spec.grid ** 2) * (spec.channels * spec.patch ** 2) * D + 2 * D * C
    return 3.0 * fwd
