"""Fused fine-tune step: forward -> cross-entropy (+ its gradient) -> backward with per-bucket gradient
all-reduce -> AdamW, with ~30 library calls and no host synchronisation.  Same arithmetic as
``logits = model(x).logits; loss = F.cross_entropy(logits, y); loss.backward(); optimizer.step()``
(TIC/ViT/ntrain.py:43-50, finetune.py:54-67), which also works on a TIC model through autograd."""
from __future__ import annotations

from typing import Optional

import torch

from .dist import BucketedGradSync
from .optim import FusedAdamW


def fused_train_step(model, optimizer: FusedAdamW, x: torch.Tensor, target: torch.Tensor,
                     sync: Optional[BucketedGradSync] = None):
    """target: int64 [B] class ids or fp32 [B,C] soft labels (after MixUp/CutMix).  Returns (loss, logits)
    as device tensors (no .item(): the reference's per-step host sync, finetune.py:67, is the caller's choice)."""
    e = model._engine
    logits = e.forward(x)
    B, C = logits.shape
    loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    hard = target.dtype == torch.int64
    tgt = target.contiguous() if hard else target.to(torch.float32).contiguous()
    e.backend.call("tic_softmax_xent", logits.data_ptr(), tgt.data_ptr() if hard else None, None if hard else tgt.data_ptr(),
                   loss.data_ptr(), dlogits.data_ptr(), B, C, sync.grad_scale if sync else 1.0, e.backend.stream())
    loss = loss[0]
    e.zero_grads(B, keep_matrices=True)   # this step's backward is the only writer: it stores the weight-matrix gradients whole
    hook = None
    if model._bucket_hook is not None:
        user = model._bucket_hook
        hook = lambda name, a, b: user(name, e.grads[a:b])   # noqa: E731
    e.backward(dlogits, bucket_hook=hook)
    if sync:
        sync.wait()
    optimizer.step()
    return loss, logits
