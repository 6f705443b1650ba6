"""CPU restatement of the mixture-of-experts gate / combine / loss arithmetic (TEST INFRASTRUCTURE ONLY).

Each function restates, with plain torch CPU ops, what the reference computes:

  gate(logits, noise, top_k)            TIC/ResMoE/model.py:33-38   noise 0.01 N(0,1) in training, top-k, softmax over the k values
  scatter(top_w, top_i, E)              TIC/ResMoE/model.py:53-54   dense [B,E] weight row
  combine(expert_out[B,E,C], gate_w)    TIC/ResMoE/model.py:56-57   bmm(gate_w[B,1,E], expert_out[B,E,C])
  symmetric_cross_entropy               TIC/ResMoE/train.py:21-25   0.1 CE + 1.0 RCE, RCE with log_softmax of the TARGETS
  load_balance_loss                     TIC/ResMoE/train.py:27-29
  total_loss                            TIC/ResMoE/train.py:31-36   classification + 0.5 balance

PARITY UNPINNED: ``TIC.ResMoE.model`` imports torchvision and ``TIC.ResMoE.train`` imports lightning -- neither is installed, and
the reference holds no test or fixture for this arithmetic.  The formulas are short enough to be restated line for
line; autograd differentiates them, as it does in the reference.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gate(logits: torch.Tensor, noise, top_k: int, noise_scale: float = 0.01):
    z = logits if noise is None else logits + noise * noise_scale
    top_v, top_i = torch.topk(z, k=top_k, dim=1)
    return torch.softmax(top_v, dim=1), top_i


def scatter(top_w: torch.Tensor, top_i: torch.Tensor, num_experts: int) -> torch.Tensor:
    dense = torch.zeros(top_w.shape[0], num_experts, dtype=top_w.dtype)
    return torch.scatter(dense, 1, top_i, top_w)


def combine(expert_out_bec: torch.Tensor, gate_w: torch.Tensor) -> torch.Tensor:
    return torch.bmm(gate_w.unsqueeze(1), expert_out_bec).squeeze(1)


def symmetric_cross_entropy(logits, targets, alpha=0.1, beta=1.0):
    ce = F.cross_entropy(logits, targets)
    rce = -(F.softmax(logits, dim=1) * F.log_softmax(targets, dim=1)).sum(dim=1).mean()
    return alpha * ce + beta * rce


def load_balance_loss(gate_w):
    usage = gate_w.mean(dim=0)
    return (gate_w @ usage.unsqueeze(1)).squeeze(1).mean()


def total_loss(logits, targets, gate_w, alpha=0.5):
    return symmetric_cross_entropy(logits, targets) + alpha * load_balance_loss(gate_w)
