"""Counterpart of ``TIC/ResMoE/model.py`` (BASELINE config 5).

As the reference CODE does (its docstring says "ResNet + MLP experts", the code does not -- SURVEY 2): the backbone is
``nn.Identity``, the gate is a ViT-Base with ``num_experts`` outputs plus 0.01 N(0,1) noise in training, top-k softmax
(model.py:24-38), and every expert is a FULL ViT classifier evaluated densely on every sample, combined with the
scattered gate weights (model.py:50-58).  Gate and experts are ``TicViTForImageClassification`` modules (HIP forward /
backward); the gate / combine arithmetic on [B, E] and [B, E, C] tensors is a few torch ops (10^-5 of the FLOPs).

``ExpertParallelMoE`` is the expert-parallel form for E experts on E ranks (new: the reference is single-GPU): every rank
owns ONE expert and a replica of the gate; images are all-gathered, each rank runs its expert on the global batch, and an
all-to-all returns to every rank the [B_local, E, C] logits of its own samples (the reverse all-to-all carries their
gradients in backward).  Gate gradients are averaged over ranks; expert gradients stay local.  RCCL over xGMI via
``torch.distributed`` ("nccl"), gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..ViT import model as vit


class GatingNetwork(nn.Module):
    def __init__(self, num_experts, top_k, random_t, pretrained=True, backend=None, model_name='google/vit-base-patch16-224'):
        super().__init__()
        self.top_k, self.num_experts, self.random_t = top_k, num_experts, random_t
        self.vit = vit.ViT(num_classes=num_experts, pretrained=pretrained, model_name=model_name, backend=backend)

    def forward(self, x):
        logits = self.vit(x).logits
        if self.training:
            logits = logits + torch.randn_like(logits) * 0.01
        top_k_weights, top_k_indeces = torch.topk(logits, k=self.top_k, dim=1)
        return torch.softmax(top_k_weights, dim=1), top_k_indeces


def scatter_gate(top_k_weights, top_k_indeces, batch, num_experts):
    gate_weights = torch.zeros(batch, num_experts, device=top_k_weights.device, dtype=top_k_weights.dtype)
    return torch.scatter(gate_weights, 1, top_k_indeces, top_k_weights)


class MoEClassifier(nn.Module):
    def __init__(self, backbone, experts: nn.ModuleList, gate, top_k, num_classes):
        super().__init__()
        self.experts, self.top_k, self.num_classes = experts, top_k, num_classes
        self.shared_backbone, self.gate = backbone, gate

    def forward(self, x):
        features = self.shared_backbone(x)
        top_k_weights, top_k_indeces = self.gate(x)
        gate_weights = scatter_gate(top_k_weights, top_k_indeces, x.shape[0], len(self.experts))
        expert_outputs = torch.stack([expert(features).logits for expert in self.experts], dim=1)    # [B, E, C]
        combined_output = torch.bmm(gate_weights.unsqueeze(1), expert_outputs).squeeze(1)
        return combined_output, gate_weights, top_k_indeces


def make_ViTMoE(num_classes: int, num_experts: int, top_k: int, gateway_t: float, pretrained: bool = True,
                model_name: Optional[str] = None, gate_pretrained: bool = True, backend=None, gate_model_name='google/vit-base-patch16-224'):
    return MoEClassifier(
        backbone=nn.Identity(),
        experts=nn.ModuleList([vit.ViT(num_classes=num_classes, pretrained=pretrained, model_name=model_name, backend=backend)
                               for _ in range(num_experts)]),
        gate=GatingNetwork(num_experts=num_experts, top_k=top_k, random_t=gateway_t, pretrained=gate_pretrained, backend=backend,
                           model_name=gate_model_name),
        top_k=top_k, num_classes=num_classes)


class _AllGatherRows(torch.autograd.Function):
    """[b, ...] per rank -> [world*b, ...]; backward: each rank keeps the gradient slice of its own rows, summed over ranks."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group, ctx.b = group, x.shape[0]
        out = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
        dist.all_gather(out, x.contiguous(), group=group)
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        r = dist.get_rank(ctx.group)
        return g[r * ctx.b:(r + 1) * ctx.b], None


class _AllToAllExpertLogits(torch.autograd.Function):
    """rank e holds its expert's logits for the GLOBAL batch [world*b, C]; afterwards rank s holds [b, E, C] for ITS rows."""

    @staticmethod
    def forward(ctx, logits, group):
        ctx.group = group
        w = dist.get_world_size(group)
        b = logits.shape[0] // w
        send = list(logits.contiguous().view(w, b, -1).unbind(0))
        recv = [torch.empty_like(send[0]) for _ in range(w)]
        if dist.get_backend(group) == "gloo":   # gloo has no all_to_all: emulate with all_gather (tests)
            every = [torch.empty_like(logits) for _ in range(w)]
            dist.all_gather(every, logits.contiguous(), group=group)
            r = dist.get_rank(group)
            recv = [e.view(w, b, -1)[r] for e in every]
        else:
            dist.all_to_all(recv, [s.contiguous() for s in send], group=group)
        return torch.stack(recv, dim=1)   # [b, E, C]

    @staticmethod
    def backward(ctx, g):
        group = ctx.group
        w = dist.get_world_size(group)
        send = [t.contiguous() for t in g.unbind(1)]        # E x [b, C]: gradient for expert e's logits of my rows
        if dist.get_backend(group) == "gloo":
            every = [torch.empty_like(g) for _ in range(w)]
            dist.all_gather(every, g.contiguous(), group=group)
            r = dist.get_rank(group)
            recv = [e[:, r] for e in every]
        else:
            recv = [torch.empty_like(send[0]) for _ in range(w)]
            dist.all_to_all(recv, send, group=group)
        return torch.cat(recv, 0), None                      # [world*b, C] for my expert


class ExpertParallelMoE(nn.Module):
    """One expert per rank (world size == num_experts).  forward(x_local) -> (combined [b,C], gate_weights [b,E], indices)."""

    def __init__(self, expert: nn.Module, gate: GatingNetwork, num_classes: int, group=None):
        super().__init__()
        self.expert, self.gate, self.num_classes, self.group = expert, gate, num_classes, group

    def forward(self, x):
        w = dist.get_world_size(self.group)
        top_k_weights, top_k_indeces = self.gate(x)
        gate_weights = scatter_gate(top_k_weights, top_k_indeces, x.shape[0], w)
        x_all = _AllGatherRows.apply(x, self.group) if x.requires_grad else torch.cat(self._gather_plain(x), 0)
        mine = self.expert(x_all).logits                                  # this rank's expert on the global batch
        expert_outputs = _AllToAllExpertLogits.apply(mine, self.group)    # [b, E, C] for my samples
        return torch.bmm(gate_weights.unsqueeze(1), expert_outputs).squeeze(1), gate_weights, top_k_indeces

    def _gather_plain(self, x):
        out = [torch.empty_like(x) for _ in range(dist.get_world_size(self.group))]
        dist.all_gather(out, x.contiguous(), group=self.group)
        return out

    def sync_gate_gradients(self):
        """gate replicas see different samples: average their gradients (call before optimizer.step())"""
        w = dist.get_world_size(self.group)
        for p in self.gate.parameters():
            if p.grad is not None:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                p.grad.div_(w)
