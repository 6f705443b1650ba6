"""Counterpart of ``TIC/ResMoE/model.py`` (BASELINE config 5).

As the reference CODE does (its docstring says "ResNet + MLP experts", the code does not -- SURVEY 2): the backbone is
``nn.Identity``, the gate is a ViT-Base with ``num_experts`` outputs plus 0.01 N(0,1) noise in training, top-k softmax
(model.py:24-38), and every expert is a FULL ViT classifier evaluated densely on every sample, combined with the
scattered gate weights (model.py:50-58).  Gate and experts are ``TicViTForImageClassification`` modules (HIP forward /
backward); the gate (noise + top-k + softmax + scatter) and the combine (weighted sum over experts) are one HIP launch each,
forward and backward (``tic_moe_gate[_bwd]``, ``tic_moe_combine[_bwd]``, csrc/moe.h).

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


def _backend_of(module):
    return module._engine.backend


class _GateFn(torch.autograd.Function):
    """logits [B,E] (+ noise) -> (dense gate weights [B,E], top-k weights [B,K], top-k indices [B,K]) in one HIP launch"""

    @staticmethod
    def forward(ctx, logits, noise, noise_scale, top_k, backend):
        B, E = logits.shape
        logits = logits.contiguous()
        gate_w = torch.empty_like(logits)
        idx = torch.empty(B, top_k, dtype=torch.int64, device=logits.device)
        topw = torch.empty(B, top_k, dtype=torch.float32, device=logits.device)
        backend.call("tic_moe_gate", logits.data_ptr(), None if noise is None else noise.contiguous().data_ptr(), float(noise_scale),
                     gate_w.data_ptr(), idx.data_ptr(), topw.data_ptr(), B, E, top_k, backend.stream())
        ctx.backend = backend
        ctx.save_for_backward(gate_w)
        ctx.mark_non_differentiable(idx, topw)
        return gate_w, topw, idx

    @staticmethod
    def backward(ctx, d_gate_w, _dtopw, _didx):
        (gate_w,) = ctx.saved_tensors
        B, E = gate_w.shape
        dlogits = torch.empty_like(gate_w)
        ctx.backend.call("tic_moe_gate_bwd", gate_w.data_ptr(), d_gate_w.contiguous().data_ptr(), dlogits.data_ptr(), B, E, ctx.backend.stream())
        return dlogits, None, None, None, None


class _CombineFn(torch.autograd.Function):
    """expert-major logits X [E,B,C] and gate weights [B,E] -> mixture logits [B,C]"""

    @staticmethod
    def forward(ctx, X, gate_w, backend):
        E, B, C = X.shape
        X, gate_w = X.contiguous(), gate_w.contiguous()
        out = torch.empty(B, C, dtype=torch.float32, device=X.device)
        backend.call("tic_moe_combine", X.data_ptr(), gate_w.data_ptr(), out.data_ptr(), B, E, C, backend.stream())
        ctx.backend = backend
        ctx.save_for_backward(X, gate_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        X, gate_w = ctx.saved_tensors
        E, B, C = X.shape
        dX, dw = torch.empty_like(X), torch.empty_like(gate_w)
        ctx.backend.call("tic_moe_combine_bwd", X.data_ptr(), gate_w.data_ptr(), dout.contiguous().data_ptr(), dX.data_ptr(), dw.data_ptr(),
                         B, E, C, ctx.backend.stream())
        return dX, dw, None


class GatingNetwork(nn.Module):
    """ViT-Base router with ``num_experts`` outputs (TIC/ResMoE/model.py:24-38).  ``forward(x)`` keeps the reference's return
    value ``(softmax weights of the top-k experts [B,K], their indices [B,K])``; the dense scattered row the classifier needs
    comes out of the same kernel and is kept in ``last_gate_weights`` (autograd flows through it)."""

    NOISE = 0.01   # model.py:35: the literal, not ``random_t`` (which the reference stores and never reads)

    def __init__(self, num_experts, top_k, random_t, pretrained=True, backend=None, model_name='google/vit-base-patch16-224'):
        super().__init__()
        self.top_k, self.num_experts, self.random_t = top_k, num_experts, random_t
        self.vit = vit.ViT(num_classes=num_experts, pretrained=pretrained, model_name=model_name, backend=backend)
        self.last_gate_weights = None

    def route(self, x):
        """-> (dense gate weights [B,E], top-k weights, top-k indices)"""
        scores = self.vit(x).logits
        noise = torch.randn_like(scores) if self.training else None   # same RNG draw as the reference's randn_like
        return _GateFn.apply(scores, noise, self.NOISE, self.top_k, _backend_of(self.vit))

    def forward(self, x):
        self.last_gate_weights, top_w, top_i = self.route(x)
        return top_w, top_i


class MoEClassifier(nn.Module):
    """Every expert is evaluated on every sample and the K routed ones are mixed by their gate weights (model.py:40-58: dense
    evaluation, sparse weights).  Returns ``(mixture logits [B,C], gate weights [B,E], top-k indices [B,K])``."""

    def __init__(self, backbone, experts: nn.ModuleList, gate, top_k, num_classes):
        super().__init__()
        self.experts, self.top_k, self.num_classes = experts, top_k, num_classes
        self.shared_backbone, self.gate = backbone, gate

    def forward(self, x):
        feats = self.shared_backbone(x)
        gate_w, _, top_i = self.gate.route(x)
        per_expert = torch.stack([ex(feats).logits for ex in self.experts], dim=0)     # [E, B, C], expert-major
        return _CombineFn.apply(per_expert, gate_w, _backend_of(self.gate.vit)), gate_w, top_i


def make_ViTMoE(num_classes: int, num_experts: int, top_k: int, gateway_t: float, pretrained: bool = True,
                model_name: Optional[str] = None, gate_pretrained: bool = True, backend=None, gate_model_name='google/vit-base-patch16-224'):
    """model.py:60-72: identity backbone, ``num_experts`` full ViT classifiers, a ViT-Base gate"""
    experts = nn.ModuleList(vit.ViT(num_classes=num_classes, pretrained=pretrained, model_name=model_name, backend=backend)
                            for _ in range(num_experts))
    gate = GatingNetwork(num_experts=num_experts, top_k=top_k, random_t=gateway_t, pretrained=gate_pretrained, backend=backend,
                         model_name=gate_model_name)
    return MoEClassifier(backbone=nn.Identity(), experts=experts, gate=gate, top_k=top_k, num_classes=num_classes)


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
    """rank e holds its expert's logits for the GLOBAL batch [world*b, C]; afterwards rank s holds [E, b, C] for ITS rows."""

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
        return torch.stack(recv, dim=0)   # [E, b, C], expert-major

    @staticmethod
    def backward(ctx, g):
        group = ctx.group
        w = dist.get_world_size(group)
        send = [t.contiguous() for t in g.unbind(0)]        # E x [b, C]: gradient for expert e's logits of my rows
        if dist.get_backend(group) == "gloo":
            every = [torch.empty_like(g) for _ in range(w)]
            dist.all_gather(every, g.contiguous(), group=group)
            r = dist.get_rank(group)
            recv = [e[r] for e in every]
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
        if self.gate.num_experts != w:
            raise ValueError(f"expert parallelism places one expert per rank: {self.gate.num_experts} experts on {w} ranks")
        gate_w, _, top_i = self.gate.route(x)
        x_all = _AllGatherRows.apply(x, self.group) if x.requires_grad else torch.cat(self._gather_plain(x), 0)
        mine = self.expert(x_all).logits                                  # this rank's expert on the global batch
        per_expert = _AllToAllExpertLogits.apply(mine, self.group)        # [E, b, C] for my samples
        return _CombineFn.apply(per_expert, gate_w, _backend_of(self.gate.vit)), gate_w, top_i

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
