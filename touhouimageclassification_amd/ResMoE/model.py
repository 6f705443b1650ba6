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

from typing import List, Optional

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

    def __init__(self, backbone, experts: nn.ModuleList, gate, top_k, num_classes, sparse: bool = False, pad_rows: int = 8):
        super().__init__()
        self.experts, self.top_k, self.num_classes = experts, top_k, num_classes
        self.shared_backbone, self.gate = backbone, gate
        # sparse=True: every expert runs only on the samples routed to it (top_k / E of the dense work) -- the same mixture, because
        # the reference multiplies the other experts' logits by gate weights that are exact zeros (model.py:53-57)
        self.sparse, self.pad_rows = sparse, pad_rows

    def forward(self, x):
        feats = self.shared_backbone(x)
        gate_w, _, top_i = self.gate.route(x)
        if not self.sparse:
            per_expert = torch.stack([ex(feats).logits for ex in self.experts], dim=0)     # [E, B, C], expert-major
            return _CombineFn.apply(per_expert, gate_w, _backend_of(self.gate.vit)), gate_w, top_i
        outs, where_e, where_s = [], [], []
        for e, ex in enumerate(self.experts):
            rows = (top_i == e).any(dim=1).nonzero().flatten()          # samples routed to expert e (one host sync per expert)
            n = int(rows.numel())
            if n == 0:
                continue
            sub = feats.index_select(0, rows)
            n_pad = (n + self.pad_rows - 1) // self.pad_rows * self.pad_rows   # few distinct batch sizes for the engine's workspaces
            if n_pad != n:
                sub = torch.cat([sub, sub.new_zeros((n_pad - n,) + tuple(sub.shape[1:]))], 0)
            outs.append(ex(sub).logits[:n])
            where_e.append(torch.full_like(rows, e))
            where_s.append(rows)
        cat = torch.cat(outs, 0)
        per_expert = cat.new_zeros((len(self.experts), x.shape[0], cat.shape[1])).index_put((torch.cat(where_e), torch.cat(where_s)), cat)
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


# ---- sparse top-k dispatch ("or sparse ... 25 % of the dense FLOPs") ---------------------------------------------------
def _exchange_rows(send: List[torch.Tensor], group) -> List[torch.Tensor]:
    """all-to-all of row blocks with DIFFERENT row counts per (source, destination) pair: send[d] goes to rank d, the result's [s] came
    from rank s.  RCCL: one `all_to_all` (xGMI is point-to-point: every pair's rows cross their own link).  gloo (CPU tests) has no
    all-to-all: the counts are all-gathered and the payload travels as an all-gather of blocks padded to the largest count."""
    w, r = dist.get_world_size(group), dist.get_rank(group)
    dev = send[0].device
    counts = torch.tensor([t.shape[0] for t in send], dtype=torch.int64, device=dev)
    every = [torch.empty_like(counts) for _ in range(w)]
    dist.all_gather(every, counts, group=group)
    cm = torch.stack(every).cpu()                      # cm[s][d] = rows s sends to d  (one small host sync per exchange)
    tail = send[0].shape[1:]
    if dist.get_backend(group) != "gloo":
        recv = [torch.empty((int(cm[s][r]),) + tuple(tail), dtype=send[0].dtype, device=dev) for s in range(w)]
        dist.all_to_all(recv, [t.contiguous() for t in send], group=group)
        return recv
    mx = max(1, int(cm.max()))
    block = torch.zeros((w, mx) + tuple(tail), dtype=send[0].dtype, device=dev)
    for d in range(w):
        block[d, :send[d].shape[0]] = send[d]
    blocks = [torch.empty_like(block) for _ in range(w)]
    dist.all_gather(blocks, block, group=group)
    return [blocks[s][r, :int(cm[s][r])].clone() for s in range(w)]


class _ReturnExpertRows(torch.autograd.Function):
    """expert rank -> source ranks: the logits an expert computed for the rows it received travel back to where the rows came from;
    backward sends the gradient of those logits the opposite way.  `splits` = rows received from each source (forward's send sizes)."""

    @staticmethod
    def forward(ctx, logits, splits, group):
        ctx.group = group
        out = _exchange_rows(list(logits.split(splits, 0)), group)
        ctx.back_splits = [t.shape[0] for t in out]
        return torch.cat(out, 0)

    @staticmethod
    def backward(ctx, g):
        back = _exchange_rows(list(g.contiguous().split(ctx.back_splits, 0)), ctx.group)
        return torch.cat(back, 0), None, None


class SparseExpertParallelMoE(nn.Module):
    """One expert per rank, SPARSE dispatch: a sample's image travels only to the `top_k` experts its gate routed it to, each expert
    runs forward / backward on the rows it received (top_k / E of the dense work: 25 % for the reference's top-2 of 8), the K logit
    rows travel back and are mixed with the gate weights.  Same results as the reference's dense evaluation
    (`TIC/ResMoE/model.py:53-57` evaluates every expert on every sample and multiplies the non-routed ones by an exact 0): the
    non-routed entries of the [E, b, C] tensor the combine kernel reads are zeros here, their weights are zeros there.
    forward(x_local [b,3,H,W]) -> (mixture logits [b,C], gate weights [b,E], top-k indices [b,K]) as `MoEClassifier`."""

    def __init__(self, expert: nn.Module, gate: GatingNetwork, num_classes: int, group=None, pad_rows: int = 8):
        super().__init__()
        self.expert, self.gate, self.num_classes, self.group, self.pad_rows = expert, gate, num_classes, group, pad_rows
        self.last_rows = 0   # rows this rank's expert processed in the last forward (load-balance diagnostics)

    def forward(self, x):
        w = dist.get_world_size(self.group)
        if self.gate.num_experts != w:
            raise ValueError(f"expert parallelism places one expert per rank: {self.gate.num_experts} experts on {w} ranks")
        b = x.shape[0]
        gate_w, _, top_i = self.gate.route(x)                                  # [b,E] (zeros off the top-k), [b,K]
        K = top_i.shape[1]
        flat_e = top_i.reshape(-1)                                             # (sample s, slot k) -> expert
        order = torch.argsort(flat_e, stable=True)                             # routed pairs grouped by destination expert
        pair_sample = torch.div(order, K, rounding_mode="floor")
        counts = torch.bincount(flat_e, minlength=w).tolist()
        send = list(x.detach()[pair_sample].split(counts, 0))                  # images are data: no gradient travels back to them
        recv = _exchange_rows(send, self.group)
        splits = [t.shape[0] for t in recv]
        rows = torch.cat(recv, 0)
        n = rows.shape[0]
        self.last_rows = n
        n_pad = max(self.pad_rows, (n + self.pad_rows - 1) // self.pad_rows * self.pad_rows)   # few distinct batch sizes for the engine's workspaces
        if n_pad != n:
            rows = torch.cat([rows, rows.new_zeros((n_pad - n,) + tuple(rows.shape[1:]))], 0)
        mine = self.expert(rows).logits[:n]                                    # this rank's expert on the rows routed to it
        back = _ReturnExpertRows.apply(mine, splits, self.group)               # [b*K, C] in `order`
        per_expert = back.new_zeros((w, b, back.shape[1]))
        per_expert = per_expert.index_put((flat_e[order], pair_sample), back)  # [E, b, C], zeros where not routed
        return _CombineFn.apply(per_expert, gate_w, _backend_of(self.gate.vit)), gate_w, top_i

    sync_gate_gradients = ExpertParallelMoE.sync_gate_gradients
