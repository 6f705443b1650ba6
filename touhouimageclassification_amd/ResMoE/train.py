"""Counterpart of ``TIC/ResMoE/train.py``: losses (train.py:21-36), the trainer module (train.py:38-77), checkpoint helpers
(train.py:79-99) and ``get_model`` (train.py:113-122).  The reference's ``train_epoch`` (train.py:101-111) is dead code that
calls ``backward()`` under ``no_grad`` and is not reproduced."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.parameter import CHECKPOINT_DIR, NUM_CLASSES
from .model import MoEClassifier, make_ViTMoE
from .parameter import *  # noqa: F401,F403


class _MoELossFn(torch.autograd.Function):
    """(logits [B,C], targets [B,C], gate weights [B,E] or None) -> loss3 = [total, classification part, balance part];
    the gradient of ``loss3[0]`` w.r.t. logits and gate weights comes out of the same launches (csrc/moe.h)."""

    @staticmethod
    def forward(ctx, logits, targets, gate_w, a_ce, b_rce, a_bal, backend):
        B, C = logits.shape
        logits, targets = logits.contiguous(), targets.to(torch.float32).contiguous()
        loss3 = torch.empty(3, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        E = 1 if gate_w is None else gate_w.shape[1]
        dgate = None if gate_w is None else torch.empty_like(gate_w)
        backend.call("tic_moe_loss", logits.data_ptr(), targets.data_ptr(), None if gate_w is None else gate_w.contiguous().data_ptr(),
                     loss3.data_ptr(), dlogits.data_ptr(), None if dgate is None else dgate.data_ptr(), B, C, E,
                     float(a_ce), float(b_rce), float(a_bal), backend.stream())
        ctx.save_for_backward(dlogits, dgate)
        return loss3

    @staticmethod
    def backward(ctx, g):
        dlogits, dgate = ctx.saved_tensors
        return g[0] * dlogits, None, (None if dgate is None else g[0] * dgate), None, None, None, None


def _loss_backend(t):
    """the losses are free functions in the reference: without an explicit backend they run on the HIP library, which needs
    the tensors on the GPU (there is no CPU path in the product package)"""
    from ..engine import _HipBackend
    be = _HipBackend()
    be.check_tensor(t)
    return be


def symmetric_cross_entropy(logits, targets, alpha=0.1, beta=1.0, backend=None):
    """alpha CE + beta RCE (train.py:21-25); RCE takes ``log_softmax`` of the TARGETS, as the reference writes it"""
    return _MoELossFn.apply(logits, targets, None, alpha, beta, 0.0, backend or _loss_backend(logits))[1]


def load_balance_loss(gate_weights, top_k_indeces, num_experts, backend=None):
    """mean_b <w_b, mean_b' w_b'> (train.py:27-29) = sum_e (mean_b w[b,e])^2"""
    avg = gate_weights.mean(dim=0)
    return (avg * avg).sum()


def total_loss(logits, targets, gate_weights, top_k_indeces, alpha=0.5, backend=None):
    """symmetric CE + alpha balance (train.py:31-36) and its gradients in two HIP launches.  The reference's NaN / Inf asserts
    on the logits are a host sync per step; the loss itself goes non-finite if they are, which the trainer's caller can test."""
    return _MoELossFn.apply(logits, targets, gate_weights, 0.1, 1.0, alpha, backend or _loss_backend(logits))[0]


class ResMoETrainerModule(nn.Module):
    def __init__(self, model: MoEClassifier, optimizer):
        super().__init__()
        self.model, self.optimizer = model, optimizer
        self.logged = {}
        self._backend = model.gate.vit._engine.backend

    def log(self, name, value, **kw):
        self.logged[name] = value

    def _targets(self, y):
        return F.one_hot(y, num_classes=self.model.num_classes).float()

    def training_step(self, batch, batch_idx):
        x, y = batch
        logits, gate_weights, top_k_indeces = self.model(x)
        loss = total_loss(logits, self._targets(y), gate_weights, top_k_indeces, backend=self._backend)
        self.log("train_loss", loss)
        return loss

    def validation_step(self, batch, batch_idx):
        x, y = batch
        logits, gate_weights, top_k_indeces = self.model(x)
        self.log("val_balance_loss", load_balance_loss(gate_weights, top_k_indeces, gate_weights.shape[1]))
        self.log("val_classification_loss", symmetric_cross_entropy(logits, self._targets(y), backend=self._backend))
        self.log("val_accuracy", (torch.argmax(logits, dim=1) == y).float().mean())

    def test_step(self, batch, batch_idx):
        x, y = batch
        logits, _, _ = self.model(x)
        self.log("test_classification_loss", symmetric_cross_entropy(logits, self._targets(y), backend=self._backend))
        self.log("test_accuracy", (torch.argmax(logits, dim=1) == y).float().mean())

    def configure_optimizers(self):
        return self.optimizer


def get_checkpoint_path(epoch: int) -> str:
    return os.path.join(CHECKPOINT_DIR, f"ResMoE_epoch{epoch}.pth")


def dump_checkpoint(model, optimizer, epoch: int, loss: float):
    os.makedirs(CHECKPOINT_DIR, exist_ok=True)
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch, "loss": loss}, get_checkpoint_path(epoch))


def load_checkpoint(model, optimizer, epoch: int) -> float:
    ck = torch.load(get_checkpoint_path(epoch), weights_only=False)
    model.load_state_dict(ck["model"])
    optimizer.load_state_dict(ck["optimizer"])
    return ck["loss"]


def get_model(backend=None):
    return make_ViTMoE(num_classes=NUM_CLASSES, num_experts=MOE_NUM_EXPERTS, top_k=MOE_TOP_K, pretrained=MOE_PRETRAINED,
                       model_name=MOE_EXPERT_MODEL_NAME, gate_pretrained=MOE_GATE_PRETRAINED, gateway_t=MOE_GATEWAY_T, backend=backend)
