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


def symmetric_cross_entropy(logits, targets, alpha=0.1, beta=1.0):
    ce = F.cross_entropy(logits, targets)
    rce = -torch.sum(F.softmax(logits, dim=1) * F.log_softmax(targets, dim=1), dim=1).mean()
    return alpha * ce + beta * rce


def load_balance_loss(gate_weights, top_k_indeces, num_experts):
    avg_expert_usages = torch.mean(gate_weights, dim=0)
    return torch.matmul(gate_weights, avg_expert_usages.unsqueeze(1)).squeeze(1).mean()


def total_loss(logits, targets, gate_weights, top_k_indeces, alpha=0.5):
    assert not torch.isnan(logits).any(), "Logits contains NaN"
    assert torch.isfinite(logits).all(), "Logits contains Inf"
    return symmetric_cross_entropy(logits, targets) + alpha * load_balance_loss(gate_weights, top_k_indeces, gate_weights.shape[1])


class ResMoETrainerModule(nn.Module):
    def __init__(self, model: MoEClassifier, optimizer):
        super().__init__()
        self.model, self.optimizer = model, optimizer
        self.logged = {}

    def log(self, name, value, **kw):
        self.logged[name] = value

    def _targets(self, y):
        return F.one_hot(y, num_classes=self.model.num_classes).float()

    def training_step(self, batch, batch_idx):
        x, y = batch
        logits, gate_weights, top_k_indeces = self.model(x)
        loss = total_loss(logits, self._targets(y), gate_weights, top_k_indeces)
        self.log("train_loss", loss)
        return loss

    def validation_step(self, batch, batch_idx):
        x, y = batch
        logits, gate_weights, top_k_indeces = self.model(x)
        self.log("val_balance_loss", load_balance_loss(gate_weights, top_k_indeces, gate_weights.shape[1]))
        self.log("val_classification_loss", symmetric_cross_entropy(logits, self._targets(y)))
        self.log("val_accuracy", (torch.argmax(logits, dim=1) == y).float().mean())

    def test_step(self, batch, batch_idx):
        x, y = batch
        logits, _, _ = self.model(x)
        self.log("test_classification_loss", symmetric_cross_entropy(logits, self._targets(y)))
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
