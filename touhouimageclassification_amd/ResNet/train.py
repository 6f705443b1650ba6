"""Counterpart of ``TIC/ResNet/train.py``: the same hand-rolled loop as the ViT fine-tune harness (the reference's two
files are twins), SGD(lr 5e-2) + StepLR(5, 0.25) on a from-scratch ResNet (train.py:210-255).
``train_step`` / ``validate_step`` / ``early_exit`` / ``get_logger`` / ``train_model`` are shared with
``touhouimageclassification_amd.ViT.finetune`` -- they accept models that return a raw logits tensor (ResNet,
train.py:52) as well as objects with ``.logits`` (ViT)."""
from __future__ import annotations

import torch

from ..ViT.finetune import early_exit, get_logger, train_model, train_step, validate_step  # noqa: F401
from .model import resnet152


def build_reference_setup(num_classes: int, lr: float = 5e-2):
    """model / optimizer / scheduler / criterion exactly as train.py:239-242 builds them"""
    model = resnet152(num_classes=num_classes)
    optimizer = torch.optim.SGD(model.parameters(), lr=lr)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=5, gamma=0.25)
    return model, optimizer, scheduler, torch.nn.CrossEntropyLoss()
