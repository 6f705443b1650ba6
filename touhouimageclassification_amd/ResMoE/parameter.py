"""Defaults of the mixture-of-experts harness.

The reference keeps them as flat ``MOE_*`` module constants (TIC/ResMoE/parameter.py:1-19) that its trainer star-imports;
here they live in one frozen dataclass grouped by concern, and the ``MOE_*`` names are generated from it so that
``from ...ResMoE.parameter import *`` keeps working with the same names and values.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Optional


@dataclass(frozen=True)
class MoEDefaults:
    # experts / gate
    num_experts: int = 8
    top_k: int = 2
    gateway_t: float = 0.01                     # noise temperature of the gate
    pretrained: bool = True                     # experts start from local pretrained weights (never downloaded here)
    gate_pretrained: bool = True
    expert_model_name: str = "google/vit-base-patch16-224"
    # optimisation loop
    batch_size: int = 4
    accumulate_grad_batches: int = 4
    max_epochs: int = 10
    limit_train_batches_per_epoch: int = 500
    limit_val_batches_per_epoch: int = 100
    train_split: float = 0.8
    enable_amp: str = "bf16-mixed"              # Lightning precision string of the reference; bf16 is native here
    train_precision: str = "high"               # torch.set_float32_matmul_precision value of the reference
    profiler: Optional[str] = None
    # checkpoints / logs
    root_dir: str = "log/resmoe"
    checkpoint_min_k: int = 3
    checkpoint_last_k: int = 3
    checkpoint_every_n_epochs: int = 3


DEFAULTS = MoEDefaults()
globals().update({"MOE_" + k.upper(): v for k, v in asdict(DEFAULTS).items()})
__all__ = ["MoEDefaults", "DEFAULTS"] + ["MOE_" + k.upper() for k in asdict(DEFAULTS)]
