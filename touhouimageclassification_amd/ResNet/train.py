"""MI355X-side counterpart of the reference's ResNet harness (``TIC/ResNet/train.py``).

  get_logger(name, log_dir)                                        train.py:14-45
  train_step(model, data, optimizer, criterion, scaler) -> float   train.py:47-57
  validate_step(model, data, criterion) -> (float, int)            train.py:59-66
  early_exit(timeline, max_tolerant_epoch, logger) -> bool         train.py:68-80
  train_model(model, dataset, optimizer, scheduler, criterion, batch_size, num_epochs, max_tolerant_epoch, save_path,
              logger, skip_optimizer_load=False)                   train.py:82-208
  main()  /  python -m touhouimageclassification_amd.ResNet.train  train.py:210-255: ResNet-152 from scratch on the 256 x 256
                                                                   thumbnails, SGD(lr 5e-2), StepLR(5, 0.25), batch 80, 25 epochs

The model returns a raw ``[B, C]`` tensor (train.py:52).  The reference's two harness files are twins: the epoch loop,
resume rule, tuple checkpoints and the early-exit rule are the ones of ``ViT/finetune.py`` with a per-epoch scheduler, so this
module drives that loop with the ResNet signatures instead of keeping a second copy of it.  Differences that are deliberate:
fp16 autocast becomes the engine's native bf16 GEMM I/O with fp32 accumulation -- the ``scaler`` protocol is honoured when an
ENABLED ``GradScaler`` is passed (scale -> backward -> unscale / inf check -> step -> update), it is just never needed; the device
is the model's device, not a hard-coded "cuda".
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional

import torch

from ..utils.parameter import CHECKPOINT_DIR, IMAGE_SIZE, LOG_DIR, UNFILTERED_DATA_DIR
from ..ViT import finetune as _loop
from ..ViT.finetune import early_exit, get_logger  # noqa: F401  (same rule / same log format in both reference files)
from .model import resnet152


def train_step(model, data, optimizer, criterion, scaler=None) -> float:
    """one SGD step on a (images, labels) batch; returns the loss as a Python float (the reference's per-step host sync)"""
    return _loop.train_step(model, data, optimizer, criterion, scaler, None)


def validate_step(model, data, criterion):
    """(loss, number of correct top-1 predictions) of one batch, under no_grad"""
    return _loop.validate_step(model, data, criterion)


def train_model(model: torch.nn.Module, dataset, optimizer, scheduler, criterion, batch_size: int, num_epochs: int,
                max_tolerant_epoch: int, save_path: str, logger: logging.Logger, skip_optimizer_load: bool = False,
                num_workers: int = 8) -> List[float]:
    """resume-from-latest, 90/10 split (seed 0), per-epoch ``(model_sd, optim_sd, sched_sd)`` checkpoints, early exit; the
    scheduler steps once per epoch (train.py:206-207).  Returns the validation-loss timeline."""
    return _loop.train_model(model, dataset, optimizer, scheduler, criterion, batch_size=batch_size, num_epochs=num_epochs,
                             max_tolerant_epoch=max_tolerant_epoch, save_path=save_path, logger=logger,
                             skip_optimizer_load=skip_optimizer_load, scheduler_per_epoch=True, num_workers=num_workers)


def build_reference_setup(num_classes: int, lr: float = 5e-2, arch=resnet152):
    """model / optimizer / scheduler / criterion as train.py:239-242 builds them"""
    model = arch(num_classes=num_classes)
    optimizer = torch.optim.SGD(model.parameters(), lr=lr)
    scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=5, gamma=0.25)
    return model, optimizer, scheduler, torch.nn.CrossEntropyLoss()


def main(data_dir: str = UNFILTERED_DATA_DIR, batch_size: int = 80, num_epochs: int = 25, lr: float = 5e-2,
         max_tolerant_epoch: int = 3, device: Optional[str] = None, dataset=None, arch=resnet152, num_workers: int = 8):
    """the reference's ``__main__`` block (train.py:210-255) with its hyper-parameters as defaults"""
    os.makedirs(CHECKPOINT_DIR, exist_ok=True)
    save_path = os.path.join(CHECKPOINT_DIR, 'ResNet_model_{epoch}.pth')
    logger = get_logger('ResNet_train', LOG_DIR)
    logger.info("Starting training script.")
    logger.info(f"Parameters: BATCH_SIZE={batch_size}, IMAGE_SIZE={IMAGE_SIZE}, NUM_EPOCHS={num_epochs}, MAX_TOLERANT_EPOCH={max_tolerant_epoch}")
    if dataset is None:
        from ..utils.preprocess import get_dataset
        logger.info(f"Loading dataset from {data_dir}...")
        dataset = get_dataset(data_dir=data_dir, image_size=IMAGE_SIZE)
    num_classes = len(dataset.classes)
    logger.info(f"Dataset loaded. Number of classes: {num_classes}")
    model, optimizer, scheduler, criterion = build_reference_setup(num_classes, lr, arch)
    model.to(device or _loop.init_distributed_from_env() or "cuda")
    logger.info("Starting model training...")
    timeline = train_model(model, dataset, optimizer, scheduler, criterion, num_epochs=num_epochs, batch_size=batch_size,
                           max_tolerant_epoch=max_tolerant_epoch, save_path=save_path, logger=logger, skip_optimizer_load=True,
                           num_workers=num_workers)
    logger.info("Training finished.")
    return model, timeline


if __name__ == '__main__':
    main()
