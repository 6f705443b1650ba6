"""MI355X-side counterpart of the reference's hand-rolled fine-tune harness (``TIC/ViT/finetune.py``).

Same entry points, argument meaning and stop/checkpoint rules, re-built around the HIP hot path:

  get_logger(name, log_dir)                                   finetune.py:21-52   file + stdout logger under log/
  train_step(model, data, optimizer, criterion, scaler, scheduler=None) -> float    :54-67
  validate_step(model, data, criterion) -> (float, int)       :69-77
  early_exit(timeline, max_tolerant_epoch, logger) -> bool    :79-91
  train_model(model, dataset, optimizer, scheduler, criterion, batch_size, num_epochs, max_tolerant_epoch,
              save_path, logger, skip_optimizer_load=False, scheduler_per_epoch=True)            :93-268
  get_linear_schedule_with_warmup(optimizer, warmup, total)   HF optimization.py:101-104 (used at finetune.py:324)

Differences that are deliberate: the model computes in bf16 with fp32 accumulation natively, so
the fp16 ``GradScaler`` of the reference is accepted for signature compatibility but never needed (pass ``None``
or a disabled scaler); ``validate_step`` really runs under ``no_grad`` (the reference's ``no_grad() and autocast``
expression never enters no_grad, finetune.py:71); the device is the model's device instead of a hard-coded "cuda".
"""
from __future__ import annotations

import logging
import math
import os
import sys
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, random_split


def get_logger(name: str, log_dir: str = "log") -> logging.Logger:
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    if logger.handlers:
        return logger
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    for handler in (logging.FileHandler(os.path.join(log_dir, f"{name}.log")), logging.StreamHandler(sys.stdout)):
        handler.setLevel(logging.INFO)
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    return logger


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, last_epoch: int = -1):
    """lr factor: step/warmup while warming up, then linear decay to 0 at num_training_steps."""
    def factor(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch)


def init_distributed_from_env() -> Optional[str]:
    """`torchrun --nproc-per-node N -m ...ViT.finetune` / `...ResNet.train` (one process per GPU): joins the RCCL group the launcher
    describes (WORLD_SIZE / RANK / LOCAL_RANK / MASTER_*) and returns this rank's device; None when not launched that way.  The
    reference is single-process (finetune.py:313 hard-codes "cuda")."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    return f"cuda:{local}" if torch.cuda.is_available() else "cpu"


def _device_of(model) -> torch.device:
    return next(model.parameters()).device


def _prepare(model, inputs):
    """uint8 thumbnails (get_dataset) are resized + normalised on the device by the transform train_model attached"""
    if inputs.dtype == torch.uint8:
        tf = getattr(model, "input_transform", None)
        if tf is None:
            raise TypeError("uint8 image batch but the model has no input_transform (use utils.preprocess.get_dataset + train_model)")
        return tf(inputs)
    return inputs


def _logits_of(outputs):
    return outputs.logits if hasattr(outputs, "logits") else outputs   # ViT returns an object, ResNet a tensor


def train_step(model, data, optimizer, criterion, scaler=None, scheduler=None) -> float:
    """One optimisation step; returns the loss as a Python float (this is the reference's per-step host sync).
    Data parallel (one process per GPU; `train_model` attaches `model._dp_sync`): the batch is this rank's shard, the loss
    gradient is scaled by 1 / world so that the per-bucket SUM all-reduce fired during backward yields the global-batch mean,
    and the optimizer waits for the buckets.  The returned loss is this rank's."""
    model.train()
    optimizer.zero_grad()
    dev = _device_of(model)
    sync = getattr(model, "_dp_sync", None)
    inputs, labels = (torch.as_tensor(t).to(dev, non_blocking=True) for t in data)
    loss = criterion(_logits_of(model(_prepare(model, inputs))), labels)
    back = loss if sync is None else loss * sync.grad_scale
    if scaler is not None and getattr(scaler, "is_enabled", lambda: False)():
        scaler.scale(back).backward()
        if sync is not None:
            sync.wait()
        scaler.step(optimizer)
        scaler.update()
    else:
        back.backward()
        if sync is not None:
            sync.wait()
        optimizer.step()
    if scheduler:
        scheduler.step()
    return loss.item()


def validate_step(model, data, criterion) -> Tuple[float, int]:
    model.eval()
    dev = _device_of(model)
    with torch.no_grad():
        inputs, labels = (torch.as_tensor(t).to(dev, non_blocking=True) for t in data)
        logits = _logits_of(model(_prepare(model, inputs)))
        loss = criterion(logits, labels)
        correct = (logits.argmax(dim=1) == labels).sum().item()
    return loss.item(), correct


def early_exit(timeline: Sequence[float], max_tolerant_epoch: int, logger) -> bool:
    """Stop when at least N epochs are recorded and none of the last N validation losses is below the loss
    N epochs earlier (the window start)."""
    if len(timeline) < max_tolerant_epoch:
        return False
    window = list(timeline[-(max_tolerant_epoch + 1):])
    anchor, rest = window[0], window[1:]
    if all(v >= anchor for v in rest):
        logger.info(f"Validation loss has not improved for {max_tolerant_epoch} epochs. Stopping training.")
        return True
    return False


def _find_resume_epoch(save_path: str, num_epochs: int) -> int:
    for e in range(num_epochs, 0, -1):
        if os.path.exists(save_path.format(epoch=e)):
            return e
    return 0


def _restore(model, optimizer, scheduler, ckpt, epoch: int, skip_optimizer_load: bool, scheduler_per_epoch: bool, logger) -> None:
    """Checkpoint = (model_sd, optim_sd[, sched_sd]) tuple, or a bare model state_dict (older files)."""
    if not (isinstance(ckpt, tuple) and len(ckpt) >= 2):
        model.load_state_dict(ckpt)
        logger.warning("Loaded checkpoint only contains model state_dict. Optimizer and scheduler state not loaded.")
        return
    model.load_state_dict(ckpt[0])
    sched_state = ckpt[2] if len(ckpt) > 2 else None
    if not skip_optimizer_load:
        optimizer.load_state_dict(ckpt[1])
        if scheduler and sched_state and scheduler_per_epoch:
            scheduler.load_state_dict(sched_state)
            logger.info("Loaded scheduler state.")
        elif scheduler and not scheduler_per_epoch:
            logger.warning("Resuming per-step scheduler state not fully implemented, may restart LR schedule.")
    elif scheduler and scheduler_per_epoch:
        logger.info(f"Skipping optimizer load, manually advancing scheduler to epoch {epoch}")
        for _ in range(epoch):
            scheduler.step()


def _nan_guard(loss: float, running: float, i: int, what: str, epoch: int, logger) -> float:
    if math.isnan(loss):
        logger.warning(f"NaN loss detected at {what} step {i} in epoch {epoch + 1}. Replacing with avg loss.")
        return running / (i + 1) if i > 0 else 0.0
    return loss


def train_model(model: torch.nn.Module, dataset, optimizer, scheduler, criterion, batch_size: int, num_epochs: int,
                max_tolerant_epoch: int, save_path: str, logger: logging.Logger, skip_optimizer_load: bool = False,
                scheduler_per_epoch: bool = True, num_workers: int = 8, val_fraction_denominator: int = 10, scaler=None) -> List[float]:
    """Epoch loop with resume-from-latest, 90/10 split (seed 0), per-epoch tuple checkpoints and the early-exit rule.
    Returns the validation-loss timeline.  ``scaler``: the reference builds a ``GradScaler`` here for its fp16 autocast
    (finetune.py:163); bf16 needs none, an enabled one is honoured by ``train_step``."""
    start_epoch = _find_resume_epoch(save_path, num_epochs)
    if start_epoch:
        logger.info(f"Resuming from epoch {start_epoch}")
        ckpt = torch.load(save_path.format(epoch=start_epoch), map_location=_device_of(model), weights_only=False)
        _restore(model, optimizer, scheduler, ckpt, start_epoch, skip_optimizer_load, scheduler_per_epoch, logger)
    else:
        logger.info("Starting training from scratch.")

    if getattr(dataset, "device_transform", None) is not None:
        model.input_transform = dataset.device_transform   # Resize + Normalize(dataset stats) as one HIP kernel per batch
    n_val = len(dataset) // val_fraction_denominator
    torch.manual_seed(0)   # split consistency across runs
    train_set, val_set = random_split(dataset, [len(dataset) - n_val, n_val])
    pin = _device_of(model).type == "cuda"
    # data parallel (absent from the reference: single process, finetune.py:313): `batch_size` is PER RANK; every rank trains on its
    # DistributedSampler shard (equal lengths: each step has a collective), validates on an unpadded shard (every sample once),
    # the epoch metrics are all-reduced, rank 0 writes the checkpoints and the log lines
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
    train_sampler = val_sampler = None
    if world > 1:
        from ..dist import BucketedGradSync, UnpaddedShardSampler
        if not hasattr(model, "register_bucket_hook"):
            raise TypeError("data-parallel train_model needs a TIC model (gradient buckets)")
        model._dp_sync = BucketedGradSync(model)
        model._dp_sync.broadcast_parameters()
        train_sampler = torch.utils.data.distributed.DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True, seed=0)
        val_sampler = UnpaddedShardSampler(val_set, world, rank)
    train_loader = DataLoader(train_set, batch_size=batch_size, shuffle=train_sampler is None, sampler=train_sampler, pin_memory=pin,
                              num_workers=num_workers)
    val_loader = DataLoader(val_set, batch_size=batch_size, shuffle=False, sampler=val_sampler, pin_memory=pin, num_workers=num_workers)

    def reduce(*vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=_device_of(model))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return tuple(t.tolist())

    def run_train(epoch: int) -> float:
        if scheduler and scheduler_per_epoch:
            logger.info(f"LR for epoch {epoch + 1}: {scheduler.get_last_lr()[0]:.6e}")
        if train_sampler is not None:
            train_sampler.set_epoch(epoch)
        running, steps = 0.0, 0
        for i, batch in enumerate(train_loader):
            loss = train_step(model, batch, optimizer, criterion, scaler, None if scheduler_per_epoch else scheduler)
            running += _nan_guard(loss, running, i, "training", epoch, logger)
            steps += 1
        running, steps = reduce(running, steps)
        return running / steps if steps else 0.0

    def run_val(epoch: int) -> Tuple[float, float]:
        optimizer.zero_grad(set_to_none=True)
        running, correct, total, steps = 0.0, 0, 0, 0
        for i, batch in enumerate(val_loader):
            loss, c = validate_step(model, batch, criterion)
            running += _nan_guard(loss, running, i, "validation", epoch, logger)
            correct += c
            total += len(batch[1])
            steps += 1
        running, correct, total, steps = reduce(running, correct, total, steps)
        return (running / steps if steps else 0.0), (100.0 * correct / total if total else 0.0)

    if start_epoch:
        logger.info(f"Validating model from loaded checkpoint (Epoch {start_epoch}) before resuming training...")
        vl, acc = run_val(start_epoch - 1)
        logger.info(f"Epoch [{start_epoch}], Validation Loss: {vl:.4f}, Accuracy: {acc:.2f}%")

    timeline: List[float] = []
    for epoch in range(start_epoch, num_epochs):
        tl = run_train(epoch)
        vl, acc = run_val(epoch)
        timeline.append(vl)
        state = (model.state_dict(), optimizer.state_dict())
        if scheduler and scheduler_per_epoch:
            state += (scheduler.state_dict(),)
        path = save_path.format(epoch=epoch + 1)
        if rank == 0:   # replicas are identical
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            torch.save(state, path)
            logger.info(f"Checkpoint saved to {path}")
        logger.info(f"Epoch [{epoch + 1}/{num_epochs}], Training Loss: {tl:.4f}, Validation Loss: {vl:.4f}, Accuracy: {acc:.2f}%")
        if early_exit(timeline, max_tolerant_epoch, logger):
            break
        if scheduler and scheduler_per_epoch:
            scheduler.step()
    return timeline


def main(data_dir: Optional[str] = None, num_epochs: int = 40, batch_size: int = 30, lr: float = 1e-5, weight_decay: float = 0.01,
         use_pretrained: bool = True, scheduler_type: str = 'linear', warmup_steps: int = 500,
         model_name: str = "google/vit-large-patch16-224-in21k", device: Optional[str] = None, dataset=None, num_workers: int = 8):
    """the reference's ``__main__`` block (finetune.py:270-342) with its hyper-parameters as defaults: AdamW + per-step linear
    warm-up schedule, early stopping disabled (tolerance = number of epochs)"""
    from ..optim import FusedAdamW
    from ..utils.parameter import CHECKPOINT_DIR, LOG_DIR, UNFILTERED_DATA_DIR, VIT_IMAGE_SIZE
    from .model import ViT
    os.makedirs(CHECKPOINT_DIR, exist_ok=True)
    save_path = os.path.join(CHECKPOINT_DIR, 'ViT_model_finetune_{epoch}.pth')
    logger = get_logger('ViT_finetune', LOG_DIR)
    logger.info("Starting ViT training script.")
    logger.info(f"Parameters: BATCH_SIZE={batch_size}, IMAGE_SIZE={VIT_IMAGE_SIZE}, NUM_EPOCHS={num_epochs}, LR={lr}, PRETRAINED={use_pretrained}")
    if dataset is None:
        from ..utils.preprocess import get_dataset
        data_dir = data_dir or UNFILTERED_DATA_DIR
        logger.info(f"Loading dataset from {data_dir}...")
        dataset = get_dataset(data_dir=data_dir, image_size=VIT_IMAGE_SIZE)
    num_classes = len(dataset.classes)
    logger.info(f"Dataset loaded. Number of classes: {num_classes}")
    model = ViT(num_classes=num_classes, pretrained=use_pretrained, model_name=model_name).to(device or init_distributed_from_env() or "cuda")
    optimizer = FusedAdamW(model, lr=lr, weight_decay=weight_decay)   # AdamW(model.parameters(), ...) semantics, one HIP kernel
    criterion = torch.nn.CrossEntropyLoss()
    num_training_steps = (len(dataset) - len(dataset) // 10) // batch_size * num_epochs
    scheduler, scheduler_per_epoch = None, True
    if scheduler_type == 'linear':
        scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=warmup_steps, num_training_steps=num_training_steps)
        scheduler_per_epoch = False
        logger.info(f"Using linear scheduler with {warmup_steps} warmup steps and {num_training_steps} total steps.")
    else:
        logger.info("Not using a learning rate scheduler.")
    logger.info("Starting model training...")
    timeline = train_model(model, dataset, optimizer, scheduler, criterion, num_epochs=num_epochs, batch_size=batch_size,
                           max_tolerant_epoch=num_epochs, save_path=save_path, logger=logger, skip_optimizer_load=False,
                           scheduler_per_epoch=scheduler_per_epoch, num_workers=num_workers)
    logger.info("Training finished.")
    return model, timeline


if __name__ == '__main__':
    main()
