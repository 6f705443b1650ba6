"""MI355X-side counterpart of the reference's headline harness (``TIC/ViT/ntrain.py``).

  ViTLModule(num_classes, pretrained, model_name, lr, weight_decay, enable_mixup=True, full_finetune=True)   ntrain.py:16-66
      configure_optimizers / training_step / validation_step / test_step
  AugmentedDataset(train_path, test_path, batch_size, train_split, num_workers, image_size, enable_augmentation,
                   enable_diversity, enable_generalization, only_grey_augmentation)                               ntrain.py:68-157
      setup(stage) / train_dataloader / val_dataloader / test_dataloader
  train_main(PRETRAINED, MODEL_NAME, LR, WEIGHT_DECAY, FULL_FINETUNE, BATCH_SIZE, NUM_WORKERS, TRAIN_SPLIT, DATA_DIR,
             MAX_EPOCHS, ENABLE_MIX_UP, ENABLE_AUGMENTATION, TRAIN_ID, PATIENCE=3, ONLY_GREY_AUGMENTATION=False,
             ENABLE_DIVERSITY=True, ENABLE_GENERALIZATION=True)   + CLI --restore/-r --test/-t --transform/-tr       ntrain.py:159-248

What changed underneath:
  * Lightning is not installed in this image: ``Trainer`` below is a minimal fit/test loop with the same callbacks
    (top-3 by val_acc + every-3-epochs checkpoints, early stopping on val_acc with PATIENCE); real Lightning modules
    are not required.  bf16 mixed precision is what the HIP engine computes natively (``precision="bf16-mixed"``).
  * the per-sample CPU transforms (4 DataLoader workers, ntrain.py:150) become ONE HIP kernel per batch: the
    dataloaders deliver raw uint8 thumbnails and ``AugmentedDataset.on_device(batch, stage)`` runs crop / flip /
    colour-jitter / grayscale / erasing / normalise on the GPU; MixUp/CutMix are HIP kernels too.
  * under ``torchrun`` every rank trains its shard of each epoch and gradients are all-reduced per bucket (RCCL).
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from ..aug import CutMixOrMixUp, GpuAugment, preset_name
from ..dist import BucketedGradSync, UnpaddedShardSampler
from ..optim import FusedAdamW
from .model import TicViTForImageClassification, ViT

from ..utils.parameter import (CHECKPOINT_DIR, DATA_DIR, FILTERED_DATA_DIR, NUM_CLASSES, TEST_DIR,  # noqa: F401
                               UNFILTERED_DATA_DIR, VIT_IMAGE_SIZE)

STAGING_SIZE = 256   # dataset thumbnails are 256x256 (report section 3.1); raw uint8 batches are staged at this size


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class ViTLModule(nn.Module):
    def __init__(self, num_classes: int, pretrained: bool, model_name: str, lr: float, weight_decay: float,
                 enable_mixup: bool = True, full_finetune: bool = True, backend=None):
        super().__init__()
        self.num_classes = num_classes
        self.vit = ViT(num_classes, pretrained, model_name, backend=backend)
        self.lr = lr
        self.weight_decay = weight_decay
        self.cutmix_or_mixup = CutMixOrMixUp(num_classes, backend=backend)
        self.enable_mixup = enable_mixup
        if not full_finetune:
            for param in self.vit.base_model.parameters():
                param.requires_grad = False
        self.logged: Dict[str, float] = {}

    def log(self, name: str, value, prog_bar: bool = False) -> None:
        self.logged[name] = value

    def configure_optimizers(self):
        if isinstance(self.vit, TicViTForImageClassification):
            return FusedAdamW(self.vit, lr=self.lr, weight_decay=self.weight_decay)
        return torch.optim.AdamW(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)

    def forward(self, x):
        return self.vit(x)

    def training_step(self, batch, batch_idx):
        x, y = batch
        if self.enable_mixup:
            x, y = self.cutmix_or_mixup(x, y)
        logits = self.vit(x).logits
        loss = F.cross_entropy(logits, y)
        self.log('train_loss', loss, prog_bar=True)
        return loss

    def validation_step(self, batch, batch_idx):
        x, y = batch
        logits = self.vit(x).logits
        loss = F.cross_entropy(logits, y)
        self.log('val_loss', loss, prog_bar=True)
        acc = (logits.argmax(dim=1) == y).float().mean()
        self.log('val_acc', acc, prog_bar=True)
        return loss, acc

    def test_step(self, batch, batch_idx):
        x, y = batch
        acc = (self.vit(x).logits.argmax(dim=1) == y).float().mean()
        self.log('test_acc', acc, prog_bar=True)
        return acc


class ImageFolderU8(torch.utils.data.Dataset):
    """ImageFolder semantics (class = sub-directory, sorted) delivering raw uint8 HWC tensors at a fixed staging
    size; all augmentation happens later on the GPU."""

    EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")

    def __init__(self, root: str, staging: int = STAGING_SIZE):
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = [(os.path.join(dp, f), self.class_to_idx[c]) for c in self.classes
                        for dp, _, fs in sorted(os.walk(os.path.join(root, c))) for f in sorted(fs) if f.lower().endswith(self.EXT)]
        self.staging = staging

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        import numpy as np
        from PIL import Image
        path, label = self.samples[i]
        with Image.open(path) as im:
            im = im.convert("RGB")
            if im.size != (self.staging, self.staging):
                im = im.resize((self.staging, self.staging), Image.BILINEAR)
            arr = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
        return arr, label


class SyntheticU8(torch.utils.data.Dataset):
    """seeded random uint8 thumbnails (benchmarks / tests: there is no dataset offline)"""

    def __init__(self, n: int, num_classes: int, size: int = STAGING_SIZE, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8, generator=g)
        self.y = torch.randint(0, num_classes, (n,), generator=g)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], int(self.y[i])


class AugmentedDataset:
    def __init__(self, train_path: str = DATA_DIR, test_path: str = TEST_DIR, batch_size: int = 8, train_split: float = 0.8,
                 num_workers: int = 8, image_size=VIT_IMAGE_SIZE, enable_augmentation: bool = True, enable_diversity: bool = True,
                 enable_generalization: bool = True, only_grey_augmentation: bool = False, backend=None, dataset=None, test_dataset=None):
        self.train_path, self.test_path = train_path, test_path
        self.batch_size, self.image_size, self.train_split, self.num_workers = batch_size, image_size, train_split, num_workers
        self.enable_augmentation, self.enable_diversity = enable_augmentation, enable_diversity
        self.enable_generalization, self.only_grey_augmentation = enable_generalization, only_grey_augmentation
        self._backend = backend
        self._dataset_override, self._test_override = dataset, test_dataset
        self._samplers: List = []

    def setup(self, stage: str):
        size = self.image_size[0]
        if stage == 'fit':
            preset = preset_name(self.enable_augmentation, self.enable_diversity, self.enable_generalization, self.only_grey_augmentation)
            self.train_transform = GpuAugment(preset, size, backend=self._backend)
            self.eval_transform = GpuAugment("test", size, backend=self._backend)
            self.dataset = self._dataset_override or ImageFolderU8(self.train_path)
            train_size = int(len(self.dataset) * self.train_split)
            self.train_dataset, self.val_dataset = torch.utils.data.random_split(self.dataset, [train_size, len(self.dataset) - train_size])
        if stage == 'test':
            self.eval_transform = GpuAugment("test", size, backend=self._backend)
            self.test_dataset = self._test_override or ImageFolderU8(self.test_path)

    def _loader(self, ds, shuffle):
        """Under torch.distributed every rank iterates its own 1/world shard of `ds` (DistributedSampler: same permutation on
        every rank, seeded by the epoch the Trainer sets, rank-strided slices); the reference is single-process (ntrain.py:240)."""
        sampler = None
        if _world() > 1 and shuffle:   # training: equal shard lengths (every step has a collective), a new permutation per epoch
            sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=_world(), rank=_rank(), shuffle=True, seed=42)
            self._samplers.append(sampler)
        elif _world() > 1:            # validation / test: every sample exactly once over the ranks, no padding (metrics = the single-process ones)
            sampler = UnpaddedShardSampler(ds, _world(), _rank())
        return torch.utils.data.DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                           num_workers=self.num_workers, pin_memory=torch.cuda.is_available())

    def set_epoch(self, epoch: int) -> None:
        for s in self._samplers:
            s.set_epoch(epoch)

    def train_dataloader(self):
        self._samplers.clear()
        return self._loader(self.train_dataset, True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, False)

    def test_dataloader(self):
        return self._loader(self.test_dataset, False)

    def on_device(self, batch, stage: str, device):
        """raw uint8 batch -> (normalised fp32 [B,3,224,224], labels) on the GPU (the transform of ntrain.py:95-147)."""
        x, y = batch
        x = x.to(device, non_blocking=True)
        y = torch.as_tensor(y).to(device, non_blocking=True)
        tf = self.train_transform if stage == "train" else self.eval_transform
        return tf(x), y


class Trainer:
    """The slice of lightning.Trainer that train_main uses (ntrain.py:219-248)."""

    def __init__(self, max_epochs: int, checkpoint_dir: Optional[str] = None, train_id: str = "run", patience: int = 3,
                 save_top_k: int = 3, every_n_epochs: int = 3, device=None, precision: str = "bf16-mixed", log=print):
        if precision != "bf16-mixed":
            raise ValueError("the HIP engine computes in bf16 with fp32 master weights and accumulation (bf16-mixed)")
        self.max_epochs, self.dir, self.train_id = max_epochs, checkpoint_dir, train_id
        self.patience, self.save_top_k, self.every_n = patience, save_top_k, every_n_epochs
        self.device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        self.log = log
        self.history: List[Dict[str, float]] = []
        self._top: List = []       # (val_acc, path)
        self._periodic: List = []  # paths

    def _save(self, lmodel, opt, epoch, val_acc):
        if not self.dir or _rank() != 0:   # replicas are identical: rank 0 writes
            return None
        os.makedirs(self.dir, exist_ok=True)
        path = os.path.join(self.dir, f"checkpoint_{self.train_id}_epoch={epoch:02d}_val_acc={val_acc:.4f}.ckpt")
        torch.save({"state_dict": lmodel.state_dict(), "optimizer": opt.state_dict(), "epoch": epoch, "val_acc": val_acc}, path)
        return path

    def _evaluate(self, lmodel, loader, data, kind):
        """sample-weighted means over the WHOLE split: sums stay on the device (no per-batch host sync) and are all-reduced
        over the ranks' shards once at the end"""
        lmodel.eval()
        tot = torch.zeros(3, dtype=torch.float64, device=self.device)   # sum loss*n, sum acc*n, n
        with torch.no_grad():
            for i, batch in enumerate(loader):
                b = data.on_device(batch, "eval", self.device)
                nb = len(b[1])
                if kind == "val":
                    loss, acc = lmodel.validation_step(b, i)
                    tot[0] += loss.detach().double() * nb
                else:
                    acc = lmodel.test_step(b, i)
                tot[1] += acc.detach().double() * nb
                tot[2] += nb
        if _world() > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot_loss, tot_acc, n = (float(v) for v in tot.cpu())
        return (tot_loss / max(n, 1.0), tot_acc / max(n, 1.0))

    def fit(self, lmodel, datamodule, ckpt_path: Optional[str] = None):
        lmodel.to(self.device)
        datamodule.setup('fit')
        opt = lmodel.configure_optimizers()
        start = 0
        if ckpt_path:
            ck = torch.load(ckpt_path, map_location=self.device, weights_only=False)
            lmodel.load_state_dict(ck["state_dict"])
            if "optimizer" in ck:
                opt.load_state_dict(ck["optimizer"])
            start = int(ck.get("epoch", -1)) + 1
        sync = BucketedGradSync(lmodel.vit) if isinstance(lmodel.vit, TicViTForImageClassification) else None
        if sync:
            sync.broadcast_parameters()
        best, bad = -1.0, 0
        for epoch in range(start, self.max_epochs):
            lmodel.train()
            run = torch.zeros(2, dtype=torch.float64, device=self.device)   # sum of step losses, steps: read ONCE per epoch
            loader = datamodule.train_dataloader()
            if hasattr(datamodule, "set_epoch"):
                datamodule.set_epoch(epoch)
            for i, batch in enumerate(loader):
                opt.zero_grad()
                loss = lmodel.training_step(datamodule.on_device(batch, "train", self.device), i)
                (loss * (sync.grad_scale if sync else 1.0)).backward()
                if sync:
                    sync.wait()
                opt.step()
                run[0] += loss.detach().double()
                run[1] += 1
            if _world() > 1:
                dist.all_reduce(run, op=dist.ReduceOp.SUM)
            run_loss, nb = (float(v) for v in run.cpu())
            val_loss, val_acc = self._evaluate(lmodel, datamodule.val_dataloader(), datamodule, "val")
            rec = dict(epoch=epoch, train_loss=run_loss / max(nb, 1.0), val_loss=val_loss, val_acc=val_acc)
            self.history.append(rec)
            if _rank() == 0:
                self.log(f"epoch {epoch}: train_loss {rec['train_loss']:.4f} val_loss {val_loss:.4f} val_acc {val_acc:.4f}")
            # ModelCheckpoint(monitor='val_acc', mode='max', save_top_k=3)
            if self.dir and (len(self._top) < self.save_top_k or val_acc > min(a for a, _ in self._top)):
                self._top.append((val_acc, self._save(lmodel, opt, epoch, val_acc)))
                self._top.sort(key=lambda t: -t[0])
                for _, p in self._top[self.save_top_k:]:
                    if p and os.path.exists(p) and p not in self._periodic:
                        os.remove(p)
                self._top = self._top[:self.save_top_k]
            # ModelCheckpoint(monitor='epoch', every_n_epochs=3, save_top_k=3)
            if self.dir and (epoch + 1) % self.every_n == 0:
                self._periodic.append(self._save(lmodel, opt, epoch, val_acc))
                for p in self._periodic[:-self.save_top_k]:
                    if p and os.path.exists(p) and p not in [q for _, q in self._top]:
                        os.remove(p)
                self._periodic = self._periodic[-self.save_top_k:]
            # EarlyStopping(monitor='val_acc', mode='max', patience=PATIENCE)
            if val_acc > best:
                best, bad = val_acc, 0
            else:
                bad += 1
                if self.patience > 0 and bad >= self.patience:
                    self.log(f"early stop: val_acc has not improved for {self.patience} epochs")
                    break
        return self.history

    def test(self, lmodel, datamodule, ckpt_path: Optional[str] = None):
        lmodel.to(self.device)
        if ckpt_path:
            lmodel.load_state_dict(torch.load(ckpt_path, map_location=self.device, weights_only=False)["state_dict"])
        datamodule.setup('test')
        _, acc = self._evaluate(lmodel, datamodule.test_dataloader(), datamodule, "test")
        if _rank() == 0:
            self.log(f"test_acc {acc:.4f}")
        return acc


def seed_everything(seed: int) -> None:
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def train_main(PRETRAINED: bool, MODEL_NAME: str, LR: float, WEIGHT_DECAY: float, FULL_FINETUNE: bool, BATCH_SIZE: int,
               NUM_WORKERS: int, TRAIN_SPLIT: float, DATA_DIR: str, MAX_EPOCHS: int, ENABLE_MIX_UP: bool, ENABLE_AUGMENTATION: bool,
               TRAIN_ID: str, PATIENCE: int = 3, ONLY_GREY_AUGMENTATION: bool = False, ENABLE_DIVERSITY: bool = True,
               ENABLE_GENERALIZATION: bool = True, argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--restore', '-r', type=str, default=None, help='Path to the checkpoint to restore')
    parser.add_argument('--test', '-t', action='store_true', help='Only test model without training')
    parser.add_argument('--transform', '-tr', type=str, default=None, help='Transform the checkpoint')
    args = parser.parse_args(argv)
    seed_everything(42)
    mk = dict(num_classes=NUM_CLASSES, pretrained=PRETRAINED, model_name=MODEL_NAME, lr=LR, weight_decay=WEIGHT_DECAY,
              enable_mixup=ENABLE_MIX_UP, full_finetune=FULL_FINETUNE)
    if args.transform:
        if not args.restore:
            print("No checkpoint to transform")
            raise SystemExit(-1)
        lmodel = ViTLModule(**mk)
        lmodel.load_state_dict(torch.load(args.restore, map_location="cpu", weights_only=False)["state_dict"])
        torch.save(lmodel.vit.state_dict(), args.transform)   # bare HF-named state_dict (ntrain.py:188-194)
        raise SystemExit(0)
    lmodel = ViTLModule(**mk)
    data = AugmentedDataset(train_path=DATA_DIR, test_path=TEST_DIR, batch_size=BATCH_SIZE, train_split=TRAIN_SPLIT,
                            num_workers=NUM_WORKERS, image_size=VIT_IMAGE_SIZE, enable_augmentation=ENABLE_AUGMENTATION,
                            enable_diversity=ENABLE_DIVERSITY, enable_generalization=ENABLE_GENERALIZATION,
                            only_grey_augmentation=ONLY_GREY_AUGMENTATION)
    trainer = Trainer(max_epochs=MAX_EPOCHS, checkpoint_dir=os.path.join(CHECKPOINT_DIR, TRAIN_ID), train_id=TRAIN_ID,
                      patience=PATIENCE, precision="bf16-mixed")
    if not args.test:
        trainer.fit(lmodel, datamodule=data, ckpt_path=args.restore)
    trainer.test(lmodel, datamodule=data, ckpt_path=args.restore if args.test else None)
    return trainer


if __name__ == '__main__':
    from .presets import run_preset
    run_preset("nViT")   # filtered dataset + full augmentation: the reference's main preset (ntrain.py:250-267)
