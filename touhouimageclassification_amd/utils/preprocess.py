"""Counterpart of ``TIC/utils/preprocess.py``: the ImageFolder front-end of the finetune / ResNet harnesses.

  get_dataset(data_dir, image_size)         preprocess.py:15-40   dataset + ``class_to_idx.pth`` side file
  get_class_to_idx(data_dir)                preprocess.py:42-46
  get_transforms(data_dir, image_size)      preprocess.py:48-77   Resize -> ToTensor -> Normalize(DATASET mean/std, cached
                                                                   in ``meta_mean_std.pth``)
  calculate_mean_std(data_dir, batch_size, image_size, num_workers)   preprocess.py:81-128

The reference resizes / normalises every sample on CPU DataLoader workers.  Here the dataset hands out raw uint8
thumbnails and the transform is ONE HIP kernel per batch (``tic_augment`` with the full-image box and every random op
off == Resize + ToTensor + Normalize); ``train_model`` / ``train_step`` apply it on the device when they see a uint8
batch.  Same on-disk side files, same statistics definition (mean over images of the per-batch channel mean / std,
accumulated in float64).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from ..ViT.ntrain import ImageFolderU8
from ..aug import GpuAugment
from .parameter import DATA_DIR, IMAGE_SIZE

META_MEAN_STD_FILENAME = "meta_mean_std.pth"
CLASS_TO_IDX_FILENAME = "class_to_idx.pth"


class DeviceTransform:
    """Resize((H, W)) -> ToTensor -> Normalize(mean, std) on the GPU for uint8 [B,h,w,3] batches."""

    def __init__(self, image_size, mean, std, backend=None):
        if image_size[0] != image_size[1]:
            raise ValueError("the HIP resize kernel produces square outputs (the reference only uses 224x224 / 256x256)")
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]
        self._aug = GpuAugment("none", image_size[0], mean=self.mean, std=self.std, backend=backend)

    def __call__(self, images_u8: torch.Tensor) -> torch.Tensor:
        return self._aug(images_u8)


def calculate_mean_std(data_dir, batch_size, image_size, num_workers=4, backend=None, device=None, dataset=None):
    """per-channel statistics of the resized [0,1] images: sum over batches of (batch mean, batch std) x batch size / N"""
    ds = dataset or ImageFolderU8(data_dir)
    device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
    loader = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=False, num_workers=num_workers)
    resize = GpuAugment("none", image_size[0], mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), backend=backend)   # Resize + ToTensor only
    mean = torch.zeros(3, dtype=torch.float64)
    std = torch.zeros(3, dtype=torch.float64)
    n = 0
    for images, _ in loader:
        x = resize(images.to(device)).double()          # [b, 3, H, W] in [0, 1]
        b = x.shape[0]
        flat = x.view(b, 3, -1)
        mean += flat.mean([0, 2]).cpu() * b
        std += flat.std([0, 2]).cpu() * b
        n += b
    mean /= n
    std /= n
    torch.save({'mean': mean, 'std': std}, os.path.join(data_dir, META_MEAN_STD_FILENAME))
    return mean, std


def get_transforms(data_dir: str = DATA_DIR, image_size=IMAGE_SIZE, backend=None) -> DeviceTransform:
    try:
        meta = torch.load(os.path.join(data_dir, META_MEAN_STD_FILENAME), weights_only=False)
        mean, std = meta['mean'], meta['std']
    except FileNotFoundError:
        print(f"{META_MEAN_STD_FILENAME} not found, calculating mean and std for the first time...")
        mean, std = calculate_mean_std(data_dir, 32, image_size, backend=backend)
    print(f"Normalizing with mean={mean} std={std}")
    return DeviceTransform(image_size, mean, std, backend=backend)


def get_dataset(data_dir: str = DATA_DIR, image_size=IMAGE_SIZE, backend=None):
    """ImageFolder-style dataset of raw uint8 thumbnails with ``.classes`` / ``.class_to_idx`` and the device-side
    transform attached as ``.device_transform`` (applied by train_step / validate_step)."""
    ds = ImageFolderU8(data_dir)
    ds.device_transform = get_transforms(data_dir, image_size, backend=backend)
    side = os.path.join(data_dir, CLASS_TO_IDX_FILENAME)
    if not os.path.exists(side):
        torch.save(ds.class_to_idx, side)
    return ds


def get_class_to_idx(data_dir: str = DATA_DIR):
    side = os.path.join(data_dir, CLASS_TO_IDX_FILENAME)
    if not os.path.exists(side):
        get_dataset(data_dir)
    return torch.load(side, weights_only=False)
