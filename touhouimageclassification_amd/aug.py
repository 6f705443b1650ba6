"""On-GPU augmentation front-end: samples the random PARAMETERS with torchvision-v2's documented distributions
and runs the pixel work in libtic_hip.so (`tic_augment`, `tic_mix`, `tic_mix_labels`).

Presets mirror ``AugmentedDataset.setup`` (TIC/ViT/ntrain.py:93-148):
    "full"            RandomResizedCrop, HFlip, ColorJitter(.2,.2,.2,.1), RandomGrayscale(.2), RandomErasing(.5)   :104-112
    "grey"            Resize, RandomGrayscale(.2)                                                                   :97-102
    "diversity"       Resize, ColorJitter, RandomGrayscale(.2)                                                       :114-120
    "generalization"  RandomResizedCrop, HFlip, RandomErasing(.5)                                                    :122-128
    "none" / "test"   Resize                                                                                         :132-136,143-147
all followed by ToTensor + Normalize(ImageNet mean/std).  Inputs are raw uint8 HWC batches already on the GPU.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch

NPARAM = 20
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

PRESETS = {
    #                  crop   flip   jitter gray_p erase_p
    "full":           (True,  True,  True,  0.2,   0.5),
    "grey":           (False, False, False, 0.2,   0.0),
    "diversity":      (False, False, True,  0.2,   0.0),
    "generalization": (True,  True,  False, 0.0,   0.5),
    "none":           (False, False, False, 0.0,   0.0),
    "test":           (False, False, False, 0.0,   0.0),
}


def preset_name(enable_augmentation: bool, enable_diversity: bool, enable_generalization: bool, only_grey_augmentation: bool) -> str:
    """The branch structure of AugmentedDataset.setup('fit') (ntrain.py:95-136)."""
    if not enable_augmentation:
        return "none"
    if only_grey_augmentation:
        return "grey"
    if enable_diversity and enable_generalization:
        return "full"
    if enable_diversity:
        return "diversity"
    if enable_generalization:
        return "generalization"
    raise Exception("Must select diversity or generalization!")   # ntrain.py:130


def _resized_crop_box(H: int, W: int, g: torch.Generator, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    area = H * W
    lr = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1], generator=g).item()
        ar = math.exp(torch.empty(1).uniform_(lr[0], lr[1], generator=g).item())
        w, h = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < w <= W and 0 < h <= H:
            return (int(torch.randint(0, H - h + 1, (1,), generator=g).item()), int(torch.randint(0, W - w + 1, (1,), generator=g).item()), h, w)
    in_ratio = W / H
    if in_ratio < ratio[0]:
        w, h = W, int(round(W / ratio[0]))
    elif in_ratio > ratio[1]:
        h, w = H, int(round(H * ratio[1]))
    else:
        w, h = W, H
    return (H - h) // 2, (W - w) // 2, h, w


def _erase_box(S: int, g: torch.Generator, p: float, scale=(0.02, 0.33), ratio=(0.3, 3.3)):
    if p <= 0 or torch.rand(1, generator=g).item() >= p:
        return None
    area = S * S
    lr = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        ea = area * torch.empty(1).uniform_(scale[0], scale[1], generator=g).item()
        ar = math.exp(torch.empty(1).uniform_(lr[0], lr[1], generator=g).item())
        h, w = int(round(math.sqrt(ea * ar))), int(round(math.sqrt(ea / ar)))
        if h < S and w < S:
            return (int(torch.randint(0, S - h + 1, (1,), generator=g).item()), int(torch.randint(0, S - w + 1, (1,), generator=g).item()), h, w)
    return None


def sample_params(B: int, H: int, W: int, S: int, preset: str, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """[B, 20] fp32 parameter table (CPU) for `tic_augment`."""
    crop, flip, jitter, gray_p, erase_p = PRESETS[preset]
    g = generator or torch.default_generator
    P = torch.zeros(B, NPARAM)
    for b in range(B):
        top, left, h, w = _resized_crop_box(H, W, g) if crop else (0, 0, H, W)
        P[b, 0:4] = torch.tensor([top, left, h, w], dtype=torch.float32)
        P[b, 4] = float(flip and torch.rand(1, generator=g).item() < 0.5)
        P[b, 5:9] = torch.arange(4, dtype=torch.float32)
        P[b, 9:12] = 1.0
        if jitter:
            P[b, 5:9] = torch.randperm(4, generator=g).float()
            P[b, 9] = torch.empty(1).uniform_(0.8, 1.2, generator=g).item()
            P[b, 10] = torch.empty(1).uniform_(0.8, 1.2, generator=g).item()
            P[b, 11] = torch.empty(1).uniform_(0.8, 1.2, generator=g).item()
            P[b, 12] = torch.empty(1).uniform_(-0.1, 0.1, generator=g).item()
            P[b, 13] = 1.0
        P[b, 14] = float(gray_p > 0 and torch.rand(1, generator=g).item() < gray_p)
        box = _erase_box(S, g, erase_p)
        if box is not None:
            P[b, 15] = 1.0
            P[b, 16:20] = torch.tensor(box, dtype=torch.float32)
    return P


class GpuAugment:
    """uint8 [B,H,W,3] (device) -> normalised fp32 [B,3,S,S] (device)."""

    def __init__(self, preset: str = "full", size: int = 224, mean=IMAGENET_MEAN, std=IMAGENET_STD, backend=None, seed: Optional[int] = None):
        if preset not in PRESETS:
            raise ValueError(f"unknown augmentation preset {preset!r}")
        self.preset, self.size = preset, size
        self._mean = (ctypes.c_float * 3)(*mean)
        self._std = (ctypes.c_float * 3)(*std)
        self.generator = torch.Generator().manual_seed(seed) if seed is not None else None
        if backend is None:
            from .engine import _HipBackend
            backend = _HipBackend()
        self.backend = backend

    def __call__(self, images_u8: torch.Tensor, params: Optional[torch.Tensor] = None) -> torch.Tensor:
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
            raise ValueError("GpuAugment expects uint8 [B,H,W,3] images")
        self.backend.check_tensor(images_u8)
        B, H, W, _ = images_u8.shape
        if params is None:
            params = sample_params(B, H, W, self.size, self.preset, self.generator)
        params = params.to(images_u8.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, 3, self.size, self.size, dtype=torch.float32, device=images_u8.device)
        self.backend.call("tic_augment", images_u8.contiguous().data_ptr(), B, H, W, params.data_ptr(), out.data_ptr(), self.size,
                          self._mean, self._std, self.backend.stream())
        return out


class CutMixOrMixUp:
    """v2.RandomChoice([v2.CutMix(num_classes), v2.MixUp(num_classes)]) (ntrain.py:30-33): uniform choice, alpha = 1
    so lambda ~ Beta(1,1) = U(0,1).  Returns (mixed images, soft labels [B, num_classes])."""

    def __init__(self, num_classes: int, backend=None, seed: Optional[int] = None):
        self.num_classes = num_classes
        self.generator = torch.Generator().manual_seed(seed) if seed is not None else None
        if backend is None:
            from .engine import _HipBackend
            backend = _HipBackend()
        self.backend = backend

    def sample(self, H: int, W: int) -> Tuple[int, float, Tuple[int, int, int, int]]:
        g = self.generator or torch.default_generator
        mode = int(torch.randint(0, 2, (1,), generator=g).item())   # 0 = CutMix, 1 = MixUp (RandomChoice order)
        lam = float(torch.rand(1, generator=g).item())
        if mode == 1:
            return 0, lam, (0, 0, 0, 0)
        rx = int(torch.randint(0, W, (1,), generator=g).item())
        ry = int(torch.randint(0, H, (1,), generator=g).item())
        r = 0.5 * math.sqrt(1.0 - lam)
        rw, rh = int(r * W), int(r * H)
        x1, y1, x2, y2 = max(rx - rw, 0), max(ry - rh, 0), min(rx + rw, W), min(ry + rh, H)
        lam_adj = 1.0 - (x2 - x1) * (y2 - y1) / float(W * H)
        return 1, lam_adj, (x1, y1, x2, y2)

    def __call__(self, x: torch.Tensor, y: torch.Tensor, choice=None):
        self.backend.check_tensor(x)
        B, C, H, W = x.shape
        kmode, lam, (x1, y1, x2, y2) = choice if choice is not None else self.sample(H, W)
        x = x.contiguous()
        out = torch.empty_like(x)
        soft = torch.empty(B, self.num_classes, dtype=torch.float32, device=x.device)
        s = self.backend.stream()
        self.backend.call("tic_mix", x.data_ptr(), out.data_ptr(), B, C, H, W, kmode, lam, x1, y1, x2, y2, s)
        self.backend.call("tic_mix_labels", y.contiguous().data_ptr(), soft.data_ptr(), B, self.num_classes, lam, s)
        return out, soft
