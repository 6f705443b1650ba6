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


def _uniform(g, shape, lo=0.0, hi=1.0) -> torch.Tensor:
    return torch.empty(shape).uniform_(lo, hi, generator=g)


def _first_valid(ok: torch.Tensor):
    """ok [B, T] bool -> (index of the first True per row (0 if none), any True per row): the 'try up to T times' loops of
    torchvision's get_params, for a whole batch at once"""
    return ok.to(torch.int8).argmax(dim=1), ok.any(dim=1)


def _resized_crop_boxes(B: int, H: int, W: int, g, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), tries: int = 10):
    """RandomResizedCrop.get_params for B images: up to 10 (area, log-uniform aspect) draws each, the first that fits wins, else
    the central crop clamped to the ratio range.  -> top, left, h, w as fp32 [B]"""
    area = float(H * W)
    target = area * _uniform(g, (B, tries), scale[0], scale[1])
    ar = torch.exp(_uniform(g, (B, tries), math.log(ratio[0]), math.log(ratio[1])))
    w, h = torch.round(torch.sqrt(target * ar)), torch.round(torch.sqrt(target / ar))
    first, found = _first_valid((w > 0) & (w <= W) & (h > 0) & (h <= H))
    rows = torch.arange(B)
    h, w = h[rows, first], w[rows, first]
    in_ratio = W / H   # fallback: whole image, clamped to the aspect range
    if in_ratio < ratio[0]:
        fw, fh = W, int(round(W / ratio[0]))
    elif in_ratio > ratio[1]:
        fh, fw = H, int(round(H * ratio[1]))
    else:
        fw, fh = W, H
    h, w = torch.where(found, h, torch.tensor(float(fh))), torch.where(found, w, torch.tensor(float(fw)))
    top = torch.minimum(torch.floor(_uniform(g, (B,)) * (H - h + 1)), H - h)      # randint(0, H - h + 1)
    left = torch.minimum(torch.floor(_uniform(g, (B,)) * (W - w + 1)), W - w)
    top = torch.where(found, top, torch.tensor(float((H - fh) // 2)))
    left = torch.where(found, left, torch.tensor(float((W - fw) // 2)))
    return top, left, h, w


def _erase_boxes(B: int, S: int, g, p: float, scale=(0.02, 0.33), ratio=(0.3, 3.3), tries: int = 10):
    """RandomErasing.get_params for B images -> (erase flag, top, left, h, w) fp32 [B]; an image whose 10 draws all miss keeps no box"""
    if p <= 0:
        z = torch.zeros(B)
        return z, z, z, z, z
    chosen = _uniform(g, (B,)) < p
    area = float(S * S)
    ea = area * _uniform(g, (B, tries), scale[0], scale[1])
    ar = torch.exp(_uniform(g, (B, tries), math.log(ratio[0]), math.log(ratio[1])))
    h, w = torch.round(torch.sqrt(ea * ar)), torch.round(torch.sqrt(ea / ar))
    first, found = _first_valid((h < S) & (w < S))
    rows = torch.arange(B)
    h, w = h[rows, first], w[rows, first]
    top = torch.minimum(torch.floor(_uniform(g, (B,)) * (S - h + 1)), S - h)
    left = torch.minimum(torch.floor(_uniform(g, (B,)) * (S - w + 1)), S - w)
    on = (chosen & found).float()
    return on, top * on, left * on, h * on, w * on


def sample_params(B: int, H: int, W: int, S: int, preset: str, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """[B, 20] fp32 parameter table (CPU) for `tic_augment`.  One batched draw per parameter -- a few dozen torch calls per BATCH and
    no host round trip per image (the per-image loop of round 1 made ~15 `.item()` calls per image: 5 000 per 332-image step)."""
    crop, flip, jitter, gray_p, erase_p = PRESETS[preset]
    g = generator
    P = torch.zeros(B, NPARAM)
    if crop:
        P[:, 0], P[:, 1], P[:, 2], P[:, 3] = _resized_crop_boxes(B, H, W, g)
    else:
        P[:, 2], P[:, 3] = float(H), float(W)
    if flip:
        P[:, 4] = (_uniform(g, (B,)) < 0.5).float()
    P[:, 5:9] = torch.arange(4, dtype=torch.float32)
    P[:, 9:12] = 1.0
    if jitter:
        P[:, 5:9] = torch.argsort(_uniform(g, (B, 4)), dim=1).float()      # a uniformly random order of the four colour ops
        P[:, 9:12] = _uniform(g, (B, 3), 0.8, 1.2)                          # brightness, contrast, saturation factors
        P[:, 12] = _uniform(g, (B,), -0.1, 0.1)                             # hue shift
        P[:, 13] = 1.0
    if gray_p > 0:
        P[:, 14] = (_uniform(g, (B,)) < gray_p).float()
    P[:, 15], P[:, 16], P[:, 17], P[:, 18], P[:, 19] = _erase_boxes(B, S, g, erase_p)
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
