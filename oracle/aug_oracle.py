"""CPU restatement of the ntrain.py augmentation pipeline (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED: the reference composes ``torchvision.transforms.v2`` ops (TIC/ViT/ntrain.py:95-136,30-33);
torchvision is not installed in this image and the reference holds no test / golden vector for them, so
this file restates their DOCUMENTED behaviour in float arithmetic with EXPLICIT parameters.
Known, deliberate deviation: torchvision's PIL path rounds to uint8 after every stage; this restatement (and
the HIP kernels) keep fp32 between stages.

Ops (full preset, ntrain.py:104-112): RandomResizedCrop(224) -> RandomHorizontalFlip -> ColorJitter(0.2,0.2,0.2,0.1)
-> RandomGrayscale(0.2) -> RandomErasing(0.5) -> ToTensor -> Normalize(ImageNet mean/std); val/test:
Resize((224,224)) -> ToTensor -> Normalize (ntrain.py:132-136,143-147); batch-level MixUp / CutMix (ntrain.py:30-33,45-46).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
GRAY_W = (0.2989, 0.587, 0.114)

# op ids used in the jitter order
OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE = 0, 1, 2, 3


@dataclass
class AugParams:
    """Explicit per-image parameters (what the random samplers produce)."""
    top: int
    left: int
    height: int
    width: int
    flip: bool = False
    order: Tuple[int, int, int, int] = (0, 1, 2, 3)      # permutation of the 4 jitter ops
    brightness: float = 1.0
    contrast: float = 1.0
    saturation: float = 1.0
    hue: float = 0.0
    jitter: bool = True
    gray: bool = False
    erase: Optional[Tuple[int, int, int, int]] = None     # (i, j, h, w) in output pixels


def _aa_weights(in_size: int, out_size: int, start: float, length: float):
    """Separable anti-aliased triangle filter (PIL / ATen _upsample_bilinear2d_aa): for output i, the
    source interval is [start + i*scale, start + (i+1)*scale), scale = length / out_size."""
    scale = length / out_size
    support = scale if scale >= 1.0 else 1.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    out = []
    for i in range(out_size):
        center = start + scale * (i + 0.5)
        xmin = max(0, int(center - support + 0.5))
        xmax = min(in_size, int(center + support + 0.5))
        ws = []
        for j in range(xmin, xmax):
            w = 1.0 - abs((j - center + 0.5) * invscale)
            ws.append(max(0.0, w))
        tot = sum(ws)
        out.append((xmin, [w / tot for w in ws]))
    return out


def resized_crop(img_u8: torch.Tensor, top: int, left: int, h: int, w: int, out: int = 224) -> torch.Tensor:
    """img_u8 [H,W,3] uint8 -> [3,out,out] float in [0,1]: crop box then antialiased bilinear resize.
    The filter never reads outside the crop box (the crop happens first, as in torchvision)."""
    H, W, _ = img_u8.shape
    src = img_u8[top:top + h, left:left + w].to(torch.float64) / 255.0   # [h,w,3]
    wy = _aa_weights(h, out, 0.0, float(h))
    wx = _aa_weights(w, out, 0.0, float(w))
    tmp = torch.zeros(out, w, 3, dtype=torch.float64)
    for i, (y0, ws) in enumerate(wy):
        for k, wk in enumerate(ws):
            tmp[i] += wk * src[y0 + k]
    res = torch.zeros(out, out, 3, dtype=torch.float64)
    for j, (x0, ws) in enumerate(wx):
        for k, wk in enumerate(ws):
            res[:, j] += wk * tmp[:, x0 + k]
    return res.permute(2, 0, 1).to(torch.float32)


def _gray(x: torch.Tensor) -> torch.Tensor:
    return GRAY_W[0] * x[0] + GRAY_W[1] * x[1] + GRAY_W[2] * x[2]


def _rgb_to_hsv(x):
    r, g, b = x[0], x[1], x[2]
    maxc = torch.max(x, dim=0).values
    minc = torch.min(x, dim=0).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return h, s, maxc


def _hsv_to_rgb(h, s, v):
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - f * s), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (1.0 - f) * s), 0.0, 1.0)
    sel = [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)]
    out = torch.zeros(3, *h.shape)
    for k, (rr, gg, bb) in enumerate(sel):
        m = i == k
        out[0] = torch.where(m, rr, out[0])
        out[1] = torch.where(m, gg, out[1])
        out[2] = torch.where(m, bb, out[2])
    return out


def color_jitter(x: torch.Tensor, p: AugParams) -> torch.Tensor:
    for op in p.order:
        if op == OP_BRIGHTNESS:
            x = torch.clamp(x * p.brightness, 0.0, 1.0)
        elif op == OP_CONTRAST:
            m = _gray(x).mean()
            x = torch.clamp(p.contrast * x + (1.0 - p.contrast) * m, 0.0, 1.0)
        elif op == OP_SATURATION:
            x = torch.clamp(p.saturation * x + (1.0 - p.saturation) * _gray(x).unsqueeze(0), 0.0, 1.0)
        else:
            h, s, v = _rgb_to_hsv(x)
            x = _hsv_to_rgb(torch.fmod(h + p.hue + 1.0, 1.0), s, v)
    return x


def augment_one(img_u8: torch.Tensor, p: AugParams, out: int = 224, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    x = resized_crop(img_u8, p.top, p.left, p.height, p.width, out)
    if p.flip:
        x = x.flip(-1)
    if p.jitter:
        x = color_jitter(x, p)
    if p.gray:
        x = _gray(x).unsqueeze(0).expand(3, -1, -1).clone()
    if p.erase is not None:
        i, j, eh, ew = p.erase
        x[:, i:i + eh, j:j + ew] = 0.0
    m = torch.tensor(mean).view(3, 1, 1)
    s = torch.tensor(std).view(3, 1, 1)
    return (x - m) / s


# ---- random parameter samplers (torchvision defaults) -------------------------------------------------
def sample_resized_crop(H: int, W: int, g: torch.Generator, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    area = H * W
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1], generator=g).item()
        ar = math.exp(torch.empty(1).uniform_(log_r[0], log_r[1], generator=g).item())
        w = int(round(math.sqrt(target * ar)))
        h = int(round(math.sqrt(target / ar)))
        if 0 < w <= W and 0 < h <= H:
            i = int(torch.randint(0, H - h + 1, (1,), generator=g).item())
            j = int(torch.randint(0, W - w + 1, (1,), generator=g).item())
            return i, j, h, w
    in_ratio = W / H
    if in_ratio < ratio[0]:
        w, h = W, int(round(W / ratio[0]))
    elif in_ratio > ratio[1]:
        h, w = H, int(round(H * ratio[1]))
    else:
        w, h = W, H
    return (H - h) // 2, (W - w) // 2, h, w


def sample_erasing(Hh: int, Ww: int, g: torch.Generator, p=0.5, scale=(0.02, 0.33), ratio=(0.3, 3.3)):
    if torch.rand(1, generator=g).item() >= p:
        return None
    area = Hh * Ww
    log_r = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        ea = area * torch.empty(1).uniform_(scale[0], scale[1], generator=g).item()
        ar = math.exp(torch.empty(1).uniform_(log_r[0], log_r[1], generator=g).item())
        h = int(round(math.sqrt(ea * ar)))
        w = int(round(math.sqrt(ea / ar)))
        if h < Hh and w < Ww:
            i = int(torch.randint(0, Hh - h + 1, (1,), generator=g).item())
            j = int(torch.randint(0, Ww - w + 1, (1,), generator=g).item())
            return i, j, h, w
    return None


# ---- batch-level MixUp / CutMix (torchvision v2, alpha = 1) ------------------------------------------------
def mixup(x: torch.Tensor, y: torch.Tensor, num_classes: int, lam: float):
    oh = torch.nn.functional.one_hot(y, num_classes).float()
    return lam * x + (1.0 - lam) * x.roll(1, 0), lam * oh + (1.0 - lam) * oh.roll(1, 0)


def cutmix_box(H: int, W: int, lam: float, rx: int, ry: int):
    r = 0.5 * math.sqrt(1.0 - lam)
    rw, rh = int(r * W), int(r * H)
    x1, y1 = max(rx - rw, 0), max(ry - rh, 0)
    x2, y2 = min(rx + rw, W), min(ry + rh, H)
    return x1, y1, x2, y2


def cutmix(x: torch.Tensor, y: torch.Tensor, num_classes: int, lam: float, rx: int, ry: int):
    H, W = x.shape[-2:]
    x1, y1, x2, y2 = cutmix_box(H, W, lam, rx, ry)
    out = x.clone()
    out[..., y1:y2, x1:x2] = x.roll(1, 0)[..., y1:y2, x1:x2]
    lam_adj = 1.0 - (x2 - x1) * (y2 - y1) / float(W * H)
    oh = torch.nn.functional.one_hot(y, num_classes).float()
    return out, lam_adj * oh + (1.0 - lam_adj) * oh.roll(1, 0)
