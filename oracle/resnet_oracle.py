"""Pure-torch CPU restatement of the reference's ResNet training step (TEST INFRASTRUCTURE ONLY).

Table-driven functional form of ``TIC/ResNet/model.py`` (the reference's own file; parity PINNED against it by
tools/gen_golden.py -> tests/golden/resnet18.npz / resnet50.npz):

  stem      conv 7x7/2 p3 -> BN -> ReLU -> maxpool 3x3/2 p1                TIC/ResNet/model.py:148-152, 212-215
  stages    [64,128,256,512] planes x block counts, stride 2 from stage 2   :155-162, _make_layer :185-208
  BasicBlock  3x3(stride) BN ReLU 3x3 BN (+identity) ReLU                   :17-63   (resnet18/34)
  Bottleneck  1x1 BN ReLU 3x3(stride) BN ReLU 1x1(x4) BN (+identity) ReLU   :66-115  (resnet50/101/152; "v1.5": stride on the 3x3, :87)
  downsample  1x1(stride) + BN when the shape changes                        :193-197
  head      global avgpool -> flatten -> Linear                              :164-165, 222-224
  BN        train mode: batch statistics, eps 1e-5, momentum 0.1, running stats (unbiased var) + num_batches_tracked
  init      kaiming-normal(fan_out, relu) convs, BN (1, 0)                   :168-173

state keys are the reference's ``state_dict`` keys (conv1.weight, bn1.*, layerN.i.convK/bnK/downsample.{0,1}.*, fc.*).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

CFG = {
    "resnet18": ("basic", [2, 2, 2, 2]),
    "resnet34": ("basic", [3, 4, 6, 3]),
    "resnet50": ("bottleneck", [3, 4, 6, 3]),
    "resnet101": ("bottleneck", [3, 4, 23, 3]),
    "resnet152": ("bottleneck", [3, 8, 36, 3]),
}
EXPANSION = {"basic": 1, "bottleneck": 4}
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def block_plan(name: str):
    """[(prefix, kind, inplanes, planes, stride, has_downsample)] in forward order."""
    kind, counts = CFG[name]
    exp = EXPANSION[kind]
    plan, inplanes = [], 64
    for si, (planes, n) in enumerate(zip([64, 128, 256, 512], counts)):
        for bi in range(n):
            stride = 2 if (bi == 0 and si > 0) else 1
            ds = bi == 0 and (stride != 1 or inplanes != planes * exp)
            plan.append((f"layer{si + 1}.{bi}", kind, inplanes, planes, stride, ds))
            inplanes = planes * exp
    return plan, inplanes


def state_shapes(name: str, num_classes: int) -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
                    (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())])

    out.append(("conv1.weight", (64, 3, 7, 7)))
    bn("bn1", 64)
    plan, feat = block_plan(name)
    for p, kind, cin, planes, stride, ds in plan:
        if kind == "basic":
            out.append((p + ".conv1.weight", (planes, cin, 3, 3))); bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3))); bn(p + ".bn2", planes)
        else:
            out.append((p + ".conv1.weight", (planes, cin, 1, 1))); bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3))); bn(p + ".bn2", planes)
            out.append((p + ".conv3.weight", (planes * 4, planes, 1, 1))); bn(p + ".bn3", planes * 4)
        if ds:
            out.append((p + ".downsample.0.weight", (planes * EXPANSION[kind], cin, 1, 1)))
            bn(p + ".downsample.1", planes * EXPANSION[kind])
    out.append(("fc.weight", (num_classes, feat)))
    out.append(("fc.bias", (num_classes,)))
    return out


def init_state(name: str, num_classes: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for k, shp in state_shapes(name, num_classes):
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_mean"):
            st[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            st[k] = torch.ones(shp)
        elif len(shp) == 4:   # kaiming normal, fan_out, relu
            std = math.sqrt(2.0 / (shp[0] * shp[2] * shp[3]))
            st[k] = torch.empty(shp).normal_(0, std, generator=g)
        elif k == "fc.weight":
            bound = 1.0 / math.sqrt(shp[1])
            st[k] = torch.empty(shp).uniform_(-bound, bound, generator=g)
        elif k == "fc.bias":
            bound = 1.0 / math.sqrt(state_shapes(name, num_classes)[-2][1][1])
            st[k] = torch.empty(shp).uniform_(-bound, bound, generator=g)
        elif k.endswith(".weight"):
            st[k] = torch.ones(shp)
        else:
            st[k] = torch.zeros(shp)
    return st


# residual gains of the well-conditioned golden cases (tools/gen_golden.py, tests/resnet_checks.py): deeper nets need smaller ones
GOLDEN_RESIDUAL_GAIN = {"resnet18_b32": 0.25, "resnet50_b32": 0.25, "resnet152_b8": 0.1}


def damp_residual_gains(st: Dict[str, torch.Tensor], name: str, gain: float) -> None:
    """scale the weight of the LAST BatchNorm of every residual branch (bn3 of a Bottleneck, bn2 of a BasicBlock) in place:
    gain 0 is the reference's ``zero_init_residual`` option (TIC/ResNet/model.py:178-183), 1 its default init"""
    plan, _ = block_plan(name)
    for p, kind, *_ in plan:
        st[p + (".bn2.weight" if kind == "basic" else ".bn3.weight")].mul_(gain)


def is_param(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var") or key.endswith("num_batches_tracked"))


def _r(t: torch.Tensor, on: bool) -> torch.Tensor:
    """straight-through bf16 round trip (value stays fp32): emulates the HIP path's bf16 activation / weight storage"""
    if not on:
        return t
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _bn(x, st, prefix, after, train):
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    if train:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        after[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * st[prefix + ".running_mean"] + BN_MOMENTUM * mean.detach()
        after[prefix + ".running_var"] = (1 - BN_MOMENTUM) * st[prefix + ".running_var"] + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1))
        after[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
    else:
        mean, var = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def forward(st: Dict[str, torch.Tensor], x: torch.Tensor, name: str, train: bool = True, emulate_bf16: bool = False):
    """x [B,3,H,W] fp32 -> (logits [B,C], {buffer key: updated value}).  emulate_bf16 rounds weights, conv outputs
    and block activations to bf16 where the HIP path stores bf16 (fp32 accumulation and BN statistics, as there)."""
    e = emulate_bf16
    after: Dict[str, torch.Tensor] = {}

    def conv(t, key, stride=1, padding=0):
        return _r(F.conv2d(t, _r(st[key], e), None, stride=stride, padding=padding), e)

    h = conv(_r(x, e), "conv1.weight", 2, 3)
    h = _r(F.relu(_bn(h, st, "bn1", after, train)), e)
    h = F.max_pool2d(h, 3, 2, 1)
    plan, _ = block_plan(name)
    for p, kind, cin, planes, stride, ds in plan:
        identity = h
        if kind == "basic":
            o = _r(F.relu(_bn(conv(h, p + ".conv1.weight", stride, 1), st, p + ".bn1", after, train)), e)
            o = _bn(conv(o, p + ".conv2.weight", 1, 1), st, p + ".bn2", after, train)
        else:
            o = _r(F.relu(_bn(conv(h, p + ".conv1.weight"), st, p + ".bn1", after, train)), e)
            o = _r(F.relu(_bn(conv(o, p + ".conv2.weight", stride, 1), st, p + ".bn2", after, train)), e)
            o = _bn(conv(o, p + ".conv3.weight"), st, p + ".bn3", after, train)
        if ds:
            identity = _r(_bn(conv(h, p + ".downsample.0.weight", stride), st, p + ".downsample.1", after, train), e)
        h = _r(F.relu(o + identity), e)
    h = _r(h.mean(dim=(2, 3)), e)
    return _r(h @ _r(st["fc.weight"], e).t() + _r(st["fc.bias"], e), e), after


def loss_and_grads(st: Dict[str, torch.Tensor], x: torch.Tensor, y: torch.Tensor, name: str, emulate_bf16: bool = False):
    leaves = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v) for k, v in st.items()}
    logits, after = forward(leaves, x, name, train=True, emulate_bf16=emulate_bf16)
    loss = F.cross_entropy(logits, y)
    pk = [k for k in leaves if is_param(k)]
    grads = torch.autograd.grad(loss, [leaves[k] for k in pk])
    return logits.detach(), loss.detach(), dict(zip(pk, grads)), {k: v.detach() for k, v in after.items()}


def train_flops_per_image(name: str, image: int = 224, num_classes: int = 120) -> float:
    """3 x forward conv+fc FLOPs (8.174 GFLOP fwd conv for resnet50 @224)."""
    fl, hw = 0.0, image // 2
    fl += 2 * hw * hw * 64 * 3 * 49
    hw //= 2
    plan, feat = block_plan(name)
    for p, kind, cin, planes, stride, ds in plan:
        ohw = hw // stride
        if kind == "basic":
            fl += 2 * ohw * ohw * planes * cin * 9 + 2 * ohw * ohw * planes * planes * 9
        else:
            fl += 2 * hw * hw * planes * cin + 2 * ohw * ohw * planes * planes * 9 + 2 * ohw * ohw * planes * 4 * planes
        if ds:
            fl += 2 * ohw * ohw * planes * EXPANSION[kind] * cin
        hw = ohw
    fl += 2 * feat * num_classes
    return 3.0 * fl
