#!/usr/bin/env python3
"""Every distinct convolution of a ResNet as the step runs it (forward, input gradient, weight gradient), one by one: microseconds,
TFLOP/s and GB/s of the bytes each MUST move (operands once + result once), i.e. how far each launch is from max(FLOP time, byte time).

  python tools/resnet_conv_table.py [--arch resnet50] [--batch 256] [--image 224]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd.ResNet import model as rm  # noqa: E402
from touhouimageclassification_amd._lib import call as _call  # noqa: E402
for _kv in filter(None, os.environ.get("TIC_PRESET", "").split(",")):   # knobs held for the run: TIC_PRESET=gemm_tile=128,...
    _k, _v = _kv.split("=")
    _call("tic_set_option", _k.encode(), int(_v))

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="resnet50")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--image", type=int, default=224)
a = ap.parse_args()
dev = torch.device("cuda")
m = getattr(rm, a.arch)(num_classes=120).to(dev).train()
m._refresh_packs(dev)
m._begin_backward(dev)
B = a.batch
flush = torch.empty(384 << 20, dtype=torch.uint8, device=dev)


def t_us(fn, n=5):
    fn()
    tot = 0.0
    for _ in range(n):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


seen = {}
H = W = (a.image + 6 - 7) // 2 + 1
H = W = (H + 2 - 3) // 2 + 1   # after the max-pool
rows = []
for blk in m._blocks():
    convs = [blk.conv1, blk.conv2] + ([blk.conv3] if blk.kind == "bottleneck" else [])
    if blk.downsample is not None:
        convs.append(blk.downsample[0])
    Hin, Win = H, W
    for cv in convs:
        Hc, Wc = (Hin, Win) if cv is (blk.downsample[0] if blk.downsample is not None else None) else (H, W)
        key = (cv.cin, cv.cout, cv.k, cv.stride, Hc)
        Ho = (Hc + 2 * cv.pad - cv.k) // cv.stride + 1
        if key not in seen:
            seen[key] = 0
            x = torch.randn(B * Hc * Wc, cv.cin, device=dev).to(torch.bfloat16)
            dy = torch.randn(B * Ho * Ho, cv.cout, device=dev).to(torch.bfloat16)
            fl = 2.0 * B * Ho * Ho * cv.cout * cv.cin * cv.k * cv.k
            by_f = 2.0 * (x.numel() + dy.numel() + cv.cout * cv.kp)
            col = {}

            def fwd():
                col["c"] = m._conv_fwd(cv, x, B, Hc, Wc)[1]
            tf = t_us(fwd)
            td = t_us(lambda: m._conv_bwd(cv, dy, col["c"], B, Hc, Wc, need_dx=True))   # wgrad + dgrad
            tw = t_us(lambda: m._conv_bwd(cv, dy, col["c"], B, Hc, Wc, need_dx=False))  # wgrad only
            rows.append((key, Ho, fl, by_f, tf, td - tw, tw))
        seen[key] += 1
        if cv is not (blk.downsample[0] if blk.downsample is not None else None):
            H = W = Ho
print(f"{a.arch} B={B} {a.image}px: cin cout k s Hin -> count |  fwd us (TF/s, GB/s) | dgrad us | wgrad us | floor us = max(flops / 1.2 PF, bytes / 5.5 TB/s)")
tot = [0.0, 0.0, 0.0, 0.0]
for key, Ho, fl, by, tf, td, tw in rows:
    n = seen[key]
    floor = max(fl / 1.2e15, by / 5.5e12) * 1e6
    print(f"  {key[0]:5d} {key[1]:5d} {key[2]} {key[3]} {key[4]:3d} x{n}: fwd {tf:7.1f} ({fl / tf / 1e6:6.0f} TF/s, {by / tf / 1e3:5.0f} GB/s)  dgrad {td:7.1f}  wgrad {tw:7.1f}  floor {floor:6.1f}", flush=True)
    for i, t in enumerate((tf, td, tw, floor)):
        tot[i] += n * t
print(f"sum over the network (x count): fwd {tot[0] / 1e3:.2f} ms  dgrad {tot[1] / 1e3:.2f} ms  wgrad {tot[2] / 1e3:.2f} ms   floor per pass {tot[3] / 1e3:.2f} ms")
