#!/usr/bin/env python3
"""A/B inside one process: FusedAdamW through tic_vit_adamw (one pass: fp32 update + w16 + wT16) against the round-2 form (tic_adamw + the
cast-transpose launch of the next forward), on the full ViT-L step at the given per-GPU batch sizes.   python tools/ab_adamw.py 8 16 32 332"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd.ViT.model import ViT  # noqa: E402
from touhouimageclassification_amd.optim import FusedAdamW  # noqa: E402
from touhouimageclassification_amd.step import fused_train_step  # noqa: E402

dev = torch.device("cuda")
m = ViT(120, pretrained=False, model_name="google/vit-large-patch16-224").to(dev)
opt = FusedAdamW(m, lr=1e-5, weight_decay=0.01)
for B in [int(a) for a in sys.argv[1:]] or [8, 32, 332]:
    x = torch.randn(B, 3, 224, 224, device=dev)
    y = torch.randint(0, 120, (B,), device=dev)
    acc = {True: [], False: []}
    for _ in range(3):
        fused_train_step(m, opt, x, y, None)
    for rnd in range(4):
        for fused in (True, False):
            opt.one_pass = fused
            fused_train_step(m, opt, x, y, None)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 10 if B < 128 else 5
            for _ in range(n):
                fused_train_step(m, opt, x, y, None)
            torch.cuda.synchronize()
            acc[fused].append((time.perf_counter() - t) / n * 1e3)
    for fused in (False, True):
        a = acc[fused]
        print(f"B={B} one_pass={fused}: {sum(a) / len(a):8.3f} ms/step = {B / (sum(a) / len(a)) * 1e3:7.1f} img/s  ({', '.join(f'{t:.2f}' for t in a)})", flush=True)
