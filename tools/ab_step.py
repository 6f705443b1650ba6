#!/usr/bin/env python3
"""A/B a tic_set_option knob on the full ViT-L step inside ONE process (boxes differ by +-1.5 %, so two bench.py runs cannot
resolve a 1 % change):  python tools/ab_step.py gemm_stagger 0 -1   -> alternates the two values, 4 x 5 steps each."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call  # noqa: E402
from touhouimageclassification_amd.ViT.model import ViT  # noqa: E402
from touhouimageclassification_amd.optim import FusedAdamW  # noqa: E402
from touhouimageclassification_amd.step import fused_train_step  # noqa: E402

name, va, vb = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
for kv in filter(None, os.environ.get("TIC_PRESET", "").split(",")):   # other knobs held fixed for the run: TIC_PRESET=a=1,b=2
    k, v = kv.split("=")
    call("tic_set_option", k.encode(), int(v))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 332
dev = torch.device("cuda")
MODEL = {"large": ("google/vit-large-patch16-224", 120), "base": ("google/vit-base-patch16-224", 10)}[os.environ.get("TIC_MODEL", "large")]   # TIC_MODEL=base: BASELINE config 2
m = ViT(MODEL[1], pretrained=False, model_name=MODEL[0]).to(dev)
opt = FusedAdamW(m, lr=1e-5, weight_decay=0.01)
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, MODEL[1], (B,), device=dev)
for _ in range(4): fused_train_step(m, opt, x, y, None)
acc = {va: [], vb: []}
for rnd in range(4):
    for v in (va, vb):
        call("tic_set_option", name, v)
        fused_train_step(m, opt, x, y, None)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): fused_train_step(m, opt, x, y, None)
        torch.cuda.synchronize(); acc[v].append((time.perf_counter() - t) / 5 * 1e3)
for v in (va, vb):
    print(f"{name.decode()}={v}: {sum(acc[v])/len(acc[v]):8.2f} ms/step  ({', '.join(f'{t:.1f}' for t in acc[v])})", flush=True)
