#!/usr/bin/env python3
"""Does running two half-batches on two HIP streams (kernels of one hiding the HBM-bound phases of the other) beat one
full batch?  Two independent ViT-L replicas, one stream each, B images each, vs one replica with 2B."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd.ViT.model import ViT  # noqa: E402
from touhouimageclassification_amd.optim import FusedAdamW  # noqa: E402
from touhouimageclassification_amd.step import fused_train_step  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=166); ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
dev = torch.device("cuda")
def make(B):
    m = ViT(120, pretrained=False, model_name="google/vit-large-patch16-224").to(dev)
    return m, FusedAdamW(m, lr=1e-5, weight_decay=0.01), torch.randn(B, 3, 224, 224, device=dev), torch.randint(0, 120, (B,), device=dev)
def timed(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
B = a.batch
m0, o0, x0, y0 = make(B)
t1 = timed(lambda: fused_train_step(m0, o0, x0, y0, None), a.steps)
print(f"one stream,  B={B}: {t1*1e3:.1f} ms/step = {B/t1:.0f} img/s", flush=True)
m1, o1, x1, y1 = make(B)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s0): fused_train_step(m0, o0, x0, y0, None)
    with torch.cuda.stream(s1): fused_train_step(m1, o1, x1, y1, None)
torch.cuda.synchronize()
t2 = timed(both, a.steps)
print(f"two streams, 2 x B={B}: {t2*1e3:.1f} ms per pair = {2*B/t2:.0f} img/s", flush=True)
del m1, o1
mb, ob, xb, yb = make(2 * B)
t3 = timed(lambda: fused_train_step(mb, ob, xb, yb, None), a.steps)
print(f"one stream,  B={2*B}: {t3*1e3:.1f} ms/step = {2*B/t3:.0f} img/s", flush=True)
