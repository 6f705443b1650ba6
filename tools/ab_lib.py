#!/usr/bin/env python3
"""A/B two BUILDS of libtic_hip.so on the full ViT-L step, on one box: alternating child processes (each loads one library through
TIC_HIP_LIB), 3 rounds.   python tools/ab_lib.py touhouimageclassification_amd/libtic_hip_prev.so touhouimageclassification_amd/libtic_hip.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from touhouimageclassification_amd.ViT.model import ViT
from touhouimageclassification_amd.optim import FusedAdamW
from touhouimageclassification_amd.step import fused_train_step
B = 332
dev = torch.device("cuda")
m = ViT(120, pretrained=False, model_name="google/vit-large-patch16-224").to(dev)
opt = FusedAdamW(m, lr=1e-5, weight_decay=0.01)
x = torch.randn(B, 3, 224, 224, device=dev); y = torch.randint(0, 120, (B,), device=dev)
for _ in range(4): fused_train_step(m, opt, x, y, None)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): fused_train_step(m, opt, x, y, None)
torch.cuda.synchronize(); print("MS", (time.perf_counter() - t) / 10 * 1e3)
''' % ROOT
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, TIC_HIP_LIB=l), capture_output=True, text=True).stdout
        ms = [float(x.split()[1]) for x in out.splitlines() if x.startswith("MS")]
        res[l].append(ms[0] if ms else float("nan"))
for l in libs:
    print(f"{os.path.basename(l):28s} {sum(res[l])/len(res[l]):8.2f} ms/step  ({', '.join(f'{t:.1f}' for t in res[l])})", flush=True)
