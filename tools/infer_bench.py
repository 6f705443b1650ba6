#!/usr/bin/env python3
"""Forward-only (evaluation / serving) latency and throughput of the HIP ViT on one MI355X: the path behind
`validate_step`, `test_step` and `utils.serve.full_judge`.   python tools/infer_bench.py [--model large] [--batches 1 8 64 256]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd.ViT.model import ViT  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="large", choices=["large", "base"])
ap.add_argument("--classes", type=int, default=120)
ap.add_argument("--batches", type=int, nargs="+", default=[1, 8, 64, 256])
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
m = ViT(a.classes, pretrained=False, model_name=f"google/vit-{a.model}-patch16-224").to(dev).eval()
L, D = (24, 1024) if a.model == "large" else (12, 768)
fwd_flops = L * (24 * 197 * D * D + 4 * 197 * 197 * D) + 2 * 196 * 768 * D + 2 * D * a.classes
rows = []
for B in a.batches:
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            logits = m(x).logits
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    rows.append(dict(batch=B, latency_ms=round(dt * 1e3, 3), images_per_sec=round(B / dt, 1), tflops=round(B / dt * fwd_flops / 1e12, 1)))
print(json.dumps({"metric": f"forward-only ViT-{a.model}/16 224px bf16 C={a.classes}", "rows": rows}))
