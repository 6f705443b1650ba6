#!/usr/bin/env python3
"""BatchNorm backward in isolation: the y-masked form against the mask-from-x form (tic_batchnorm_bwd_relu), microseconds per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402
dev = torch.device("cuda")
only = sys.argv[1] if len(sys.argv) > 1 else ""   # "y" / "x": run one form only (per-kernel times under rocprofv3 --stats)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)   # > Infinity Cache


def t_us(fn, n=10):
    for _ in range(3):
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


for B, HW, C in ((256, 112 * 112, 64), (256, 56 * 56, 64), (256, 56 * 56, 256), (256, 28 * 28, 512), (256, 14 * 14, 1024), (256, 7 * 7, 2048)):
    M = B * HW
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    mean, rstd = x.float().mean(0), 1.0 / (x.float().var(0, unbiased=False) + 1e-5).sqrt()
    y = torch.relu((x.float() - mean) * rstd * gamma + beta).to(torch.bfloat16)
    scr = torch.empty(513 * 2 * C, device=dev)
    SB = scr.numel() * 4
    dx = torch.empty_like(x)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    a = b = float("nan")
    if only != "x":
        a = t_us(lambda: call("tic_batchnorm_bwd", dy.data_ptr(), y.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), scr.data_ptr(), SB,
                              dx.data_ptr(), None, 0, dg.data_ptr(), db.data_ptr(), M, C, current_stream()))
    if only != "y":
        b = t_us(lambda: call("tic_batchnorm_bwd_relu", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), scr.data_ptr(), SB,
                              dx.data_ptr(), dg.data_ptr(), db.data_ptr(), M, C, current_stream()))
    print(f"M={M:8d} C={C:5d}: y-masked {a:8.1f} us ({M * C * 16 / a / 1e6:5.2f} TB/s of 16 B/elem)   from-x {b:8.1f} us ({M * C * 12 / b / 1e6:5.2f} TB/s of 12 B/elem)", flush=True)
