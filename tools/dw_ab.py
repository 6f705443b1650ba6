#!/usr/bin/env python3
"""A/B of the grouped dW launch (the step's dominant kernel) inside ONE process on random data: alternates a tic_set_option knob
over interleaved rounds and prints ms / TFLOP/s per value.   python tools/dw_ab.py tn_mfma 16 32 [--batch 332]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("knob")
ap.add_argument("values", type=int, nargs="+")
ap.add_argument("--batch", type=int, default=332)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
M, D, F = args.batch * 197, 1024, 4096
shapes = [(D, F), (F, D), (D, D), (3 * D, D)]
dev = torch.device("cuda")
As = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n, k in shapes]
Bs = [torch.randn(M, k, device=dev).to(torch.bfloat16) for n, k in shapes]
Cs = [torch.zeros(n, k, device=dev) for n, k in shapes]
n = len(shapes)
PA = (ctypes.c_void_p * n)(*[a.data_ptr() for a in As])
PB = (ctypes.c_void_p * n)(*[b.data_ptr() for b in Bs])
PC = (ctypes.c_void_p * n)(*[c.data_ptr() for c in Cs])
NN = (ctypes.c_int * n)(*[s[0] for s in shapes])
KK = (ctypes.c_int * n)(*[s[1] for s in shapes])
flops = 2.0 * M * sum(a * b for a, b in shapes)


def run():
    call("tic_gemm_tn_group_bf16", n, PA, PB, PC, NN, KK, M, current_stream())


acc = {v: [] for v in args.values}
for v in args.values:
    call("tic_set_option", args.knob.encode(), v)
    run()
torch.cuda.synchronize()
for _ in range(args.rounds):
    for v in args.values:
        call("tic_set_option", args.knob.encode(), v)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        acc[v].append(e0.elapsed_time(e1) / args.reps)
for v in args.values:
    ms = sorted(acc[v])[len(acc[v]) // 2]
    print(f"{args.knob}={v}: median {ms:.4f} ms = {flops / ms / 1e9:.0f} TFLOP/s   ({', '.join(f'{t:.4f}' for t in acc[v])})  M={M}", flush=True)
