#!/usr/bin/env python3
"""Run ONE GEMM configuration repeatedly (for rocprofv3 --pmc / --kernel-trace passes).
  python tools/gemm_one.py --kind nt --M 12608 --N 1024 --K 4096 --epi 2 --tile 256 --reps 20"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="nt")
ap.add_argument("--M", type=int, default=12608)
ap.add_argument("--N", type=int, default=1024)
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--epi", type=int, default=0)
ap.add_argument("--tile", type=int, default=256)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--streamk", type=int, default=1)
ap.add_argument("--phase", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda")
call("tic_set_option", b"gemm_tile", a.tile)
call("tic_set_option", b"tn_streamk", a.streamk)
call("tic_set_option", b"tn_phase", a.phase)
M, N, K = a.M, a.N, a.K
if a.kind == "nt":
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    of = torch.empty(M, N, device=dev)
    resid = torch.randn(M, N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for _ in range(a.reps):
        call("tic_gemm_nt_bf16", A.data_ptr(), W.data_ptr(), M, N, K, a.epi, None if a.epi == 3 else bias.data_ptr(), o1.data_ptr(), o2.data_ptr(),
             of.data_ptr(), resid.data_ptr(), aux.data_ptr(), None, 0, current_stream())
else:
    import ctypes
    D, F = N, K
    shapes = [(D, F), (F, D), (D, D), (3 * D, D)]
    As = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n, k in shapes]
    Bs = [torch.randn(M, k, device=dev).to(torch.bfloat16) for n, k in shapes]
    Cs = [torch.zeros(n, k, device=dev) for n, k in shapes]
    PA = (ctypes.c_void_p * 4)(*[x.data_ptr() for x in As])
    PB = (ctypes.c_void_p * 4)(*[x.data_ptr() for x in Bs])
    PC = (ctypes.c_void_p * 4)(*[x.data_ptr() for x in Cs])
    NN = (ctypes.c_int * 4)(*[s[0] for s in shapes])
    KK = (ctypes.c_int * 4)(*[s[1] for s in shapes])
    for _ in range(a.reps):
        call("tic_gemm_tn_group_bf16", 4, PA, PB, PC, NN, KK, M, current_stream())
torch.cuda.synchronize()
print("done")
