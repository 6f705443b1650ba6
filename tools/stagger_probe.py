#!/usr/bin/env python3
"""Experiment: does delaying every other first-wave workgroup (epilogue HBM bursts of one half of the CUs inside the main
loops of the other half) shorten the epilogue-heavy NT GEMMs?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402
dev = torch.device("cuda")
M = 65404
for (N, K, epi) in ((4096, 1024, 5), (4096, 1024, 6), (1024, 1024, 2), (1024, 4096, 2), (3072, 1024, 0), (1024, 4096, 0)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev); o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    of = torch.empty(M, N, device=dev); resid = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for st in (0, 1, 2, 3, 4, 6):
        call("tic_set_option", b"gemm_stagger", st)
        def run():
            call("tic_gemm_nt_bf16", A.data_ptr(), W.data_ptr(), M, N, K, epi, bias.data_ptr(), o1.data_ptr(), o2.data_ptr(), of.data_ptr(), resid.data_ptr(), aux.data_ptr(), None, 0, current_stream())
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} K={K} epi={epi} stagger={st}: {e0.elapsed_time(e1)/10*1e3:8.1f} us", flush=True)
call("tic_set_option", b"gemm_stagger", -1)
