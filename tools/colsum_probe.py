#!/usr/bin/env python3
"""What does the fused column-sum (bias-gradient) epilogue cost on the 256x256 NT kernel?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402
dev = torch.device("cuda")
M = 65404
for (N, K, epi) in ((4096, 1024, 6), (1024, 1024, 0)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev); aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    cs = torch.zeros(N, device=dev)
    for with_cs in (0, 1, 0, 1):
        def run():
            call("tic_gemm_nt_bf16_ex", A.data_ptr(), W.data_ptr(), M, N, K, epi, None, o1.data_ptr(), None, None, None, aux.data_ptr(), None, 0,
                 cs.data_ptr() if with_cs else None, current_stream())
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(f"N={N} K={K} epi={epi} colsum={with_cs}: {e0.elapsed_time(e1)/10*1e3:8.1f} us", flush=True)
