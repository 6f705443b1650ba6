#!/usr/bin/env python3
"""Which pipe bounds the 256x256 NT main loop?  Times the EPI_BF16 kernel with parts of the loop compiled out
(gemm256.h DBG: 1 = no LDS-DMA, 2 = no fragment ds_reads, 4 = no MFMA).  Outputs are garbage for dbg != 0."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd import build  # noqa: E402
os.environ["TIC_HIP_LIB"] = build.build_hip_dbg()   # the ablation variants live only in the measurement library (-DTIC_MEASURE)
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402

dev = torch.device("cuda")
call("tic_set_option", b"gemm_tile", 256)
for (M, N, K) in ((32702, 3072, 1024), (32702, 4096, 4096)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for dbg in (0, 16, 17, 18, 19, 24):
        call("tic_set_option", b"gemm_dbg", dbg)
        def run():
            call("tic_gemm_nt_bf16", A.data_ptr(), W.data_ptr(), M, N, K, 0, bias.data_ptr(), out.data_ptr(), None, None, None, None, None, 0, current_stream())
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"M={M} N={N} K={K} dbg={dbg} ({'noDMA ' if dbg&1 else ''}{'noLDSread ' if dbg&2 else ''}{'noMFMA ' if dbg&4 else ''}{'vmcnt12 ' if dbg&8 else ''}{'1barrier/phase' if dbg&16 else ''}): {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF-equivalent", flush=True)
call("tic_set_option", b"gemm_dbg", 0)

# the same isolation for the dW (TN) kernel: one 4096 x 4096 problem = 256 tiles, one per CU, full reduction over M
import ctypes
call("tic_set_option", b"tn_streamk", 0)
M, N, K = 32702, 4096, 4096
A = torch.randn(M, N, device=dev).to(torch.bfloat16); B = torch.randn(M, K, device=dev).to(torch.bfloat16); C = torch.zeros(N, K, device=dev)
PA = (ctypes.c_void_p * 1)(A.data_ptr()); PB = (ctypes.c_void_p * 1)(B.data_ptr()); PC = (ctypes.c_void_p * 1)(C.data_ptr())
NN = (ctypes.c_int * 1)(N); KK = (ctypes.c_int * 1)(K)
for dbg in (0,):
    call("tic_set_option", b"gemm_dbg", dbg)
    def run():
        call("tic_gemm_tn_group_bf16", 1, PA, PB, PC, NN, KK, M, current_stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"TN M={M} N={N} K={K} dbg={dbg} ({'noDMA ' if dbg&1 else ''}{'noLDSread ' if dbg&2 else ''}{'noMFMA' if dbg&4 else ''}): {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF-equivalent", flush=True)
call("tic_set_option", b"gemm_dbg", 0)
call("tic_set_option", b"tn_streamk", 1)
