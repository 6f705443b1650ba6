#!/usr/bin/env python3
"""Experiment (measurement build: TIC_HIP_LIB=.../libtic_hip_dbg.so): do staggered starts of the first-round workgroups -- G groups,
group g delayed by g x r sleep rounds (~4.8 us each), so that the HBM bursts of some CUs' epilogues fall into other CUs' main loops --
shorten the epilogue-heavy NT GEMMs?  Each shape is timed in a ring of 4 operand sets (2-4 GB: no launch finds its operands in the
256 MiB Infinity Cache, as in the training step)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402
dev = torch.device("cuda")
M = int(os.environ.get("TIC_M", 65404))
RING = 4
for (N, K, epi) in ((4096, 1024, 5), (4096, 1024, 6), (1024, 1024, 2), (1024, 4096, 2), (3072, 1024, 0), (1024, 4096, 0)):
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    sets = []
    for _ in range(RING):
        sets.append(dict(A=torch.randn(M, K, device=dev).to(torch.bfloat16), o1=torch.empty(M, N, dtype=torch.bfloat16, device=dev),
                         o2=torch.empty(M, N, dtype=torch.bfloat16, device=dev) if epi == 5 else None,
                         of=torch.empty(M, N, device=dev) if epi == 2 else None, resid=torch.randn(M, N, device=dev) if epi == 2 else None,
                         aux=torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == 6 else None))
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    for (G, st) in ((2, 0), (2, 1), (2, 2), (4, 1), (4, 2), (8, 1), (8, 2), (16, 1)):
        call("tic_set_option", b"gemm_stagger_groups", G)
        call("tic_set_option", b"gemm_stagger", st)

        def run(i):
            s = sets[i % RING]
            call("tic_gemm_nt_bf16", P(s["A"]), W.data_ptr(), M, N, K, epi, bias.data_ptr(), P(s["o1"]), P(s["o2"]), P(s["of"]), P(s["resid"]), P(s["aux"]), None, 0, current_stream())
        for i in range(4):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        print(f"N={N} K={K} epi={epi} groups={G} rounds={st}: {e0.elapsed_time(e1) / 12 * 1e3:8.1f} us", flush=True)
    del sets
    torch.cuda.empty_cache()
call("tic_set_option", b"gemm_stagger", -1)
