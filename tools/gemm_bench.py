#!/usr/bin/env python3
"""GPU micro-benchmark + race screen for the GEMM kernels (interleaved A/B in ONE process, random data).

  python tools/gemm_bench.py [--batch 64] [--screen 20]
Prints TFLOP/s of the 128x128 and the 256x256 NT kernels and of the TN kernel at the ViT-L hot shapes,
and checks that the 256 kernel is BITWISE equal to the 128 kernel over repeated launches (same fp32
accumulation order => any difference is a pipeline race)."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd import ops  # noqa: E402
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402


def time_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[64])
    ap.add_argument("--screen", type=int, default=10)
    ap.add_argument("--hidden", type=int, default=1024)
    args = ap.parse_args()
    dev = torch.device("cuda")
    D, F = args.hidden, 4 * args.hidden
    for B in args.batch:
        M = B * 197
        shapes = [("qkv", M, 3 * D, D, 0), ("o_proj+res", M, D, D, 2), ("fc1+gelu", M, F, D, 1), ("fc2+res", M, D, F, 2),
                  ("dgelu", M, F, D, 3), ("dX(K=3D)", M, D, 3 * D, 0)]
        for name, m, n, k, epi in shapes:
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            bias = torch.randn(n, device=dev)
            resid = torch.randn(m, n, device=dev) if epi == 2 else None
            aux = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi == 3 else None
            o1 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            o2 = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            of = torch.empty(m, n, dtype=torch.float32, device=dev) if epi == 2 else None

            def run():
                call("tic_gemm_nt_bf16", a.data_ptr(), w.data_ptr(), m, n, k, epi, None if epi == 3 else bias.data_ptr(), o1.data_ptr(), o2.data_ptr(),
                     None if of is None else of.data_ptr(), None if resid is None else resid.data_ptr(), None if aux is None else aux.data_ptr(), None, 0, current_stream())
            res = {}
            outs = {}
            for tile in (128, 256, 128, 256):
                call("tic_set_option", b"gemm_tile", tile)
                ms = time_ms(run)
                res.setdefault(tile, []).append(2.0 * m * n * k / ms / 1e9)
                outs[tile] = (of if epi == 2 else o1).clone()
            same = torch.equal(outs[128], outs[256])
            bad = 0
            call("tic_set_option", b"gemm_tile", 256)
            for _ in range(args.screen):
                run()
                bad += int(not torch.equal(of if epi == 2 else o1, outs[128]))
            call("tic_set_option", b"gemm_tile", 0)
            print(f"B={B:4d} {name:12s} M={m:6d} N={n:5d} K={k:5d}  128: {max(res[128]):7.1f} TF  256: {max(res[256]):7.1f} TF  "
                  f"bitwise_equal={same} race_screen_bad={bad}/{args.screen}", flush=True)
        # TN (dW) shapes
        for name, n, k in (("dWqkv", 3 * D, D), ("dWo", D, D), ("dW1", F, D), ("dW2", D, F)):
            a = torch.randn(M, n, device=dev).to(torch.bfloat16)
            x = torch.randn(M, k, device=dev).to(torch.bfloat16)
            c = torch.zeros(n, k, device=dev)
            ms = time_ms(lambda: call("tic_gemm_tn_bf16", a.data_ptr(), x.data_ptr(), c.data_ptr(), M, n, k, current_stream()))
            print(f"B={B:4d} {name:12s} M={M:6d} N={n:5d} K={k:5d}  TN: {2.0 * M * n * k / ms / 1e9:7.1f} TF", flush=True)
        tn_group(M, D, F, dev)


def tn_group(M, D, F, dev):
    import ctypes
    shapes = [(D, F), (F, D), (D, D), (3 * D, D)]
    As = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n, k in shapes]
    Bs = [torch.randn(M, k, device=dev).to(torch.bfloat16) for n, k in shapes]
    Cs = [torch.zeros(n, k, device=dev) for n, k in shapes]
    n = 4
    PA = (ctypes.c_void_p * n)(*[a.data_ptr() for a in As])
    PB = (ctypes.c_void_p * n)(*[b.data_ptr() for b in Bs])
    PC = (ctypes.c_void_p * n)(*[c.data_ptr() for c in Cs])
    NN = (ctypes.c_int * n)(*[s_[0] for s_ in shapes])
    KK = (ctypes.c_int * n)(*[s_[1] for s_ in shapes])
    fl = sum(2.0 * M * a * b for a, b in shapes)
    for name, sk, ph in (("one tile per CU", 0, 0), ("stream-K flat", 1, 0), ("stream-K phase-aligned", 1, 1)) * 2:
        call("tic_set_option", b"gemm_tile", 256)
        call("tic_set_option", b"tn_streamk", sk)
        call("tic_set_option", b"tn_phase", ph)
        ms = time_ms(lambda: call("tic_gemm_tn_group_bf16", n, PA, PB, PC, NN, KK, M, current_stream()))
        print(f"M={M} dW group of 4 ({name}): {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF", flush=True)
    call("tic_set_option", b"gemm_tile", 0)
    call("tic_set_option", b"tn_streamk", 1)
    call("tic_set_option", b"tn_phase", 1)


if __name__ == "__main__":
    main()
