#!/usr/bin/env python3
"""Per-launch microseconds of the ViT-L block GEMMs at the per-GPU batches the reference's launchers use (8-32 images): the 128x128
kernel, the 256x256 kernel unsplit and split-K 2 / 4, and the four dW products per-problem vs the grouped stream-K launch.

  python tools/small_batch_bench.py [--batch 8 16 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd import _capi  # noqa: E402
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402


def time_us(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[8, 16, 32])
    args = ap.parse_args()
    dev = torch.device("cuda")
    D, F = 1024, 4096
    scratch = torch.zeros(_capi.NT_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    call("tic_gemm_nt_scratch", scratch.data_ptr(), scratch.numel())
    try:
        call("tic_set_option", b"nt_deep", -1)
        HAVE_RING_KNOB = True
    except Exception:
        HAVE_RING_KNOB = False
    for B in args.batch:
        M = B * 197
        shapes = [("qkv", 3 * D, D, 0), ("o_proj+res", D, D, 2), ("fc1+gelu", F, D, 1), ("fc2+res", D, F, 2),
                  ("fc2T.dgelu", F, D, 3), ("fc1T", D, F, 0), ("o_projT", D, D, 0), ("qkvT", D, 3 * D, 0)]
        tot = {}
        for name, n, k, epi in shapes:
            a = torch.randn(M, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            bias = torch.randn(n, device=dev)
            resid = torch.randn(M, n, device=dev) if epi == 2 else None
            aux = torch.randn(M, n, device=dev).to(torch.bfloat16) if epi == 3 else None
            o1 = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
            o2 = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
            of = torch.empty(M, n, dtype=torch.float32, device=dev) if epi == 2 else None

            def run():
                call("tic_gemm_nt_bf16", a.data_ptr(), w.data_ptr(), M, n, k, epi, None if epi == 3 else bias.data_ptr(), o1.data_ptr(), o2.data_ptr(),
                     None if of is None else of.data_ptr(), None if resid is None else resid.data_ptr(), None if aux is None else aux.data_ptr(), None, 0, current_stream())
            row = []
            # ring = the 4-stage form of the 128x128 kernel (gemm.h NST = 4); forcing it on / off needs the measurement library
            # (TIC_HIP_LIB=.../libtic_hip_dbg.so); the product library chooses by itself ("auto")
            cfgs = (("128", 128, 0, 0), ("128r", 128, 0, 1), ("128/2", 128, 2, 0), ("128r/2", 128, 2, 1), ("128r/4", 128, 4, 1), ("256", 256, 0, -1), ("256/2", 256, 2, -1), ("auto", 0, -1, -1))
            for label, tile, sp, ring in cfgs:
                call("tic_set_option", b"gemm_tile", tile)
                call("tic_set_option", b"gemm_split", sp)
                if HAVE_RING_KNOB:
                    call("tic_set_option", b"nt_deep", ring)
                elif ring == 1:
                    continue
                us = time_us(run)
                row.append(f"{label} {us:6.1f}")
                tot[label] = tot.get(label, 0.0) + us
            ideal = 2.0 * M * n * k / 1.3e15 * 1e6
            print(f"B={B:3d} {name:11s} N={n:5d} K={k:5d}  " + "  ".join(row) + f"   (at 1300 TF: {ideal:5.1f})", flush=True)
        print(f"B={B:3d} sum over the 8 NT launches of a block: " + "  ".join(f"{k} {v:6.1f}" for k, v in tot.items()), flush=True)
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"gemm_split", -1)
        if HAVE_RING_KNOB:
            call("tic_set_option", b"nt_deep", -1)
        # dW: per problem (128x128 split-M) vs the grouped stream-K launch
        ns, ks = [3 * D, D, F, D], [D, D, D, F]
        As = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n in ns]
        Xs = [torch.randn(M, k, device=dev).to(torch.bfloat16) for k in ks]
        Cs = [torch.zeros(n, k, device=dev) for n, k in zip(ns, ks)]
        import ctypes
        pa = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in As])
        pb = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in Xs])
        pc = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in Cs])
        pn = (ctypes.c_int * 4)(*ns)
        pk = (ctypes.c_int * 4)(*ks)
        for tile, sk, mf in ((128, 1, 0), (256, 1, 0), (256, 1, 16), (256, 0, 0), (0, 1, 0)):
            call("tic_set_option", b"gemm_tile", tile)
            call("tic_set_option", b"tn_streamk", sk)
            call("tic_set_option", b"tn_mfma", mf)
            us = time_us(lambda: call("tic_gemm_tn_group_bf16", 4, pa, pb, pc, pn, pk, M, current_stream()))
            print(f"B={B:3d} dW group (4 products) gemm_tile={tile:3d} tn_streamk={sk} tn_mfma={mf:2d}: {us:7.1f} us   (at 1300 TF: {2.0 * M * 12582912 / 1.3e15 * 1e6:5.1f})", flush=True)
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"tn_streamk", 1)
        call("tic_set_option", b"tn_mfma", 0)
    call("tic_gemm_nt_scratch", None, 0)
    for B in args.batch:   # LayerNorm backward: rows per wave against the dgamma / dbeta atomic rows
        rows = B * 197
        dy = torch.randn(rows, D, device=dev).to(torch.bfloat16)
        x, dres = torch.randn(rows, D, device=dev), torch.randn(rows, D, device=dev)
        gamma, mean, rstd = torch.randn(D, device=dev), torch.randn(rows, device=dev), torch.rand(rows, device=dev) + 0.5
        dx, dxb = torch.empty(rows, D, device=dev), torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
        dg, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        row = []
        for rpw in (1, 2, 4, 8, 16):
            call("tic_set_option", b"ln_bwd_rows", rpw)
            us = time_us(lambda: call("tic_layernorm_bwd_ex", dy.data_ptr(), x.data_ptr(), D, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dres.data_ptr(),
                                      dx.data_ptr(), dxb.data_ptr(), dg.data_ptr(), db.data_ptr(), cs.data_ptr(), rows, D, current_stream()))
            row.append(f"{rpw}: {us:5.1f}")
        call("tic_set_option", b"ln_bwd_rows", 4)
        print(f"B={B:3d} ln_bwd rows={rows:6d} us by rows-per-wave  " + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
