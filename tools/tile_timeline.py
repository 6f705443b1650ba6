#!/usr/bin/env python3
"""Where does a 256x256 NT tile spend its time?  Runs the measurement library (-DTIC_MEASURE): lane 0 of every workgroup stamps the
100 MHz clock at kernel entry / after the prologue / after the K loop / after the staging pass / after the second (store) pass, plus
its HW_ID, so that the gap between one workgroup's end and the next one's start on the SAME CU can be read too.

  python tools/tile_timeline.py [--batch 332]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd import build  # noqa: E402
os.environ["TIC_HIP_LIB"] = build.build_hip_dbg()
from touhouimageclassification_amd._lib import call, current_stream, lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=332)
ap.add_argument("--in-step", action="store_true", help="stamp the launches of transformer block 12 INSIDE a full ViT-L training step "
                "(what the previous kernel's tail and the next one's start do to the workgroups' phase is part of the answer)")
args = ap.parse_args()
M, D, F = args.batch * 197, 1024, 4096
dev = torch.device("cuda")
h = lib()
h.tic_dbg_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int]
h.tic_dbg_set_stamp_buffer.restype = ctypes.c_int
h.tic_dbg_launch_log.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int * 4)]
h.tic_dbg_launch_log.restype = ctypes.c_int
HDR = (f"{'case':34s} {'rounds':>6s} {'prologue':>16s} {'K loop':>16s} {'drain+stage':>16s} {'2nd pass':>16s} {'tile':>8s} "
       f"{'gap->next WG':>16s} {'span':>8s}")


def report(name, s, grid):
    t = s[:, :5].astype(np.float64) * 0.01   # 100 MHz -> us
    seg = np.diff(t, axis=1)                  # prologue, loop, drain+stage, 2nd pass
    cu = (s[:, 7] & 0xF) * (1 << 16) + (s[:, 6] & 0xFF00)   # XCC id, HW_ID bits 8..15 = CU / SH / SE ids
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        order = idx[np.argsort(t[idx, 0])]
        gaps += list(t[order[1:], 0] - t[order[:-1], 4])

    def q(v):
        v = np.asarray(v)
        return f"{np.median(v):6.2f} ({np.percentile(v, 10):4.1f}..{np.percentile(v, 90):4.1f})" if len(v) else "      -"
    # how spread out are the workgroups of one round?  std of the loop-end time over the first 256 workgroups to start
    first = np.argsort(t[:, 0])[:256]
    print(f"{name:34s} {grid / 256:6.2f} {q(seg[:, 0]):>16s} {q(seg[:, 1]):>16s} {q(seg[:, 2]):>16s} {q(seg[:, 3]):>16s} "
          f"{np.median(t[:, 4] - t[:, 0]):8.2f} {q(gaps):>16s} {t[:, 4].max() - t[:, 0].min():8.1f}   round-1 start spread {np.std(t[first, 0]):.1f} us, "
          f"loop-end spread {np.std(t[first, 2]):.1f} us, CUs seen {len(np.unique(cu))}", flush=True)


if args.in_step:
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    B = args.batch
    m = ViT(120, pretrained=False, model_name="google/vit-large-patch16-224").to(dev)
    opt = FusedAdamW(m, lr=1e-5, weight_decay=0.01)
    x = torch.randn(B, 3, 224, 224, device=dev)
    y = torch.randint(0, 120, (B,), device=dev)
    for _ in range(3):
        fused_train_step(m, opt, x, y, None)
    torch.cuda.synchronize()
    h.tic_dbg_set_stamp_buffer(None, -1)
    fused_train_step(m, opt, x, y, None)
    torch.cuda.synchronize()
    log = []
    rec = (ctypes.c_int * 4)()
    i = 0
    while h.tic_dbg_launch_log(i, ctypes.byref(rec)) == 0:
        log.append(tuple(rec))
        i += 1
    n = len(log)
    print(f"{n} 256x256 NT launches per step; stamping the 8 launches of the middle transformer block (forward 4, backward 4)")
    fwd0 = 1 + 4 * 12                       # patch embed, then 4 per block
    bwd_first = 1 + 4 * 24                  # first backward launch
    # backward walks blocks 23 .. 0, 4 launches each
    picks = list(range(fwd0, fwd0 + 4)) + list(range(bwd_first + 4 * 11, bwd_first + 4 * 12))
    print(HDR)
    for idx in picks:
        epi, M_, N, K = log[idx]
        grid = ((M_ + 255) // 256) * (N // 256)
        stamps = torch.zeros(grid, 8, dtype=torch.int64, device=dev)
        h.tic_dbg_set_stamp_buffer(stamps.data_ptr(), idx)
        fused_train_step(m, opt, x, y, None)
        torch.cuda.synchronize()
        h.tic_dbg_set_stamp_buffer(None, -1)
        report(f"launch {idx:3d}: <{epi}> N={N} K={K}", stamps.cpu().numpy(), grid)
    sys.exit(0)

call("tic_set_option", b"gemm_tile", 256)

cases = [("qkv      <0> N=3072 K=1024", 3 * D, D, 0), ("o^T dX   <0> N=1024 K=1024", D, D, 0), ("o+resid  <2> N=1024 K=1024", D, D, 2),
         ("fc1+gelu <5> N=4096 K=1024", F, D, 5), ("fc2^T*dg <6> N=4096 K=1024", F, D, 6), ("fc2+res  <2> N=1024 K=4096", D, F, 2),
         ("fc1^T dX <0> N=1024 K=4096", D, F, 0)]
print(f"M = {M}; durations in us, median over workgroups (p10 .. p90)")
print(HDR)
for name, N, K, epi in cases:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if epi == 5 else None
    of = torch.empty(M, N, device=dev) if epi == 2 else None
    resid = torch.randn(M, N, device=dev) if epi == 2 else None
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == 6 else None
    grid = ((M + 255) // 256) * (N // 256)
    stamps = torch.zeros(grid, 8, dtype=torch.int64, device=dev)

    def run():
        call("tic_gemm_nt_bf16", a.data_ptr(), w.data_ptr(), M, N, K, epi, None if epi == 6 else bias.data_ptr(), o1.data_ptr(),
             None if o2 is None else o2.data_ptr(), None if of is None else of.data_ptr(), None if resid is None else resid.data_ptr(),
             None if aux is None else aux.data_ptr(), None, 0, current_stream())
    for _ in range(3):
        run()
    h.tic_dbg_set_stamp_buffer(stamps.data_ptr(), -1)
    run()
    torch.cuda.synchronize()
    h.tic_dbg_set_stamp_buffer(None, -1)
    report(name, stamps.cpu().numpy(), grid)
