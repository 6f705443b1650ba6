#!/usr/bin/env python3
"""LayerNorm forward / backward and AdamW in isolation: microseconds and TB/s of algorithmic bytes (fwd 6 B, bwd 16 B per element;
AdamW 30 B per parameter).   python tools/ln_bench.py [rows=65404] [knob value ...]
A/B of two builds: TIC_HIP_LIB=<path to the other libtic_hip.so> python tools/ln_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65404
for k, v in zip(sys.argv[2::2], sys.argv[3::2]):
    call("tic_set_option", k.encode(), int(v))
D = 1024
dev = torch.device("cuda")


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x, dres = torch.randn(rows, D, device=dev), torch.randn(rows, D, device=dev)
gamma, beta = torch.randn(D, device=dev), torch.randn(D, device=dev)
y = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
dy = torch.randn(rows, D, device=dev).to(torch.bfloat16)
dx, dxb = torch.empty(rows, D, device=dev), torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
dg, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)   # > Infinity Cache: every timed launch reads from HBM


def fwd():
    flush.zero_()
    call("tic_layernorm_fwd", x.data_ptr(), D, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, 1e-12, current_stream())


def bwd():
    flush.zero_()
    call("tic_layernorm_bwd_ex", dy.data_ptr(), x.data_ptr(), D, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dres.data_ptr(), dx.data_ptr(), dxb.data_ptr(),
         dg.data_ptr(), db.data_ptr(), cs.data_ptr(), rows, D, current_stream())


base = t_us(lambda: flush.zero_())
f, b = t_us(fwd) - base, t_us(bwd) - base
print(f"lib={os.environ.get('TIC_HIP_LIB', 'default')} rows={rows}: ln_fwd {f:7.1f} us = {rows * D * 6 / f / 1e6:5.2f} TB/s   ln_bwd {b:7.1f} us = {rows * D * 16 / b / 1e6:5.2f} TB/s"
      f"   (flush {base:.1f} us subtracted)", flush=True)
