#!/usr/bin/env python3
"""Attention forward / backward kernels in isolation at the bench shape (B images x 16 heads, N = 197)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 332
H, N = 16, 197
D = H * 64
dev = torch.device("cuda")
qkv = torch.randn(B * N, 3 * D, device=dev).to(torch.bfloat16)
o = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * H, N, device=dev)
do = torch.randn(B * N, D, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dbias = torch.zeros(3 * D, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
if len(sys.argv) > 2:
    call("tic_set_option", b"attn_fwd_waves", int(sys.argv[2]))   # 4 (default) or 8 waves per forward workgroup
fwd = t(lambda: call("tic_attention_fwd", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, H, N, 0.125, current_stream()))
bwd = t(lambda: call("tic_attention_bwd_ex", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), B, H, N, 0.125, current_stream()))
bwd0 = t(lambda: call("tic_attention_bwd_ex", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dqkv.data_ptr(), None, B, H, N, 0.125, current_stream()))
part = torch.empty(B, 3 * D, device=dev)
try:
    bws = t(lambda: call("tic_attention_bwd_ws", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), part.data_ptr(), 1, B, H, N, 0.125, current_stream()))
except Exception:
    bws = float("nan")
print(f"B={B}: bwd (q,k bias grad via scratch, v skipped) {bws:7.1f} us")
print(f"B={B}: attn fwd {fwd:7.1f} us   bwd (+bias grad) {bwd:7.1f} us   bwd (no bias grad) {bwd0:7.1f} us", flush=True)
# interleaved A/B of the tile-to-wave assignment (measurement library only: TIC_HIP_LIB=.../libtic_hip_dbg.so)
try:
    call("tic_set_option", b"attn_legacy", 0)
    run = lambda: call("tic_attention_bwd_ws", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(), part.data_ptr(), 1, B, H, N, 0.125, current_stream())
    rows = {0: [], 1: []}
    for rnd in range(4):
        for legacy in (1, 0):
            call("tic_set_option", b"attn_legacy", legacy)
            rows[legacy].append(t(run))
    call("tic_set_option", b"attn_legacy", 0)
    for legacy in (1, 0):
        v = sorted(rows[legacy])
        print(f"B={B}: bwd, {'round-1 assignment (waves 0-7 phase A, 8-15 phase B)' if legacy else 'one A tile per wave + B tiles on the rest      '}: median {0.5 * (v[1] + v[2]):7.1f} us  ({', '.join(f'{x:.1f}' for x in rows[legacy])})")
except Exception as e:   # product library: no such knob
    print("(no attn_legacy knob in this library)")
