#!/usr/bin/env python3
"""Race screen for the pipelined GEMM kernels: every epilogue of the 256x256 NT kernel (no atomics involved) must give
bit-identical outputs over repeated launches and agree with the 128x128 kernel; then step-level run-to-run spread."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
bad = 0
for (M, N, K) in ((1576, 2304, 768), (1576, 3072, 768), (3136, 1024, 768), (32702, 1024, 1024), (16351, 4096, 1024), (5000, 256, 64), (5000, 512, 192)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    resid = torch.randn(M, N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for epi in (0, 1, 2, 3, 5, 6):
        outs = {}
        for tile in (128, 256):
            call("tic_set_option", b"gemm_tile", tile)
            res = []
            for rep in range(6 if tile == 256 else 1):
                o1 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); o2 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); of = torch.zeros(M, N, device=dev)
                call("tic_gemm_nt_bf16", A.data_ptr(), W.data_ptr(), M, N, K, epi, None if epi in (3, 6) else bias.data_ptr(), o1.data_ptr(), o2.data_ptr(),
                     of.data_ptr(), resid.data_ptr(), aux.data_ptr(), None, 0, current_stream())
                torch.cuda.synchronize()
                res.append((o1, o2, of))
            outs[tile] = res
        ref = outs[128][0]
        for i, r in enumerate(outs[256]):
            for a, b, nm in zip(r, ref, ("out", "out2", "out_f32")):
                if not torch.equal(a, b):
                    d = (a.float() - b.float()).abs()
                    print(f"MISMATCH M={M} N={N} K={K} epi={epi} rep={i} {nm}: {int((d > 0).sum())} elements, max {float(d.max()):.4g}", flush=True)
                    bad += 1
call("tic_set_option", b"gemm_tile", 0)
print("gemm race screen:", "CLEAN" if bad == 0 else f"{bad} mismatching outputs", flush=True)

# dW (TN) kernels: the one-tile-per-workgroup form has no atomics -> repeated launches must agree bit for bit; the
# stream-K forms add partial tiles with fp32 atomics -> equal to the plain result up to fp32 summation order
import ctypes
bad_tn = 0
try:   # `tn_phase` (flat instead of phase-aligned stream-K shares) is a measurement-library knob (TIC_HIP_LIB=.../libtic_hip_dbg.so)
    call("tic_set_option", b"tn_phase", 1)
    HAVE_PHASE_KNOB = True
except Exception:
    HAVE_PHASE_KNOB = False
for (M, shapes) in ((32702, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]), (16351, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                    (4099, [(256, 512), (512, 256)]), (65, [(256, 256)])):
    As = [torch.randn(M, n, device=dev).to(torch.bfloat16) for n, k in shapes]
    Bs = [torch.randn(M, k, device=dev).to(torch.bfloat16) for n, k in shapes]
    G = len(shapes)
    PA = (ctypes.c_void_p * G)(*[t.data_ptr() for t in As]); PB = (ctypes.c_void_p * G)(*[t.data_ptr() for t in Bs])
    NN = (ctypes.c_int * G)(*[s_[0] for s_ in shapes]); KK = (ctypes.c_int * G)(*[s_[1] for s_ in shapes])
    def run(streamk, phase):
        call("tic_set_option", b"gemm_tile", 256); call("tic_set_option", b"tn_streamk", streamk)
        if HAVE_PHASE_KNOB:
            call("tic_set_option", b"tn_phase", phase)
        Cs = [torch.zeros(n, k, device=dev) for n, k in shapes]
        PC = (ctypes.c_void_p * G)(*[t.data_ptr() for t in Cs])
        call("tic_gemm_tn_group_bf16", G, PA, PB, PC, NN, KK, M, current_stream())
        torch.cuda.synchronize()
        return Cs
    plain = [run(0, 1) for _ in range(5)]
    for i in range(1, 5):
        for a, b in zip(plain[i], plain[0]):
            if not torch.equal(a, b):
                print(f"TN MISMATCH plain kernel M={M} rep {i}: {int((a != b).sum())} elements differ", flush=True); bad_tn += 1
    ref = [A.float().t() @ B.float() for A, B in zip(As, Bs)]
    for a, r in zip(plain[0], ref):
        err = float((a - r).abs().max() / r.abs().max())
        if err > 2e-5:
            print(f"TN plain kernel vs fp32 matmul M={M}: rel err {err:.3e}", flush=True); bad_tn += 1
    for sk, ph in ((1, 1), (1, 0), (7, 0)):
        for rep in range(3):
            out = run(sk, ph)
            for a, b in zip(out, plain[0]):
                err = float((a - b).abs().max() / b.abs().max())
                if err > 2e-5:
                    print(f"TN MISMATCH stream-K({sk},{ph}) M={M} rep {rep}: rel err {err:.3e}", flush=True); bad_tn += 1
call("tic_set_option", b"gemm_tile", 0); call("tic_set_option", b"tn_streamk", 1)
if HAVE_PHASE_KNOB:
    call("tic_set_option", b"tn_phase", 1)
print("dW race screen:", "CLEAN" if bad_tn == 0 else f"{bad_tn} problems", flush=True)

from touhouimageclassification_amd.ViT.model import ViT
from touhouimageclassification_amd.optim import FusedAdamW
from touhouimageclassification_amd.step import fused_train_step
x = torch.randn(8, 3, 224, 224, device=dev); y = torch.randint(0, 10, (8,), device=dev)
res = []
for rep in range(4):
    torch.manual_seed(0)
    m = ViT(10, pretrained=False, model_name="google/vit-base-patch16-224").to(dev)
    opt = FusedAdamW(m, lr=1e-4, weight_decay=0.01)
    losses = [float(fused_train_step(m, opt, x, y, None)[0]) for _ in range(2)]
    torch.cuda.synchronize()
    res.append((losses, m._engine.params.clone()))
    print("rep", rep, losses, "max|dparam| vs rep0", float((res[-1][1] - res[0][1]).abs().max()), flush=True)

# gradient-level determinism: same weights, same batch, repeated fwd+bwd; report the spread per parameter block
torch.manual_seed(0)
m = ViT(10, pretrained=False, model_name="google/vit-base-patch16-224").to(dev)
e = m._engine
def grads_once():
    logits = e.forward(x)
    B, C = logits.shape
    loss = torch.zeros(1, device=dev); dl = torch.empty_like(logits)
    call("tic_softmax_xent", logits.data_ptr(), y.data_ptr(), None, loss.data_ptr(), dl.data_ptr(), B, C, 1.0, current_stream())
    e.grads.zero_()
    e.backward(dl)
    torch.cuda.synchronize()
    return e.grads.clone()
g0 = grads_once()
scale = float(g0.abs().max())
for rep in range(4):
    g = grads_once()
    d = (g - g0).abs()
    i = int(d.argmax())
    print(f"grad rep {rep}: max|dg| {float(d.max()):.3e} at flat index {i} (g0 there {float(g0[i]):.3e}); global max|g| {scale:.3e}; elements differing {int((d > 0).sum())}/{d.numel()}", flush=True)
