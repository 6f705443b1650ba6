#!/usr/bin/env python3
"""Randomised shape sweep of the GEMM entry points against torch (fp32 math on the bf16-rounded operands): NT with every
epilogue on both tile families, grouped TN (plain / stream-K), implicit-GEMM convolution.  Prints mismatches; exit code 1 if any."""
import ctypes, os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream  # noqa: E402
dev = torch.device("cuda")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
def bf(t): return t.to(torch.bfloat16)
def close(a, b, atol, rtol, what):
    global bad
    d = (a - b).abs()
    tol = atol + rtol * b.abs()
    if not bool((d <= tol).all()):
        bad += 1
        print(f"MISMATCH {what}: max err {float(d.max()):.4g} (tol {float(tol.flatten()[d.flatten().argmax()]):.4g})", flush=True)
for it in range(60):
    tile = rng.choice([0, 128, 256])
    N = rng.choice([256, 512, 768, 1024]) if tile == 256 else rng.choice([8, 64, 72, 128, 200, 256, 384, 1024])
    K = 64 * rng.choice([1, 2, 3, 4, 5, 8, 12, 16])
    M = rng.choice([1, 7, 64, 127, 128, 129, 255, 256, 257, 300, 511, 777, 1000, 2049, 4097])
    epi = rng.choice([0, 1, 2, 3, 5, 6])
    call("tic_set_option", b"gemm_tile", tile)
    A, W, bias = bf(torch.randn(M, K, device=dev) * 0.5), bf(torch.randn(N, K, device=dev) * 0.2), torch.randn(N, device=dev)
    o1 = torch.full((M, N), 7.0, device=dev).to(torch.bfloat16); o2 = torch.full((M, N), 7.0, device=dev).to(torch.bfloat16)
    of = torch.full((M, N), 7.0, device=dev); resid = torch.randn(M, N, device=dev); aux = bf(torch.randn(M, N, device=dev))
    cs = torch.zeros(N, device=dev)
    use_cs = epi in (0, 3, 6) and rng.random() < 0.5
    call("tic_gemm_nt_bf16_ex", A.data_ptr(), W.data_ptr(), M, N, K, epi, None if epi in (3, 6) else bias.data_ptr(), o1.data_ptr(), o2.data_ptr(), of.data_ptr(),
         resid.data_ptr(), aux.data_ptr(), None, 0, cs.data_ptr() if use_cs else None, current_stream())
    torch.cuda.synchronize()
    acc = A.float() @ W.float().t()
    u = bf(acc + (0 if epi in (3, 6) else bias)).float()
    tag = f"nt tile={tile} M={M} N={N} K={K} epi={epi}"
    if epi == 0: close(o1.float(), u, 0.03, 0.02, tag)
    if epi == 1: close(o1.float(), u, 0.03, 0.02, tag); close(o2.float(), torch.nn.functional.gelu(o1.float()), 0.02, 0.02, tag + " gelu")
    if epi == 2: close(of, resid + u, 0.03, 0.02, tag)
    if epi == 3:
        x = aux.float().double().requires_grad_(True); torch.nn.functional.gelu(x).sum().backward()
        close(o1.float(), bf(u * x.grad.float()).float(), 0.03, 0.03, tag)
    if epi == 5:
        x = u.double().requires_grad_(True); torch.nn.functional.gelu(x).sum().backward()
        close(o1.float(), x.grad.float(), 0.01, 0.01, tag + " dgelu"); close(o2.float(), torch.nn.functional.gelu(u), 0.02, 0.02, tag + " gelu")
    if epi == 6: close(o1.float(), bf(u * aux.float()).float(), 0.03, 0.03, tag)
    if use_cs: close(cs, o1.float().sum(0), 0.02 * M ** 0.5 + 0.05, 0.02, tag + " colsum")
call("tic_set_option", b"gemm_tile", 0)
for it in range(12):
    G = rng.choice([1, 2, 4]); M = rng.choice([1, 63, 64, 65, 200, 1000, 4099, 9000])
    shapes = [(256 * rng.choice([1, 2, 4]), 256 * rng.choice([1, 2, 3])) for _ in range(G)]
    As = [bf(torch.randn(M, n, device=dev)) for n, k in shapes]; Bs = [bf(torch.randn(M, k, device=dev)) for n, k in shapes]
    PA = (ctypes.c_void_p * G)(*[t.data_ptr() for t in As]); PB = (ctypes.c_void_p * G)(*[t.data_ptr() for t in Bs])
    NN = (ctypes.c_int * G)(*[s[0] for s in shapes]); KK = (ctypes.c_int * G)(*[s[1] for s in shapes])
    for sk, ph in ((0, 1), (1, 1), (1, 0), (5, 0)):
        call("tic_set_option", b"gemm_tile", 256); call("tic_set_option", b"tn_streamk", sk); call("tic_set_option", b"tn_phase", ph)
        Cs = [torch.ones(n, k, device=dev) for n, k in shapes]
        PC = (ctypes.c_void_p * G)(*[t.data_ptr() for t in Cs])
        call("tic_gemm_tn_group_bf16", G, PA, PB, PC, NN, KK, M, current_stream()); torch.cuda.synchronize()
        for A, B, C in zip(As, Bs, Cs):
            ref = 1 + A.float().t() @ B.float()
            close(C, ref, 2e-4 * float(ref.abs().max()) + 1e-4, 1e-4, f"tn group M={M} shapes={shapes} streamk={sk} phase={ph}")
call("tic_set_option", b"gemm_tile", 0); call("tic_set_option", b"tn_streamk", 1); call("tic_set_option", b"tn_phase", 1)
from tests import resnet_checks as rc  # noqa: E402
def c(name, *a): call(name, *a[:-1], current_stream())
for it in range(10):
    B, H, W = rng.choice([1, 2, 5]), rng.choice([5, 8, 14, 23]), rng.choice([5, 9, 14])
    Cin, Cout, stride = 64 * rng.choice([1, 2, 4]), 64 * rng.choice([1, 2, 3]), rng.choice([1, 2])
    try:
        rc.check_implicit_conv(c, dev, B, H, W, Cin, Cout, stride)
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH conv B={B} H={H} W={W} Cin={Cin} Cout={Cout} stride={stride}: {str(e)[:200]}", flush=True)
print("gemm fuzz:", "CLEAN" if bad == 0 else f"{bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
