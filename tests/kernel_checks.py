"""Kernel checks shared by the CPU-simulator tests (test_sim_kernels.py) and the GPU tests
(test_gpu_ops.py): every C-ABI operator against plain torch fp32 math on the same device."""
import torch

from tests.simlib import bf, bfr, ptr


class Env:
    def __init__(self, dev, call, stream=None, seed=0):
        self.dev, self._call, self.stream = torch.device(dev), call, stream
        self.gen = torch.Generator().manual_seed(seed)

    def rnd(self, *shape, scale=1.0):
        return (torch.randn(*shape, generator=self.gen) * scale).to(self.dev)

    def call(self, name, *args):
        # the trailing argument of every entry point is the stream
        args = list(args)
        if self.stream is not None:
            args[-1] = self.stream()
        self._call(name, *args)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()




def check_gemm_nt_bias_bf16(env, M, N, K):
    rnd, call, dev = env.rnd, env.call, env.dev
    A, B, bias = bf(rnd(M, K)), bf(rnd(N, K)), rnd(N)
    out = torch.full((M, N), 7.0, device=dev).to(torch.bfloat16)
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), M, N, K, 0, ptr(bias), ptr(out), None, None, None, None, None, 0, None)
    ref = A.float() @ B.float().t() + bias
    torch.testing.assert_close(out.float(), ref, atol=0.06, rtol=0.02)


def check_gemm_nt_gelu_resid_dgelu_patch(env, N=128, K=128, imgs=3, Pn=50):
    rnd, call, dev = env.rnd, env.call, env.dev
    M = imgs * Pn
    A, B, bias = bf(rnd(M, K, scale=0.3)), bf(rnd(N, K, scale=0.3)), rnd(N, scale=0.1)
    u, g = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), M, N, K, 1, ptr(bias), ptr(u), ptr(g), None, None, None, None, 0, None)
    ref_u = bfr(A.float() @ B.float().t() + bias)
    torch.testing.assert_close(u.float(), ref_u, atol=0.03, rtol=0.02)
    torch.testing.assert_close(g.float(), torch.nn.functional.gelu(u.float()), atol=0.02, rtol=0.02)
    # residual
    resid, out = rnd(M, N), torch.empty(M, N, device=dev)
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), M, N, K, 2, ptr(bias), None, None, ptr(out), ptr(resid), None, None, 0, None)
    torch.testing.assert_close(out, resid + ref_u, atol=0.03, rtol=0.02)
    # dgelu
    d = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), M, N, K, 3, None, ptr(d), None, None, None, ptr(u), None, 0, None)
    uu = u.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).backward(bfr(A.float() @ B.float().t()))
    torch.testing.assert_close(d.float(), uu.grad, atol=0.03, rtol=0.03)
    # fc1 + GELU with the derivative saved (EPI 5) and the one-multiply backward through it (EPI 6, with fused column sums)
    dg5, g5 = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), M, N, K, 5, ptr(bias), ptr(dg5), ptr(g5), None, None, None, None, 0, None)
    torch.testing.assert_close(g5.float(), g.float(), atol=1e-3, rtol=8e-3)   # EPI_GELU's activation up to one bf16 ulp
    u64 = ref_u.double().requires_grad_(True)
    torch.nn.functional.gelu(u64).sum().backward()
    torch.testing.assert_close(dg5.float(), u64.grad.float(), atol=5e-3, rtol=8e-3)   # bf16 rounding of gelu'
    d6, cs6 = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.zeros(N, device=dev)
    call("tic_gemm_nt_bf16_ex", ptr(A), ptr(B), M, N, K, 6, None, ptr(d6), None, None, None, ptr(dg5), None, 0, ptr(cs6), None)
    # both sides round twice (the product, then product x derivative) from accumulators that differ in their fp32 summation order: up
    # to 2 bf16 ulps (2 x 2^-7 relative) apart by construction; the band is twice that (tests/headroom.py: no band more than half used)
    torch.testing.assert_close(d6.float(), bfr(bfr(A.float() @ B.float().t()) * dg5.float()), atol=4e-3, rtol=3.2e-2)
    torch.testing.assert_close(d6.float(), uu.grad, atol=0.03, rtol=0.03)
    torch.testing.assert_close(cs6, d6.float().sum(0), atol=0.05, rtol=0.02)
    # ADDAUX (7): the product added onto a bf16 tensor IN PLACE, exactly bf16(bf16(acc) + aux) = what a separate add pass would store
    acc7 = bf(rnd(M, N))
    want7 = bfr(bfr(A.float() @ B.float().t()) + acc7.float())
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), M, N, K, 7, None, ptr(acc7), None, None, None, ptr(acc7), None, 0, None)
    # one bf16 ulp of the PRODUCT (it rounds before the add, which may cancel) + one of the sum, by construction; the band is twice that
    torch.testing.assert_close(acc7.float(), want7, atol=4e-2, rtol=3.2e-2)
    # patch epilogue: M = images * patches, rows remapped past the CLS slot
    pos = rnd(Pn + 1, N)
    h = torch.zeros(imgs * (Pn + 1), N, device=dev)
    call("tic_gemm_nt_bf16", ptr(A), ptr(B), imgs * Pn, N, K, 4, ptr(bias), None, None, ptr(h), None, None, ptr(pos), Pn, None)
    h = h.view(imgs, Pn + 1, N)
    assert h[:, 0].abs().max() == 0
    torch.testing.assert_close(h[:, 1:], ref_u.view(imgs, Pn, N) + pos[1:], atol=0.03, rtol=0.02)


def check_gemm_tn(env, M, N=128, K=256):
    rnd, call, dev = env.rnd, env.call, env.dev
    A, B = bf(rnd(M, N)), bf(rnd(M, K))
    Cm = rnd(N, K)
    ref = Cm + A.float().t() @ B.float()
    call("tic_gemm_tn_bf16", ptr(A), ptr(B), ptr(Cm), M, N, K, None)
    torch.testing.assert_close(Cm, ref, atol=2e-3, rtol=1e-3)


def check_layernorm_fwd_bwd(env, D, rows=9):
    rnd, call, dev = env.rnd, env.call, env.dev
    x, gamma, beta = rnd(rows, D) * 2 + 0.5, 1 + rnd(D, scale=0.1), rnd(D, scale=0.1)
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    call("tic_layernorm_fwd", ptr(x), D, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, D, 1e-12, None)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-12)
    torch.testing.assert_close(y.float(), ref.detach(), atol=0.02, rtol=0.01)
    torch.testing.assert_close(mean, x.mean(-1), atol=1e-5, rtol=1e-5)
    dy = bf(rnd(rows, D))
    ref.backward(dy.float())
    dres = rnd(rows, D)
    dx, dxb = torch.empty(rows, D, device=dev), torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    call("tic_layernorm_bwd", ptr(dy), ptr(x), D, ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dxb), ptr(dg), ptr(db), rows, D, None)
    torch.testing.assert_close(dx, dres + xr.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(dxb.float(), bfr(dx), atol=0, rtol=0)
    torch.testing.assert_close(dg, gr.grad, atol=2e-3, rtol=1e-3)   # fp32 sums over `rows` terms, atomics order
    torch.testing.assert_close(db, br.grad, atol=2e-3, rtol=1e-3)


def check_layernorm_strided_cls_rows(env):
    rnd, call, dev = env.rnd, env.call, env.dev
    B, N, D = 3, 5, 128
    h = rnd(B, N, D)
    gamma, beta = 1 + rnd(D, scale=0.1), rnd(D, scale=0.1)
    y = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(B, device=dev), torch.empty(B, device=dev)
    call("tic_layernorm_fwd", ptr(h), N * D, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), B, D, 1e-12, None)
    torch.testing.assert_close(y.float(), torch.nn.functional.layer_norm(h[:, 0], (D,), gamma, beta, 1e-12), atol=0.02, rtol=0.01)


def _attn_ref(qkv, B, H, N):
    D = H * 64
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    return o, torch.logsumexp(s, -1)


def check_attention_fwd_bwd(env, B, H, N):
    rnd, call, dev = env.rnd, env.call, env.dev
    D = H * 64
    qkv = bf(rnd(B * N, 3 * D))
    o = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H, N, device=dev)
    call("tic_attention_fwd", ptr(qkv), ptr(o), ptr(lse), B, H, N, 0.125, None)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, H, N)
    torch.testing.assert_close(o.float(), o_ref.detach(), atol=0.02, rtol=0.02)
    torch.testing.assert_close(lse.view(B, H, N), lse_ref.detach(), atol=2e-3, rtol=1e-3)
    do = bf(rnd(B * N, D))
    o_ref.backward(do.float())
    dqkv = torch.full((B * N, 3 * D), 9.0, device=dev).to(torch.bfloat16)
    call("tic_attention_bwd", ptr(qkv), ptr(o), ptr(lse), ptr(do), ptr(dqkv), B, H, N, 0.125, None)
    torch.testing.assert_close(dqkv.float(), qr.grad, atol=0.03, rtol=0.05)



def check_elementwise_ops(env):
    rnd, call, dev = env.rnd, env.call, env.dev
    # patchify
    B, C, img, patch = 2, 3, 32, 16
    x = rnd(B, C, img, img)
    G = img // patch
    P = torch.empty(B * G * G, C * patch * patch, dtype=torch.bfloat16, device=dev)
    call("tic_patchify", ptr(x), ptr(P), B, C, img, patch, None)
    ref = x.reshape(B, C, G, patch, G, patch).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, -1)
    torch.testing.assert_close(P.float(), bfr(ref), atol=0, rtol=0)
    # colsum
    M, N = 77, 384
    a = bf(rnd(M, N))
    out = torch.ones(N, device=dev)
    call("tic_colsum_bf16", ptr(a), ptr(out), M, N, None)
    torch.testing.assert_close(out, 1 + a.float().sum(0), atol=1e-3, rtol=1e-4)
    # casts
    w = rnd(128, 192)
    w16, wT = torch.empty(128, 192, dtype=torch.bfloat16, device=dev), torch.empty(192, 128, dtype=torch.bfloat16, device=dev)
    call("tic_cast_bf16", ptr(w), ptr(w16), w.numel(), None)
    call("tic_cast_transpose_bf16", ptr(w), ptr(wT), 128, 192, None)
    assert torch.equal(w16, bf(w)) and torch.equal(wT, bf(w).t().contiguous())
    # embed cls / bwd / gather
    Bn, N_, D = 2, 5, 128
    cls, pos, h = rnd(D), rnd(N_, D), torch.zeros(Bn, N_, D, device=dev)
    call("tic_embed_cls", ptr(cls), ptr(pos), ptr(h), Bn, N_, D, None)
    torch.testing.assert_close(h[:, 0], (cls + pos[0]).expand(Bn, D))
    dh = rnd(Bn, N_, D)
    dcls, dpos = torch.zeros(D, device=dev), torch.zeros(N_, D, device=dev)
    call("tic_embed_bwd", ptr(dh), ptr(dcls), ptr(dpos), Bn, N_, D, None)
    torch.testing.assert_close(dcls, dh[:, 0].sum(0))
    torch.testing.assert_close(dpos, dh.sum(0))
    g = torch.empty(Bn * (N_ - 1), D, dtype=torch.bfloat16, device=dev)
    call("tic_gather_patch_rows", ptr(dh), ptr(g), Bn, N_, D, None)
    assert torch.equal(g, bf(dh[:, 1:].reshape(-1, D)))


def check_adamw_matches_torch(env):
    rnd, call, dev = env.rnd, env.call, env.dev
    n = 1000
    p0, g = rnd(n), rnd(n)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    w16 = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for step in (1, 2, 3):
        gs = g * step   # keep alive across the call
        ref.grad = gs
        opt.step()
        call("tic_adamw", ptr(p), ptr(gs), ptr(m), ptr(v), ptr(w16), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, None)
    torch.testing.assert_close(p, ref.detach(), atol=1e-6, rtol=1e-5)
    assert torch.equal(w16, bf(p))


def check_head_and_xent(env, soft):
    rnd, call, dev = env.rnd, env.call, env.dev
    B, Cc, D = 5, 10, 128
    z, W, b = bf(rnd(B, D)), rnd(Cc, D, scale=0.1), rnd(Cc, scale=0.1)
    logits = torch.empty(B, Cc, device=dev)
    call("tic_head_fwd", ptr(z), ptr(W), ptr(b), ptr(logits), B, Cc, D, None)
    torch.testing.assert_close(logits, bfr(z.float() @ bfr(W).t() + bfr(b)), atol=0.02, rtol=0.01)
    y = torch.randint(0, Cc, (B,), device=dev)
    t = torch.softmax(rnd(B, Cc), -1)
    lr = logits.clone().requires_grad_(True)
    loss_ref = torch.nn.functional.cross_entropy(lr, t if soft else y)
    loss_ref.backward()
    loss, dl = torch.zeros(1, device=dev), torch.empty(B, Cc, device=dev)
    call("tic_softmax_xent", ptr(logits), None if soft else ptr(y), ptr(t) if soft else None, ptr(loss), ptr(dl), B, Cc, 1.0, None)
    torch.testing.assert_close(loss[0], loss_ref.detach(), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(dl, lr.grad, atol=1e-6, rtol=1e-4)
    dz = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    dW, db = torch.zeros(Cc, D, device=dev), torch.zeros(Cc, device=dev)
    call("tic_head_bwd", ptr(dl), ptr(z), ptr(W), ptr(dz), ptr(dW), ptr(db), B, Cc, D, None)
    torch.testing.assert_close(dz.float(), bfr(dl) @ bfr(W), atol=1e-3, rtol=0.02)
    torch.testing.assert_close(dW, bfr(dl).t() @ z.float(), atol=1e-4, rtol=1e-3)
    torch.testing.assert_close(db, bfr(dl).sum(0), atol=1e-5, rtol=1e-4)


def check_gemm_tn_group(env, M, shapes):
    """grouped dW: C_g += A_g^T B_g for several (N, K) sharing M, accumulate semantics"""
    import ctypes
    rnd, call, dev = env.rnd, env.call, env.dev
    As = [bf(rnd(M, n)) for n, k in shapes]
    Bs = [bf(rnd(M, k)) for n, k in shapes]
    Cs = [rnd(n, k) for n, k in shapes]
    refs = [c + a.float().t() @ b.float() for a, b, c in zip(As, Bs, Cs)]
    n = len(shapes)
    PA = (ctypes.c_void_p * n)(*[ptr(a) for a in As])
    PB = (ctypes.c_void_p * n)(*[ptr(b) for b in Bs])
    PC = (ctypes.c_void_p * n)(*[ptr(c) for c in Cs])
    NN = (ctypes.c_int * n)(*[s_[0] for s_ in shapes])
    KK = (ctypes.c_int * n)(*[s_[1] for s_ in shapes])
    call("tic_gemm_tn_group_bf16", n, PA, PB, PC, NN, KK, M, None)
    for c, r in zip(Cs, refs):
        torch.testing.assert_close(c, r, atol=2e-3 * max(1.0, (M / 200) ** 0.5), rtol=1e-3)
    # overwrite form (the first backward after the gradients were cleared): whatever C held -- NaNs here -- is dropped, on every route
    for c in Cs:
        c.fill_(float("nan"))
    call("tic_gemm_tn_group_bf16_ex", n, PA, PB, PC, NN, KK, M, 1, None)
    for a, b, c in zip(As, Bs, Cs):
        torch.testing.assert_close(c, a.float().t() @ b.float(), atol=2e-3 * max(1.0, (M / 200) ** 0.5), rtol=1e-3)


def aug_cases(H, W):
    from oracle import aug_oracle as ao
    return [
        ao.AugParams(0, 0, H, W, jitter=False),                                                     # plain Resize (val/test)
        ao.AugParams(3, 5, 30, 20, flip=True, order=(2, 0, 3, 1), brightness=1.15, contrast=0.85, saturation=1.1, hue=0.07, gray=False, erase=(4, 6, 10, 9)),
        ao.AugParams(10, 0, 17, 48, flip=False, order=(1, 3, 0, 2), brightness=0.8, contrast=1.2, saturation=0.8, hue=-0.1, gray=True),
        ao.AugParams(0, 8, 40, 33, flip=True, order=(3, 2, 1, 0), brightness=1.0, contrast=1.19, saturation=1.2, hue=0.0, erase=(0, 0, 31, 5)),
    ]


def aug_param_table(cases):
    """explicit AugParams -> the [B, 20] fp32 rows tic_augment takes"""
    P = torch.zeros(len(cases), 20)
    for b, c in enumerate(cases):
        P[b, 0:4] = torch.tensor([c.top, c.left, c.height, c.width], dtype=torch.float32)
        P[b, 4] = float(c.flip)
        P[b, 5:9] = torch.tensor(c.order, dtype=torch.float32)
        P[b, 9:13] = torch.tensor([c.brightness, c.contrast, c.saturation, c.hue])
        P[b, 13], P[b, 14] = float(c.jitter), float(c.gray)
        if c.erase is not None:
            P[b, 15] = 1.0
            P[b, 16:20] = torch.tensor(c.erase, dtype=torch.float32)
    return P


def _aug_close(out, ref, b):
    # hue is discontinuous at sector boundaries: allow a few outlier pixels there, everything else tight
    diff = (out - ref).abs()
    assert diff.median() < 1e-5 and (diff > 2e-3).float().mean() < 2e-3, (b, diff.max().item(), (diff > 2e-3).float().mean().item())


def check_augment(env, S=32, H=40, W=48):
    """HIP augmentation vs the float CPU restatement (oracle/aug_oracle.py) for explicit parameters, and -- at the fixture size --
    vs the committed vectors of tests/golden/aug_cases.npz (inputs + expected outputs: the oracle cannot drift unnoticed)"""
    from oracle import aug_oracle as ao
    import ctypes
    import os
    import numpy as np
    dev, call = env.dev, env.call
    g = torch.Generator().manual_seed(5)
    cases = aug_cases(H, W)
    B = len(cases)
    imgs = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    P = aug_param_table(cases)
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aug_cases.npz")
    gold = np.load(gold_path) if (S, H, W) == (32, 40, 48) else None
    if gold is not None:
        assert np.array_equal(gold["imgs"], imgs.numpy()) and np.array_equal(gold["params"], P.numpy())
    out = torch.empty(B, 3, S, S, device=dev)
    mean = (ctypes.c_float * 3)(*ao.IMAGENET_MEAN)
    std = (ctypes.c_float * 3)(*ao.IMAGENET_STD)
    imgs_d, P_d = imgs.to(dev), P.to(dev)
    call("tic_augment", ptr(imgs_d), B, H, W, ptr(P_d), ptr(out), S, mean, std, None)
    for b, c in enumerate(cases):
        ref = ao.augment_one(imgs[b], c, out=S)
        _aug_close(out[b].cpu(), ref, b)
        if gold is not None:
            torch.testing.assert_close(ref, torch.from_numpy(gold["out"][b]), atol=1e-6, rtol=1e-6)   # oracle == committed vector
            _aug_close(out[b].cpu(), torch.from_numpy(gold["out"][b]), b)                              # HIP == committed vector


def check_mix(env):
    from oracle import aug_oracle as ao
    dev, call, rnd = env.dev, env.call, env.rnd
    B, C, H, W, ncls = 5, 3, 16, 24, 7
    x = rnd(B, C, H, W)
    y = torch.randint(0, ncls, (B,), device=dev)
    out, soft = torch.empty_like(x), torch.empty(B, ncls, device=dev)
    call("tic_mix", ptr(x), ptr(out), B, C, H, W, 0, 0.3, 0, 0, 0, 0, None)
    call("tic_mix_labels", ptr(y), ptr(soft), B, ncls, 0.3, None)
    rx, ry_ = ao.mixup(x.cpu(), y.cpu(), ncls, 0.3)
    torch.testing.assert_close(out.cpu(), rx)
    torch.testing.assert_close(soft.cpu(), ry_)
    lam, cx, cy = 0.4, 17, 5
    x1, y1, x2, y2 = ao.cutmix_box(H, W, lam, cx, cy)
    call("tic_mix", ptr(x), ptr(out), B, C, H, W, 1, 0.0, x1, y1, x2, y2, None)
    rx, ry_ = ao.cutmix(x.cpu(), y.cpu(), ncls, lam, cx, cy)
    assert torch.equal(out.cpu(), rx)
    lam_adj = 1.0 - (x2 - x1) * (y2 - y1) / float(W * H)
    call("tic_mix_labels", ptr(y), ptr(soft), B, ncls, lam_adj, None)
    torch.testing.assert_close(soft.cpu(), ry_)
    # the committed vectors (tests/golden/aug_cases.npz): HIP on the fixture inputs == the fixture outputs
    import os
    import numpy as np
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aug_cases.npz"))
    gx, gy = torch.from_numpy(gold["mix_x"]).to(dev), torch.from_numpy(gold["mix_y"]).to(dev)
    gout, gsoft = torch.empty_like(gx), torch.empty(B, ncls, device=dev)
    call("tic_mix", ptr(gx), ptr(gout), B, C, H, W, 0, 0.3, 0, 0, 0, 0, None)
    call("tic_mix_labels", ptr(gy), ptr(gsoft), B, ncls, 0.3, None)
    torch.testing.assert_close(gout.cpu(), torch.from_numpy(gold["mixup_x"]))
    torch.testing.assert_close(gsoft.cpu(), torch.from_numpy(gold["mixup_y"]))
    bx1, by1, bx2, by2 = (int(v) for v in gold["cutmix_box"])
    assert (bx1, by1, bx2, by2) == (x1, y1, x2, y2)
    call("tic_mix", ptr(gx), ptr(gout), B, C, H, W, 1, 0.0, bx1, by1, bx2, by2, None)
    call("tic_mix_labels", ptr(gy), ptr(gsoft), B, ncls, lam_adj, None)
    assert torch.equal(gout.cpu(), torch.from_numpy(gold["cutmix_x"]))
    torch.testing.assert_close(gsoft.cpu(), torch.from_numpy(gold["cutmix_y"]))


def check_fused_bias_gradients(env):
    """column sums fused into producers: GEMM epilogue (BF16 / DGELU), LayerNorm backward, attention backward"""
    rnd, call, dev = env.rnd, env.call, env.dev
    # GEMM epilogues, both tile sizes
    for tile, (M, N, K) in ((128, (150, 192, 64)), (256, (300, 256, 128))):
        call("tic_set_option", b"gemm_tile", tile, ) if False else None
        A, B, u = bf(rnd(M, K, scale=0.3)), bf(rnd(N, K, scale=0.3)), bf(rnd(M, N))
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        cs = torch.ones(N, device=dev)
        env._call("tic_set_option", b"gemm_tile", tile)
        try:
            call("tic_gemm_nt_bf16_ex", ptr(A), ptr(B), M, N, K, 3, None, ptr(out), None, None, None, ptr(u), None, 0, ptr(cs), None)
            torch.testing.assert_close(cs, 1 + out.float().sum(0), atol=0.1, rtol=0.02)   # sums of the fp32 values vs sums of their bf16 roundings over M rows
            cs2 = torch.zeros(N, device=dev)
            bias = rnd(N)
            call("tic_gemm_nt_bf16_ex", ptr(A), ptr(B), M, N, K, 0, ptr(bias), ptr(out), None, None, None, None, None, 0, ptr(cs2), None)
            torch.testing.assert_close(cs2, out.float().sum(0), atol=0.3, rtol=0.02)
        finally:
            env._call("tic_set_option", b"gemm_tile", 0)
    # LayerNorm backward
    rows, D = 37, 256
    x, gamma = rnd(rows, D), 1 + rnd(D, scale=0.1)
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    call("tic_layernorm_fwd", ptr(x), D, ptr(gamma), ptr(gamma), ptr(y), ptr(mean), ptr(rstd), rows, D, 1e-12, None)
    dy, dres = bf(rnd(rows, D)), rnd(rows, D)
    dx, dxb = torch.empty(rows, D, device=dev), torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
    dg, db, cs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.full((D,), 2.0, device=dev)
    call("tic_layernorm_bwd_ex", ptr(dy), ptr(x), D, ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dxb), ptr(dg), ptr(db), ptr(cs), rows, D, None)
    torch.testing.assert_close(cs, 2 + dx.sum(0), atol=1e-4, rtol=1e-4)
    # attention backward
    B, H, N = 2, 2, 197
    Dm = H * 64
    qkv = bf(rnd(B * N, 3 * Dm))
    o = torch.empty(B * N, Dm, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B * H, N, device=dev)
    call("tic_attention_fwd", ptr(qkv), ptr(o), ptr(lse), B, H, N, 0.125, None)
    do = bf(rnd(B * N, Dm))
    dqkv = torch.empty_like(qkv)
    dbias = torch.zeros(3 * Dm, device=dev)
    call("tic_attention_bwd_ex", ptr(qkv), ptr(o), ptr(lse), ptr(do), ptr(dqkv), ptr(dbias), B, H, N, 0.125, None)
    torch.testing.assert_close(dbias, dqkv.float().sum(0), atol=0.05, rtol=0.02)
    # the atomics-free form: per-image sums in a caller-owned scratch, added up by a second kernel (+= into dbias)
    dbias2, dqkv2 = torch.full((3 * Dm,), 0.5, device=dev), torch.empty_like(qkv)
    scratch = torch.full((B, 3 * Dm), 9.0, device=dev)
    call("tic_attention_bwd_ws", ptr(qkv), ptr(o), ptr(lse), ptr(do), ptr(dqkv2), ptr(dbias2), ptr(scratch), 0, B, H, N, 0.125, None)
    assert torch.equal(dqkv2, dqkv)
    torch.testing.assert_close(dbias2, 0.5 + dbias, atol=1e-3, rtol=1e-4)
    # skip_v_bias: the v third is left alone; it equals the column sums of dO (rows of P sum to 1) up to the bf16 rounding of P
    dbias3 = torch.full((3 * Dm,), 0.25, device=dev)
    call("tic_attention_bwd_ws", ptr(qkv), ptr(o), ptr(lse), ptr(do), ptr(dqkv2), ptr(dbias3), ptr(scratch), 1, B, H, N, 0.125, None)
    torch.testing.assert_close(dbias3[:2 * Dm], 0.25 + dbias[:2 * Dm], atol=1e-3, rtol=1e-4)
    assert torch.equal(dbias3[2 * Dm:], torch.full((Dm,), 0.25, device=dev))
    torch.testing.assert_close(do.float().sum(0), dbias[2 * Dm:], atol=0.02 * float(do.float().abs().sum(0).max()) + 0.05, rtol=0.02)


def check_moe_ops(env, B=7, E=8, K=2, C=120):
    """gate / combine / loss kernels (csrc/moe.h) and their backward launches vs oracle/moe_oracle.py + autograd"""
    from oracle import moe_oracle as mo
    rnd, call, dev = env.rnd, env.call, env.dev
    logits, noise = rnd(B, E), rnd(B, E)
    if E >= 6:                           # a tie between the two largest scores of row 0: both are selected, order is the kernel's
        logits[0, 3] = logits[0, 5] = 5.0   # (lowest index first); rows 1.. have distinct values
        noise[0, 3] = noise[0, 5]
    gw = torch.empty(B, E, device=dev)
    idx = torch.empty(B, K, dtype=torch.int64, device=dev)
    tw = torch.empty(B, K, device=dev)
    call("tic_moe_gate", ptr(logits), ptr(noise), 0.01, ptr(gw), ptr(idx), ptr(tw), B, E, K, None)
    lr = logits.cpu().clone().requires_grad_(True)
    rw, ri = mo.gate(lr, noise.cpu(), K)
    rg = mo.scatter(rw, ri, E)
    assert torch.equal(idx.cpu()[1:], ri[1:]) and set(idx.cpu()[0].tolist()) == set(ri[0].tolist())
    torch.testing.assert_close(tw.cpu()[1:], rw.detach()[1:], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(gw.cpu(), rg.detach(), atol=1e-6, rtol=1e-5)
    dgw = rnd(B, E)
    rg.backward(dgw.cpu())
    dl = torch.empty(B, E, device=dev)
    call("tic_moe_gate_bwd", ptr(gw), ptr(dgw), ptr(dl), B, E, None)
    torch.testing.assert_close(dl.cpu(), lr.grad, atol=1e-6, rtol=1e-4)
    # eval mode: no noise pointer
    call("tic_moe_gate", ptr(logits), None, 0.01, ptr(gw), ptr(idx), ptr(tw), B, E, K, None)
    torch.testing.assert_close(gw.cpu()[1:], mo.scatter(*mo.gate(logits.cpu(), None, K), E)[1:], atol=1e-6, rtol=1e-5)
    # combine
    X = rnd(E, B, C)
    w = torch.softmax(rnd(B, E), -1) * (rnd(B, E) > 0)
    out = torch.empty(B, C, device=dev)
    call("tic_moe_combine", ptr(X), ptr(w), ptr(out), B, E, C, None)
    xr, wr = X.cpu().clone().requires_grad_(True), w.cpu().clone().requires_grad_(True)
    ro = mo.combine(xr.permute(1, 0, 2), wr)
    torch.testing.assert_close(out.cpu(), ro.detach(), atol=1e-5, rtol=1e-5)
    dout = rnd(B, C)
    ro.backward(dout.cpu())
    dX, dw = torch.empty_like(X), torch.empty_like(w)
    call("tic_moe_combine_bwd", ptr(X), ptr(w), ptr(dout), ptr(dX), ptr(dw), B, E, C, None)
    torch.testing.assert_close(dX.cpu(), xr.grad, atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(dw.cpu(), wr.grad, atol=1e-5 * max(1.0, C / 100), rtol=1e-4)   # fp32 dot products of C terms
    # loss: one-hot and soft targets
    for soft in (False, True):
        z = rnd(B, C)
        t = torch.softmax(rnd(B, C), -1) if soft else torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=env.gen), C).float().to(dev)
        g = torch.softmax(rnd(B, E), -1)
        loss3, dz, dg = torch.empty(3, device=dev), torch.empty(B, C, device=dev), torch.empty(B, E, device=dev)
        call("tic_moe_loss", ptr(z), ptr(t), ptr(g), ptr(loss3), ptr(dz), ptr(dg), B, C, E, 0.1, 1.0, 0.5, None)
        zr, gr = z.cpu().clone().requires_grad_(True), g.cpu().clone().requires_grad_(True)
        ref = mo.total_loss(zr, t.cpu(), gr)
        ref.backward()
        torch.testing.assert_close(loss3.cpu()[0], ref.detach(), atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(loss3.cpu()[1], mo.symmetric_cross_entropy(z.cpu(), t.cpu()), atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(loss3.cpu()[2], mo.load_balance_loss(g.cpu()), atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(dz.cpu(), zr.grad, atol=1e-6, rtol=1e-4)
        torch.testing.assert_close(dg.cpu(), gr.grad, atol=1e-7, rtol=1e-4)


def check_splitk_nt(env, M, N, K, split, tile=128):
    """the split-K form of the 128x128 NT kernel (2 or 4 workgroups per tile, fp32 partial accumulators handed over through a
    caller-owned scratch, flags that count launches) against the unsplit kernel and fp32 math: same single rounding to bf16, only the
    fp32 summation order differs"""
    from touhouimageclassification_amd import _capi
    rnd, call, dev = env.rnd, env.call, env.dev
    A, B, bias = bf(rnd(M, K, scale=0.3)), bf(rnd(N, K, scale=0.3)), rnd(N, scale=0.1)
    resid = rnd(M, N)
    scratch = torch.zeros(_capi.NT_SCRATCH_BYTES, dtype=torch.uint8, device=dev)   # flags (last 4 KiB) zeroed once

    def run(epi, sp):
        env._call("tic_set_option", b"gemm_split", sp)
        o1 = torch.full((M, N), 3.0, device=dev).to(torch.bfloat16)
        of = torch.full((M, N), 7.0, device=dev)
        cs = torch.zeros(N, device=dev)
        call("tic_gemm_nt_bf16_ex", ptr(A), ptr(B), M, N, K, epi, ptr(bias), ptr(o1) if epi == 0 else None, None, ptr(of) if epi == 2 else None,
             ptr(resid) if epi == 2 else None, None, None, 0, ptr(cs) if epi == 0 else None, None)
        return (o1 if epi == 0 else of), cs

    env._call("tic_set_option", b"gemm_tile", tile)   # 256: gemm256.h's split-K form; 128: gemm.h's (any epilogue; 0 and 2 are checked)
    env._call("tic_gemm_nt_scratch", ptr(scratch), scratch.numel())
    try:
        ref32 = A.float() @ B.float().t() + bias
        for epi in (0, 2):
            base, bcs = run(epi, 0)
            for rep in range(3):   # repeated launches: the flags count launches, nothing is cleared in between
                got, gcs = run(epi, split)
                if epi == 0:
                    d = (got.float() - base.float()).abs()
                    assert float((d > 0).float().mean()) < 0.02 and float(d.max()) <= 0.0625 * max(1.0, float(base.float().abs().max()) / 8), (epi, rep)
                    torch.testing.assert_close(got.float(), ref32, atol=0.06, rtol=0.02)
                    torch.testing.assert_close(gcs, bcs, atol=2e-2 * max(1.0, float(bcs.abs().max()) / 10), rtol=1e-3)
                else:
                    torch.testing.assert_close(got, resid + bfr(ref32), atol=0.06, rtol=0.02)
                    assert float(((got - base).abs() > 0).float().mean()) < 0.02
    finally:
        env._call("tic_gemm_nt_scratch", None, 0)
        env._call("tic_set_option", b"gemm_tile", 0)
        env._call("tic_set_option", b"gemm_split", -1)


def check_nt_ring_matches(env, M, N, K, split=0):
    """the 4-stage ring of the 128x128 NT kernel (gemm.h NST = 4: at most 256 workgroups, each alone on its CU) forms the same products in
    the same order as the 2-stage loop: every epilogue's outputs agree BIT FOR BIT.  Test / measurement builds switch the form with the
    `nt_deep` knob (also under the split-K hand-off); the product library has no such knob -- there the SAME rows are computed once as an
    M-row product (<= 256 tiles: ring) and once as the first M rows of a taller product (> 256 tiles: 2-stage loop)."""
    from touhouimageclassification_amd import _capi
    rnd, call, dev = env.rnd, env.call, env.dev
    tiles_n = (N + 127) // 128
    try:
        env._call("tic_set_option", b"nt_deep", -1)
        knob = True
    except _capi.TicError:
        knob = False
        assert split == 0, "product library: the split-K ring is covered by check_splitk_nt (auto route) against fp32 math"
    assert ((M + 127) // 128) * tiles_n * max(1, split) <= 256
    Mb = M if knob else 128 * (256 // tiles_n + 1)   # taller product: more than 256 tiles
    A, B, bias = bf(rnd(Mb, K, scale=0.3)), bf(rnd(N, K, scale=0.3)), rnd(N, scale=0.1)
    resid, aux = rnd(Mb, N), bf(rnd(Mb, N))
    scratch = torch.zeros(_capi.NT_SCRATCH_BYTES, dtype=torch.uint8, device=dev)

    def run(epi, rows, deep):
        if knob:
            env._call("tic_set_option", b"nt_deep", deep)
        o1 = torch.full((rows, N), 3.0, device=dev).to(torch.bfloat16)
        o2 = torch.full((rows, N), 5.0, device=dev).to(torch.bfloat16)
        of = torch.full((rows, N), 7.0, device=dev)
        cs = torch.zeros(N, device=dev)
        has_cs = epi in (0, 3, 6)
        call("tic_gemm_nt_bf16_ex", ptr(A), ptr(B), rows, N, K, epi, ptr(bias) if epi in (0, 1, 2, 5) else None, ptr(o1), ptr(o2), ptr(of),
             ptr(resid) if epi == 2 else None, ptr(aux) if epi in (3, 6, 7) else None, None, 0, ptr(cs) if has_cs else None, None)
        return o1[:M], o2[:M], of[:M], cs

    env._call("tic_set_option", b"gemm_tile", 128)
    env._call("tic_set_option", b"gemm_split", split)
    if split:
        env._call("tic_gemm_nt_scratch", ptr(scratch), scratch.numel())
    try:
        for epi in (0, 1, 2, 3, 5, 6, 7):
            a, b = run(epi, Mb, 0), run(epi, M, 1)
            for x, y in zip(a[:3], b[:3]):
                assert torch.equal(x, y), (epi, M, N, K, split)
            if knob:   # the column sums are fp32 atomics of per-wave sums: same addends, the order of the adds is the hardware's
                torch.testing.assert_close(a[3], b[3], atol=1e-3 * max(1.0, float(a[3].abs().max())), rtol=1e-5)
        ref = A[:M].float() @ B.float().t() + bias
        first = run(0, M, 1)[0]
        torch.testing.assert_close(first.float(), ref, atol=0.06, rtol=0.02)
        # the ring's hazards (a stage refilled while a slow wave still reads it; a stage read before every wave's pieces have landed) would
        # show as launch-to-launch differences: 12 more launches, each bit-identical to the first
        for _ in range(12):
            assert torch.equal(run(0, M, 1)[0], first), (M, N, K, split)
    finally:
        env._call("tic_gemm_nt_scratch", None, 0)
        env._call("tic_set_option", b"gemm_tile", 0)
        env._call("tic_set_option", b"gemm_split", -1)
        if knob:
            env._call("tic_set_option", b"nt_deep", -1)


def check_refresh_weights(model):
    """after a forward, the bf16 operand copies are W (flat) and, per layer, W^T of the four Linear weights: exactly bf16(W) transposed
    (tic_vit_refresh_weights: one grouped cast+transpose launch over all layers)"""
    eng = model._engine
    lay = eng.lay
    D, F, L = eng.D, eng.F, eng.L
    flat = eng.params.detach()
    torch.testing.assert_close(eng.w16.float(), flat.to(torch.bfloat16).float(), atol=0, rtol=0)
    for l in range(L):
        for off, toff, R, C in ((lay.wqkv, lay.t_wqkv, 3 * D, D), (lay.wo, lay.t_wo, D, D), (lay.w1, lay.t_w1, F, D), (lay.w2, lay.t_w2, D, F)):
            w = flat[lay.layer0 + l * lay.layer_stride + off:][: R * C].view(R, C)
            wt = eng.wT16[l * lay.t_layer_stride + toff:][: R * C].view(C, R)
            assert torch.equal(wt, w.to(torch.bfloat16).t().contiguous()), (l, R, C)


def check_gemm_tn_slab(env, M, N, K, slab_mb=64):
    """few-tile weight gradient through a registered slab scratch (row parts STORE their partial tiles, one more launch adds them up)
    against the stream-K atomics form and fp32 math; accumulate semantics (C +=) in both"""
    rnd, call, dev = env.rnd, env.call, env.dev
    A, B = bf(rnd(M, N)), bf(rnd(M, K))
    C0 = rnd(N, K)
    ref = C0 + A.float().t() @ B.float()
    Ca, Cs = C0.clone(), C0.clone()
    call("tic_gemm_tn_bf16", ptr(A), ptr(B), ptr(Ca), M, N, K, None)
    scratch = torch.full((slab_mb << 20,), 0x7f, dtype=torch.uint8, device=dev)   # garbage on purpose: every part must overwrite what the reduce reads
    env._call("tic_gemm_tn_scratch", ptr(scratch), scratch.numel())
    try:
        call("tic_gemm_tn_bf16", ptr(A), ptr(B), ptr(Cs), M, N, K, None)
        call("tic_gemm_tn_bf16", ptr(A), ptr(B), ptr(Cs), M, N, K, None)   # accumulates
    finally:
        env._call("tic_gemm_tn_scratch", None, 0)
    tol = dict(atol=2e-3 * max(1.0, (M / 200) ** 0.5), rtol=1e-3)
    torch.testing.assert_close(Ca, ref, **tol)
    torch.testing.assert_close(Cs, ref + (ref - C0), atol=2 * tol["atol"], rtol=1e-3)


def check_splitk_timeout_is_reported(env, M=200, N=256, K=512):
    """a split-K consumer whose producer never publishes its flag must not sum an unwritten slab and report success.
    Test builds can withhold part 0's flag ("nt_fault"): the consumer's bounded poll runs out, it stores 0xDEADxxxx into the host-mapped
    error word, and every later call fails with TIC_ELAUNCH until a scratch is registered again (the caller re-zeroes the flags)."""
    import pytest
    from touhouimageclassification_amd import _capi
    rnd, dev = env.rnd, env.dev
    A, B, bias = bf(rnd(M, K, scale=0.3)), bf(rnd(N, K, scale=0.3)), rnd(N, scale=0.1)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    scratch = torch.zeros(_capi.NT_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    args = (ptr(A), ptr(B), M, N, K, 0, ptr(bias), ptr(out), None, None, None, None, None, 0, None, None)
    env._call("tic_set_option", b"gemm_tile", 128)
    env._call("tic_set_option", b"gemm_split", 2)
    env._call("tic_gemm_nt_scratch", ptr(scratch), scratch.numel())
    try:
        env.call("tic_gemm_nt_bf16_ex", *args)                       # healthy hand-off
        ref = out.clone()
        env._call("tic_set_option", b"nt_fault", 1)
        with pytest.raises(Exception, match="timed out"):            # the faulty launch itself (simulator: synchronous) or the next call
            env.call("tic_gemm_nt_bf16_ex", *args)
            env.call("tic_gemm_nt_bf16_ex", *args)
        env._call("tic_set_option", b"nt_fault", 0)
        b16 = torch.empty(N, dtype=torch.bfloat16, device=dev)
        with pytest.raises(Exception, match="timed out"):            # sticky: unrelated calls fail too
            env.call("tic_cast_bf16", ptr(bias), ptr(b16), N, None)
        scratch.zero_()
        env._call("tic_gemm_nt_scratch", ptr(scratch), scratch.numel())   # re-registration clears the error
        env.call("tic_gemm_nt_bf16_ex", *args)
        assert torch.equal(out, ref)
    finally:
        env._call("tic_set_option", b"nt_fault", 0)
        env._call("tic_gemm_nt_scratch", None, 0)
        env._call("tic_set_option", b"gemm_tile", 0)
        env._call("tic_set_option", b"gemm_split", -1)
