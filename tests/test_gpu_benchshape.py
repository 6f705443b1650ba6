"""Parity AT THE BENCHMARK SHAPE: bench.py runs ViT-L with 332 images per GPU, i.e. token matrices of
M = 332 x 197 = 65 404 rows -- 256 row tiles of 256, the phase-aligned stream-K split of the grouped dW launch, 2 M-thread attention
grids and 0.5 GB [M, F] activations.  Every kernel family is checked here at exactly that M against fp32 torch math on the same
device (full tensors where a reduction over M is what is being tested, sampled rows / images for the row-independent ones), and
one whole ViT-L step at B = 332 is tied to the CPU oracle (logits of a 4-image slice) and, through linearity of the gradient in
the batch, to the M = 16 351 regime the other parity tests cover."""
import ctypes

import pytest
import torch

from oracle import vit_oracle as vo
from tests import headroom as hr
from tests import kernel_checks as kc
from tests.simlib import bf, bfr, ptr

pytestmark = pytest.mark.gpu
B_BENCH, N_TOK = 332, 197
M = B_BENCH * N_TOK          # 65 404
D, F = 1024, 4096


@pytest.fixture()
def env():
    from touhouimageclassification_amd._lib import call, current_stream
    return kc.Env("cuda", call, stream=current_stream, seed=65404)


def _rows(gen):
    """sampled rows: the first and last row tiles (ragged edge: 65 404 = 255 x 256 + 124), and 1 500 rows anywhere"""
    r = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M), torch.randint(0, M, (1500,), generator=gen)])
    return r.unique().cuda()


@pytest.mark.parametrize("N,K,epi", [(3 * D, D, "bf16"), (D, D, "resid"), (D, F, "resid"), (F, D, "gelu_dg"), (F, D, "mulaux"), (D, F, "bf16_colsum")])
def test_nt_gemm_epilogues_at_bench_m(env, N, K, epi):
    rnd, call = env.rnd, env.call
    A, W, bias = bf(rnd(M, K, scale=0.5)), bf(rnd(N, K, scale=0.05)), rnd(N, scale=0.1)
    rows = _rows(env.gen)
    acc = A[rows].float() @ W.float().t()          # fp32 reference of the sampled rows
    if epi == "bf16":
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        call("tic_gemm_nt_bf16", ptr(A), ptr(W), M, N, K, 0, ptr(bias), ptr(out), None, None, None, None, None, 0, None)
        torch.testing.assert_close(out[rows].float(), acc + bias, atol=0.03, rtol=0.02)
    elif epi == "bf16_colsum":      # dX through fc1 with the bias-gradient column sums fused: a reduction over ALL 65 404 rows
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        cs = torch.zeros(N, device="cuda")
        call("tic_gemm_nt_bf16_ex", ptr(A), ptr(W), M, N, K, 0, None, ptr(out), None, None, None, None, None, 0, ptr(cs), None)
        torch.testing.assert_close(out[rows].float(), acc, atol=0.03, rtol=0.02)
        ref = out.float().sum(0)
        torch.testing.assert_close(cs, ref, atol=2e-3 * float(out.float().abs().sum(0).max()), rtol=1e-3)
    elif epi == "resid":
        resid, out = rnd(M, N), torch.empty(M, N, device="cuda")
        call("tic_gemm_nt_bf16", ptr(A), ptr(W), M, N, K, 2, ptr(bias), None, None, ptr(out), ptr(resid), None, None, 0, None)
        torch.testing.assert_close(out[rows], resid[rows] + bfr(acc + bias), atol=0.03, rtol=0.02)
    elif epi == "gelu_dg":
        dg, g = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        call("tic_gemm_nt_bf16", ptr(A), ptr(W), M, N, K, 5, ptr(bias), ptr(dg), ptr(g), None, None, None, None, 0, None)
        u = bfr(acc + bias).double().requires_grad_(True)
        gu = torch.nn.functional.gelu(u)
        gu.sum().backward()
        torch.testing.assert_close(g[rows].float(), gu.detach().float(), atol=0.02, rtol=0.02)
        torch.testing.assert_close(dg[rows].float(), u.grad.float(), atol=0.02, rtol=0.02)
    else:   # mulaux: d(fc1 out) = bf16(bf16(acc) * saved gelu'), with fused column sums
        aux = bf(rnd(M, N, scale=0.5))
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        cs = torch.zeros(N, device="cuda")
        call("tic_gemm_nt_bf16_ex", ptr(A), ptr(W), M, N, K, 6, None, ptr(out), None, None, None, ptr(aux), None, 0, ptr(cs), None)
        torch.testing.assert_close(out[rows].float(), bfr(bfr(acc) * aux[rows].float()), atol=5e-3, rtol=3.2e-2)   # 2 bf16 ulps by construction (kernel_checks.py), band 4
        torch.testing.assert_close(cs, out.float().sum(0), atol=2e-3 * float(out.float().abs().sum(0).max()), rtol=1e-3)


def test_grouped_dw_stream_k_at_bench_m(env):
    """the step's dominant kernel at its bench shape: dW_g += dY_g^T X_g for the four Linear layers of a ViT-L block, ONE launch,
    phase-aligned stream-K with fp32 atomics -- against full fp32 matmuls over all 65 404 rows"""
    kc.check_gemm_tn_group(env, M, [(D, F), (F, D), (D, D), (3 * D, D)])
    kc.check_gemm_tn_group(env, M, [(D, F), (F, D), (D, D), (3 * D, D)])   # a second launch into fresh accumulators: same answer


def test_layernorm_at_bench_m(env):
    """forward + backward (residual add, fp32 + bf16 gradient, dgamma / dbeta and the fused column sums) over all 65 404 rows"""
    rnd, call = env.rnd, env.call
    x, gamma, beta = rnd(M, D) * 2 + 0.5, 1 + rnd(D, scale=0.1), rnd(D, scale=0.1)
    y = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    call("tic_layernorm_fwd", ptr(x), D, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, D, 1e-12, None)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-12)
    torch.testing.assert_close(y.float(), ref.detach(), atol=0.02, rtol=0.01)
    dy, dres = bf(rnd(M, D)), rnd(M, D)
    ref.backward(dy.float())
    dx, dxb = torch.empty(M, D, device="cuda"), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    dg, db, cs = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    call("tic_layernorm_bwd_ex", ptr(dy), ptr(x), D, ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dxb), ptr(dg), ptr(db), ptr(cs), M, D, None)
    torch.testing.assert_close(dx, dres + xr.grad, atol=1e-4, rtol=1e-4)
    assert torch.equal(dxb.float(), bfr(dx))
    # sums of 65 404 terms of magnitude ~1 in fp32, atomics order: absolute error ~1e-2 on sums of magnitude ~250
    torch.testing.assert_close(dg, gr.grad, atol=5e-2, rtol=2e-3)
    torch.testing.assert_close(db, br.grad, atol=5e-2, rtol=2e-3)
    torch.testing.assert_close(cs, dx.double().sum(0).float(), atol=5e-2, rtol=2e-3)


def test_attention_at_bench_batch(env):
    """B = 332 x 16 heads = 5 312 workgroups; images are independent, so four of them are re-computed in fp32"""
    rnd, call = env.rnd, env.call
    H = 16
    qkv = bf(rnd(M, 3 * D))
    o = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B_BENCH * H, N_TOK, device="cuda")
    call("tic_attention_fwd", ptr(qkv), ptr(o), ptr(lse), B_BENCH, H, N_TOK, 0.125, None)
    do = bf(rnd(M, D))
    dqkv = torch.empty_like(qkv)
    dbias = torch.zeros(3 * D, device="cuda")
    call("tic_attention_bwd_ex", ptr(qkv), ptr(o), ptr(lse), ptr(do), ptr(dqkv), ptr(dbias), B_BENCH, H, N_TOK, 0.125, None)
    for img in (0, 1, 165, 331):
        sl = slice(img * N_TOK, (img + 1) * N_TOK)
        qr = qkv[sl].float().requires_grad_(True)
        o_ref, lse_ref = kc._attn_ref(qr, 1, H, N_TOK)
        torch.testing.assert_close(o[sl].float(), o_ref.detach(), atol=0.02, rtol=0.02)
        torch.testing.assert_close(lse.view(B_BENCH, H, N_TOK)[img], lse_ref.detach()[0], atol=2e-3, rtol=1e-3)
        o_ref.backward(do[sl].float())
        torch.testing.assert_close(dqkv[sl].float(), qr.grad, atol=0.03, rtol=0.05)
    # the fused bias-gradient column sums reduce over every image
    torch.testing.assert_close(dbias, dqkv.float().sum(0), atol=2e-3 * float(dqkv.float().abs().sum(0).max()), rtol=2e-3)


def test_vit_large_step_at_bench_batch_ties_to_oracle_and_to_small_batches():
    """One full ViT-L/16 C=120 training step at B = 332 (the bench configuration):
      (1) the logits of a 4-image slice equal the fp32 CPU oracle's on those 4 images (images are independent);
      (2) the gradient of the 332-image batch equals the mean of the gradients of its four 83-image quarters (M = 16 351: the
          regime the golden / operator tests cover) -- linearity of the gradient in the batch, a size-independent property;
      (3) the fused step (forward, CE, backward, AdamW) moves every parameter tensor."""
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    from touhouimageclassification_amd import ops
    dev = torch.device("cuda")
    C = 120
    spec = vo.ViTSpec(**vo.VIT_LARGE, num_labels=C)
    params = vo.randomize_small_params(vo.init_params(spec, seed=20), seed=21)
    model = ViT(C, pretrained=False, model_name="google/vit-large-patch16-224")
    model.load_state_dict(params)
    model.to(dev)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B_BENCH, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (B_BENCH,), generator=g)
    xd, yd = x.to(dev), y.to(dev)
    pick = [0, 1, 330, 331]
    e = model._engine
    logits = e.forward(xd)
    with torch.no_grad():
        ref = vo.forward(params, x[pick], spec)
    torch.testing.assert_close(logits[pick].cpu(), ref, atol=2e-2 * max(1.0, float(ref.abs().max())), rtol=2e-2)   # bf16 GEMM I/O: the error scales with the logits
    loss, dl = ops.softmax_xent(logits, yd)
    hr.le("test_gpu_benchshape.py:155", abs(float(loss) - float(vo.cross_entropy(logits.cpu(), y))), 1e-4)
    e.grads.zero_()
    e.backward(dl)
    big = e.grads.clone()
    assert torch.isfinite(big).all()
    acc = torch.zeros_like(big)
    q = B_BENCH // 4
    for i in range(4):
        lg = e.forward(xd[i * q:(i + 1) * q].contiguous())
        _, dli = ops.softmax_xent(lg, yd[i * q:(i + 1) * q].contiguous())
        e.grads.zero_()
        e.backward(dli)
        acc += e.grads
    acc /= 4
    for name, a, b in e.buckets():   # per DP bucket (head, every layer, embeddings): relative L2 error of the gradient
        rel = ((big[a:b] - acc[a:b]).norm() / acc[a:b].norm()).item()
        hr.le("test_gpu_benchshape.py:171", rel, 2e-2, ctx=(name, rel))
    opt = FusedAdamW(model, lr=1e-5, weight_decay=0.01)
    before = e.params.clone()
    fused_train_step(model, opt, xd, yd, None)
    moved = (e.params - before).abs()
    assert float(moved.max()) <= 1.3e-5 and all(float(moved[a:b].max()) > 0 for _, a, b in e.buckets())


# ---- BASELINE config 2: ViT-Base/16, global batch 256 on one GPU (M = 256 x 197 = 50 432 rows) ---------------------------------
B_CFG2 = 256
M2 = B_CFG2 * N_TOK          # 50 432
DB, FB = 768, 3072


def test_vit_base_grouped_dw_equal_parts_split_at_config2_m(env):
    """ViT-B's four weight gradients are 108 tiles of 256 x 256: not a multiple of the 8 XCDs, so the launch takes the equal-parts
    split (`tn_parts`, two row halves per tile, 216 workgroups) instead of the phase-aligned stream-K one.  Checked at config 2's
    M = 50 432 against full fp32 matmuls over all rows, default routing and every forced form of the split."""
    from touhouimageclassification_amd._lib import call
    shapes = [(DB, FB), (FB, DB), (DB, DB), (3 * DB, DB)]
    kc.check_gemm_tn_group(env, M2, shapes)          # what the step launches
    try:
        for parts in (2, 3, 0):                      # forced: 2 / 3 equal row parts per tile, the flat stream-K split
            call("tic_set_option", b"tn_parts", parts)
            kc.check_gemm_tn_group(env, M2, shapes)
    finally:
        call("tic_set_option", b"tn_parts", -1)


@pytest.mark.parametrize("N,K,epi", [(3 * DB, DB, "bf16"), (DB, FB, "resid"), (FB, DB, "gelu_dg")])
def test_vit_base_nt_gemms_at_config2_m(env, N, K, epi):
    """N = 768 = 3 column tiles x 197 row tiles = 591 tiles (2.3 rounds of the 256 CUs), K = 768 = 12 K tiles: the shapes only ViT-B has"""
    rnd, call = env.rnd, env.call
    A, W, bias = bf(rnd(M2, K, scale=0.5)), bf(rnd(N, K, scale=0.05)), rnd(N, scale=0.1)
    r = torch.cat([torch.arange(0, 300), torch.arange(M2 - 300, M2), torch.randint(0, M2, (1500,), generator=env.gen)]).unique().cuda()
    acc = A[r].float() @ W.float().t()
    if epi == "bf16":
        out = torch.empty(M2, N, dtype=torch.bfloat16, device="cuda")
        call("tic_gemm_nt_bf16", ptr(A), ptr(W), M2, N, K, 0, ptr(bias), ptr(out), None, None, None, None, None, 0, None)
        torch.testing.assert_close(out[r].float(), acc + bias, atol=0.03, rtol=0.02)
    elif epi == "resid":
        resid, out = rnd(M2, N), torch.empty(M2, N, device="cuda")
        call("tic_gemm_nt_bf16", ptr(A), ptr(W), M2, N, K, 2, ptr(bias), None, None, ptr(out), ptr(resid), None, None, 0, None)
        torch.testing.assert_close(out[r], resid[r] + bfr(acc + bias), atol=0.03, rtol=0.02)
    else:
        dg, g = torch.empty(M2, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M2, N, dtype=torch.bfloat16, device="cuda")
        call("tic_gemm_nt_bf16", ptr(A), ptr(W), M2, N, K, 5, ptr(bias), ptr(dg), ptr(g), None, None, None, None, 0, None)
        u = bfr(acc + bias).double().requires_grad_(True)
        gu = torch.nn.functional.gelu(u)
        gu.sum().backward()
        torch.testing.assert_close(g[r].float(), gu.detach().float(), atol=0.02, rtol=0.02)
        torch.testing.assert_close(dg[r].float(), u.grad.float(), atol=0.02, rtol=0.02)


def test_vit_base_step_at_config2_batch_ties_to_oracle_and_to_small_batches():
    """One ViT-B/16 C=10 training step at B = 256 (BASELINE config 2): the logits of a 4-image slice against the fp32 CPU oracle on
    those images, the loss against the oracle's CE of the GPU logits, the 256-image gradient against the mean of its four 64-image
    quarters per DP bucket (M = 12 608, the regime of the operator / golden tests), and the fused step moves every bucket."""
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    from touhouimageclassification_amd import ops
    dev = torch.device("cuda")
    C = 10
    spec = vo.ViTSpec(**vo.VIT_BASE, num_labels=C)
    params = vo.randomize_small_params(vo.init_params(spec, seed=30), seed=31)
    model = ViT(C, pretrained=False, model_name="google/vit-base-patch16-224")
    model.load_state_dict(params)
    model.to(dev)
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(B_CFG2, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (B_CFG2,), generator=g)
    xd, yd = x.to(dev), y.to(dev)
    pick = [0, 1, 254, 255]
    e = model._engine
    logits = e.forward(xd)
    with torch.no_grad():
        ref = vo.forward(params, x[pick], spec)
    torch.testing.assert_close(logits[pick].cpu(), ref, atol=2e-2 * max(1.0, float(ref.abs().max())), rtol=2e-2)   # bf16 GEMM I/O: the error scales with the logits
    margin = ref.sort(-1, descending=True).values
    decided = (margin[:, 0] - margin[:, 1]) > 4e-2
    assert torch.equal(logits[pick].cpu().argmax(-1)[decided], ref.argmax(-1)[decided])
    loss, dl = ops.softmax_xent(logits, yd)
    hr.le("loss vs oracle CE of the same logits", abs(float(loss) - float(vo.cross_entropy(logits.cpu(), y))), 1e-4)
    e.grads.zero_()
    e.backward(dl)
    big = e.grads.clone()
    assert torch.isfinite(big).all()
    acc = torch.zeros_like(big)
    q = B_CFG2 // 4
    for i in range(4):
        lg = e.forward(xd[i * q:(i + 1) * q].contiguous())
        _, dli = ops.softmax_xent(lg, yd[i * q:(i + 1) * q].contiguous())
        e.grads.zero_()
        e.backward(dli)
        acc += e.grads
    acc /= 4
    for name, a, b in e.buckets():
        hr.le(f"bucket {name}: 256-image gradient vs mean of quarters", ((big[a:b] - acc[a:b]).norm() / acc[a:b].norm()).item(), 2e-2)
    opt = FusedAdamW(model, lr=1e-5, weight_decay=0.01)
    before = e.params.clone()
    fused_train_step(model, opt, xd, yd, None)
    moved = (e.params - before).abs()
    assert float(moved.max()) <= 1.3e-5 and all(float(moved[a:b].max()) > 0 for _, a, b in e.buckets())


def test_resnet50_step_at_config4_batch_ties_to_the_oracle():
    """BASELINE config 4 at its full size -- ResNet-50 @224, 256 images (M = 802 816 rows in the first stage: the implicit stem, the 3x3
    implicit GEMMs, the parity-class stride-2 input gradients, the 256x256 and slab routes of the 1x1 weight gradients, BatchNorm
    reductions over 512 row splits) -- one training step against the fp32 oracle run on the same device.  Bands as the 32-image golden
    case of the same model (tests/test_gpu_resnet.py): what they bound is bf16 storage noise through 50 layers, not the batch size."""
    from tests import resnet_checks as rc
    rc.check_against_live_oracle("resnet50_b256", "resnet50", 120, 256, 224, torch.device("cuda"),
                                 dict(logits=4.5e-2, loss=1e-2, gnorm=0.18, fc=0.05, cos=0.69, stats=1.5e-2))
