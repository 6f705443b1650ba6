"""ResNet-18 (64x64, B=4) through the simulator backend vs the reference-generated golden file, plus kernel-level
checks of the conv-path ops against torch."""
import pytest
import torch

from tests import resnet_checks as rc
from tests.simlib import SimBackend, bf, bfr, call, ptr


def test_conv_ops_against_torch():
    torch.manual_seed(0)
    B, H, W, Ci, Co, k, s, p = 2, 9, 10, 16, 24, 3, 2, 1
    x = bf(torch.randn(B, H, W, Ci))
    w = torch.randn(Co, Ci, k, k) * 0.2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    Kp = (k * k * Ci + 63) // 64 * 64
    col = torch.empty(B * Ho * Wo, Kp, dtype=torch.bfloat16)
    call("tic_im2col_bf16", ptr(x), ptr(col), B, H, W, Ci, k, k, s, p, None)
    wp, wpT = torch.empty(Co, Kp, dtype=torch.bfloat16), torch.empty(Kp, Co, dtype=torch.bfloat16)
    call("tic_conv_weight_pack", ptr(w), ptr(wp), Co, Ci, k, k, 0, None)
    call("tic_conv_weight_pack", ptr(w), ptr(wpT), Co, Ci, k, k, 1, None)
    assert torch.equal(wp.t().contiguous(), wpT)
    y = col.float() @ wp.float().t()
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), bfr(w), None, stride=s, padding=p).permute(0, 2, 3, 1).reshape(-1, Co)
    torch.testing.assert_close(y, ref, atol=1e-3, rtol=1e-3)
    # the stem's shape class (Ci = 3, 7x7 / 2, pad 3): 8 consecutive k per thread, K = 147 padded to 192 with zeros
    Hs, Ws = 13, 12
    xs = bf(torch.randn(B, Hs, Ws, 3))
    ws = torch.randn(8, 3, 7, 7) * 0.2
    Hso, Wso = (Hs + 6 - 7) // 2 + 1, (Ws + 6 - 7) // 2 + 1
    cols = torch.full((B * Hso * Wso, 192), 7.0).to(torch.bfloat16)
    call("tic_im2col_bf16", ptr(xs), ptr(cols), B, Hs, Ws, 3, 7, 7, 2, 3, None)
    unf = torch.nn.functional.unfold(xs.float().permute(0, 3, 1, 2), 7, padding=3, stride=2)           # [B, 3*49, L]
    unf = unf.view(B, 3, 49, Hso * Wso).permute(0, 3, 2, 1).reshape(B * Hso * Wso, 147)                  # tap-major
    assert torch.equal(cols[:, :147].float(), unf) and float(cols[:, 147:].abs().max()) == 0.0
    # col2im = adjoint of im2col
    dcol = bf(torch.randn(B * Ho * Wo, Kp))
    dx = torch.empty(B * H * W, Ci, dtype=torch.bfloat16)
    call("tic_col2im_bf16", ptr(dcol), ptr(dx), B, H, W, Ci, k, k, s, p, 0, None)
    xr = x.float().requires_grad_(True)
    colr = torch.nn.functional.unfold(xr.permute(0, 3, 1, 2), k, padding=p, stride=s)          # [B, Ci*k*k, L] (c-major)
    colr = colr.view(B, Ci, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, k * k * Ci)  # -> tap-major
    (colr * dcol.float()[:, :k * k * Ci]).sum().backward()
    torch.testing.assert_close(dx.float().view(B, H, W, Ci), xr.grad, atol=0.05, rtol=0.02)
    # weight gradient layout round trip
    dwp = torch.randn(Co, Kp)
    gr = torch.ones(Co, Ci, k, k)
    call("tic_conv_weight_grad", ptr(dwp), ptr(gr), Co, Ci, k, k, 0, None)
    torch.testing.assert_close(gr - 1, dwp[:, :k * k * Ci].view(Co, k, k, Ci).permute(0, 3, 1, 2))


def test_batchnorm_and_pools_against_torch():
    torch.manual_seed(1)
    B, H, W, C = 3, 8, 6, 16
    M = B * H * W
    x = bf(torch.randn(M, C) * 2 + 0.3)
    ident = bf(torch.randn(M, C))
    gamma, beta = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    rm, rv, nb = torch.zeros(C), torch.ones(C), torch.tensor(0)
    mean, rstd = torch.empty(C), torch.empty(C)
    scr = torch.full((513 * 2 * C,), float('nan'))   # partial-sum scratch: garbage on entry on purpose (no zero-on-entry contract)
    SB = scr.numel() * 4
    y = torch.empty(M, C, dtype=torch.bfloat16)
    call("tic_batchnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nb), ptr(mean), ptr(rstd), ptr(scr), SB, ptr(ident), ptr(y), M, C, 1e-5, 0.1, 1, 1, None)
    xr = x.float().requires_grad_(True)
    gr, br, ir = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), ident.float().requires_grad_(True)
    bn = torch.nn.functional.batch_norm(xr, torch.zeros(C), torch.ones(C), gr, br, True, 0.1, 1e-5)
    ref = torch.relu(bn + ir)
    torch.testing.assert_close(y.float(), ref.detach(), atol=0.03, rtol=0.02)
    torch.testing.assert_close(rm, 0.1 * x.float().mean(0), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(rv, 0.9 + 0.1 * x.float().var(0, unbiased=True), atol=1e-4, rtol=1e-4)
    assert int(nb) == 1
    dy = bf(torch.randn(M, C))
    ref.backward(dy.float())
    dx, dskip = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
    dg, db = torch.zeros(C), torch.zeros(C)
    call("tic_batchnorm_bwd", ptr(dy), ptr(y), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(scr), SB, ptr(dx), ptr(dskip), 0, ptr(dg), ptr(db), M, C, None)
    # the ReLU mask comes from the bf16 output; compare where the reference is not at the kink
    torch.testing.assert_close(dx.float(), xr.grad, atol=0.06, rtol=0.05)
    torch.testing.assert_close(dskip.float(), ir.grad, atol=0.02, rtol=0.02)
    torch.testing.assert_close(dg, gr.grad, atol=0.05, rtol=0.02)
    torch.testing.assert_close(db, br.grad, atol=0.05, rtol=0.02)
    # y = relu(bn(x)) without the residual add: the backward recomputes the mask from x and must equal the y-masked form exactly
    y2 = torch.empty(M, C, dtype=torch.bfloat16)
    call("tic_batchnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nb), ptr(mean), ptr(rstd), ptr(scr), SB, None, ptr(y2), M, C, 1e-5, 0.1, 1, 1, None)
    dxa_, dxb_ = torch.empty(M, C, dtype=torch.bfloat16), torch.empty(M, C, dtype=torch.bfloat16)
    dga, dba, dgb, dbb = torch.zeros(C), torch.zeros(C), torch.zeros(C), torch.zeros(C)
    call("tic_batchnorm_bwd", ptr(dy), ptr(y2), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(scr), SB, ptr(dxa_), None, 0, ptr(dga), ptr(dba), M, C, None)
    call("tic_batchnorm_bwd_relu", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(scr), SB, ptr(dxb_), ptr(dgb), ptr(dbb), M, C, None)
    assert torch.equal(dxa_, dxb_) and torch.equal(dga, dgb) and torch.equal(dba, dbb)
    assert 0.2 < float((y2 == 0).float().mean()) < 0.8
    # pools
    xp = bf(torch.relu(torch.randn(B, H, W, C)))   # post-ReLU input: many ties at 0 -> first-maximum rule matters
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    yp = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16)
    call("tic_maxpool3x3s2_fwd", ptr(xp), ptr(yp), B, H, W, C, None)
    xpr = xp.float().permute(0, 3, 1, 2).requires_grad_(True)
    mp = torch.nn.functional.max_pool2d(xpr, 3, 2, 1)
    assert torch.equal(yp.float(), mp.detach().permute(0, 2, 3, 1))
    dyp = bf(torch.randn(B, Ho, Wo, C))
    mp.backward(dyp.float().permute(0, 3, 1, 2))
    dxp = torch.empty(B, H, W, C, dtype=torch.bfloat16)
    call("tic_maxpool3x3s2_bwd", ptr(xp), ptr(yp), ptr(dyp), ptr(dxp), B, H, W, C, None)
    torch.testing.assert_close(dxp.float(), xpr.grad.permute(0, 2, 3, 1), atol=0.02, rtol=0.01)
    # the index form: same y, and a backward from (argmax position, dy) alone that equals the scanning backward bit for bit
    yp2, pidx = torch.empty_like(yp), torch.empty(B, Ho, Wo, C, dtype=torch.uint8)
    call("tic_maxpool3x3s2_fwd_idx", ptr(xp), ptr(yp2), ptr(pidx), B, H, W, C, None)
    dxp2 = torch.empty_like(dxp)
    call("tic_maxpool3x3s2_bwd_idx", ptr(pidx), ptr(dyp), ptr(dxp2), B, H, W, C, None)
    assert torch.equal(yp2, yp) and torch.equal(dxp2, dxp) and int(pidx.max()) <= 8
    # the stem's fused tail: bn -> relu -> maxpool (+ positions), bit-identical to the two-kernel sequence
    xs = bf(torch.randn(B * H * W, C) * 2 + 0.3)
    rm2, rv2, nb2 = torch.zeros(C), torch.ones(C), torch.tensor(0)
    rm3, rv3, nb3 = torch.zeros(C), torch.ones(C), torch.tensor(0)
    mA, rA, mB, rB = torch.empty(C), torch.empty(C), torch.empty(C), torch.empty(C)
    aA = torch.empty(B * H * W, C, dtype=torch.bfloat16)
    call("tic_batchnorm_fwd", ptr(xs), ptr(gamma), ptr(beta), ptr(rm2), ptr(rv2), ptr(nb2), ptr(mA), ptr(rA), ptr(scr), SB, None, ptr(aA), B * H * W, C, 1e-5, 0.1, 1, 1, None)
    hA, iA = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16), torch.empty(B, Ho, Wo, C, dtype=torch.uint8)
    call("tic_maxpool3x3s2_fwd_idx", ptr(aA), ptr(hA), ptr(iA), B, H, W, C, None)
    hB, iB = torch.empty_like(hA), torch.empty_like(iA)
    call("tic_bn_relu_maxpool_fwd", ptr(xs), ptr(gamma), ptr(beta), ptr(rm3), ptr(rv3), ptr(nb3), ptr(mB), ptr(rB), ptr(scr), SB, ptr(hB), ptr(iB), B, H, W, C, 1e-5, 0.1, 1, None)
    assert torch.equal(hA, hB) and torch.equal(iA, iB) and torch.equal(mA, mB) and torch.equal(rA, rB) and torch.equal(rm2, rm3) and torch.equal(rv2, rv3)
    z = torch.empty(B, C, dtype=torch.bfloat16)
    call("tic_avgpool_fwd", ptr(xp), ptr(z), B, H * W, C, None)
    torch.testing.assert_close(z.float(), xp.float().mean((1, 2)), atol=0.01, rtol=0.01)
    dxa = torch.empty(B, H * W, C, dtype=torch.bfloat16)
    call("tic_avgpool_bwd", ptr(z), ptr(dxa), B, H * W, C, None)
    torch.testing.assert_close(dxa.float(), (z.float() / (H * W)).unsqueeze(1).expand(B, H * W, C), atol=1e-3, rtol=0.01)
    a, b2 = bf(torch.randn(64)), bf(torch.randn(64))
    exp = bf(a.float() + b2.float())
    call("tic_add_bf16", ptr(a), ptr(b2), 64, None)
    assert torch.equal(a, exp)


@pytest.mark.timeout(900)
def test_resnet18_step_matches_reference_golden(golden_dir):
    worst = rc.check_against_golden("resnet18", golden_dir, SimBackend(), torch.device("cpu"), logit_tol=5e-2, gnorm_tol=0.12)
    print("worst grad-norm rel err", worst)


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 9, 7, 64, 64, 1), (1, 12, 12, 128, 72, 2), (3, 6, 6, 64, 192, 1)])
def test_implicit_gemm_conv3x3(B, H, W, Cin, Cout, stride):
    """tic_conv_igemm_fwd / _wgrad (and the stride-1 dgrad through the flipped filter) against F.conv2d on CPU tensors"""
    rc.check_implicit_conv(lambda name, *a: call(name, *a), torch.device("cpu"), B, H, W, Cin, Cout, stride)


@pytest.mark.parametrize("M,C", [(5000, 16), (2100, 64), (700, 512)])
def test_batchnorm_row_splits_sum_in_fixed_order(M, C):
    """several row splits (10 / 17 / 11): every split stores its partial sums into the NaN-filled scratch, the consuming kernel adds
    them in a fixed order -- statistics and gradients vs torch, and bit-identical across two calls that use DIFFERENT scratch buffers"""
    torch.manual_seed(M)
    x, dy = bf(torch.randn(M, C) * 1.5 + 0.2), bf(torch.randn(M, C))
    gamma, beta = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
    outs = []
    for rep in range(2):
        scr = torch.full((513 * 2 * C + 7 * rep,), float("nan"))
        rm, rv, nb = torch.zeros(C), torch.ones(C), torch.tensor(0)
        mean, rstd, y = torch.empty(C), torch.empty(C), torch.empty(M, C, dtype=torch.bfloat16)
        call("tic_batchnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nb), ptr(mean), ptr(rstd), ptr(scr), scr.numel() * 4, None, ptr(y), M, C,
             1e-5, 0.1, 1, 1, None)
        dx, dg, db = torch.empty(M, C, dtype=torch.bfloat16), torch.zeros(C), torch.zeros(C)
        call("tic_batchnorm_bwd_relu", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(scr), scr.numel() * 4, ptr(dx), ptr(dg), ptr(db), M, C, None)
        outs.append((mean, rstd, y, dx, dg, db, rm, rv))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    mean, rstd, y, dx, dg, db, rm, rv = outs[0]
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5))
    torch.testing.assert_close(mean, x.float().mean(0), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(rstd, 1 / torch.sqrt(x.float().var(0, unbiased=False) + 1e-5), atol=1e-5, rtol=1e-4)
    torch.testing.assert_close(y.float(), ref.detach(), atol=0.03, rtol=0.02)
    ref.backward(dy.float())
    torch.testing.assert_close(dx.float(), xr.grad, atol=0.06, rtol=0.05)
    torch.testing.assert_close(dg, gr.grad, atol=0.02 * float(gr.grad.abs().max()) + 0.05, rtol=0.02)
    torch.testing.assert_close(db, br.grad, atol=0.02 * float(br.grad.abs().max()) + 0.05, rtol=0.02)
    # a scratch one byte short of tic_batchnorm_scratch_bytes is refused, not overrun
    import ctypes
    from tests.simlib import sim
    need = sim().tic_batchnorm_scratch_bytes(M, C)
    assert 2 * C * 4 * 2 <= need <= 513 * 2 * C * 4
    with pytest.raises(Exception, match="scratch"):
        call("tic_batchnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nb), ptr(mean), ptr(rstd), ptr(scr), need - 1, None, ptr(y), M, C,
             1e-5, 0.1, 1, 1, None)


# ---- one-layer-deep goldens from the reference's own modules (tests/resnet_unit_checks.py) -------------------------------------
def test_reference_conv_units(golden_dir):
    from tests import resnet_unit_checks as ru
    ru.check_conv_units(SimBackend(), torch.device("cpu"), golden_dir)


def test_reference_batchnorm_units(golden_dir):
    from tests import resnet_unit_checks as ru
    ru.check_bn_units(SimBackend(), torch.device("cpu"), golden_dir)


def test_reference_block_units(golden_dir):
    from tests import resnet_unit_checks as ru
    ru.check_block_units(SimBackend(), torch.device("cpu"), golden_dir)
