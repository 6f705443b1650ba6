"""GPU parity tests of every C-ABI operator (libtic_hip.so on a real MI355X) against plain torch
fp32 math on the same device, at hot-path shapes (ViT-B / ViT-L token matrices, N = 197 ragged M)
and at ragged edge shapes.  Tolerances are bf16-level and written in tests/kernel_checks.py."""
import pytest
import torch

from tests import kernel_checks as kc

pytestmark = pytest.mark.gpu


@pytest.fixture()
def env():
    from touhouimageclassification_amd._lib import call, current_stream
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return kc.Env("cuda", call, stream=current_stream)


@pytest.mark.parametrize("M,N,K", [(200, 128, 64), (130, 256, 192), (12608, 3072, 1024), (1970, 768, 3072), (197 * 3 + 1, 1024, 4096)])
def test_gemm_nt_bias_bf16(env, M, N, K):
    kc.check_gemm_nt_bias_bf16(env, M, N, K)


@pytest.mark.parametrize("N,K,imgs,Pn", [(128, 128, 3, 50), (1024, 768, 16, 196), (4096, 1024, 7, 197),
                                         (1024, 768, 44, 196), (3072, 1024, 13, 197)])   # the last two: 136 / 132 tiles -> the 256x256 kernel's epilogues
def test_gemm_nt_epilogues(env, N, K, imgs, Pn):
    kc.check_gemm_nt_gelu_resid_dgelu_patch(env, N, K, imgs, Pn)


@pytest.mark.parametrize("M,N,K", [(64, 128, 256), (200, 128, 256), (333, 256, 128), (12608, 1024, 1024), (1970, 3072, 768), (3940, 1024, 4096)])
def test_gemm_tn(env, M, N, K):
    kc.check_gemm_tn(env, M, N, K)


@pytest.mark.parametrize("D,rows", [(128, 9), (768, 1001), (1024, 12608)])
def test_layernorm_fwd_bwd(env, D, rows):
    kc.check_layernorm_fwd_bwd(env, D, rows)


def test_layernorm_strided_cls_rows(env):
    kc.check_layernorm_strided_cls_rows(env)


@pytest.mark.parametrize("B,H,N", [(1, 2, 197), (2, 1, 50), (8, 16, 197), (3, 12, 197), (3, 2, 5), (1, 1, 33)])
def test_attention_fwd_bwd(env, B, H, N):
    kc.check_attention_fwd_bwd(env, B, H, N)


def test_elementwise_ops(env):
    kc.check_elementwise_ops(env)


def test_adamw_matches_torch(env):
    kc.check_adamw_matches_torch(env)


@pytest.mark.parametrize("soft", [False, True])
def test_head_and_xent(env, soft):
    kc.check_head_and_xent(env, soft)


@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (12608, 1024, 4096), (16351, 3072, 1024), (1000, 4096, 1024)])
def test_gemm_nt256_pipelined(env, M, N, K):
    """the deep-pipelined 256x256 kernel forced on: same checks + bitwise equality with the 128x128 kernel
    (identical fp32 accumulation order, so any difference is a pipeline race)."""
    from touhouimageclassification_amd._lib import call
    from touhouimageclassification_amd import ops
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    try:
        call("tic_set_option", b"gemm_tile", 128)
        ref = ops.gemm_nt(a, w, ops.EPI_BF16, bias=bias)
        call("tic_set_option", b"gemm_tile", 256)
        kc.check_gemm_nt_bias_bf16(env, M, N, K)
        for _ in range(5):
            assert torch.equal(ops.gemm_nt(a, w, ops.EPI_BF16, bias=bias), ref)
    finally:
        call("tic_set_option", b"gemm_tile", 0)


@pytest.mark.parametrize("M,shapes", [(200, [(256, 256)]), (12608, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]),
                                      (1970, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                                      (12608, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])])   # ViT-B: 108 tiles -> the equal-parts split (2 x 108 workgroups)
def test_gemm_tn_group(env, M, shapes):
    kc.check_gemm_tn_group(env, M, shapes)
    from touhouimageclassification_amd._lib import call
    call("tic_set_option", b"gemm_tile", 256)
    try:
        for mfma in (16, 32):           # MFMA shape of the stream-K launch: 16x16x32 (default) and 32x32x16
            call("tic_set_option", b"tn_mfma", mfma)
            for sk in (0, 1, 7, 100):   # full-M tiles; 256 / 7 / 100 stream-K shares (partial tiles, shares crossing problems)
                call("tic_set_option", b"tn_streamk", sk)
                kc.check_gemm_tn_group(env, M, shapes)
            call("tic_set_option", b"tn_streamk", 1)
            for parts in (0, 2, 3):         # flat stream-K / every tile in 2 / 3 equal row parts (where the tile count has no phase-aligned split)
                call("tic_set_option", b"tn_parts", parts)
                kc.check_gemm_tn_group(env, M, shapes)
            call("tic_set_option", b"tn_parts", -1)
    finally:
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"tn_streamk", 1)
        call("tic_set_option", b"tn_mfma", 0)
        call("tic_set_option", b"tn_parts", -1)


@pytest.mark.parametrize("S,H,W", [(32, 40, 48), (224, 256, 256)])
def test_augment(env, S, H, W):
    kc.check_augment(env, S, H, W) if S == 32 else kc.check_augment(env, S, H, W)


def test_mix(env):
    kc.check_mix(env)


def test_fused_bias_gradients(env):
    kc.check_fused_bias_gradients(env)




@pytest.mark.parametrize("M,N,K", [(40960, 1024, 256), (50176, 256, 1024), (12544, 2048, 512), (12544, 512, 2048), (8200, 256, 256), (200704, 512, 256), (20480, 1024, 2048), (50176, 512, 1024)])
def test_gemm_tn_parts_slab_route(env, M, N, K):
    """ResNet 1x1 weight-gradient shapes over 8 200 .. 200 704 rows: from 8 tiles of 256 x 256 on (the 512 <-> 1024 / 2048 layers) row parts
    stored to the slab + one reduce launch; below (256 <-> 1024: 4 tiles) the split-M 128 x 128 kernel, which is faster there -- same check"""
    kc.check_gemm_tn_slab(env, M, N, K)


@pytest.mark.parametrize("M,N,K,split", [(1576, 1024, 4096, 2), (1576, 1024, 4096, 4), (3152, 1024, 3072, 2), (1576, 1024, 1024, 2)])
def test_splitk_nt_kernel_128(env, M, N, K, split):
    """the 128x128 kernel's split-K form at ViT-L's 8 / 16-image shapes (104 / 200 tiles)"""
    kc.check_splitk_nt(env, M, N, K, split, tile=128)


@pytest.mark.parametrize("M,N,K", [(1576, 1024, 4096), (3152, 1024, 3072), (1576, 1024, 1024), (130, 128, 64), (300, 256, 192)])
def test_nt_ring_of_four_stages_matches_two_stage_loop(env, M, N, K):
    """the 4-stage ring of the 128x128 NT kernel (what ViT-L's N = 1024 products run on at 8-16 images per GPU) against the 2-stage loop:
    bit-identical for every epilogue; 1 and 3 K tiles leave the ring partly empty.  (Its split-K form at these shapes:
    test_splitk_nt_kernel_128, whose 104 x 2 workgroups take the ring by the same rule; bit-equality under the hand-off: simulator.)"""
    kc.check_nt_ring_matches(env, M, N, K, 0)
