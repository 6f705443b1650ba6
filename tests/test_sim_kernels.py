"""CPU checks of the HIP kernel sources through the wave-lockstep simulator (no GPU needed).

The kernel SOURCE that ships in libtic_hip.so is compiled against tests/sim/sim_runtime.h and run on
tiny shapes: this validates indexing, MFMA / transposed-LDS lane maps, swizzles and ragged-edge
handling against plain torch fp32 math.  The same checks run against the real library on the GPU
box in test_gpu_ops.py.
"""
import pytest

from tests import kernel_checks as kc
from tests.simlib import call


@pytest.fixture()
def env():
    return kc.Env("cpu", call)


@pytest.mark.parametrize("M,N,K", [(200, 128, 64), (130, 256, 192), (100, 64, 64), (70, 200, 128)])
def test_gemm_nt_bias_bf16(env, M, N, K):
    kc.check_gemm_nt_bias_bf16(env, M, N, K)


def test_gemm_nt_epilogues(env):
    kc.check_gemm_nt_gelu_resid_dgelu_patch(env)


@pytest.mark.parametrize("M", [64, 200, 333])
def test_gemm_tn(env, M):
    kc.check_gemm_tn(env, M)


def test_gemm_tn_ragged_nk(env):
    kc.check_gemm_tn(env, 150, N=64, K=192)     # conv-shaped: Cout = 64, K = 3 x 64
    kc.check_gemm_tn(env, 90, N=200, K=72)


@pytest.mark.parametrize("D", [128, 768, 1024])
def test_layernorm_fwd_bwd(env, D):
    kc.check_layernorm_fwd_bwd(env, D)


def test_layernorm_strided_cls_rows(env):
    kc.check_layernorm_strided_cls_rows(env)


@pytest.mark.parametrize("B,H,N", [(1, 2, 197), (2, 1, 50), (3, 2, 5), (1, 1, 33)])
def test_attention_fwd_bwd(env, B, H, N):
    kc.check_attention_fwd_bwd(env, B, H, N)


def test_elementwise_ops(env):
    kc.check_elementwise_ops(env)


def test_adamw_matches_torch(env):
    kc.check_adamw_matches_torch(env)


@pytest.mark.parametrize("soft", [False, True])
def test_head_and_xent(env, soft):
    kc.check_head_and_xent(env, soft)


@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (520, 256, 256), (257, 512, 128)])
def test_gemm_nt256_pipelined(env, M, N, K):
    """the 256x256 deep-pipelined kernel (forced), incl. ragged M, nk = 1 and odd/even K-tile counts"""
    call("tic_set_option", b"gemm_tile", 256)
    try:
        kc.check_gemm_nt_bias_bf16(env, M, N, K)
        if K == 256:
            kc.check_gemm_nt_gelu_resid_dgelu_patch(env, N=256, K=192, imgs=3, Pn=50)
    finally:
        call("tic_set_option", b"gemm_tile", 0)


@pytest.mark.parametrize("streamk", [0, 3, 7])   # 3 / 7 shares: partial tiles, shares crossing tile and problem boundaries
@pytest.mark.parametrize("M,shapes", [(200, [(256, 256)]), (333, [(256, 512), (512, 256), (256, 256)]), (64, [(128, 256), (256, 256)])])
def test_gemm_tn_group(env, M, shapes, streamk):
    call("tic_set_option", b"gemm_tile", 256)
    call("tic_set_option", b"tn_streamk", streamk)
    try:
        kc.check_gemm_tn_group(env, M, shapes)   # last case has a 128-multiple shape -> per-problem fallback
    finally:
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"tn_streamk", 1)


def test_augment(env):
    kc.check_augment(env)


def test_mix(env):
    kc.check_mix(env)


def test_fused_bias_gradients(env):
    kc.check_fused_bias_gradients(env)


@pytest.mark.parametrize("mfma", [16, 32])
def test_gemm_tn_group_phase_aligned_split(env, mfma):
    """16 shares over 8 tiles: per 'XCD' one main workgroup (steps [0, S/2)) and one tail workgroup (steps [S/2, S)); both MFMA
    shapes of the stream-K launch (16x16x32: the default; 32x32x16: kept for A/B measurements)"""
    call("tic_set_option", b"gemm_tile", 256)
    call("tic_set_option", b"tn_streamk", 16)
    call("tic_set_option", b"tn_mfma", mfma)
    try:
        kc.check_gemm_tn_group(env, 333, [(512, 512), (256, 1024)])
    finally:
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"tn_streamk", 1)
        call("tic_set_option", b"tn_mfma", 0)


def test_gemm_tn_group_four_wave_experiment(env):
    """the measured-and-rejected 4-wave x 128 x 128 form of the stream-K launch (gemm_tn256_w4.h, simulator + measurement builds only,
    DESIGN 4c: 685 vs 1 231 TFLOP/s): same results as the shipped kernel's checks, flat and phase-aligned shares, ragged last step"""
    call("tic_set_option", b"gemm_tile", 256)
    call("tic_set_option", b"tn_waves", 4)
    try:
        for shares in (16, 7):
            call("tic_set_option", b"tn_streamk", shares)
            kc.check_gemm_tn_group(env, 333, [(512, 512), (256, 1024)])
    finally:
        call("tic_set_option", b"tn_waves", 8)
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"tn_streamk", 1)


@pytest.mark.parametrize("parts", [2, 3])
def test_gemm_tn_group_equal_parts_split(env, parts):
    """9 tiles (not a multiple of 8: XCD runs of 2 and 1 tiles), every tile cut into 2 / 3 equal row ranges; 333 rows = 6 steps"""
    call("tic_set_option", b"gemm_tile", 256)
    call("tic_set_option", b"tn_streamk", 16)
    call("tic_set_option", b"tn_parts", parts)
    try:
        for mfma in (16, 32):
            call("tic_set_option", b"tn_mfma", mfma)
            kc.check_gemm_tn_group(env, 333, [(512, 512), (256, 1024), (256, 256)])
    finally:
        call("tic_set_option", b"gemm_tile", 0)
        call("tic_set_option", b"tn_streamk", 1)
        call("tic_set_option", b"tn_parts", -1)
        call("tic_set_option", b"tn_mfma", 0)




def test_gemm_tn_parts_slab_route(env):
    """2 tiles x 129 steps (ragged last step): 32 row parts per tile store into the slab, one reduce launch adds them to C
    (the library takes this route from 8 tiles on; the test knob lowers that so that 2 tiles reach it in the simulator)"""
    call("tic_set_option", b"tn_slab_min_tiles", 1)
    try:
        kc.check_gemm_tn_slab(env, 8200, 256, 512, slab_mb=24)
    finally:
        call("tic_set_option", b"tn_slab_min_tiles", 8)


@pytest.mark.parametrize("M,N,K,split", [(200, 256, 512, 2), (130, 128, 1024, 4)])
def test_splitk_nt_kernel_128(env, M, N, K, split):
    """the 128x128 kernel's split-K form: 4 tiles x 2 parts (4 K tiles each) / 2 tiles x 4 parts"""
    kc.check_splitk_nt(env, M, N, K, split, tile=128)


def test_splitk_timeout_is_reported(env):
    kc.check_splitk_timeout_is_reported(env)


@pytest.mark.parametrize("M,N,K,split", [(200, 256, 64, 0), (130, 128, 128, 0), (70, 128, 192, 0), (200, 256, 512, 0), (200, 128, 512, 2), (130, 128, 1024, 4)])
def test_nt_ring_of_four_stages_matches_two_stage_loop(env, M, N, K, split):
    """1, 2, 3 K tiles (ring not full), 8 K tiles, and the split-K hand-off on top of it: bit-identical outputs for every epilogue"""
    kc.check_nt_ring_matches(env, M, N, K, split)
