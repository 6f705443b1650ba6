"""ctypes signatures for the C ABI declared in include/tic_hip.h.

`bind(lib)` attaches argtypes/restype to every exported symbol and returns the list of names, so
the product loader (`_lib.py`, libtic_hip.so) and the test-only simulator loader (tests/sim) share
one definition.  No torch types cross the boundary: pointers are integers (`tensor.data_ptr()`),
the stream is the raw hipStream_t handle.
"""
from __future__ import annotations

import ctypes as C

P = C.c_void_p
I = C.c_int
L = C.c_long
F = C.c_float
SZ = C.c_size_t


NT_SCRATCH_BYTES = 192 * 32 * 512 * 16 + 4096   # TIC_NT_SCRATCH_BYTES
NT_FLAG_BYTES = 4096


class TicVitDims(C.Structure):
    _fields_ = [("B", I), ("D", I), ("H", I), ("F", I), ("L", I), ("C", I),
                ("img", I), ("patch", I), ("chans", I), ("eps", F)]


class TicVitLayout(C.Structure):
    _fields_ = [(n, L) for n in (
        "cls", "pos", "patch_w", "patch_b", "layer0", "layer_stride",
        "ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2",
        "lnf_g", "lnf_b", "cls_w", "cls_b", "n_params",
        "t_layer_stride", "t_wqkv", "t_wo", "t_w1", "t_w2", "t_total")] + [(n, SZ) for n in (
        "P", "hs", "hs_stride", "layer_ws", "layer_ws_stride",
        "a1", "mean1", "rstd1", "qkv", "lse", "o", "hmid", "a2", "mean2", "rstd2", "u", "g",
        "zf", "meanf", "rstdf", "logits", "dlogits", "dzf", "dh", "dhb", "dhb2", "du", "da", "dqkv", "dpatch",
        "nt_scratch", "ws_bytes")]


class TicVitState(C.Structure):
    _fields_ = [("dims", TicVitDims), ("params", P), ("grads", P), ("w16", P), ("wT16", P), ("workspace", P)]


SIGNATURES = {
    "tic_version": ([], I),
    "tic_last_error_string": ([], C.c_char_p),
    "tic_set_option": ([C.c_char_p, I], I),
    "tic_probe_stream": ([P, P, P, C.c_long, I, I, P], I),
    "tic_kernel_timer_enable": ([I], I),
    "tic_kernel_timer_read": ([P, P], I),
    "tic_gemm_nt_scratch": ([P, SZ], I),
    "tic_gemm_tn_scratch": ([P, SZ], I),
    "tic_gemm_nt_bf16": ([P, P, I, I, I, I, P, P, P, P, P, P, P, I, P], I),
    "tic_gemm_nt_bf16_ex": ([P, P, I, I, I, I, P, P, P, P, P, P, P, I, P, P], I),
    "tic_layernorm_bwd_ex": ([P, P, L, P, P, P, P, P, P, P, P, P, I, I, P], I),
    "tic_attention_bwd_ex": ([P, P, P, P, P, P, I, I, I, F, P], I),
    "tic_attention_bwd_ws": ([P, P, P, P, P, P, P, I, I, I, I, F, P], I),
    "tic_gemm_tn_bf16": ([P, P, P, I, I, I, P], I),
    "tic_gemm_tn_group_bf16": ([I, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(I), C.POINTER(I), I, P], I),
    "tic_gemm_tn_group_bf16_ex": ([I, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(I), C.POINTER(I), I, I, P], I),
    "tic_layernorm_fwd": ([P, L, P, P, P, P, P, I, I, F, P], I),
    "tic_layernorm_bwd": ([P, P, L, P, P, P, P, P, P, P, P, I, I, P], I),
    "tic_attention_fwd": ([P, P, P, I, I, I, F, P], I),
    "tic_attention_bwd": ([P, P, P, P, P, I, I, I, F, P], I),
    "tic_patchify": ([P, P, I, I, I, I, P], I),
    "tic_embed_cls": ([P, P, P, I, I, I, P], I),
    "tic_embed_bwd": ([P, P, P, I, I, I, P], I),
    "tic_gather_patch_rows": ([P, P, I, I, I, P], I),
    "tic_colsum_bf16": ([P, P, I, I, P], I),
    "tic_cast_bf16": ([P, P, L, P], I),
    "tic_cast_transpose_bf16": ([P, P, I, I, P], I),
    "tic_adamw": ([P, P, P, P, P, L, F, F, F, F, F, I, P], I),
    "tic_head_fwd": ([P, P, P, P, I, I, I, P], I),
    "tic_head_bwd": ([P, P, P, P, P, P, I, I, I, P], I),
    "tic_softmax_xent": ([P, P, P, P, P, I, I, F, P], I),
    "tic_augment": ([P, I, I, I, P, P, I, C.POINTER(F), C.POINTER(F), P], I),
    "tic_mix": ([P, P, I, I, I, I, I, F, I, I, I, I, P], I),
    "tic_mix_labels": ([P, P, I, I, F, P], I),
    "tic_conv_weight_pack": ([P, P, I, I, I, I, I, P], I),
    "tic_conv_weight_grad": ([P, P, I, I, I, I, I, P], I),
    "tic_conv_weight_pack_many": ([P, I, P], I),
    "tic_conv_weight_grad_many": ([P, I, P], I),
    "tic_conv_igemm_fwd": ([P, P, P, I, I, I, I, I, I, I, I, I, P], I),
    "tic_conv_igemm_dgrad_s2": ([P, P, P, I, I, I, I, I, I, I, I, I, P], I),
    "tic_conv_igemm_wgrad": ([P, P, P, I, I, I, I, I, I, I, I, I, P], I),
    "tic_nchw_to_nhwc_bf16": ([P, P, I, I, I, I, P], I),
    "tic_nchw_to_nhwc_pad_bf16": ([P, P, I, I, I, I, I, P], I),
    "tic_im2col_bf16": ([P, P, I, I, I, I, I, I, I, I, P], I),
    "tic_col2im_bf16": ([P, P, I, I, I, I, I, I, I, I, I, P], I),
    "tic_batchnorm_scratch_bytes": ([L, I], SZ),
    "tic_batchnorm_fwd": ([P, P, P, P, P, P, P, P, P, SZ, P, P, L, I, F, F, I, I, P], I),
    "tic_batchnorm_bwd": ([P, P, P, P, P, P, P, SZ, P, P, I, P, P, L, I, P], I),
    "tic_batchnorm_bwd_relu": ([P, P, P, P, P, P, P, SZ, P, P, P, L, I, P], I),
    "tic_maxpool3x3s2_fwd": ([P, P, I, I, I, I, P], I),
    "tic_maxpool3x3s2_bwd": ([P, P, P, P, I, I, I, I, P], I),
    "tic_bn_relu_maxpool_fwd": ([P, P, P, P, P, P, P, P, P, SZ, P, P, I, I, I, I, F, F, I, P], I),
    "tic_maxpool3x3s2_fwd_idx": ([P, P, P, I, I, I, I, P], I),
    "tic_maxpool3x3s2_bwd_idx": ([P, P, P, I, I, I, I, P], I),
    "tic_avgpool_fwd": ([P, P, I, I, I, P], I),
    "tic_avgpool_bwd": ([P, P, I, I, I, P], I),
    "tic_add_bf16": ([P, P, L, P], I),
    "tic_moe_gate": ([P, P, F, P, P, P, I, I, I, P], I),
    "tic_moe_gate_bwd": ([P, P, P, I, I, P], I),
    "tic_moe_combine": ([P, P, P, I, I, I, P], I),
    "tic_moe_combine_bwd": ([P, P, P, P, P, I, I, I, P], I),
    "tic_moe_loss": ([P, P, P, P, P, P, I, I, I, F, F, F, P], I),
    "tic_vit_layout": ([C.POINTER(TicVitDims), C.POINTER(TicVitLayout)], I),
    "tic_vit_refresh_weights": ([C.POINTER(TicVitState), I, P], I),
    "tic_vit_adamw": ([C.POINTER(TicVitState), P, P, F, F, F, F, F, I, P], I),
    "tic_vit_forward": ([C.POINTER(TicVitState), P, P, P], I),
    "tic_vit_forward_infer": ([C.POINTER(TicVitState), P, P, P], I),
    "tic_vit_backward_head": ([C.POINTER(TicVitState), P, P], I),
    "tic_vit_backward_layer": ([C.POINTER(TicVitState), I, P], I),
    "tic_vit_backward_layer_ex": ([C.POINTER(TicVitState), I, I, P], I),
    "tic_vit_zero_grads": ([C.POINTER(TicVitState), I, P], I),
    "tic_vit_backward_embed": ([C.POINTER(TicVitState), P], I),
}


def bind(lib: C.CDLL):
    for name, (args, res) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == a declared symbol is not exported
        fn.argtypes = args
        fn.restype = res
    return list(SIGNATURES)


class TicError(RuntimeError):
    pass


def check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise TicError(f"libtic error {rc}: {lib.tic_last_error_string().decode(errors='replace')}")
