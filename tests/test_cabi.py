"""CPU checks of the boundary: the product library builds for gfx950, loads, and exports every symbol
include/tic_hip.h declares (no compute calls without a GPU); argument validation returns TIC_EINVAL."""
import ctypes
import re

from touhouimageclassification_amd import _capi, build


def _header_symbols():
    src = open(build.ROOT + "/include/tic_hip.h").read()
    return sorted(set(re.findall(r"\b(tic_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import torch  # noqa: F401  (same load order as the product loader)
    lib = ctypes.CDLL(build.build_hip())
    declared = _header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tic_hip.h but not exported"
    assert sorted(_capi.SIGNATURES) == declared, "ctypes signature table out of sync with the header"
    _capi.bind(lib)
    assert lib.tic_version() == 1


def test_argument_validation_without_gpu():
    import torch  # noqa: F401
    lib = ctypes.CDLL(build.build_hip())
    _capi.bind(lib)
    # N not a multiple of 128 -> TIC_EINVAL before any launch
    rc = lib.tic_gemm_nt_bf16(16, 16, 10, 200, 100, 0, None, 16, None, None, None, None, None, 0, None)
    assert rc == -1 and b"K % 64" in lib.tic_last_error_string()
    rc = lib.tic_attention_fwd(16, 16, 16, 1, 1, 300, 0.125, None)
    assert rc == -1 and b"208" in lib.tic_last_error_string()
    d = _capi.TicVitDims(4, 1024, 16, 4096, 24, 120, 224, 16, 3, 1e-12)
    lay = _capi.TicVitLayout()
    assert lib.tic_vit_layout(ctypes.byref(d), ctypes.byref(lay)) == 0
    assert lay.n_params >= 303_424_632 and lay.n_params - 303_424_632 < 4096   # ViT-L/16 C=120 (+ padding)
    d.D = 1000
    assert lib.tic_vit_layout(ctypes.byref(d), ctypes.byref(lay)) == -1
    # newer entry points reject bad shapes / pointer combinations the same way
    rc = lib.tic_conv_igemm_fwd(16, 16, 16, 2, 8, 8, 48, 64, 3, 3, 1, 1, None)          # Cin = 48: not a multiple of 64
    assert rc == -1 and b"Cin" in lib.tic_last_error_string()
    rc = lib.tic_conv_igemm_wgrad(16, 16, 16, 2, 8, 8, 64, 60, 3, 3, 1, 1, None)        # Cout = 60: not a multiple of 8
    assert rc == -1 and b"Cout" in lib.tic_last_error_string()
    rc = lib.tic_attention_bwd_ws(16, 16, 16, 16, 16, None, 16, 0, 1, 1, 197, 0.125, None)   # scratch without dbias
    assert rc == -1 and b"scratch" in lib.tic_last_error_string()
    rc = lib.tic_gemm_nt_bf16_ex(16, 16, 10, 256, 64, 2, None, None, None, 16, 16, None, None, 0, 16, None)   # colsum with EPI_RESID
    assert rc == -1 and b"colsum" in lib.tic_last_error_string()
    assert lib.tic_kernel_timer_read(None, None) == -1
    assert lib.tic_set_option(b"no_such_knob", 1) == -1 and lib.tic_set_option(b"stream_nt", 13) == 0


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from touhouimageclassification_amd.ViT.model import ViT
    m = ViT(10, pretrained=False, model_name="tiny")
    with pytest.raises(Exception) as e:
        m(torch.zeros(1, 3, 224, 224))
    assert "MI355X" in str(e.value) or "no CPU path" in str(e.value)
