"""Loader of the product HIP library (libtic_hip.so).  There is NO fallback: if the library is
missing and cannot be built, or no MI355X is visible, the hot path raises."""
from __future__ import annotations

import ctypes
import functools
import os

from . import _capi, build


class TicLibraryError(RuntimeError):
    pass


@functools.lru_cache(maxsize=1)
def lib() -> ctypes.CDLL:
    path = os.environ.get("TIC_HIP_LIB", build.LIB)   # TIC_HIP_LIB: A/B a previously built libtic_hip*.so (measurement only)
    if not os.path.exists(path):
        try:
            build.build_hip()
        except Exception as e:  # noqa: BLE001
            raise TicLibraryError(
                f"{path} is missing and could not be built with hipcc ({e}). The TIC hot path has no CPU or "
                "PyTorch fallback: build it with `python -m touhouimageclassification_amd.build hip`.") from e
    # torch bundles its own libamdhip64.so.7; it must be in the process BEFORE our library so that our
    # NEEDED libamdhip64.so.7 resolves to the SAME runtime (one HIP context, shared streams / pointers)
    import torch  # noqa: F401
    try:
        handle = ctypes.CDLL(path)
    except OSError as e:
        raise TicLibraryError(f"cannot load {path}: {e}") from e
    _capi.bind(handle)
    if handle.tic_version() != 1:
        raise TicLibraryError(f"{path}: ABI version {handle.tic_version()} != 1")
    return handle


def call(name: str, *args) -> None:
    h = lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise _capi.TicError(f"{name} failed ({rc}): {h.tic_last_error_string().decode(errors='replace')}")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t) -> None:
    if not t.is_cuda:
        raise TicLibraryError("TIC HIP kernels need tensors on an MI355X (`.to('cuda')`); there is no CPU path "
                              "in the product package (the CPU oracle lives under oracle/ and is test-only).")
