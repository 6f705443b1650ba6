"""Test-only loader for the CPU simulator build of the kernel sources (tests/sim/libtic_sim.so).

The simulator executes the *same kernel source* as libtic_hip.so with 64-lane lock-step fibers, so
lane maps, LDS swizzles and range-check handling are checked on CPU.  Host pointers, tiny shapes.
"""
import ctypes
import functools

import torch

from touhouimageclassification_amd import _capi, build


@functools.lru_cache(maxsize=1)
def sim():
    lib = ctypes.CDLL(build.build_sim())
    _capi.bind(lib)
    return lib


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous()
    return t.data_ptr()


def call(name, *args):
    lib = sim()
    rc = getattr(lib, name)(*args)
    _capi.check(lib, rc)


def bf(t):
    return t.to(torch.bfloat16)


def bfr(t):
    """bf16 round-trip, value kept in fp32."""
    return t.to(torch.bfloat16).to(torch.float32)


class SimBackend:
    """Engine backend that routes the C ABI to the CPU simulator build (tests only)."""

    def call(self, name, *args):
        call(name, *args)

    def stream(self):
        return None

    def check_tensor(self, t):
        assert not t.is_cuda
