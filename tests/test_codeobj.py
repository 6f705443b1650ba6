"""Static guard on the built gfx950 code object: no kernel may touch scratch or spill, and the kernels whose launch shapes
assume a register budget keep it.  (A by-value argument struct forwarded to a helper once put the dominant weight-gradient
kernel's lookup table in scratch: 232 -> 253 VGPRs, 136 B/lane, 8 % slower -- invisible to every numerical test.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    from touhouimageclassification_amd import build
    import codeobj_report
    ks = codeobj_report.kernels(build.build_hip())
    names = codeobj_report.demangle([k["name"] for k in ks])
    return {n: k for n, k in zip(names, ks)}


def test_no_scratch_no_spills(kernels):
    assert len(kernels) > 40
    # (SGPR spills go to VGPR lanes, not to memory: attn_fwd_kernel has a few and they are harmless)
    def cold_path(name):
        # the split-K forms of the 256x256 NT kernel (<EPI, 0, 1>): a few accumulator quads are spilled AROUND the flag-wait loop of
        # the hand-off, once per tile and after the K loop (checked in the ISA: no scratch instruction inside the loop); bounded here
        return name.startswith("void gemm_nt256_kernel<") and name.rstrip().endswith(", 0, 1>(GemmNtParams)")
    bad = {n: (k["private_segment_fixed_size"], k["vgpr_spill_count"]) for n, k in kernels.items()
           if (k["private_segment_fixed_size"] or k["vgpr_spill_count"]) and not (cold_path(n) and k["private_segment_fixed_size"] <= 384)}
    assert not bad, bad


def test_register_budgets(kernels):
    def get(prefix):
        hit = [k for n, k in kernels.items() if n.startswith(prefix)]
        assert hit, prefix
        return hit

    # 512-thread workgroups, 2 waves per SIMD: the 256-VGPR half of the unified file (accumulators live in arch VGPRs)
    for prefix in ("void gemm_nt256_kernel<", "gemm_tn256_streamk_kernel", "gemm_tn256_streamk_single_kernel", "void gemm_tn256_kernel<0>"):
        for k in get(prefix):
            assert k["vgpr_count"] + k["agpr_count"] <= 256, (prefix, k)
    for k in get("gemm_tn256_streamk_mfma32"):   # the 32x32x16 form: 232 when the argument table stays in SGPRs
        assert k["vgpr_count"] <= 236, k
    # the 16x16x32 form (default) under its two names is the same code; it fills the 256-register half (6 lane-address registers
    # instead of 3) and must not touch scratch (test_no_scratch_no_spills)
    assert len({k["vgpr_count"] for n, k in kernels.items() if n.startswith("gemm_tn256_streamk") and "mfma32" not in n}) == 1
    for k in get("attn_bwd_kernel"):             # 1024 threads -> 4 waves/SIMD -> 128 registers
        assert k["vgpr_count"] <= 128, k
    for k in get("void attn_fwd_kernel<8>"):     # two 8-wave workgroups per CU = 4 waves per SIMD
        assert k["vgpr_count"] <= 128, k
    for k in get("void attn_fwd_kernel<"):            # 4 or 8 waves, launch bound 2 waves per SIMD
        assert k["vgpr_count"] <= 256, k
    for prefix in ("void gemm_nt_kernel<", "void gemm_tn_kernel<"):   # 256-thread 128^2 tiles: 4 waves/SIMD wanted
        for n, k in kernels.items():
            if n.startswith(prefix):
                # (the split-K forms <EPI, false, 1> batch the 16 slab loads of the hand-off: LDS allows two workgroups per CU = 2 waves
                # per SIMD anyway, so they may use the 256-register half)
                # the 4-stage ring forms <EPI, false, SPLITK, 4> use 128 KiB of LDS: one workgroup = one wave per SIMD, any register count
                # up to 256 costs nothing; they hold two k-halves of fragments (64 registers) beside the 64 accumulators
                limit = 256 if n.rstrip().endswith((", false, 1, 2>(GemmNtParams)", ", 4>(GemmNtParams)")) else 128
                assert k["vgpr_count"] <= limit, (prefix, n, k)
