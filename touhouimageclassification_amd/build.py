"""Build the native libraries IN-TREE (they travel to the GPU box with the repo snapshot).

  libtic_hip.so          hipcc --offload-arch=gfx950, the product library (device code + C ABI)
  tests/sim/libtic_sim.so  clang++ -x c++ -DTIC_SIM, the test-only CPU simulator build of the same sources
  libtic_hip_dbg.so      hipcc -DTIC_MEASURE, measurement variants for tools/ (built on demand: `build dbg`)

hipcc cross-compiles gfx950 without a GPU, so this runs in the authoring container.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libtic_hip.so")
DBG_LIB = os.path.join(PKG, "libtic_hip_dbg.so")   # -DTIC_MEASURE: measurement variants (tools/ only), never loaded by the product
SIM_DIR = os.path.join(ROOT, "tests", "sim")
SIM_LIB = os.path.join(SIM_DIR, "libtic_sim.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _sources():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    deps.append(os.path.join(ROOT, "include", "tic_hip.h"))
    return deps


def build_hip(force: bool = False, verbose: bool = False) -> str:
    deps = _sources()
    if not force and _newer(LIB, deps):
        return LIB
    cmd = [os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", os.path.join(CSRC, "tic_hip.hip"), "-o", LIB]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return LIB


def build_hip_dbg(force: bool = False) -> str:
    """the measurement library: the same sources with -DTIC_MEASURE (main-loop ablation variants of the 256x256 GEMMs, in-kernel
    stage stamps).  Selected by the tools through TIC_HIP_LIB; the product loader never picks it up by itself."""
    deps = _sources()
    if not force and _newer(DBG_LIB, deps):
        return DBG_LIB
    cmd = [os.path.join(ROCM, "bin", "hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTIC_MEASURE",
           "-Wno-unused-value", os.path.join(CSRC, "tic_hip.hip"), "-o", DBG_LIB]
    subprocess.run(cmd, check=True)
    return DBG_LIB


def build_sim(force: bool = False) -> str:
    deps = _sources() + [os.path.join(SIM_DIR, "sim_runtime.h"), os.path.join(SIM_DIR, "tic_sim.cpp")]
    if not force and _newer(SIM_LIB, deps):
        return SIM_LIB
    cmd = [os.path.join(ROCM, "lib", "llvm", "bin", "clang++"), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared",
           "-DTIC_SIM", "-I", SIM_DIR, "-I", CSRC, os.path.join(SIM_DIR, "tic_sim.cpp"), "-o", SIM_LIB]
    subprocess.run(cmd, check=True)
    return SIM_LIB


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("hip", "all"):
        print(build_hip(force=True, verbose="-v" in sys.argv))
    if what in ("sim", "all"):
        print(build_sim(force=True))
    if what == "dbg":
        print(build_hip_dbg(force=True))
