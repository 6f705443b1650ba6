"""Per-kernel resource table (VGPRs, LDS, scratch, spills) read back from the BUILT libtic_hip.so -- the cheap guard against
codegen regressions that cost 8 % once: a by-value kernel argument struct handed to a helper function landed in scratch.

    python tools/codeobj_report.py [path/to/libtic_hip.so]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("private_segment_fixed_size", "group_segment_fixed_size", "vgpr_count", "agpr_count", "sgpr_count",
          "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size")


def kernels(lib):
    """[{name, vgpr_count, ...}] for every kernel in the gfx950 code object embedded in `lib`"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co.o")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(d, "unused")],
                       check=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", f"--targets={TARGET}", f"--input={fat}",
                        f"--output={co}", "--unbundle"], check=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    import yaml
    doc = notes[notes.index("amdhsa.kernels:"):notes.index("\n...")]
    meta = yaml.safe_load(doc)
    return [{"name": k[".name"], **{f: int(k.get("." + f, 0)) for f in FIELDS}} for k in meta["amdhsa.kernels"]]


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.splitlines()


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "touhouimageclassification_amd", "libtic_hip.so")
    ks = kernels(lib)
    for k, n in sorted(zip(ks, demangle([k["name"] for k in ks])), key=lambda t: -t[0].get("vgpr_count", 0)):
        print(f"{n[:70]:70s} vgpr {k.get('vgpr_count', 0):3d} agpr {k.get('agpr_count', 0):3d} lds {k.get('group_segment_fixed_size', 0):6d} "
              f"scratch {k.get('private_segment_fixed_size', 0):4d} spill {k.get('vgpr_spill_count', 0)}")
