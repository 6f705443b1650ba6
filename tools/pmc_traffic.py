#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes over the grouped dW launch into profiles/rNN_traffic.json (what bench.py's roofline.traffic reads).

  # four SEPARATE counter passes (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots), each:  rocprofv3 --pmc <set> -d DIR/<tag> --output-format csv -- python3 tools/dw_ab.py tn_block -1 --rounds 1 --reps 6
  python tools/pmc_traffic.py DIR OUT.json [--batch 332]

FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced stream), WRITE_SIZE is exact for 16-byte streaming stores and float
atomics; both are in KiB.  The record carries the sha256 of csrc/gemm_tn256.h: bench.py prints traffic = null once the kernel changes."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 332
M, D, F = batch * 197, 1024, 4096
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("gemm_tn256_streamk_kernel"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in acc.items()}
print({k: (round(v, 1), len(acc[k])) for k, v in mean.items()})
fetch_b = 2.0 * mean["FETCH_SIZE"] * 1024.0
write_b = mean["WRITE_SIZE"] * 1024.0
shapes = [(D, F), (F, D), (D, D), (3 * D, D)]
alg = sum(M * (n + k) * 2 + n * k * 4 for n, k in shapes)   # every operand once + the fp32 dW
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "touhouimageclassification_amd", "csrc", "gemm_tn256.h")
rec = {
    "kernel": "gemm_tn256_streamk_kernel (grouped dW of one ViT-L block: 4 problems, one launch, phase-aligned stream-K, blocked tile walk)",
    "M": M, "hidden": D,
    "command": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE} "
               "--output-format csv -- python3 tools/dw_ab.py tn_mfma 0 --rounds 1 --reps 6   (four separate passes, product library)",
    "src_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
    "FETCH_SIZE_KB_mean": mean["FETCH_SIZE"], "WRITE_SIZE_KB_mean": mean["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced stream -> doubled (MI355X_MICROARCH.md 'HBM'); Infinity-Cache hits are included",
    "hbm_bytes_per_launch": int(fetch_b + write_b), "algorithmic_bytes_per_launch": int(alg),
    "traffic_over_algorithmic": round((fetch_b + write_b) / alg, 3),
}
if "TCC_HIT_sum" in mean:
    rec["L2_hit"] = round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 3)
if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over 1024 SIMDs
    rec["mfma_busy_per_simd"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0), 3)
if "SQ_WAIT_ANY" in mean and "SQ_WAVE_CYCLES" in mean:
    rec["wait_any_over_wave_cycles"] = round(mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"], 3)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
