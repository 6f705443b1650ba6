import os, sys, torch
sys.path.insert(0, os.getcwd())
from tests import kernel_checks as kc
from touhouimageclassification_amd._lib import call, current_stream
env = kc.Env("cuda", lambda name, *a: call(name, *a), stream=current_stream, seed=3)
bad = 0
for N in (1, 2, 5, 15, 16, 17, 31, 32, 33, 64, 100, 150, 196, 197, 200, 208):
    for (B, H) in ((1, 1), (3, 2), (2, 5)):
        try:
            kc.check_attention_fwd_bwd(env, B, H, N)
        except AssertionError as e:
            bad += 1; print("MISMATCH attention", B, H, N, str(e)[:300], flush=True)
for D in (128, 256, 384, 512, 768, 1024):
    for rows in (1, 3, 4, 5, 197, 1000):
        try:
            kc.check_layernorm_fwd_bwd(env, D, rows)
        except AssertionError as e:
            bad += 1; print("MISMATCH layernorm", D, rows, str(e)[:300], flush=True)
print("attention / layernorm sweep:", "CLEAN" if bad == 0 else f"{bad} mismatches")
