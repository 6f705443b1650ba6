"""Host-side logic of bench.py that needs no GPU: the in-run parity block and the self-launch of N ranks."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_block_counts():
    import bench
    ref = torch.tensor([[0.9, 0.1, 0.0, -0.2, -0.5, -0.9], [0.0, 0.5, 0.49, 0.1, -0.1, -0.3]])
    hip = ref + torch.tensor([[0.01, -0.01, 0.0, 0.0, 0.0, 0.0], [0.0, -0.01, 0.01, 0.0, 0.0, 0.0]])   # row 1: top-1 flips inside the noise
    p = bench.parity_block(hip, 1.001, ref, 1.0, torch.tensor([0, 1]))
    assert abs(p["max_abs_dlogits"] - 0.01) < 1e-6 and abs(p["loss_rel"] - 1e-3) < 1e-6
    assert p["top1_match"] == 0.5 and p["top5_match"] == 1.0
    assert p["top1_decidable_rows"] == 1 and p["top1_match_on_decidable"] is True


def test_self_launch_starts_ranks_and_forwards_failure():
    """`python bench.py --gpus 2` with no launcher env: the parent must start two rank processes itself (they then stop because
    this container has no GPU) and exit non-zero -- never an AssertionError about WORLD_SIZE (VERDICT r1 missing #1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr
    assert "bench.py needs an MI355X" in r.stderr and "AssertionError" not in r.stderr


def test_resnet_work_per_image_matches_the_survey():
    """ResNet-50 @224 = 8.174 GFLOP forward conv (+ fc); 53 BatchNorm-ed tensors = 11.1 M elements per image"""
    import bench
    from touhouimageclassification_amd.ResNet.model import resnet50
    fl, by = bench.resnet_work_per_image(resnet50(num_classes=120), 224)
    assert abs(fl / 3.0 - (8.174e9 + 2 * 2048 * 120)) < 2e7
    assert 10.5e6 < by / 22.0 < 11.5e6
