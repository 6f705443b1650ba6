"""CPU check of the committed tolerance-headroom log of the last full `-m gpu` run (profiles/r03_headroom.jsonl, written by
tests/headroom.py on the GPU box): no tolerance band may be more than HALF used.  A band that passes at 0.9 on the box the
suite was developed on is a failure on the next box (round 2: one 15 % band at ratio 1.12 hid 86 tests from the record)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "profiles", "r03_headroom.jsonl")
LIMIT = 0.5


def _rows():
    with open(LOG) as f:
        return [json.loads(line) for line in f if line.strip()]


def test_no_gpu_tolerance_band_is_more_than_half_used():
    rows = _rows()
    assert len(rows) >= 500, "the log must come from a FULL -m gpu run"
    tests = {r["test"].split("[")[0] for r in rows}
    for needed in ("test_gpu_model.py", "test_gpu_ops.py", "test_gpu_resnet.py", "test_gpu_harness.py", "test_gpu_benchshape.py", "test_gpu_dp.py"):
        assert any(needed in t for t in tests), f"no entries from {needed}: the run did not reach it"
    over = [(r["ratio"], r["test"], r["site"], r["name"]) for r in rows if r["ratio"] is None or r["ratio"] > LIMIT]
    assert not over, f"{len(over)} tolerance bands more than half used, worst: {sorted(over, key=lambda t: -(t[0] or 9e9))[:5]}"


def test_worst_ratio_is_the_one_stated_in_design_md():
    worst = max(r["ratio"] for r in _rows())
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert f"worst observed / allowed = {worst:.2f}" in text, f"DESIGN.md must state the worst ratio of the committed log ({worst:.2f})"
