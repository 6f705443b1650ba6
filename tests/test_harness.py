"""Host logic of the harness mirrors (TIC/ViT/finetune.py, TIC/ViT/ntrain.py counterparts) on CPU.
Model steps, where needed, run through the simulator backend on the tiny config."""
import logging
import math
import os

import pytest
import torch

from oracle import vit_oracle as vo
from tests.simlib import SimBackend
from touhouimageclassification_amd import aug
from touhouimageclassification_amd.ViT import finetune as ft

LOG = logging.getLogger("tic-test")


def test_early_exit_rule():
    # fewer than N recorded epochs -> never
    assert not ft.early_exit([1.0, 2.0], 3, LOG)
    # last N losses all >= the loss N epochs earlier -> stop
    assert ft.early_exit([0.5, 0.6, 0.7, 0.5], 3, LOG)
    assert ft.early_exit([9.0, 0.5, 0.6, 0.7, 0.8], 3, LOG)
    # any improvement inside the window -> continue
    assert not ft.early_exit([0.5, 0.6, 0.4, 0.7], 3, LOG)
    # exactly N recorded: the window is the whole timeline
    assert ft.early_exit([0.5, 0.5, 0.5], 3, LOG)


def test_linear_warmup_schedule_matches_hf_lambda():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = ft.get_linear_schedule_with_warmup(opt, 5, 20)
    for step in range(25):
        assert abs(opt.param_groups[0]["lr"] - vo.linear_warmup_lambda(step, 5, 20)) < 1e-12
        opt.step()
        sch.step()


def test_preset_selection_follows_reference_branches():
    assert aug.preset_name(False, True, True, False) == "none"
    assert aug.preset_name(True, True, True, True) == "grey"
    assert aug.preset_name(True, True, True, False) == "full"
    assert aug.preset_name(True, True, False, False) == "diversity"
    assert aug.preset_name(True, False, True, False) == "generalization"
    with pytest.raises(Exception):
        aug.preset_name(True, False, False, False)


def test_param_samplers_respect_torchvision_ranges():
    g = torch.Generator().manual_seed(0)
    P = aug.sample_params(400, 256, 256, 224, "full", g)
    top, left, h, w = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
    assert (h >= 1).all() and (w >= 1).all() and (top + h <= 256).all() and (left + w <= 256).all()
    area = h * w / (256 * 256)
    assert area.min() >= 0.07 and area.max() <= 1.0 and 0.3 < area.mean() < 0.75
    ratio = w / h
    assert ratio.min() > 0.7 and ratio.max() < 1.4
    assert 0.35 < P[:, 4].mean() < 0.65                      # flip p = 0.5
    assert (P[:, 9:12] >= 0.8).all() and (P[:, 9:12] <= 1.2).all() and (P[:, 12].abs() <= 0.1).all()
    assert all(sorted(r.tolist()) == [0, 1, 2, 3] for r in P[:, 5:9])
    assert 0.1 < P[:, 14].mean() < 0.3                       # grayscale p = 0.2
    assert 0.35 < P[:, 15].mean() < 0.65                     # erasing p = 0.5
    er = P[P[:, 15] > 0]
    ea = er[:, 18] * er[:, 19] / (224 * 224)
    assert ea.min() > 0.015 and ea.max() < 0.36 and (er[:, 16] + er[:, 18] <= 224).all() and (er[:, 17] + er[:, 19] <= 224).all()
    T = aug.sample_params(8, 200, 300, 224, "test", g)
    assert (T[:, 0:4] == torch.tensor([0., 0., 200., 300.])).all() and T[:, 4].sum() == 0 and T[:, 13:16].sum() == 0


def test_cutmix_mixup_sampler():
    m = aug.CutMixOrMixUp(10, backend=SimBackend(), seed=1)
    modes = set()
    for _ in range(50):
        mode, lam, (x1, y1, x2, y2) = m.sample(224, 224)
        modes.add(mode)
        assert 0.0 <= lam <= 1.0
        if mode == 1:
            assert 0 <= x1 <= x2 <= 224 and 0 <= y1 <= y2 <= 224
            assert abs(lam - (1 - (x2 - x1) * (y2 - y1) / 224 ** 2)) < 1e-6
    assert modes == {0, 1}


class _TinyFloatSet(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(3)
        self.x = torch.randn(n, 3, 32, 32, generator=g)   # the 'micro' preset: 32 px, 5 tokens
        self.y = torch.randint(0, 10, (n,), generator=g)

    def __len__(self):
        return len(self.y)

    def __getitem__(self, i):
        return self.x[i], self.y[i]


def test_train_model_checkpoint_and_resume(tmp_path):
    """two epochs of the finetune loop on the micro model (simulator backend): tuple checkpoints, resume-from-latest"""
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    torch.manual_seed(0)
    model = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.01)
    sch = torch.optim.lr_scheduler.StepLR(opt, 1, 0.5)
    save = str(tmp_path / "ViT_model_finetune_{epoch}.pth")
    tl = ft.train_model(model, _TinyFloatSet(10), opt, sch, torch.nn.CrossEntropyLoss(), batch_size=3, num_epochs=1, max_tolerant_epoch=3,
                        save_path=save, logger=LOG, num_workers=0)
    assert len(tl) == 1 and math.isfinite(tl[0]) and os.path.exists(save.format(epoch=1))
    ck = torch.load(save.format(epoch=1), weights_only=False)
    assert isinstance(ck, tuple) and len(ck) == 3 and "vit.layers.0.mlp.fc1.weight" in ck[0]
    # resume: a fresh model picks up epoch 1 and trains epoch 2 only
    model2 = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    opt2 = FusedAdamW(model2, lr=1e-3, weight_decay=0.01)
    sch2 = torch.optim.lr_scheduler.StepLR(opt2, 1, 0.5)
    tl2 = ft.train_model(model2, _TinyFloatSet(10), opt2, sch2, torch.nn.CrossEntropyLoss(), batch_size=3, num_epochs=2, max_tolerant_epoch=3,
                         save_path=save, logger=LOG, num_workers=0)
    assert len(tl2) == 1 and os.path.exists(save.format(epoch=2))
    # optimizer state (incl. the fused step counter) resumed; the scheduler state was saved BEFORE its epoch-end step
    # (reference order: save -> early_exit -> scheduler.step(), finetune.py:249-268), so epoch 2 re-runs at the saved LR
    assert opt2._step == 2 * opt._step and abs(opt2.param_groups[0]["lr"] - 5e-4) < 1e-12


def test_ntrain_module_and_trainer(tmp_path):
    """ViTLModule + AugmentedDataset + Trainer for one epoch on synthetic uint8 thumbnails (simulator backend)"""
    from touhouimageclassification_amd.ViT import ntrain
    be = SimBackend()
    lm = ntrain.ViTLModule(10, False, "tiny", lr=1e-3, weight_decay=0.01, enable_mixup=True, full_finetune=True, backend=be)
    ds = ntrain.SyntheticU8(5, 10, size=64, seed=1)
    data = ntrain.AugmentedDataset(batch_size=2, train_split=0.8, num_workers=0, backend=be, dataset=ds, test_dataset=ds)
    tr = ntrain.Trainer(max_epochs=1, checkpoint_dir=str(tmp_path), train_id="t", patience=3, device=torch.device("cpu"), log=lambda s: None)
    hist = tr.fit(lm, data)
    assert len(hist) == 1 and math.isfinite(hist[0]["train_loss"]) and 0.0 <= hist[0]["val_acc"] <= 1.0
    assert any(f.endswith(".ckpt") for f in os.listdir(tmp_path))
    assert 0.0 <= tr.test(lm, data) <= 1.0
    # frozen base model (full_finetune=False, ntrain.py:35-37): only the classifier is trainable
    lm2 = ntrain.ViTLModule(10, False, "tiny", lr=1e-3, weight_decay=0.01, full_finetune=False, backend=be)
    assert [n for n, p in lm2.named_parameters() if p.requires_grad] == ["vit.classifier.weight", "vit.classifier.bias"]


def test_preprocess_front_end(tmp_path):
    """ImageFolder front-end: side files, dataset statistics (reference definition), device-side Resize+Normalize"""
    import numpy as np
    from PIL import Image
    from touhouimageclassification_amd.utils import preprocess as pp
    g = np.random.default_rng(0)
    for cls in ("alice", "marisa", "reimu"):
        os.makedirs(tmp_path / cls)
        for i in range(3):
            Image.fromarray(g.integers(0, 256, (40, 52, 3), dtype=np.uint8)).save(tmp_path / cls / f"{i}.png")
    be = SimBackend()
    ds = pp.get_dataset(str(tmp_path), (32, 32), backend=be)
    assert ds.classes == ["alice", "marisa", "reimu"] and len(ds) == 9
    assert pp.get_class_to_idx(str(tmp_path)) == {"alice": 0, "marisa": 1, "reimu": 2}
    meta = torch.load(tmp_path / pp.META_MEAN_STD_FILENAME, weights_only=False)
    # reference definition on the same staged thumbnails
    xs = torch.stack([ds[i][0] for i in range(9)])
    from touhouimageclassification_amd.aug import GpuAugment
    r = GpuAugment("none", 32, mean=(0, 0, 0), std=(1, 1, 1), backend=be)(xs).double().view(9, 3, -1)
    torch.testing.assert_close(meta["mean"], r.mean([0, 2]), atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(meta["std"], r.std([0, 2]), atol=1e-6, rtol=1e-6)
    y = ds.device_transform(xs[:2])
    ref = (r[:2].view(2, 3, 32, 32) - meta["mean"].view(1, 3, 1, 1)) / meta["std"].view(1, 3, 1, 1)
    torch.testing.assert_close(y.double(), ref, atol=1e-5, rtol=1e-5)


def test_serve_full_judge(tmp_path):
    """serve / full_judge on a tiny ViT through the simulator: CSV format, accuracy, tuple checkpoints"""
    import numpy as np
    from PIL import Image
    from touhouimageclassification_amd.utils import preprocess as pp
    from touhouimageclassification_amd.utils import serve as sv
    from touhouimageclassification_amd.ViT.model import ViT
    g = np.random.default_rng(1)
    data = tmp_path / "data"
    for cls in ("alice", "reimu"):
        os.makedirs(data / cls)
        for i in range(2):
            Image.fromarray(g.integers(0, 256, (48, 48, 3), dtype=np.uint8)).save(data / cls / f"{i}.png")
    be = SimBackend()
    tf = pp.get_transforms(str(data), (224, 224), backend=be)
    c2i = pp.get_class_to_idx(str(data)) if os.path.exists(data / pp.CLASS_TO_IDX_FILENAME) else {"alice": 0, "reimu": 1}
    torch.manual_seed(0)
    model = ViT(2, pretrained=False, model_name="tiny", backend=be)
    # tuple checkpoint round trip (finetune.py:249-258 format) through the 4.x-key-tolerant loader
    torch.save((model.state_dict(), {}), tmp_path / "ck.pth")
    m2 = ViT(2, pretrained=False, model_name="tiny", backend=be)
    ck = torch.load(tmp_path / "ck.pth", weights_only=False)
    m2.load_state_dict(ck[0])
    acc = sv.full_judge(m2, tf, c2i, image=str(data), device="cpu", output=str(tmp_path / "out.csv"), batch_size=3, staging=64)
    lines = open(tmp_path / "out.csv").read().strip().split("\n")
    assert lines[0] == "filename,predicted_class,confidence,actual_class,correct,path" and len(lines) == 5
    assert 0.0 <= acc <= 1.0 and abs(acc - sum(l.split(",")[4] == "True" for l in lines[1:]) / 4) < 1e-9
    pred, conf = sv.full_judge(m2, tf, c2i, image=str(data / "alice" / "0.png"), device="cpu", output="x", staging=64)
    assert pred in c2i and 0.0 < conf <= 1.0
    with pytest.raises(ValueError):
        sv.get_model("no-such-model", 2)
