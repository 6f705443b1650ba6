"""Data-parallel path on CPU: world_size 2 and 4 over gloo, micro model (32 px, 5 tokens) through the simulator backend.
Two ranks x B images with bucketed SUM all-reduce (mean folded into the loss gradient) must produce the same
gradients and the same AdamW update as one process on the concatenated 2B batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, compress=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.dist import BucketedGradSync
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    torch.manual_seed(100 + rank)   # replicas start DIFFERENT on purpose: broadcast must fix that
    model = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    sync = BucketedGradSync(model, compress=compress)
    sync.broadcast_parameters()
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2 * world, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (2 * world,), generator=g)
    xs, ys = x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2]
    buckets = []
    inner = model._bucket_hook
    model.register_bucket_hook(lambda name, gs: (buckets.append(name), inner(name, gs)))
    loss, _ = fused_train_step(model, opt, xs, ys, sync)
    torch.save({"grads": model._engine.grads.clone(), "params": model._engine.params.clone(), "loss": float(loss), "buckets": buckets},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_dp_ranks_match_single_process(tmp_path, world):
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    ranks = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    r0 = ranks[0]
    # replicas agree bit-for-bit after the all-reduce and the update
    for rk in ranks[1:]:
        assert torch.equal(r0["grads"], rk["grads"]) and torch.equal(r0["params"], rk["params"])
    assert r0["buckets"] == ["head", "layer1", "layer0", "embed"]
    # single-process reference on the global batch, starting from rank 0's initial weights
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    torch.manual_seed(100)
    model = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    opt = FusedAdamW(model, lr=1e-3, weight_decay=0.01)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2 * world, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (2 * world,), generator=g)
    loss, _ = fused_train_step(model, opt, x, y, None)
    ref = model._engine.grads
    rel = ((r0["grads"] - ref).norm() / ref.norm()).item()
    assert rel < 2e-2, rel     # same math; bf16 rounding of per-rank partial sums differs from the global-batch order
    assert abs(sum(rk["loss"] for rk in ranks) / world - float(loss)) < 1e-3


@pytest.mark.timeout(600)
def test_dp_bf16_buckets_stay_identical_and_close_to_fp32(tmp_path):
    """compress="bf16": the buckets cross the wire as bf16 -- replicas still bit-identical, gradients within bf16 rounding of the
    fp32 exchange"""
    world = 2
    (tmp_path / "bf").mkdir()
    (tmp_path / "fp").mkdir()
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path / "bf"), "bf16"), nprocs=world, join=True, start_method="spawn")
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path / "fp"), None), nprocs=world, join=True, start_method="spawn")
    b = [torch.load(tmp_path / "bf" / f"rank{r}.pt") for r in range(world)]
    f = torch.load(tmp_path / "fp" / "rank0.pt")
    assert torch.equal(b[0]["grads"], b[1]["grads"]) and torch.equal(b[0]["params"], b[1]["params"])
    rel = ((b[0]["grads"] - f["grads"]).norm() / f["grads"].norm()).item()
    assert 0 < rel < 8e-3, rel   # bf16 has 8 significant bits: each value moves by <= 2^-9 relative, twice


class _Recording(torch.utils.data.Dataset):
    """records which samples a rank really touched"""

    def __init__(self, ds, log):
        self.ds, self.log = ds, log

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        self.log.append(int(i))
        return self.ds[i]


def _fit_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ViT import ntrain
    be = SimBackend()
    ntrain.seed_everything(42)
    lm = ntrain.ViTLModule(10, False, "micro", lr=1e-3, weight_decay=0.01, enable_mixup=False, backend=be)
    seen = []
    ds = _Recording(ntrain.SyntheticU8(12, 10, size=40, seed=1), seen)
    data = ntrain.AugmentedDataset(batch_size=2, train_split=0.67, num_workers=0, image_size=(32, 32), backend=be, dataset=ds, test_dataset=ds)
    tr = ntrain.Trainer(max_epochs=2, checkpoint_dir=os.path.join(out_dir, "ck"), train_id="dp", patience=0, device=torch.device("cpu"),
                        log=lambda s: open(os.path.join(out_dir, f"log{rank}.txt"), "a").write(s + "\n"))
    hist = tr.fit(lm, data)
    split = list(data.train_dataset.indices), list(data.val_dataset.indices)
    torch.save({"params": lm.vit._engine.params.clone(), "hist": hist, "seen": seen, "split": split}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_trainer_fit_shards_the_epoch_and_keeps_replicas_identical(tmp_path):
    """Trainer.fit under torch.distributed -- rank-sharded loaders, reduced metrics, rank-0 checkpoints"""
    world = 2
    mp.start_processes(_fit_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{k}.pt", weights_only=False) for k in range(world)]
    assert torch.equal(r[0]["params"], r[1]["params"])                 # same broadcast start, same all-reduced gradients
    assert r[0]["hist"] == r[1]["hist"] and len(r[0]["hist"]) == 2      # metrics are global means, identical on every rank
    assert r[0]["split"] == r[1]["split"]
    train_idx, val_idx = r[0]["split"]
    # every epoch each rank touches half of the split, the two halves are disjoint, together they cover it
    per_epoch = (len(train_idx) + len(val_idx)) // world
    for e in range(2):
        a = [r[k]["seen"][e * per_epoch:(e + 1) * per_epoch] for k in range(world)]
        assert not set(a[0]) & set(a[1])
        assert set(a[0]) | set(a[1]) == set(train_idx) | set(val_idx)
    assert not os.path.exists(tmp_path / "log1.txt") and len(open(tmp_path / "log0.txt").read().strip().split("\n")) == 2
    assert len([f for f in os.listdir(tmp_path / "ck") if f.endswith(".ckpt")]) >= 1


# ---- ResNet (BASELINE config 4: "ResNet-50 ... DDP scaling 1 -> 8") -------------------------------------------------------------
def _resnet_inputs(world, per_rank=2, img=32):
    g = torch.Generator().manual_seed(9)
    return torch.randn(world * per_rank, 3, img, img, generator=g), torch.randint(0, 10, (world * per_rank,), generator=g)


def _resnet_step(model, xs, ys, sync, lr=5e-2):
    opt = torch.optim.SGD(model.parameters(), lr=lr)     # TIC/ResNet/train.py:240
    model.train()
    opt.zero_grad()
    loss = torch.nn.functional.cross_entropy(model(xs), ys)
    (loss * (sync.grad_scale if sync else 1.0)).backward()
    if sync:
        sync.wait()
    flat = model.__dict__["_flat_grad"].clone()
    opt.step()
    return float(loss), flat


def _resnet_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResNet.model import resnet18
    from touhouimageclassification_amd.dist import BucketedGradSync
    torch.manual_seed(100 + rank)   # replicas start DIFFERENT on purpose: broadcast_parameters must fix that (buffers included)
    model = resnet18(num_classes=10, backend=SimBackend())
    with torch.no_grad():
        model.bn1.running_mean.add_(rank)
    sync = BucketedGradSync(model)
    sync.broadcast_parameters()
    start = {k: v.clone() for k, v in model.state_dict().items()}
    fired = []
    inner = model._bucket_hook
    model.register_bucket_hook(lambda name, gs: (fired.append((name, gs.numel())), inner(name, gs)))
    x, y = _resnet_inputs(world)
    loss, flat = _resnet_step(model, x[2 * rank:2 * rank + 2], y[2 * rank:2 * rank + 2], sync)
    torch.save({"start": start, "grads": flat, "after": {k: v.clone() for k, v in model.state_dict().items()}, "loss": loss, "fired": fired,
                "buckets": model.buckets()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_resnet_dp_ranks_match_sum_of_shard_gradients(tmp_path):
    """TicResNet under BucketedGradSync, world 2 over gloo (simulator backend): six buckets fire in backward completion order and tile
    the flat gradient buffer; replicas end bit-identical in every PARAMETER; the all-reduced gradient equals the mean of the two
    single-process shard gradients.  BatchNorm uses PER-REPLICA batch statistics (DDP's default, no SyncBN -- SURVEY 8e): the running
    statistics differ between the ranks and equal those of a single process that saw only that rank's shard."""
    world = 2
    mp.start_processes(_resnet_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{k}.pt", weights_only=False) for k in range(world)]
    names = [n for n, _ in r[0]["fired"]]
    assert names == ["fc", "layer4", "layer3", "layer2", "layer1", "stem"] == [n for n, _, _ in r[0]["buckets"]]
    spans = sorted((a, b) for _, a, b in r[0]["buckets"])
    assert spans[0][0] == 0 and all(spans[i][1] == spans[i + 1][0] for i in range(5)) and spans[-1][1] == r[0]["grads"].numel()
    assert [n for _, n in r[0]["fired"]] == [b - a for _, a, b in r[0]["buckets"]]
    for k in r[0]["start"]:
        assert torch.equal(r[0]["start"][k], r[1]["start"][k]), k            # broadcast: parameters and buffers
    assert torch.equal(r[0]["grads"], r[1]["grads"])
    is_param = lambda k: not ("running_" in k or "num_batches" in k)         # noqa: E731
    assert all(torch.equal(r[0]["after"][k], r[1]["after"][k]) for k in r[0]["after"] if is_param(k))
    assert not torch.equal(r[0]["after"]["bn1.running_mean"], r[1]["after"]["bn1.running_mean"])   # per-replica statistics
    # single process, one shard at a time, from the broadcast start
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResNet.model import resnet18
    x, y = _resnet_inputs(world)
    shard = []
    for k in range(world):
        m = resnet18(num_classes=10, backend=SimBackend())
        m.load_state_dict(r[0]["start"])
        loss, flat = _resnet_step(m, x[2 * k:2 * k + 2], y[2 * k:2 * k + 2], None)
        assert abs(loss - r[k]["loss"]) < 1e-6
        for kk, v in m.state_dict().items():
            if not is_param(kk):
                assert torch.equal(v, r[k]["after"][kk]), kk               # this rank's own batch statistics
        shard.append(flat)
    torch.testing.assert_close(r[0]["grads"], 0.5 * shard[0] + 0.5 * shard[1], atol=1e-7, rtol=1e-5)


def _resnet_loop_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import logging
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResNet import train as rt
    from touhouimageclassification_amd.ResNet.model import resnet18
    torch.manual_seed(5 + rank)
    model = resnet18(num_classes=10, backend=SimBackend())
    opt = torch.optim.SGD(model.parameters(), lr=5e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=5, gamma=0.25)
    g = torch.Generator().manual_seed(3)
    seen = []

    class DS(torch.utils.data.Dataset):
        classes = [str(i) for i in range(10)]
        x, y = torch.randn(13, 3, 32, 32, generator=g), torch.randint(0, 10, (13,), generator=g)

        def __len__(self):
            return 13

        def __getitem__(self, i):
            seen.append(int(i))
            return self.x[i], self.y[i]

    save = os.path.join(out_dir, "ck", "ResNet_model_{epoch}.pth")
    tl = rt.train_model(model, DS(), opt, sched, torch.nn.CrossEntropyLoss(), batch_size=2, num_epochs=1, max_tolerant_epoch=3, save_path=save,
                        logger=logging.getLogger(f"dp{rank}"), num_workers=0)
    torch.save({"params": [p.detach().clone() for p in model.parameters()], "timeline": tl, "seen": seen}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_resnet_train_model_shards_epoch_under_dp(tmp_path):
    """ResNet/train.py::train_model under torch.distributed (reference harness: TIC/ResNet/train.py:82-208, single process): the
    train split is sharded by a DistributedSampler (13 - 1 = 12 samples, 6 per rank), the validation split WITHOUT padding (ONE sample:
    rank 1 validates nothing and still reports the global mean), the timeline is identical on every rank, rank 0 alone writes the
    checkpoint"""
    world = 2
    mp.start_processes(_resnet_loop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{k}.pt", weights_only=False) for k in range(world)]
    assert all(torch.equal(a, b) for a, b in zip(r[0]["params"], r[1]["params"]))
    assert r[0]["timeline"] == r[1]["timeline"] and len(r[0]["timeline"]) == 1
    assert os.listdir(tmp_path / "ck") == ["ResNet_model_1.pth"]
    train0, val0 = r[0]["seen"][:6], r[0]["seen"][6:]
    train1, val1 = r[1]["seen"][:6], r[1]["seen"][6:]
    assert len(set(train0) | set(train1)) == 12 and not set(train0) & set(train1) and len(val0) == 1 and val1 == []
    assert not set(val0) & (set(train0) | set(train1))


def _eval_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _evaluate_micro()
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _evaluate_micro():
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ViT import ntrain
    be = SimBackend()
    ntrain.seed_everything(42)
    lm = ntrain.ViTLModule(10, False, "micro", lr=1e-3, weight_decay=0.01, enable_mixup=False, backend=be)
    ds = ntrain.SyntheticU8(7, 10, size=40, seed=1)     # 7 % 2 != 0
    data = ntrain.AugmentedDataset(batch_size=2, train_split=0.0, num_workers=0, image_size=(32, 32), backend=be, dataset=ds, test_dataset=ds)
    data.setup("fit")
    tr = ntrain.Trainer(max_epochs=1, device=torch.device("cpu"), log=lambda s: None)
    return tr._evaluate(lm, data.val_dataloader(), data, "val")


@pytest.mark.timeout(600)
def test_dp_evaluation_counts_every_sample_once(tmp_path):
    """7 validation samples on 2 ranks -- the all-reduced val_loss / val_acc equal the single-process ones (a padded
    DistributedSampler would count one sample twice)"""
    world = 2
    mp.start_processes(_eval_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"rank{k}.pt", weights_only=False) for k in range(world)]
    single = _evaluate_micro()
    assert r[0] == r[1]
    assert abs(r[0][0] - single[0]) < 1e-6 and abs(r[0][1] - single[1]) < 1e-9, (r[0], single)
