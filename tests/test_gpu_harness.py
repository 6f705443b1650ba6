"""The harness surface of the reference ON THE GPU (libtic_hip.so underneath), against the CPU oracle on the same inputs:

  a13  train_step / validate_step (TIC/ViT/finetune.py:54-77), ViTLModule.training_step / validation_step / test_step (ntrain.py:43-66)
  a18  the ResNet twin (TIC/ResNet/train.py:47-66, :82-208)
  a19  train_model (finetune.py:93-268), Trainer.fit / Trainer.test as train_main drives them (ntrain.py:219-248)
  f2   serve / full_judge (TIC/utils/serve.py:83-114, :158-230)
  f3   the dense mixture of ViT experts + its loss (TIC/ResMoE/model.py:24-72, train.py:21-36), and the expert-parallel form on two ranks

Tolerances are the ones of tests/test_gpu_model.py: bf16 GEMM I/O against the fp32 oracle, logits atol 2e-2 + rtol 2e-2, loss rtol 1e-2
(trajectories: the band widens with Adam's sign-like first steps)."""
import logging
import math
import os
import socket

import numpy as np
import pytest
import torch

from tests import headroom as hr

from oracle import moe_oracle as mo
from oracle import vit_oracle as vo

pytestmark = pytest.mark.gpu
LOG = logging.getLogger("tic-gpu-harness")
DEV = torch.device("cuda")
TINY = vo.ViTSpec(**vo.VIT_TINY, num_labels=10)


def _tiny_params(golden_dir):
    gold = np.load(f"{golden_dir}/vit_tiny.npz")
    params = {k[len("param/"):]: torch.from_numpy(gold[k]).clone() for k in gold.files if k.startswith("param/")}
    return gold, params


def _cpu_params(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def _oracle_eval(params, x, y, spec=TINY):
    with torch.no_grad():
        logits = vo.forward(params, x, spec)
        return logits, float(vo.cross_entropy(logits, y)), int((logits.argmax(-1) == y).sum())


class _FloatSet(torch.utils.data.Dataset):
    classes = [str(i) for i in range(10)]

    def __init__(self, n, size=224, C=10, seed=3):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(n, 3, size, size, generator=g)
        self.y = torch.randint(0, C, (n,), generator=g)

    def __len__(self):
        return len(self.y)

    def __getitem__(self, i):
        return self.x[i], self.y[i]


# ---- a13: finetune.train_step / validate_step --------------------------------------------------------------------------------
@pytest.mark.parametrize("optim_kind", ["fused", "torch_adamw", "torch_adamw_gradscaler"])
def test_train_step_trajectory_and_validate_step_match_oracle(golden_dir, optim_kind):
    """four `train_step`s on one batch: the returned losses follow the oracle's AdamW trajectory; `validate_step` afterwards
    equals the oracle's loss / correct count at the trained weights.  The torch.optim.AdamW variants run AFTER `.to('cuda')`
    (stock optimizers must invalidate the bf16 operand copies), one of them through an ENABLED GradScaler
    (finetune.py:62-64: scale -> backward -> unscale + inf check -> step -> update)."""
    from touhouimageclassification_amd.ViT import finetune as ft
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    gold, params = _tiny_params(golden_dir)
    x, y = torch.from_numpy(gold["x"]), torch.from_numpy(gold["y"])
    model = ViT(10, pretrained=False, model_name="tiny")
    model.load_state_dict(params)
    model.to(DEV)
    lr = 1e-3
    opt = FusedAdamW(model, lr=lr, weight_decay=0.01) if optim_kind == "fused" else torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=0.01)
    scaler = torch.amp.GradScaler("cuda", init_scale=1024.0) if optim_kind.endswith("gradscaler") else None
    crit = torch.nn.CrossEntropyLoss()
    got = [ft.train_step(model, (x, y), opt, crit, scaler) for _ in range(4)]
    ref = {k: v.clone() for k, v in params.items()}
    mom = {k: torch.zeros_like(v) for k, v in ref.items()}
    var = {k: torch.zeros_like(v) for k, v in ref.items()}
    want = []
    for t in range(1, 5):
        _, l, g = vo.loss_and_grads(ref, x, y, TINY, emulate_autocast=True)
        want.append(float(l))
        for k in ref:
            vo.adamw_step(ref[k], g[k], mom[k], var[k], t, lr)
    hr.le("test_gpu_harness.py:89", abs(got[0] - want[0]), 1e-2 * want[0])
    for a, b in zip(got, want):
        hr.le("test_gpu_harness.py:91", abs(a - b), 0.08 * max(b, 0.2), ctx=(optim_kind, got, want))
    assert got[-1] < 0.85 * got[0]          # the weights the GEMMs use really moved (stale operand copies would freeze the loss)
    if scaler is not None:
        assert scaler.get_scale() == 1024.0 and scaler.is_enabled()   # no inf/nan was found, no growth yet
    vl, correct = ft.validate_step(model, (x, y), crit)
    _, o_loss, o_correct = _oracle_eval(_cpu_params(model), x, y)
    hr.le("validate_step loss vs oracle", abs(vl - o_loss), 1e-2 * max(o_loss, 0.1) + 1e-3)
    assert correct == o_correct


# ---- a19: finetune.train_model ----------------------------------------------------------------------------------------------
def test_train_model_epochs_checkpoints_resume(tmp_path, golden_dir):
    from touhouimageclassification_amd.ViT import finetune as ft
    from touhouimageclassification_amd.ViT.model import ViT
    from touhouimageclassification_amd.optim import FusedAdamW
    _, params = _tiny_params(golden_dir)
    ds = _FloatSet(20)
    save = str(tmp_path / "ViT_model_finetune_{epoch}.pth")

    def setup():
        m = ViT(10, pretrained=False, model_name="tiny")
        m.load_state_dict(params)
        m.to(DEV)
        o = FusedAdamW(m, lr=1e-3, weight_decay=0.01)
        return m, o, torch.optim.lr_scheduler.StepLR(o, 1, 0.5)

    m, o, s = setup()
    tl = ft.train_model(m, ds, o, s, torch.nn.CrossEntropyLoss(), batch_size=6, num_epochs=2, max_tolerant_epoch=3, save_path=save,
                        logger=LOG, num_workers=0)
    assert len(tl) == 2 and all(math.isfinite(v) for v in tl) and os.path.exists(save.format(epoch=2))
    ck = torch.load(save.format(epoch=2), map_location="cpu", weights_only=False)
    assert isinstance(ck, tuple) and len(ck) == 3
    # the recorded validation loss of epoch 2 is what the ORACLE computes on the held-out split with the checkpointed weights
    torch.manual_seed(0)
    _, val = torch.utils.data.random_split(ds, [len(ds) - len(ds) // 10, len(ds) // 10])
    xs = torch.stack([val[i][0] for i in range(len(val))])
    ys = torch.stack([val[i][1] for i in range(len(val))])
    _, o_loss, _ = _oracle_eval({k: v.float() for k, v in ck[0].items()}, xs, ys)
    hr.le("test_gpu_harness.py:128", abs(tl[-1] - o_loss), 1e-2 * max(o_loss, 0.1) + 2e-3, ctx=(tl, o_loss))
    # resume: a fresh model continues at epoch 3 from the epoch-2 file
    m2, o2, s2 = setup()
    tl2 = ft.train_model(m2, ds, o2, s2, torch.nn.CrossEntropyLoss(), batch_size=6, num_epochs=3, max_tolerant_epoch=3, save_path=save,
                         logger=LOG, num_workers=0)
    assert len(tl2) == 1 and os.path.exists(save.format(epoch=3)) and o2._step == o._step + 3   # 18 train samples / 6 = 3 steps per epoch


# ---- a13 / a19: ViTLModule steps + Trainer ------------------------------------------------------------------------------------
def test_vitlmodule_steps_and_trainer_fit(tmp_path):
    from touhouimageclassification_amd.ViT import ntrain
    ntrain.seed_everything(42)
    lm = ntrain.ViTLModule(10, False, "tiny", lr=1e-3, weight_decay=0.01, enable_mixup=True, full_finetune=True)
    ds = ntrain.SyntheticU8(24, 10, size=64, seed=1)
    data = ntrain.AugmentedDataset(batch_size=6, train_split=0.75, num_workers=0, dataset=ds, test_dataset=ds)
    tr = ntrain.Trainer(max_epochs=2, checkpoint_dir=str(tmp_path), train_id="gpu", patience=3, device=DEV, log=lambda s: None)
    hist = tr.fit(lm, data)
    assert len(hist) == 2 and all(math.isfinite(h["train_loss"]) and math.isfinite(h["val_loss"]) for h in hist)
    ckpts = [f for f in os.listdir(tmp_path) if f.endswith(".ckpt")]
    assert ckpts and all(f.startswith("checkpoint_gpu_epoch=") for f in ckpts)
    # validation_step / test_step on one device-transformed batch vs the oracle at the trained weights
    lm.eval()
    batch = next(iter(data.val_dataloader()))
    xb, yb = data.on_device(batch, "eval", DEV)
    with torch.no_grad():
        loss, acc = lm.validation_step((xb, yb), 0)
        tacc = lm.test_step((xb, yb), 0)
    o_logits, o_loss, o_correct = _oracle_eval(_cpu_params(lm.vit), xb.cpu(), yb.cpu())
    hr.le("test_gpu_harness.py:156", abs(float(loss) - o_loss), 1e-2 * max(o_loss, 0.1) + 2e-3)
    assert abs(float(acc) - o_correct / len(yb)) < 1e-6 and float(tacc) == float(acc)
    # the epoch metrics the Trainer recorded are the sample-weighted means of exactly these steps
    tot_l = tot_a = n = 0
    with torch.no_grad():
        for i, b in enumerate(data.val_dataloader()):
            xb, yb = data.on_device(b, "eval", DEV)
            l, a = lm.validation_step((xb, yb), i)
            tot_l, tot_a, n = tot_l + float(l) * len(yb), tot_a + float(a) * len(yb), n + len(yb)
    assert abs(hist[-1]["val_loss"] - tot_l / n) < 1e-5 and abs(hist[-1]["val_acc"] - tot_a / n) < 1e-6
    # training_step with MixUp/CutMix returns the soft-label CE of the mixed batch (ntrain.py:45-48)
    lm.train()
    xb, yb = data.on_device(next(iter(data.train_dataloader())), "train", DEV)
    loss = lm.training_step((xb, yb), 0)
    assert loss.requires_grad and math.isfinite(float(loss))
    loss.backward()
    assert all(p.grad is not None for p in lm.parameters())
    assert 0.0 <= tr.test(lm, data) <= 1.0
    # restore a reference-era Lightning checkpoint (4.x names under the extra `vit.` prefix) into the module on the GPU
    sd = {}
    for k, v in lm.vit.state_dict().items():
        k = k.replace("vit.layers.", "vit.encoder.layer.").replace(".attention.q_proj.", ".attention.attention.query.")
        k = k.replace(".attention.k_proj.", ".attention.attention.key.").replace(".attention.v_proj.", ".attention.attention.value.")
        k = k.replace(".attention.o_proj.", ".attention.output.dense.").replace(".mlp.fc1.", ".intermediate.dense.").replace(".mlp.fc2.", ".output.dense.")
        sd["vit." + k] = v.detach().cpu() * 0.5
    lm.load_state_dict(sd)
    lm.eval()
    with torch.no_grad():
        got = lm(xb).logits.cpu()
    torch.testing.assert_close(got, vo.forward(_cpu_params(lm.vit), xb.cpu(), TINY), atol=2e-2, rtol=2e-2)
    assert torch.equal(lm.vit.state_dict()["classifier.weight"].cpu(), sd["vit.classifier.weight"])


# ---- a18: the ResNet harness ---------------------------------------------------------------------------------------------------
def test_resnet_harness_steps_and_train_model(tmp_path):
    """ResNet/train.py on the GPU, every comparison ONE step deep (teacher forcing): before each `train_step` the model's own
    state_dict (weights + BatchNorm buffers) is loaded into the fp32 CPU oracle, and that step's loss, every parameter gradient,
    the SGD update and the BatchNorm running statistics are compared with the oracle's at the SAME state.  A free-running
    3-step trajectory of a BatchNorm net under SGD lr 5e-2 (TIC/ResNet/train.py:240) multiplies any bf16-level difference by
    7-20x per step (round 2's red test: 0.1 % -> 2.4 % -> 17 %) and cannot tell a kernel bug from rounding; one step deep the
    difference is the storage rounding itself.  Init: the reference's own, with the trained-like residual gains of the
    well-conditioned goldens (tests/resnet_checks.build(damp=0.25)).  Tolerances = the fixed ones of
    test_step_matches_reference_golden_fixed_tolerances, ~2x what the bf16-emulating oracle shows on this case (loss 0.14 %,
    gradient norms 5.0 %, classifier gradients 2.1 %, cosines 0.958; tools: the same loop with emulate_bf16=True).
    Order-nondeterministic reductions on this path -- the fp32 atomics of `bn_stats` / `bn_bwd_reduce` (<= 512 row splits), of the
    split-M / stream-K weight gradients and of the classifier gradient -- are bounded by measuring them: the same step run twice
    from the same state must agree within a tenth of the loss band and to 5e-3 (relative L2) on every gradient tensor."""
    from tests import resnet_checks as rc
    from touhouimageclassification_amd.ResNet import train as rt
    from touhouimageclassification_amd.ResNet.model import resnet18
    C, img = 10, 64
    rc.check_teacher_forced_steps(rt, None, DEV, C=C, B=32, img=img, tol=dict(loss=1e-2, gnorm=0.14, fc=0.05, cos=0.90, stats=1.5e-2, rerun=1e-4))
    # the loop: two epochs, (model_sd, optim_sd, sched_sd) checkpoints, scheduler stepped per epoch
    ds = _FloatSet(40, size=img, C=C, seed=5)
    m2, o2, s2, c2 = rt.build_reference_setup(C, lr=5e-2, arch=resnet18)
    m2.to(DEV)
    save = str(tmp_path / "ResNet_model_{epoch}.pth")
    tl = rt.train_model(m2, ds, o2, s2, c2, batch_size=12, num_epochs=2, max_tolerant_epoch=3, save_path=save, logger=LOG, num_workers=0)
    assert len(tl) == 2 and all(math.isfinite(v) for v in tl) and os.path.exists(save.format(epoch=2))
    ck = torch.load(save.format(epoch=2), map_location="cpu", weights_only=False)
    assert isinstance(ck, tuple) and len(ck) == 3 and "layer4.1.bn2.running_var" in ck[0] and int(ck[0]["bn1.num_batches_tracked"]) == 6
    assert abs(s2.get_last_lr()[0] - 5e-2) < 1e-12 and s2.last_epoch == 2


# ---- f2: serve / full_judge ----------------------------------------------------------------------------------------------------
def test_serve_and_full_judge_match_oracle(tmp_path):
    from PIL import Image
    from touhouimageclassification_amd.utils import preprocess as pp
    from touhouimageclassification_amd.utils import serve as sv
    from touhouimageclassification_amd.ViT.model import ViT
    g = np.random.default_rng(1)
    data = tmp_path / "data"
    classes = ["alice", "cirno", "marisa", "reimu", "sakuya", "youmu"]
    for cls in classes:
        os.makedirs(data / cls)
        for i in range(3):
            Image.fromarray(g.integers(0, 256, (72, 60, 3), dtype=np.uint8)).save(data / cls / f"{i}.png")
    tf = pp.get_transforms(str(data), (224, 224))
    c2i = pp.get_class_to_idx(str(data))
    assert c2i == {c: i for i, c in enumerate(classes)}
    torch.manual_seed(0)
    model = ViT(len(classes), pretrained=False, model_name="tiny")
    with torch.no_grad():
        model.classifier.weight.mul_(8.0)   # widen the logit margins of the random-init head
    torch.save((model.state_dict(), {}), tmp_path / "ck.pth")   # finetune.py:249-258 tuple format
    m2 = ViT(len(classes), pretrained=False, model_name="tiny")
    m2.load_state_dict(torch.load(tmp_path / "ck.pth", weights_only=False)[0])
    m2.to(DEV)
    out_csv = str(tmp_path / "out.csv")
    acc = sv.full_judge(m2, tf, c2i, image=str(data), device="cuda", output=out_csv, batch_size=7, staging=64)
    lines = open(out_csv).read().strip().split("\n")
    assert lines[0] == "filename,predicted_class,confidence,actual_class,correct,path" and len(lines) == 1 + 18
    # oracle on the SAME preprocessed pixels (the HIP resize/normalise is checked against its own oracle in test_gpu_ops)
    spec = vo.ViTSpec(**vo.VIT_TINY, num_labels=len(classes))
    params = _cpu_params(m2)
    idx_to_class = {v: k for k, v in c2i.items()}
    n_ok = n_decided = 0
    for ln in lines[1:]:
        fname, pred, conf, actual, correct, path = ln.split(",")
        raw = sv._load_u8(path, 64).unsqueeze(0).to(DEV)
        with torch.no_grad():
            xin = tf(raw)
            hip = m2(xin).logits[0].cpu()
            ref = vo.forward(params, xin.cpu(), spec)[0]
        torch.testing.assert_close(hip, ref, atol=4e-2, rtol=4e-2)   # the head was widened 8x above and the bf16 error with it: twice the usual band (unwidened: 1/4 of it)
        assert pred == idx_to_class[int(hip.argmax())]                              # the CSV row is this forward's argmax ...
        assert abs(float(conf) - float(torch.softmax(ref, -1)[c2i[pred]])) < 2e-2   # ... and the oracle's confidence for it
        srt = ref.sort(descending=True).values
        if srt[0] - srt[1] > 2 * float((hip - ref).abs().max()) + 1e-3:             # top-1 decided by more than the observed difference
            assert pred == idx_to_class[int(ref.argmax())], ln
            n_decided += 1
        assert actual == os.path.basename(os.path.dirname(path)) and correct == str(pred == actual)
        n_ok += pred == actual
    assert n_decided >= 9, n_decided
    hr.le("test_gpu_harness.py:270", abs(acc - n_ok / 18), 1e-9)
    # single image -> (class, confidence) through `serve`
    p0 = str(data / "reimu" / "0.png")
    pred, conf = sv.full_judge(m2, tf, c2i, image=p0, device="cuda", output="x", staging=64)
    row = [l for l in lines[1:] if l.endswith(p0)][0].split(",")
    assert pred == row[1] and abs(conf - float(row[2])) < 1e-3


# ---- f3: mixture of ViT experts --------------------------------------------------------------------------------------------------
def test_moe_kernels_on_gpu():
    from tests import kernel_checks as kc
    from touhouimageclassification_amd._lib import call, current_stream
    kc.check_moe_ops(kc.Env("cuda", call, stream=current_stream), B=7, E=8, K=2, C=120)
    kc.check_moe_ops(kc.Env("cuda", call, stream=current_stream, seed=2), B=300, E=64, K=8, C=1000)


def _moe(E, C, seed=11):
    from touhouimageclassification_amd.ResMoE.model import make_ViTMoE
    torch.manual_seed(seed)
    return make_ViTMoE(num_classes=C, num_experts=E, top_k=2, gateway_t=0.01, pretrained=False, model_name="tiny", gate_pretrained=False,
                       gate_model_name="tiny")


def test_sparse_dispatch_equals_dense_mixture_at_config5_size():
    """BASELINE config 5 at the reference's size on one GPU -- 8 ViT-B/16 experts + a ViT-B/16 gate, 120 classes, top-2, 32 images
    (TIC/ResMoE/model.py:60-72, train.py:150-176).  `sparse=True` evaluates each expert on the ~8 images routed to it; the reference
    evaluates all 8 on all 32 and multiplies six of them by exact zeros.  Same mixture: identical routing, gate weights equal, logits
    and the gradients of every expert head and of the gate equal up to the bf16 rounding of GEMMs that ran at another row count."""
    from touhouimageclassification_amd.ResMoE.model import make_ViTMoE
    from touhouimageclassification_amd.ResMoE import train as mt
    E, C, B = 8, 120, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, 224, 224, generator=g).to(DEV)
    tgt = torch.nn.functional.one_hot(torch.randint(0, C, (B,), generator=g), C).float().to(DEV)
    res = []
    for sparse in (False, True):
        torch.manual_seed(21)
        m = make_ViTMoE(num_classes=C, num_experts=E, top_k=2, gateway_t=0.01, pretrained=False, model_name="google/vit-base-patch16-224",
                        gate_pretrained=False).to(DEV)
        m.eval()              # deterministic gate (training mode adds noise to the gate logits)
        with torch.no_grad():   # a freshly initialised gate head sends (nearly) every image to the same two experts: spread the routing
            w = torch.randn(m.gate.vit.classifier.weight.shape, generator=torch.Generator().manual_seed(5)) * 2.0
            m.gate.vit.classifier.weight.copy_(w.to(DEV))
        m.sparse = sparse
        logits, gw, idx = m(x)
        mt.total_loss(logits, tgt, gw, idx).backward()
        grad = lambda p: torch.zeros_like(p) if p.grad is None else p.grad.clone()   # noqa: E731  (an expert nobody was routed to never runs)
        res.append((logits.detach().float(), gw.detach(), idx, [grad(e.classifier.weight) for e in m.experts], grad(m.gate.vit.classifier.weight)))
        del m
        torch.cuda.empty_cache()
    d, s = res
    assert torch.equal(d[2], s[2]), "routing differs"
    used = torch.bincount(d[2].flatten(), minlength=E)
    print("images per expert:", used.tolist())
    assert int((used > 0).sum()) >= 6 and int(used.max()) <= 24, "routing collapsed: the case would be vacuous"
    torch.testing.assert_close(s[1], d[1], atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(s[0], d[0], atol=2e-2 * max(1.0, float(d[0].abs().max())), rtol=2e-2)
    scale = max(float(b.norm()) for b in d[3])
    for k, (a, b) in enumerate(zip(s[3], d[3])):
        hr.le(f"expert {k} head gradient, sparse vs dense (rel L2)", float((a - b).norm()), 0.02 * float(b.norm()) + 1e-3 * scale)
    hr.le("gate head gradient, sparse vs dense (rel L2)", float((s[4] - d[4]).norm() / (d[4].norm() + 1e-12)), 0.02)


def test_dense_moe_training_step_matches_oracle():
    """config 5's model, dense form: gate ViT + E expert ViTs + gate/combine/loss kernels, one `training_step` + SGD on the GPU.
    Reference: every ViT through oracle/vit_oracle.py, the mixture arithmetic through oracle/moe_oracle.py, autograd end to end."""
    from touhouimageclassification_amd.ResMoE import train as mt
    E, C, B = 3, 10, 4
    moe = _moe(E, C).to(DEV)
    moe.eval()      # deterministic gate (training mode adds 0.01 N(0,1) to the gate logits; covered by check_moe_ops)
    opt = torch.optim.SGD(moe.parameters(), lr=5e-2)   # train.py:176
    mod = mt.ResMoETrainerModule(moe, opt)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (B,), generator=g)
    gate_p = _cpu_params(moe.gate.vit)
    exp_p = [_cpu_params(e) for e in moe.experts]
    loss = mod.training_step((x.to(DEV), y.to(DEV)), 0)
    loss.backward()
    # oracle
    spec_e, spec_g = vo.ViTSpec(**vo.VIT_TINY, num_labels=C), vo.ViTSpec(**vo.VIT_TINY, num_labels=E)
    gl = {k: v.clone().requires_grad_(True) for k, v in gate_p.items()}
    el = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in exp_p]
    scores = vo.forward(gl, x, spec_g)
    tw, ti = mo.gate(scores, None, 2)
    gw = mo.scatter(tw, ti, E)
    eo = torch.stack([vo.forward(p, x, spec_e) for p in el], 1)
    ref_logits = mo.combine(eo, gw)
    ref_loss = mo.total_loss(ref_logits, torch.nn.functional.one_hot(y, C).float(), gw)
    ref_loss.backward()
    with torch.no_grad():
        logits, gate_w, idx = moe(x.to(DEV))
    srt = scores.detach().sort(-1, descending=True).values
    assert bool(((srt[:, 1] - srt[:, 2]) > 2e-2).all()), "test data: routing must be decided by more than the logit tolerance"
    assert torch.equal(idx.cpu(), ti)
    torch.testing.assert_close(gate_w.cpu(), gw.detach(), atol=1e-2, rtol=1e-2)
    torch.testing.assert_close(logits.cpu(), ref_logits.detach(), atol=2e-2, rtol=2e-2)
    hr.le("test_gpu_harness.py:327", abs(float(loss) - float(ref_loss)), 1e-2 * abs(float(ref_loss)) + 1e-3)
    # gradients: every expert and the gate, per-parameter norms (the test_gpu_model criterion)
    for name, got_m, ref_p in [("gate", moe.gate.vit, gl)] + [(f"expert{i}", moe.experts[i], el[i]) for i in range(E)]:
        gmax = max(v.grad.norm().item() for v in ref_p.values())
        for k, p in got_m.named_parameters():
            r = ref_p[k].grad
            hr.le("test_gpu_harness.py:333", (p.grad.cpu() - r).norm().item(), 0.08 * r.norm().item() + 3e-3 * gmax, ctx=(name, k))
    opt.step()
    with torch.no_grad():
        after, _, _ = moe(x.to(DEV))
    assert not torch.equal(after, logits)     # SGD through the stock optimizer reached the GEMM operand copies
    mod.validation_step((x.to(DEV), y.to(DEV)), 0)
    assert set(mod.logged) >= {"train_loss", "val_balance_loss", "val_classification_loss", "val_accuracy"}


def _ep_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)   # two ranks on ONE GPU: RCCL refuses a duplicate device
    from touhouimageclassification_amd.ResMoE.model import ExpertParallelMoE
    from touhouimageclassification_amd.ResMoE import train as mt
    E, C, B = 2, 10, 2
    dense = _moe(E, C).to(DEV)
    ep = ExpertParallelMoE(dense.experts[rank], dense.gate, C)
    ep.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E * B, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (E * B,), generator=g)
    xs, ys = x[rank * B:(rank + 1) * B].to(DEV), y[rank * B:(rank + 1) * B].to(DEV)
    logits, gw, idx = ep(xs)
    loss = mt.total_loss(logits, torch.nn.functional.one_hot(ys, C).float(), gw, idx) / world
    loss.backward()
    ep.sync_gate_gradients()
    torch.cuda.synchronize()
    torch.save({"logits": logits.detach().cpu(), "expert_grad": dense.experts[rank].classifier.weight.grad.cpu().clone(),
                "gate_grad": dense.gate.vit.classifier.weight.grad.cpu().clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_expert_parallel_two_ranks_match_dense_on_gpu(tmp_path):
    """ExpertParallelMoE with one expert per rank on CUDA tensors (image all-gather, logits all-to-all and their reverses in
    backward, gate-gradient averaging) against the dense model in this process"""
    import torch.multiprocessing as mp
    from touhouimageclassification_amd.ResMoE import train as mt
    E, C, B = 2, 10, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(_ep_worker, args=(E, port, str(tmp_path)), nprocs=E, join=True, start_method="spawn")
    dense = _moe(E, C).to(DEV)
    dense.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E * B, 3, 224, 224, generator=g).to(DEV)
    y = torch.randint(0, C, (E * B,), generator=g).to(DEV)
    logits, gw, idx = dense(x)
    loss = sum(mt.total_loss(logits[r * B:(r + 1) * B], torch.nn.functional.one_hot(y[r * B:(r + 1) * B], C).float(), gw[r * B:(r + 1) * B],
                             idx[r * B:(r + 1) * B]) for r in range(E)) / E
    loss.backward()
    for r in range(E):
        got = torch.load(tmp_path / f"r{r}.pt")
        torch.testing.assert_close(got["logits"], logits[r * B:(r + 1) * B].detach().cpu(), atol=1e-2, rtol=1e-2)   # expert ran B*E rows vs B*E rows: same tiles
        ref = dense.experts[r].classifier.weight.grad.cpu()
        hr.le("test_gpu_harness.py:391", (got["expert_grad"] - ref).norm(), 0.03 * ref.norm() + 1e-6)
        refg = dense.gate.vit.classifier.weight.grad.cpu() / E
        hr.le("test_gpu_harness.py:393", (got["gate_grad"] - refg).norm(), 0.05 * refg.norm() + 1e-6)


def _sparse_ep_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)   # two ranks on ONE GPU: RCCL refuses a duplicate device
    from touhouimageclassification_amd.ResMoE.model import SparseExpertParallelMoE, make_ViTMoE
    from touhouimageclassification_amd.ResMoE import train as mt
    E, C, B = 2, 10, 3
    torch.manual_seed(11)
    dense = make_ViTMoE(num_classes=C, num_experts=E, top_k=1, gateway_t=0.01, pretrained=False, model_name="tiny", gate_pretrained=False,
                        gate_model_name="tiny").to(DEV)
    ep = SparseExpertParallelMoE(dense.experts[rank], dense.gate, C)
    ep.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E * B, 3, 224, 224, generator=g)
    y = torch.randint(0, C, (E * B,), generator=g)
    xs, ys = x[rank * B:(rank + 1) * B].to(DEV), y[rank * B:(rank + 1) * B].to(DEV)
    logits, gw, idx = ep(xs)
    loss = mt.total_loss(logits, torch.nn.functional.one_hot(ys, C).float(), gw, idx) / world
    loss.backward()
    ep.sync_gate_gradients()
    torch.cuda.synchronize()
    torch.save({"logits": logits.detach().cpu(), "rows": ep.last_rows, "idx": idx.cpu(),
                "expert_grad": dense.experts[rank].classifier.weight.grad.cpu().clone(),
                "gate_grad": dense.gate.vit.classifier.weight.grad.cpu().clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sparse_expert_parallel_two_ranks_match_dense_on_gpu(tmp_path):
    """SparseExpertParallelMoE (top-1 of 2 experts: each image travels to ONE rank, the expert there runs on the rows it received,
    the logits travel back) on CUDA tensors against the dense model in this process"""
    import torch.multiprocessing as mp
    from touhouimageclassification_amd.ResMoE.model import make_ViTMoE
    from touhouimageclassification_amd.ResMoE import train as mt
    E, C, B = 2, 10, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(_sparse_ep_worker, args=(E, port, str(tmp_path)), nprocs=E, join=True, start_method="spawn")
    torch.manual_seed(11)
    dense = make_ViTMoE(num_classes=C, num_experts=E, top_k=1, gateway_t=0.01, pretrained=False, model_name="tiny", gate_pretrained=False,
                        gate_model_name="tiny").to(DEV)
    dense.eval()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E * B, 3, 224, 224, generator=g).to(DEV)
    y = torch.randint(0, C, (E * B,), generator=g).to(DEV)
    logits, gw, idx = dense(x)
    loss = sum(mt.total_loss(logits[r * B:(r + 1) * B], torch.nn.functional.one_hot(y[r * B:(r + 1) * B], C).float(), gw[r * B:(r + 1) * B],
                             idx[r * B:(r + 1) * B]) for r in range(E)) / E
    loss.backward()
    got = [torch.load(tmp_path / f"r{r}.pt") for r in range(E)]
    assert [gt["rows"] for gt in got] == [int((idx == e).sum()) for e in range(E)] and sum(gt["rows"] for gt in got) == E * B
    for r in range(E):
        assert torch.equal(got[r]["idx"], idx[r * B:(r + 1) * B].cpu())
        torch.testing.assert_close(got[r]["logits"], logits[r * B:(r + 1) * B].detach().cpu(), atol=2e-2, rtol=2e-2)   # different batch sizes: other tiles
        ref = dense.experts[r].classifier.weight.grad.cpu()
        hr.le(f"sparse EP expert {r} head gradient", float((got[r]["expert_grad"] - ref).norm()), 0.06 * float(ref.norm()) + 1e-6)
        refg = dense.gate.vit.classifier.weight.grad.cpu() / E
        hr.le(f"sparse EP gate gradient (rank {r})", float((got[r]["gate_grad"] - refg).norm()), 0.06 * float(refg.norm()) + 1e-6)


# ---- a15: the augmentation presets on the GPU, as distributions ----------------------------------------------------------------
def test_augmentation_presets_on_gpu_have_torchvision_statistics():
    """`GpuAugment` end to end (batched parameter sampling on the host + `tic_augment` on the GPU) on 1024 copies of one asymmetric
    colour pattern: RandomGrayscale p = 0.2 (grey-only preset, ntrain.py:97-102); RandomHorizontalFlip p = 0.5 and RandomErasing
    p = 0.5 with an area in (0.02, 0.33) of the image (generalization preset, ntrain.py:122-128); no augmentation = resize +
    normalise (ntrain.py:132-136).  Parity of the pixels themselves: tests/golden/aug_cases.npz."""
    from touhouimageclassification_amd.aug import GpuAugment, IMAGENET_MEAN, IMAGENET_STD
    B, S = 1024, 64
    img = torch.zeros(96, 96, 3, dtype=torch.uint8)
    img[:, :48, 0] = 220      # left half red, right half blue: a flip swaps them
    img[:, 48:, 2] = 220
    img[:, :, 1] = 60
    raw = img.unsqueeze(0).repeat(B, 1, 1, 1).to(DEV)
    mean = torch.tensor(IMAGENET_MEAN, device=DEV).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=DEV).view(1, 3, 1, 1)
    # grayscale
    px = GpuAugment("grey", S, seed=1)(raw) * std + mean
    gray = ((px[:, 0] - px[:, 1]).abs().amax((1, 2)) < 2e-3) & ((px[:, 1] - px[:, 2]).abs().amax((1, 2)) < 2e-3)
    assert 0.15 < gray.float().mean().item() < 0.25, gray.float().mean().item()
    # erasing + flip (random resized crop on: only crops that straddle the colour boundary tell left from right)
    norm = GpuAugment("generalization", S, seed=2)(raw)
    px = norm * std + mean
    erased = px.abs().amax(1) < 1e-3      # RandomErasing(value = 0) sits BEFORE ToTensor / Normalize in the reference's Compose: pixel value 0
    frac = erased.float().mean((1, 2))
    has = frac > 0
    assert 0.44 < has.float().mean().item() < 0.56, has.float().mean().item()
    assert frac[has].min().item() > 0.012 and frac[has].max().item() < 0.36
    w = (~erased).float()
    red_l = (px[:, 0, :, : S // 2] * w[:, :, : S // 2]).sum((1, 2)) / w[:, :, : S // 2].sum((1, 2)).clamp_min(1)
    red_r = (px[:, 0, :, S // 2:] * w[:, :, S // 2:]).sum((1, 2)) / w[:, :, S // 2:].sum((1, 2)).clamp_min(1)
    decided = (red_l - red_r).abs() > 0.2
    assert decided.float().mean().item() > 0.5
    flipped = (red_r > red_l)[decided]
    assert 0.43 < flipped.float().mean().item() < 0.57, flipped.float().mean().item()
    # no augmentation: the pattern comes through resized and normalised
    plain = GpuAugment("none", S)(raw[:4]) * std + mean
    assert (plain[:, 0, :, : S // 2 - 2] > 0.8).all() and (plain[:, 2, :, S // 2 + 2:] > 0.8).all() and (plain[:, 1] - 60 / 255).abs().max() < 2e-2
