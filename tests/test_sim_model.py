"""Whole-model check on CPU: the product's Python surface (ViT factory, module tree, engine, phase
drivers in C) driven through the simulator build, against the oracle and the HF-pinned golden file."""
import numpy as np
from tests import kernel_checks as kc
import pytest
import torch

from oracle import vit_oracle as vo
from tests.simlib import SimBackend
from touhouimageclassification_amd.ViT.model import ViT


def _tiny(gold):
    spec = vo.ViTSpec(**vo.VIT_TINY, num_labels=10)
    params = {k[len("param/"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("param/")}
    return spec, params


def test_state_dict_surface():
    m = ViT(10, pretrained=False, model_name="tiny", backend=SimBackend())
    spec = vo.ViTSpec(**vo.VIT_TINY, num_labels=10)
    names = [n for n, _ in vo.param_shapes(spec)]
    sd = {k: v.clone() for k, v in m.state_dict().items()}   # state_dict() aliases the flat buffer
    assert set(sd.keys()) == set(names)
    for n, shp in vo.param_shapes(spec):
        assert tuple(sd[n].shape) == shp
    assert m.config.image_size == 224
    assert sum(p.numel() for p in m.parameters()) == sum(int(np.prod(s)) for _, s in vo.param_shapes(spec))
    assert len(list(m.base_model.parameters())) == len(names) - 2
    # 4.x checkpoint keys + Lightning prefix load too
    old = {}
    for k, v in sd.items():
        k = k.replace("vit.layers.", "vit.encoder.layer.").replace(".attention.q_proj.", ".attention.attention.query.")
        k = k.replace(".attention.k_proj.", ".attention.attention.key.").replace(".attention.v_proj.", ".attention.attention.value.")
        k = k.replace(".attention.o_proj.", ".attention.output.dense.").replace(".mlp.fc1.", ".intermediate.dense.").replace(".mlp.fc2.", ".output.dense.")
        old["vit." + k] = v + 1
    m.load_state_dict(old)
    torch.testing.assert_close(m.state_dict()["vit.layers.1.mlp.fc2.bias"], sd["vit.layers.1.mlp.fc2.bias"] + 1)
    with pytest.raises(FileNotFoundError):
        ViT(10, pretrained=True, model_name="google/vit-base-patch16-224", backend=SimBackend())
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 224, 224))
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 32, 32))


def test_tiny_vit_step_matches_golden(golden_dir):
    gold = np.load(f"{golden_dir}/vit_tiny.npz")
    spec, params = _tiny(gold)
    m = ViT(10, pretrained=False, model_name="tiny", backend=SimBackend())
    m.load_state_dict(params)
    B = 2
    x = torch.from_numpy(gold["x"][:B])
    y = torch.from_numpy(gold["y"][:B])
    # oracle with the bf16 rounding points of autocast == what the kernels compute
    o_logits, o_loss, o_grads = vo.loss_and_grads(params, x, y, spec, emulate_autocast=True)
    f_logits, _, f_grads = vo.loss_and_grads(params, x, y, spec)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-5, weight_decay=0.01)
    opt.zero_grad()
    logits = m(x).logits
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    torch.testing.assert_close(logits, o_logits, atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(logits, f_logits, atol=3e-2, rtol=3e-2)          # fp32 reference path (golden == HF)
    torch.testing.assert_close(f_logits, torch.from_numpy(gold["logits_hard"][:B]), atol=1e-5, rtol=1e-5)
    gmax = max(g.norm().item() for g in f_grads.values())
    for k, p in m.named_parameters():
        ref = f_grads[k]
        err = (p.grad - ref).norm().item()
        assert err <= 0.08 * ref.norm().item() + 2e-3 * gmax, (k, err, ref.norm().item())
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    logits2 = m(x).logits   # weights refreshed automatically after the optimizer step
    assert not torch.equal(logits2, logits)
    kc.check_refresh_weights(m)
    for k, p in m.named_parameters():
        assert (p.detach() - before[k]).abs().max() <= 1.2e-5 + 1e-7


def _old_names(sd, prefix=""):
    out = {}
    for k, v in sd.items():
        k = k.replace("vit.layers.", "vit.encoder.layer.").replace(".attention.q_proj.", ".attention.attention.query.")
        k = k.replace(".attention.k_proj.", ".attention.attention.key.").replace(".attention.v_proj.", ".attention.attention.value.")
        k = k.replace(".attention.o_proj.", ".attention.output.dense.").replace(".mlp.fc1.", ".intermediate.dense.").replace(".mlp.fc2.", ".output.dense.")
        out[prefix + k] = v
    return out


def test_stock_optimizer_after_relink_refreshes_operand_copies():
    """after `.to()` re-points every Parameter at a new flat buffer, the Parameters keep their own version
    counters; an in-place update by a stock torch optimizer must still invalidate the bf16 GEMM-operand copies."""
    torch.manual_seed(0)
    m = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    e = m._engine
    x = torch.randn(2, 3, 32, 32)
    y = torch.tensor([1, 2])
    m(x)                                   # operand copies built for the current weights
    e.replace_params(e.params.clone())     # what `.to(device)` does: a NEW flat buffer ...
    m._relink()                            # ... and every Parameter re-pointed at it (p.data = view)
    opt = torch.optim.SGD(m.parameters(), lr=0.5)
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x).logits, y).backward()
        opt.step()
    with torch.no_grad():
        got = m(x).logits.clone()
        e.mark_weights_dirty()             # forced refresh
        want = m(x).logits
    assert torch.equal(got, want)
    # and the update really moved the logits (the check above is not vacuous)
    m2 = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    assert not torch.allclose(got, m2(x).logits.detach(), atol=1e-3)


def test_lightning_4x_checkpoint_loads_through_wrapper_module():
    """`ViTLModule.load_state_dict(ck['state_dict'])` with the reference's Lightning + transformers-4.x keys"""
    from touhouimageclassification_amd.ViT import ntrain
    be = SimBackend()
    lm = ntrain.ViTLModule(10, False, "micro", lr=1e-3, weight_decay=0.0, backend=be)
    src = ViT(10, pretrained=False, model_name="micro", backend=be)
    src.reset_parameters(seed=7)
    ck = _old_names({k: v.clone() for k, v in src.state_dict().items()}, prefix="vit.")   # vit.vit.encoder.layer.0.attention.attention.query...
    assert any(".attention.attention.query." in k for k in ck)
    lm.load_state_dict(ck)
    for k, v in src.state_dict().items():
        assert torch.equal(lm.vit.state_dict()[k], v), k
    x = torch.randn(1, 3, 32, 32)
    with torch.no_grad():
        assert torch.equal(lm(x).logits, src(x).logits)


def test_stale_activations_are_detected_not_differentiated():
    """a second forward at the same batch size before the first backward must raise, unless the engine was
    told to keep two graphs alive -- in which case both backwards are right."""
    torch.manual_seed(1)
    m = ViT(10, pretrained=False, model_name="micro", backend=SimBackend())
    x1, x2 = torch.randn(2, 3, 32, 32), torch.randn(2, 3, 32, 32)
    l1 = m(x1).logits.sum()
    m(x2)
    with pytest.raises(RuntimeError, match="activations saved by this forward are gone"):
        l1.backward()
    # reference gradients, one graph at a time
    refs = []
    for x in (x1, x2):
        m.zero_grad()
        m(x).logits.square().sum().backward()
        refs.append(m._engine.grads.clone())
    m._engine.live_graphs = 2
    m.zero_grad()
    a = m(x1).logits.square().sum()
    b = m(x2).logits.square().sum()
    a.backward()
    g1 = m._engine.grads.clone()
    m.zero_grad()
    b.backward()
    assert torch.equal(g1, refs[0]) and torch.equal(m._engine.grads, refs[1])


def test_inference_forward_equals_training_forward_and_refuses_backward(golden_dir):
    """under torch.no_grad the forward takes the inference form (fc1 stores gelu(u) only): same logits bit for bit; a backward on
    those activations is refused instead of reading a stale derivative"""
    gold = np.load(f"{golden_dir}/vit_tiny.npz")
    spec, params = _tiny(gold)
    m = ViT(10, pretrained=False, model_name="tiny", backend=SimBackend())
    m.load_state_dict(params)
    x = torch.from_numpy(gold["x"][:2])
    train_logits = m(x).logits.detach().clone()
    with torch.no_grad():
        infer_logits = m(x).logits
    assert torch.equal(train_logits, infer_logits)
    e = m._engine
    with pytest.raises(RuntimeError, match="inference mode"):
        e.backward(torch.zeros_like(infer_logits))
    logits = m(x).logits   # a training forward at the same batch size clears the mark
    logits.sum().backward()


def test_fused_adamw_writes_both_operand_copies_and_equals_the_flat_rule():
    """`FusedAdamW.step` on a fully trainable model = `tic_vit_adamw`: the fp32 update of the flat rule (`tic_adamw`, itself checked against
    torch.optim.AdamW) bit for bit, with w16 AND the transposed copies written in the same pass -- the next forward refreshes nothing."""
    from touhouimageclassification_amd.optim import FusedAdamW
    from tests.simlib import call
    torch.manual_seed(3)
    m = ViT(10, pretrained=False, model_name="tiny", backend=SimBackend())
    e = m._engine
    x, y = torch.randn(2, 3, 224, 224), torch.tensor([3, 7])
    opt = FusedAdamW(m, lr=1e-3, weight_decay=0.01)
    ref_p, ref_m, ref_v = e.params.clone(), torch.zeros_like(e.params), torch.zeros_like(e.params)
    for step in (1, 2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x).logits, y).backward()
        grads = e.grads.clone()
        opt.step()
        call("tic_adamw", ref_p.data_ptr(), grads.data_ptr(), ref_m.data_ptr(), ref_v.data_ptr(), None, ref_p.numel(), 1e-3, 0.9, 0.999, 1e-8, 0.01, step, None)
        assert torch.equal(e.params, ref_p) and torch.equal(opt._m, ref_m) and torch.equal(opt._v, ref_v)
        assert e._weights_version == e._version()          # clean: the forward below must not launch a refresh ...
        kc.check_refresh_weights(m)                         # ... and both bf16 copies are already bf16(W) / its transposes
    refreshed = []
    orig = e.backend.call
    e.backend.call = lambda name, *a: (refreshed.append(name), orig(name, *a))[1]
    with torch.no_grad():
        m(x)
    assert "tic_vit_refresh_weights" not in refreshed and "tic_vit_forward_infer" in refreshed


def test_fused_step_stores_matrix_gradients_over_stale_values_and_zeroes_the_rest():
    """`fused_train_step` clears only the gradient ranges that accumulate (`tic_vit_zero_grads(st, 1)`: biases, LayerNorm, embeddings, head)
    and its backward STORES the weight-matrix gradients (`tic_vit_backward_layer_ex(..., 1)`): with the whole buffer poisoned beforehand,
    every gradient equals, bit for bit, the one the autograd path accumulates into a zeroed buffer."""
    from touhouimageclassification_amd.optim import FusedAdamW
    from touhouimageclassification_amd.step import fused_train_step
    torch.manual_seed(5)
    m = ViT(10, pretrained=False, model_name="tiny", backend=SimBackend())
    e = m._engine
    x, y = torch.randn(2, 3, 224, 224), torch.tensor([1, 8])
    m.zero_grad()
    torch.nn.functional.cross_entropy(m(x).logits, y).backward()
    want = e.grads.clone()
    assert float(want.abs().max()) > 0
    opt = FusedAdamW(m, lr=0.0, weight_decay=0.0)    # lr 0: the parameters stay, only the gradients are compared
    e.grads.fill_(float("nan"))
    fused_train_step(m, opt, x, y)
    assert torch.equal(e.grads, want)
    assert not e._dw_overwrite
    # a head-only backward after the partial clear would leave the matrix ranges undefined: refused
    e.zero_grads(2, keep_matrices=True)
    with pytest.raises(RuntimeError, match="FULL backward"):
        e.backward(torch.zeros(2, 10), head_only=True)
    e.zero_grads(2)
    assert float(e.grads.abs().max()) == 0 and not e._dw_overwrite
