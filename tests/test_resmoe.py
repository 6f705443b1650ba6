"""ResMoE counterpart: dense MoE on micro ViTs through the simulator backend, and the expert-parallel form on
2 gloo ranks (one expert per rank, all-gather + all-to-all) against the single-process dense model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

E, C, B = 2, 5, 2


def _build(backend):
    from touhouimageclassification_amd.ResMoE.model import make_ViTMoE
    torch.manual_seed(11)
    m = make_ViTMoE(num_classes=C, num_experts=E, top_k=2, gateway_t=0.01, pretrained=False, model_name="micro", gate_pretrained=False,
                    backend=backend, gate_model_name="micro")
    m.eval()   # no gate noise: deterministic comparison
    return m


def _data():
    g = torch.Generator().manual_seed(5)
    return torch.randn(E * B, 3, 32, 32, generator=g), torch.randint(0, C, (E * B,), generator=g)


def test_dense_moe_step_and_losses():
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResMoE import train as mt
    m = _build(SimBackend())
    x, y = _data()
    logits, gw, idx = m(x[:B])
    assert logits.shape == (B, C) and gw.shape == (B, E) and idx.shape == (B, 2)
    torch.testing.assert_close(gw.sum(1), torch.ones(B))
    tgt = torch.nn.functional.one_hot(y[:B], C).float()
    loss = mt.total_loss(logits, tgt, gw, idx, backend=SimBackend())
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # gate, combination and loss are the reference's arithmetic (oracle/moe_oracle.py restates it line for line)
    from oracle import moe_oracle as mo
    eo = torch.stack([e(x[:B]).logits for e in m.experts], 1).detach()
    scores = m.gate.vit(x[:B]).logits.detach()
    rw, ri = mo.gate(scores, None, 2)
    assert torch.equal(idx, ri)
    torch.testing.assert_close(gw.detach(), mo.scatter(rw, ri, E), atol=1e-6, rtol=1e-5)
    torch.testing.assert_close(logits.detach(), mo.combine(eo, gw.detach()), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(loss.detach(), mo.total_loss(logits.detach(), tgt, gw.detach()), atol=1e-5, rtol=1e-5)
    assert abs(mt.load_balance_loss(gw, idx, E).item() - mo.load_balance_loss(gw.detach()).item()) < 1e-6
    # the reference-shaped gate call still returns (top-k weights, top-k indices)
    tw, ti = m.gate(x[:B])
    assert tw.shape == (B, 2) and torch.equal(ti, ri) and torch.allclose(tw.sum(1), torch.ones(B))


def test_moe_kernels_match_oracle_in_the_simulator():
    from tests import kernel_checks as kc
    from tests.simlib import call
    kc.check_moe_ops(kc.Env("cpu", call), B=5, E=8, K=2, C=70)
    kc.check_moe_ops(kc.Env("cpu", call, seed=1), B=3, E=3, K=3, C=5)


def _ep_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResMoE.model import ExpertParallelMoE
    from touhouimageclassification_amd.ResMoE import train as mt
    dense = _build(SimBackend())
    ep = ExpertParallelMoE(dense.experts[rank], dense.gate, C)
    ep.eval()
    x, y = _data()
    xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
    logits, gw, idx = ep(xs)
    loss = mt.total_loss(logits, torch.nn.functional.one_hot(ys, C).float(), gw, idx, backend=SimBackend()) / world
    loss.backward()
    ep.sync_gate_gradients()
    torch.save({"logits": logits.detach(), "expert_grad": dense.experts[rank].classifier.weight.grad.clone(),
                "gate_grad": dense.gate.vit.classifier.weight.grad.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_expert_parallel_matches_dense(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(_ep_worker, args=(E, port, str(tmp_path)), nprocs=E, join=True, start_method="spawn")
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResMoE import train as mt
    dense = _build(SimBackend())
    x, y = _data()
    logits, gw, idx = dense(x)
    # per-rank losses are means over the local batch; the EP run scaled each by 1/world
    loss = sum(mt.total_loss(logits[r * B:(r + 1) * B], torch.nn.functional.one_hot(y[r * B:(r + 1) * B], C).float(), gw[r * B:(r + 1) * B],
                             idx[r * B:(r + 1) * B], backend=SimBackend()) for r in range(E)) / E
    loss.backward()
    for r in range(E):
        got = torch.load(tmp_path / f"r{r}.pt")
        torch.testing.assert_close(got["logits"], logits[r * B:(r + 1) * B].detach(), atol=2e-3, rtol=2e-3)
        ref = dense.experts[r].classifier.weight.grad
        assert (got["expert_grad"] - ref).norm() <= 0.03 * ref.norm() + 1e-6
        refg = dense.gate.vit.classifier.weight.grad / E   # sync_gate_gradients averages
        assert (got["gate_grad"] - refg).norm() <= 0.03 * refg.norm() + 1e-6


# ---- sparse top-2 dispatch on three ranks (three experts: every sample visits two of them) --------------------------------------
E3 = 3


def _build3(backend):
    from touhouimageclassification_amd.ResMoE.model import make_ViTMoE
    torch.manual_seed(13)
    m = make_ViTMoE(num_classes=C, num_experts=E3, top_k=2, gateway_t=0.01, pretrained=False, model_name="micro", gate_pretrained=False,
                    backend=backend, gate_model_name="micro")
    m.eval()
    return m


def _data3():
    g = torch.Generator().manual_seed(6)
    return torch.randn(E3 * B, 3, 32, 32, generator=g), torch.randint(0, C, (E3 * B,), generator=g)


def _sparse_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResMoE.model import SparseExpertParallelMoE
    from touhouimageclassification_amd.ResMoE import train as mt
    dense = _build3(SimBackend())
    ep = SparseExpertParallelMoE(dense.experts[rank], dense.gate, C, pad_rows=2)
    ep.eval()
    x, y = _data3()
    xs, ys = x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B]
    logits, gw, idx = ep(xs)
    loss = mt.total_loss(logits, torch.nn.functional.one_hot(ys, C).float(), gw, idx, backend=SimBackend()) / world
    loss.backward()
    ep.sync_gate_gradients()
    torch.save({"logits": logits.detach(), "gw": gw.detach(), "idx": idx, "rows": ep.last_rows,
                "expert_grad": dense.experts[rank].classifier.weight.grad.clone(),
                "expert_deep_grad": dense.experts[rank].vit.layers[0].mlp.fc1.weight.grad.clone(),
                "gate_grad": dense.gate.vit.classifier.weight.grad.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sparse_expert_parallel_matches_dense(tmp_path):
    """SparseExpertParallelMoE (images travel only to their top-2 experts: 2/3 of the dense expert work here, 1/4 for the reference's
    top-2 of 8) against the single-process DENSE model of TIC/ResMoE/model.py:50-58: mixture logits, gate weights, routing, and the
    gradients of every expert (head and first MLP layer) and of the gate"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(_sparse_worker, args=(E3, port, str(tmp_path)), nprocs=E3, join=True, start_method="spawn")
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResMoE import train as mt
    dense = _build3(SimBackend())
    x, y = _data3()
    logits, gw, idx = dense(x)
    loss = sum(mt.total_loss(logits[r * B:(r + 1) * B], torch.nn.functional.one_hot(y[r * B:(r + 1) * B], C).float(), gw[r * B:(r + 1) * B],
                             idx[r * B:(r + 1) * B], backend=SimBackend()) for r in range(E3)) / E3
    loss.backward()
    got = [torch.load(tmp_path / f"r{r}.pt") for r in range(E3)]
    assert sum(g["rows"] for g in got) == E3 * B * 2                      # every sample visited exactly two experts ...
    assert [g["rows"] for g in got] == [int((idx == e).sum()) for e in range(E3)]   # ... the ones its gate chose
    for r in range(E3):
        sl = slice(r * B, (r + 1) * B)
        assert torch.equal(got[r]["idx"], idx[sl])
        torch.testing.assert_close(got[r]["gw"], gw[sl].detach(), atol=1e-6, rtol=1e-5)
        torch.testing.assert_close(got[r]["logits"], logits[sl].detach(), atol=2e-3, rtol=2e-3)
        for key, ref in (("expert_grad", dense.experts[r].classifier.weight.grad), ("expert_deep_grad", dense.experts[r].vit.layers[0].mlp.fc1.weight.grad)):
            assert (got[r][key] - ref).norm() <= 0.03 * ref.norm() + 1e-6, (r, key)
        refg = dense.gate.vit.classifier.weight.grad / E3
        assert (got[r]["gate_grad"] - refg).norm() <= 0.03 * refg.norm() + 1e-6


def test_sparse_single_process_equals_dense():
    """MoEClassifier(sparse=True): each expert on its routed samples only -- same mixture logits and gradients as the dense evaluation"""
    from tests.simlib import SimBackend
    from touhouimageclassification_amd.ResMoE import train as mt
    x, y = _data3()
    tgt = torch.nn.functional.one_hot(y, C).float()
    res = []
    for sparse in (False, True):
        m = _build3(SimBackend())
        m.sparse, m.pad_rows = sparse, 2
        logits, gw, idx = m(x)
        mt.total_loss(logits, tgt, gw, idx, backend=SimBackend()).backward()
        res.append((logits.detach(), gw.detach(), idx, [e.classifier.weight.grad.clone() for e in m.experts], m.gate.vit.classifier.weight.grad.clone()))
    d, s = res
    assert torch.equal(d[2], s[2])
    torch.testing.assert_close(s[0], d[0], atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(s[1], d[1], atol=1e-6, rtol=1e-5)
    for a, b in zip(s[3], d[3]):
        assert (a - b).norm() <= 0.03 * b.norm() + 1e-6
    assert (s[4] - d[4]).norm() <= 0.03 * d[4].norm() + 1e-6
