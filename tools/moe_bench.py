#!/usr/bin/env python3
"""BASELINE config 5: the mixture of ViT experts (TIC/ResMoE: ViT-Base gate + E ViT-Base experts, top-2 routing, symmetric CE +
balance loss, SGD lr 5e-2).  One GPU: the dense form the reference computes (every expert on every sample).  `--gpus E` (one expert
per rank, starts its own ranks like bench.py): `ExpertParallelMoE` -- image all-gather, each rank's expert on the global batch,
logits all-to-all, gate gradients averaged.  Prints one JSON line: images/sec of the training step.

  python tools/moe_bench.py [--experts 8] [--batch 32] [--gpus 1|E]"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--experts", type=int, default=8)          # MOE_NUM_EXPERTS (TIC/ResMoE/parameter.py)
ap.add_argument("--batch", type=int, default=32, help="images per rank per step (the reference: 4 x 4 accumulation steps)")
ap.add_argument("--classes", type=int, default=120)
ap.add_argument("--model", default="google/vit-base-patch16-224")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--sparse", action="store_true", help="every expert runs only on the samples routed to it (top-2 of E: 2/E of the dense work); N > 1: images travel only to their experts")
a = ap.parse_args()
if "WORLD_SIZE" not in os.environ and a.gpus > 1:   # our own ranks, before any GPU call
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
                                     "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode)

import torch.distributed as dist  # noqa: E402
from touhouimageclassification_amd.ResMoE import train as mt  # noqa: E402
from touhouimageclassification_amd.ResMoE.model import ExpertParallelMoE, SparseExpertParallelMoE, make_ViTMoE  # noqa: E402

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
if world > 1:
    if world != a.experts:
        raise SystemExit(f"expert parallelism places one expert per rank: --experts {a.experts} on {world} ranks")
    dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(11)   # every rank builds the same gate; rank r keeps expert r
dense = make_ViTMoE(num_classes=a.classes, num_experts=a.experts if world == 1 else world, top_k=2, gateway_t=0.01, pretrained=False,
                    model_name=a.model, gate_pretrained=False, gate_model_name=a.model)
if world == 1:
    dense.sparse = a.sparse
    model = dense.to(dev)
else:
    model = (SparseExpertParallelMoE if a.sparse else ExpertParallelMoE)(dense.experts[rank], dense.gate, a.classes).to(dev)
model.train()
opt = torch.optim.SGD(model.parameters(), lr=5e-2)   # TIC/ResMoE/train.py:176
g = torch.Generator().manual_seed(5 + rank)
x = torch.randn(a.batch, 3, 224, 224, generator=g).to(dev)
y = torch.randint(0, a.classes, (a.batch,), generator=g).to(dev)
tgt = torch.nn.functional.one_hot(y, a.classes).float()


def step():
    opt.zero_grad()
    logits, gw, idx = model(x)
    loss = mt.total_loss(logits, tgt, gw, idx) / world
    loss.backward()
    if world > 1:
        model.sync_gate_gradients()
    opt.step()
    return loss


for _ in range(a.warmup):
    step()
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
if world > 1:
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
dt = dt.item() / a.steps
if rank == 0:
    E = a.experts if world == 1 else world
    # dense: every expert sees every image of the (global) batch, the gate sees each image once: (E + 1) ViT-B training passes per image;
    # sparse: top-2 experts + the gate = 3 passes per image
    fl = 105.38e9 * ((2 if a.sparse else E) + 1)
    ips = world * a.batch / dt
    print(json.dumps({"metric": "images/sec mixture of ViT experts train step (BASELINE config 5)", "value": round(ips, 1), "unit": "images/sec",
                      "n_gpus": world, "ms_per_step": round(dt * 1e3, 2), "loss": round(float(loss) * world, 4),
                      "config": {"workload": f"gate ViT-B + {E} ViT-B experts, top-2, C={a.classes}, {'sparse' if a.sparse else 'dense'}{'' if world == 1 else ', expert-parallel (1 expert per rank)'}",
                                 "per_rank_batch": a.batch, "optimizer": "SGD lr 5e-2"},
                      "tflops_per_gpu": round(ips * fl / world / 1e12, 1)}), flush=True)
if world > 1:
    dist.destroy_process_group()
