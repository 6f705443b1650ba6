#!/usr/bin/env python3
"""ResNet training-step throughput (BASELINE config 4: ResNet-50 @224 data-parallel 1 -> 8 GPUs; and the configuration the reference
itself trains, ResNet-152 @256 px with SGD lr 5e-2, TIC/ResNet/train.py:210-255): images/sec and achieved TFLOP/s.

  python tools/resnet_bench.py [--arch resnet50|resnet152] [--image 224|256] [--batch 128] [--gpus N] [--graph]

`--gpus N` (N > 1) without a launcher starts its own N ranks exactly as bench.py does (a torch.distributed.run child, started before
this process touches the GPU).  Data-parallel form = the package's own (`dist.BucketedGradSync` on `TicResNet.buckets()`): every rank
steps its own B images, the six gradient buckets (fc, layer4 .. layer1, stem) are all-reduced on a side stream as the backward
completes them (SUM of 1/world-scaled gradients), SGD waits for them, BatchNorm keeps per-replica statistics.
Prints one JSON line on rank 0."""
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
ap.add_argument("--arch", default="resnet50", choices=["resnet18", "resnet34", "resnet50", "resnet101", "resnet152"])
ap.add_argument("--image", type=int, default=224)
ap.add_argument("--batch", type=int, default=128, help="images per GPU per step")
ap.add_argument("--classes", type=int, default=120)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--graph", action="store_true", help="capture the whole step (forward, CE, backward, SGD) in one HIP graph and replay it (1 GPU)")
a = ap.parse_args()

if "WORLD_SIZE" not in os.environ and a.gpus > 1:   # start our own ranks BEFORE any GPU call
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
                                     "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env).returncode)

import torch.distributed as dist  # noqa: E402
from touhouimageclassification_amd.ResNet import model as rm  # noqa: E402
from touhouimageclassification_amd._lib import call as _call  # noqa: E402
for _kv in filter(None, os.environ.get("TIC_PRESET", "").split(",")):   # knobs held for the run: TIC_PRESET=gemm_tile=128,...
    _k, _v = _kv.split("=")
    _call("tic_set_option", _k.encode(), int(_v))


def train_flops_per_image(model, image: int) -> float:
    """3 x the forward convolution + classifier FLOPs (8.174 GFLOP forward for ResNet-50 @224), walked off the
    module tree: each convolution costs 2 Ho Wo Cout Cin k^2"""
    def conv(c, h_in):
        h_out = (h_in + 2 * c.pad - c.k) // c.stride + 1
        return 2.0 * h_out * h_out * c.cout * c.cin * c.k * c.k, h_out
    fl, h = conv(model.conv1, image)
    h = (h + 2 - 3) // 2 + 1   # max-pool 3x3 / 2
    for blk in model._blocks():
        f, h_mid = conv(blk.conv1, h)
        fl += f
        f, h_out = conv(blk.conv2, h_mid)
        fl += f
        if blk.kind != "basic":
            f, h_out = conv(blk.conv3, h_out)
            fl += f
        if blk.downsample is not None:
            fl += conv(blk.downsample[0], h)[0]
        h = h_out
    fl += 2.0 * model.fc.weight.shape[1] * model.fc.weight.shape[0]
    return 3.0 * fl

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if world != a.gpus:
    raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
m = getattr(rm, a.arch)(num_classes=a.classes).to(dev).train()
from touhouimageclassification_amd.dist import BucketedGradSync  # noqa: E402
sync = BucketedGradSync(m, force=bool(os.environ.get("TIC_FORCE_BUCKETS")))   # inactive (no hooks) at world 1 unless forced
sync.broadcast_parameters()
opt = torch.optim.SGD(m.parameters(), lr=5e-2)   # TIC/ResNet/train.py:240
g = torch.Generator().manual_seed(1234 + rank)
x = torch.randn(a.batch, 3, a.image, a.image, generator=g).to(dev)
y = torch.randint(0, a.classes, (a.batch,), generator=g).to(dev)


def step(set_to_none=True):
    opt.zero_grad(set_to_none=set_to_none)
    (torch.nn.functional.cross_entropy(m(x), y) * sync.grad_scale).backward()
    sync.wait()
    opt.step()


for _ in range(a.warmup):
    step()
run = step
if a.graph and world == 1:   # ~1000 launches per step: one graph launch instead
    opt.zero_grad(set_to_none=False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step(False)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(False)
    run = graph.replay
    for _ in range(2):
        run()
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    run()
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
if world > 1:
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
dt = dt.item() / a.steps
if rank == 0:
    fl = train_flops_per_image(m, a.image)
    ips = world * a.batch / dt
    print(json.dumps({"metric": f"images/sec {a.arch} {a.image}px bf16 train step (SGD lr 5e-2)", "value": round(ips, 1), "unit": "images/sec",
                      "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 3), "scaling": "weak",
                      "config": {"workload": f"{a.arch} C={a.classes} {a.image}px fwd+CE+bwd+SGD" + ("+grad all-reduce" if world > 1 else ""),
                                 "per_gpu_batch": a.batch, "hip_graph": bool(a.graph and world == 1)},
                      "tflops": round(ips * fl / world / 1e12, 1), "frac_of_bf16_peak": round(ips * fl / world / 2.5e15, 4),
                      "flops_per_image": fl}), flush=True)
if world > 1:
    dist.destroy_process_group()
