#!/usr/bin/env python3
"""ResNet-50 @224 training-step throughput probe (BASELINE config 4, single GPU): images/sec and achieved TFLOP/s."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd.ResNet.model import resnet50  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--graph", action="store_true", help="capture the whole step (forward, CE, backward, SGD) in one HIP graph and replay it")
a = ap.parse_args()
torch.manual_seed(0)
m = resnet50(num_classes=120).to("cuda").train()
opt = torch.optim.SGD(m.parameters(), lr=5e-2)
x = torch.randn(a.batch, 3, 224, 224, device="cuda"); y = torch.randint(0, 120, (a.batch,), device="cuda")
def step_static():
    opt.zero_grad(set_to_none=False)
    torch.nn.functional.cross_entropy(m(x), y).backward()
    opt.step()
def step():
    opt.zero_grad()
    torch.nn.functional.cross_entropy(m(x), y).backward()
    opt.step()
for _ in range(3): step()
if a.graph:   # ~1000 launches per step: at B = 128 the host is the bottleneck -> one graph launch per step
    opt.zero_grad(set_to_none=False)   # static .grad tensors for the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): step_static()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step_static()
    step = graph.replay
    for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
tag = " (graph)" if a.graph else ""
print(f"resnet50{tag} B={a.batch}: {dt*1e3:.1f} ms/step = {a.batch/dt:.0f} img/s = {a.batch/dt*24.5e9/1e12:.1f} TFLOP/s (24.5 GF/img)")
