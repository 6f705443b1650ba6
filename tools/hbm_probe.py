import torch, time
x = torch.empty(512*1024*1024, dtype=torch.float32, device="cuda")  # 2 GiB
y = torch.empty_like(x)
for fn, name, bytes_ in ((lambda: y.copy_(x), "copy (r+w)", 2*x.numel()*4), (lambda: x.zero_(), "fill (w)", x.numel()*4), (lambda: x.sum(), "sum (r)", x.numel()*4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {bytes_*10/ (e0.elapsed_time(e1)*1e-3)/1e12:.2f} TB/s")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from touhouimageclassification_amd._lib import call, current_stream
sink = torch.zeros(4, device="cuda")
n = x.numel()
for blocks in (1024, 2048, 4096, 8192, 16384):
    for mode in (0, 1):
        for dst, name, bytes_ in ((None, "read ", n * 4), (y, "copy ", 2 * n * 4)):
            def fn(): call("tic_probe_stream", x.data_ptr(), None if dst is None else dst.data_ptr(), sink.data_ptr(), n, blocks, mode, current_stream())
            for _ in range(2): fn()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"probe {name} blocks={blocks:6d} nt={mode}: {bytes_*5/(e0.elapsed_time(e1)*1e-3)/1e12:.2f} TB/s")
