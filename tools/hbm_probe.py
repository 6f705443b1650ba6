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
