import torch, time
x = torch.empty(256*1024*1024, dtype=torch.int32, device='cuda')  # 1 GiB
for name, fn in (("fill_", lambda: x.fill_(7)), ("zero_", lambda: x.zero_()), ("arange-like add", lambda: x.add_(1)), ("copy", None)):
    if fn is None:
        y = torch.empty_like(x)
        fn = lambda: y.copy_(x)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    b = x.numel() * 4 * (2 if name in ("arange-like add", "copy") else 1)
    print(f"{name:18s} {ms:.3f} ms  {b / ms / 1e9:.2f} TB/s")
