import torch, time
x = torch.empty(3_276_800_000 // 8, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
for name, fn in (("zero_", lambda: x.zero_()), ("fill_(1.5)", lambda: x.fill_(1.5)), ("copy_", lambda: y.copy_(x))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:12s} 3.28 GB: {ms:.3f} ms -> {3.2768/ms:.2f} TB/s (stores{' + same loads' if name=='copy_' else ''})")
