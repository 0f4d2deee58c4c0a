"""practical HBM bandwidth on the box: torch device copy (read + write) and a read-only reduction, 1 GiB operands"""
import torch
x = torch.randn(256, 1024, 1024, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in (("copy_", lambda: y.copy_(x), 2 * x.numel() * 4), ("sum", lambda: x.sum(), x.numel() * 4),
                         ("add 3 streams", lambda: torch.add(x, y, out=y), 3 * x.numel() * 4), ("fill", lambda: y.fill_(1.0), x.numel() * 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[5]
    print(f"{name}: {ms:.3f} ms, {nbytes / ms / 1e6:.0f} GB/s")
