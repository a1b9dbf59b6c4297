"""Practical HBM ceilings on this box: device copy (read+write) and read-only reduction, 4 GiB buffers."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 30
x = torch.randn(n, device=dev); y = torch.empty_like(x)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tc = t(lambda: y.copy_(x)); print(f"copy 4 GiB -> 4 GiB: {2*4*n/tc/1e12:.2f} TB/s total traffic")
ts = t(lambda: x.sum()); print(f"read-only sum of 4 GiB: {4*n/ts/1e12:.2f} TB/s")
xh = x.half()
tcv = t(lambda: torch.add(x, 1.0, out=y)); print(f"y = x + 1: {2*4*n/tcv/1e12:.2f} TB/s")
tz = t(lambda: y.zero_()); print(f"write-only zero 4 GiB: {4*n/tz/1e12:.2f} TB/s")
