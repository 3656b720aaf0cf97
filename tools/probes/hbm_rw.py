"""Streaming HBM rates on the box (torch kernels: fill = write only, sum = read only, copy = read + write), 1.6 GB buffers."""
import time, torch
n = 400_000_000
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
w = t(lambda: a.fill_(1.0)); r = t(lambda: a.sum()); c = t(lambda: b.copy_(a))
print(f"write-only {n*4/w/1e12:.2f} TB/s ({w*1e6:.0f} us)   read-only {n*4/r/1e12:.2f} TB/s ({r*1e6:.0f} us)   copy {2*n*4/c/1e12:.2f} TB/s total ({c*1e6:.0f} us)")
