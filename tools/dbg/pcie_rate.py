"""Design-time: what hipMemcpyAsync between pinned host memory and HBM sustains on this box (one direction, both at once)."""
import time, torch
n = 80 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1 = torch.cuda.Stream(); s2 = torch.cuda.Stream()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
def h2d():
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
def d2h():
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
def both(): h2d(); d2h()
a, b, c = t(h2d), t(d2h), t(both)
print(f"80 MiB: H2D {a*1e3:.2f} ms ({n/a/1e9:.1f} GB/s)  D2H {b*1e3:.2f} ms ({n/b/1e9:.1f} GB/s)  both at once {c*1e3:.2f} ms ({2*n/c/1e9:.1f} GB/s)")
