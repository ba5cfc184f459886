import sys, torch
sys.path.insert(0, ".")
from ganecdotes_amd import ops
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
dev="cuda"
m, n, k = 5000, 512, 320000
a = torch.randn(k, m, device=dev) * 1e-3; b = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
for sk in (4, 6, 8, 12, 13, 16, 19, 26, 32):
    t = timeit(lambda: ops.gemm(a, b, True, False, out=out, split_k=sk)); fl = 2.0 * m * n * k
    print(f"dWc TN split {sk:2d}: {t:.3f} ms {fl / t / 1e9:.0f} TF/s  ({80 * sk} workgroups)", flush=True)
