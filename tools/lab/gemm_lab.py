import sys, os, torch
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
dev = "cuda"
which = sys.argv[1:] or ["f1", "f2", "sc", "dz", "dw", "b1"]
for w in which:
    if w == "f1":
        m, n, k1, k2 = 1 << 21, 512, 128, 128
        a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev); b = torch.randn(n, k1 + k2, device=dev)
        out = torch.empty(m, n, device=dev); co = torch.randn(m // 4, n, device=dev)
        t = timeit(lambda: ops.gemm_upadd(a1, a2, b, out, co, 256)); fl = 2.0 * m * n * (k1 + k2)
        t2 = timeit(lambda: ops.gemm_cat(a1, a2, b, out))
        print(f"f1 upadd {t:.3f} ms {fl / t / 1e9:.0f} TF/s | cat (no up) {t2:.3f} ms {fl / t2 / 1e9:.0f} TF/s", flush=True)
        del a1, a2, out, co
    if w == "f2":
        m, n, k1, k2 = 1 << 19, 512, 256, 256
        a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev); b = torch.randn(n, k1 + k2, device=dev)
        out = torch.empty(m, n, device=dev); co = torch.randn(m // 4, n, device=dev)
        t = timeit(lambda: ops.gemm_upadd(a1, a2, b, out, co, 128)); fl = 2.0 * m * n * (k1 + k2)
        print(f"f2 upadd {t:.3f} ms {fl / t / 1e9:.0f} TF/s", flush=True)
        del a1, a2, out, co
    if w == "sc":
        m, n, k = 320000, 5000, 512
        a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
        t = timeit(lambda: ops.gemm(a, b, False, True, out=out)); fl = 2.0 * m * n * k
        print(f"sc NT {t:.3f} ms {fl / t / 1e9:.0f} TF/s", flush=True)
        del a, out
    if w == "dz":
        m, n, k = 320000, 512, 5000
        a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
        t = timeit(lambda: ops.gemm(a, b, False, False, out=out)); fl = 2.0 * m * n * k
        print(f"dz NN {t:.3f} ms {fl / t / 1e9:.0f} TF/s", flush=True)
        del a, out
    if w == "dw":
        m, n, k = 5000, 512, 320000
        a = torch.randn(k, m, device=dev); b = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
        for sk in (8, 16):
            t = timeit(lambda: ops.gemm(a, b, True, False, out=out, split_k=sk)); fl = 2.0 * m * n * k
            print(f"dw TN split {sk} {t:.3f} ms {fl / t / 1e9:.0f} TF/s", flush=True)
        del a, out
    if w == "b1":
        m, n, k = 512, 128, 1 << 21
        a = torch.randn(k, m, device=dev); b = torch.randn(k, n, device=dev); out = torch.empty(m, n, device=dev)
        for sk in (64, 128, 256):
            t = timeit(lambda: ops.gemm(a, b, True, False, out=out, split_k=sk)); fl = 2.0 * m * n * k
            print(f"b1 TN split {sk} {t:.3f} ms {fl / t / 1e9:.0f} TF/s", flush=True)
        del a, out
