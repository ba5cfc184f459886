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
dev = "cuda"
m, n, k, kp = 320000, 512, 5000, 5008
buf = torch.zeros(m, kp, device=dev); buf[:, :k] = torch.randn(m, k, device=dev) * 1e-3
wc = torch.randn(k, n, device=dev)
out = torch.empty(m, n, device=dev); out2 = torch.empty(m, n, device=dev)
t0 = timeit(lambda: ops.gemm(buf[:, :k], wc, False, False, out=out))
wct = torch.zeros(n, kp, device=dev); wct[:, :k] = wc.t()
bp = ops.split_gemm_weights(wct)
t1 = timeit(lambda: ops.gemm_nt_planes(buf, None, bp, out2))
fl = 2.0 * m * n * k
print(f"dz NN ws {t0:.3f} ms {fl/t0/1e9:.0f} TF/s | xw {t1:.3f} ms {fl/t1/1e9:.0f} TF/s  maxdiff {(out-out2).abs().max().item():g} ref {out.abs().max().item():g}")

