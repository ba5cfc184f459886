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


# the same GEMMs with a device-side live-row bound: time against the share of live rows
m2 = 1048576
a2 = torch.zeros(m2, kp, device=dev); a2[:, :k] = torch.randn(m2, k, device=dev) * 1e-3
o2 = torch.empty(m2, n, device=dev)
zt = torch.randn(m2, n, device=dev)
gw = torch.empty(k, n, device=dev)
for frac in (1.0, 0.84, 0.5):
    live = torch.tensor([int(m2 * frac)], dtype=torch.int32, device=dev)
    t = timeit(lambda: ops.gemm_nt_planes_live(a2, bp, o2, live, frac))
    t3 = timeit(lambda: ops.gemm(a2[:, :k], zt, True, False, out=gw, split_k=10, live=live, live_frac=frac))
    print(f"  live {frac:4.2f}: dz xw {t:.2f} ms   dWc TN {t3:.2f} ms", flush=True)
