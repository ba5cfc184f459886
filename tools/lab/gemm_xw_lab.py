"""gemm_xw_kernel against the in-kernel-split kernels: bitwise equality and time, on the pyramid / score shapes."""
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
torch.manual_seed(0)
# correctness on small odd shapes first
for (m, n, k1, k2, up) in [(300, 132, 32, 0, 0), (1024, 512, 64, 16, 8), (4096 + 64, 260, 128, 128, 0), (4096, 512, 256, 0, 16)]:
    a1 = torch.randn(m, k1 + 8, device=dev)[:, :k1]
    a2 = torch.randn(m, k2, device=dev) if k2 else None
    b = torch.randn(n, k1 + k2 + 4, device=dev)[:, :k1 + k2]
    rs = torch.rand(m, device=dev); cb = torch.randn(n, device=dev)
    co = torch.randn(m // 4, n, device=dev) if up else None
    ref = torch.randn(m, n, device=dev); out = ref.clone()
    bp = ops.split_gemm_weights(b)
    if up:
        ops.gemm_upadd(a1, a2, b, ref, co, up, beta=0.5)
        ops.gemm_nt_planes(a1, a2, bp, out, beta=0.5, coarse=co, fine_res=up)
    elif a2 is not None:
        ops.gemm_cat(a1, a2, b, ref, alpha=0.7, beta=0.25)
        ops.gemm_nt_planes(a1, a2, bp, out, alpha=0.7, beta=0.25)
    else:
        ops.gemm(a1, b, False, True, out=ref, alpha=1.3, beta=0.0, row_scale=rs, col_bias=cb)
        ops.gemm_nt_planes(a1, None, bp, out, alpha=1.3, beta=0.0, row_scale=rs, col_bias=cb)
    d = (out - ref).abs().max().item()
    print(f"check m={m} n={n} k={k1}+{k2} up={up}: max|diff| = {d:g} equal={torch.equal(out, ref)}", flush=True)
    assert torch.isfinite(out).all()

for name, m, n, k1, k2, res in [("f1", 1 << 21, 512, 128, 128, 256), ("f2", 1 << 19, 512, 256, 256, 128),
                                ("f3", 1 << 17, 512, 512, 512, 64), ("f4", 1 << 15, 512, 512, 512, 32)]:
    a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev); b = torch.randn(n, k1 + k2, device=dev)
    out = torch.empty(m, n, device=dev); co = torch.randn(m // 4, n, device=dev)
    bp = ops.split_gemm_weights(b)
    fl = 2.0 * m * n * (k1 + k2)
    t0 = timeit(lambda: ops.gemm_upadd(a1, a2, b, out, co, res)); r0 = out.clone()
    t1 = timeit(lambda: ops.gemm_nt_planes(a1, a2, bp, out, coarse=co, fine_res=res))
    print(f"{name} upadd old {t0:.3f} ms {fl / t0 / 1e9:.0f} TF/s | xw {t1:.3f} ms {fl / t1 / 1e9:.0f} TF/s equal={torch.equal(out, r0)}", flush=True)
    del a1, a2, out, co, r0
m, n, k = 320000, 5000, 512
a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
rs = torch.rand(m, device=dev)
bp = ops.split_gemm_weights(b); fl = 2.0 * m * n * k
t0 = timeit(lambda: ops.gemm(a, b, False, True, out=out, row_scale=rs))
chk = out[::997].clone()
t1 = timeit(lambda: ops.gemm_nt_planes(a, None, bp, out, row_scale=rs))
print(f"sc NT old {t0:.3f} ms {fl / t0 / 1e9:.0f} TF/s | xw {t1:.3f} ms {fl / t1 / 1e9:.0f} TF/s equal={torch.equal(out[::997], chk)}", flush=True)
