"""The many-prototype score GEMM alone (gemm_nt_planes: rows x 5000 x 512 with a column bias, as _score_tables runs it) and the dz
GEMM (rows x 512 x 5008), timed.   usage: python tools/lab/sc_lab.py [rows] [K]"""
import sys

import torch

sys.path.insert(0, ".")
from ganecdotes_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 880000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dev = "cuda"
torch.manual_seed(0)
a = torch.randn(rows, 512, device=dev)
a = a / a.norm(dim=1, keepdim=True)
w = torch.randn(K, 512, device=dev)
w = w / w.norm(dim=1, keepdim=True)
bias = torch.randn(K, device=dev) * 0.01
out = torch.empty(rows, K, device=dev)
wp = ops.split_gemm_weights(w)


def timeit(f, n=4, warm=2):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t = timeit(lambda: ops.gemm_nt_planes(a, None, wp, out, col_bias=bias))
print(f"scores {rows} x {K} x 512: {t:7.3f} ms  {2.0 * rows * K * 512 / t / 1e9:6.1f} TFLOP/s   checksum {float(out[::997].double().sum()):.6e}")
kp = (K + 15) // 16 * 16
ds = torch.randn(rows, kp, device=dev) * 1e-3
wt = ops.split_gemm_weights(torch.nn.functional.pad(w, (0, 0, 0, kp - K)).t().contiguous())
dz = torch.empty(rows, 512, device=dev)
t = timeit(lambda: ops.gemm_nt_planes(ds, None, wt, dz))
print(f"dz     {rows} x 512 x {kp}: {t:7.3f} ms  {2.0 * rows * kp * 512 / t / 1e9:6.1f} TFLOP/s")
t = timeit(lambda: ops.gemm_nt_planes(a, None, wp, out, col_bias=bias, want_colmax=True, sk_eps=0.005))
print(f"scores + per-slice column maxima and first-pass sums (and their merge over all rows): {t:7.3f} ms")


# round 5: the same two GEMMs on the f16x3 form (DESIGN par. 4.21) — 128- / 256-row tiles, N tiles per group of the tile order
L = ops.lib()
wp3 = ops.split_gemm_weights(w, h2=True)
wt3 = ops.split_gemm_weights(torch.nn.functional.pad(w, (0, 0, 0, kp - K)).t().contiguous(), h2=True)
for tall in (1, 0):
    L.gx_set_gemm_tall(tall)
    for ng in (0, 1, 2, 4, 8, 40):
        L.gx_set_gemm_ngroup(ng)
        t1 = timeit(lambda: ops.gemm_nt_planes(a, None, wp3, out, col_bias=bias, a_range=1.0))
        t2 = timeit(lambda: ops.gemm_nt_planes(ds, None, wt3, dz, a_range=8e-3)) if ng <= 4 else float("nan")
        print(f"f16x3 tall={tall} ngroup={ng:2d}: scores {t1:7.3f} ms {2.0 * rows * K * 512 / t1 / 1e9:6.1f} TFLOP/s   dz {t2:7.3f} ms {2.0 * rows * kp * 512 / t2 / 1e9:6.1f} TFLOP/s", flush=True)
L.gx_set_gemm_tall(1)
L.gx_set_gemm_ngroup(0)
