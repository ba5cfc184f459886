"""Sinkhorn + swapped CE between the small-K kernels (K <= 64) and the many-prototype ones: the column-spread kernels of gx_bigk.h
against the team-per-row kernels (taken when K % 4 != 0: timed at K + 1).  python tools/lab/midk_lab.py"""
import sys
import time
import torch
sys.path.insert(0, "/root/repo")
from ganecdotes_amd import ops   # noqa: E402

rows, eps = 320000, 0.005
dev = "cuda"


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for K in (68, 100, 200, 400, 800, 1200, 2000, 4000):
    res = []
    for k in (K, K + 1):                               # K % 4 == 0: the column-spread kernels; K + 1: the team-per-row ones
        torch.manual_seed(K)
        s = (torch.randn(rows, k, device=dev) * 0.1).contiguous()
        sk = ops.Sinkhorn(s, eps, 10)
        res.append(timeit(lambda: sk.run()))
        del s, sk
    print(f"K={K:5d}: Sinkhorn (10 iterations, {rows} rows) column-spread {res[0]:7.2f} ms   team-per-row (K + 1) {res[1]:7.2f} ms", flush=True)
