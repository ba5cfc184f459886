"""Sinkhorn problems as row selections of one score table: what the Infinity Cache gives when the problems of
a view are swept side by side.  python tools/lab/sinkhorn_rows_lab.py [K] [rows] [npix]"""
import sys
import time
import torch
sys.path.insert(0, "/root/repo")
from ganecdotes_amd import ops   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 320000
npix = int(sys.argv[3]) if len(sys.argv) > 3 else 1048576
P, eps, niters = 5, 0.005, 10
dev = "cuda"
torch.manual_seed(0)
tab = torch.empty(npix, K, device=dev)
for i in range(0, npix, 65536):
    tab[i:i + 65536] = torch.randn(min(65536, npix - i), K, device=dev) * 0.1
live = int(npix * 0.84)


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def maps(kind):
    if kind == "same":
        m = torch.sort(torch.randperm(live, device=dev)[:rows]).values.to(torch.int32)
        return [m.clone() for _ in range(P)]
    out = [torch.randperm(live, device=dev)[:rows].to(torch.int32) for _ in range(P)]
    return [torch.sort(m).values for m in out] if kind == "sorted" else out


for kind, side in (("unsorted", False), ("sorted", False), ("sorted", True), ("same", True)):
    ml = maps(kind)
    skb = ops.SinkhornBatch.from_rows([tab] * P, ml, eps, niters, 1.0 / rows, sorted_rows=side)
    for sk_ in skb.probs:
        sk_.cmax_preset, sk_.u_first_preset = ops.sinkhorn_first_rows(sk_.S, eps, sk_.rowmap)
    t = timeit(skb.run)
    gb = 4.0 * P * rows * K * (niters - 1) / 1e9
    print(f"{kind:9s} side by side {side!s:5s}: {t:7.2f} ms for {P} problems x {niters - 1} sweeps = {gb / t:6.2f} TB/s of row bytes", flush=True)
