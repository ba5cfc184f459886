#!/usr/bin/env python3
"""Row kernels of the many-prototype path (K = 5000, the reference's nprototypes) in isolation: one Sinkhorn
(colmax + first pass + niters-1 sweeps + updates) and the swapped-CE with d(scores), as GB/s of score-matrix
bytes moved.  usage: python tools/bench_bigk.py [rows] [K]"""
import sys

import torch

sys.path.insert(0, ".")
from ganecdotes_amd import ops  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    dev = torch.device("cuda")
    torch.manual_seed(0)
    z = torch.nn.functional.normalize(torch.randn(rows, 64, device=dev), dim=1)
    w = torch.nn.functional.normalize(torch.randn(K, 64, device=dev), dim=1)
    Ss = (z @ w.t()).contiguous()
    St = (torch.roll(z, 1, 0) @ w.t()).contiguous()
    gb = rows * K * 4 / 1e9

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    sk = [ops.Sinkhorn(S, 0.005, 10, tag=f"bk{i}") for i, S in enumerate((Ss, St))]
    t = timed(lambda: sk[0].run())
    print(f"sinkhorn (11 reads of {gb:.2f} GB): {t:.2f} ms = {11 * gb / t * 1e3:.0f} GB/s")
    sk[1].run()
    loss = torch.zeros(1, device=dev)
    dS, dT = torch.empty_like(Ss), torch.empty_like(St)
    t = timed(lambda: ops.swapped_ce(sk[0], sk[1], 0.01, 1.0 / rows, loss, 1.0 / rows, 0.0, dS, dT))
    print(f"swapped CE + dS (2 reads + 2 writes): {t:.2f} ms = {4 * gb / t * 1e3:.0f} GB/s   loss {loss.item():.6f}")
    out = torch.empty(K, device=dev)
    t = timed(lambda: ops.colsum(dS, out))
    print(f"colsum (1 read): {t:.2f} ms = {gb / t * 1e3:.0f} GB/s")
    # fp64 check of the codes and the loss on a slice
    n = min(rows, 4096)
    q = torch.softmax(Ss[:n].double() / 0.005 + sk[0].la.double(), 1)
    col = torch.softmax(Ss.double()[:0], 1)
    print("la finite:", bool(torch.isfinite(sk[0].la).all()), " code row sums:", float(q.sum(1).min()), float(q.sum(1).max()))


if __name__ == "__main__":
    main()
