#!/usr/bin/env python3
"""Can an HBM-bound pass (K = 5000 Sinkhorn) run UNDER an MFMA-bound GEMM of another patch?  Times the two alone
and together on two streams.  usage: python tools/bench_overlap.py"""
import sys

import torch

sys.path.insert(0, ".")
from ganecdotes_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda")
    rows, K, D = 160000, 5000, 512
    torch.manual_seed(0)
    z = torch.nn.functional.normalize(torch.randn(rows, D, device=dev), dim=1)
    wc = torch.nn.functional.normalize(torch.randn(K, D, device=dev), dim=1)
    bias = torch.zeros(K, device=dev)
    S = ops.gemm(z, wc, False, True, col_bias=bias)
    S2 = torch.empty_like(S)
    sk = ops.Sinkhorn(S, 0.005, 10, tag="ov")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def gemms():
        for _ in range(3):
            ops.gemm(z, wc, False, True, col_bias=bias, out=S2)

    def timed(fa, fb):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if fa is not None:
            s1.wait_event(a)
            with torch.cuda.stream(s1):
                fa()
                e1 = torch.cuda.Event(); e1.record()
            torch.cuda.current_stream().wait_event(e1)
        if fb is not None:
            s2.wait_event(a)
            with torch.cuda.stream(s2):
                fb()
                e2 = torch.cuda.Event(); e2.record()
            torch.cuda.current_stream().wait_event(e2)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    for _ in range(2):
        timed(gemms, None); timed(None, lambda: sk.run())
    tg = min(timed(gemms, None) for _ in range(3))
    ts = min(timed(None, lambda: sk.run()) for _ in range(3))
    tb = min(timed(gemms, lambda: sk.run()) for _ in range(3))
    print(f"3 score GEMMs alone {tg:.2f} ms | Sinkhorn alone {ts:.2f} ms | both on two streams {tb:.2f} ms "
          f"(sum {tg + ts:.2f}, max {max(tg, ts):.2f})")


if __name__ == "__main__":
    main()
