#!/usr/bin/env python3
"""Micro-benchmarks of single kernels on the GPU box (TFLOP/s or GB/s), for tuning.
usage: python tools/bench_kernels.py conv|gemm|blur|torgb|all"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ganecdotes_amd import ops


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def conv_cases(B=16):
    return [(64, 512, 512, False), (128, 256, 256, False), (256, 128, 128, False), (32, 512, 512, False),
            (16, 512, 512, False), (64, 512, 256, True), (128, 256, 128, True), (32, 512, 512, True)]


def bench_conv(B=16):
    for res, cin, cout, up in conv_cases(B):
        x = torch.randn(B, res, res, cin, device="cuda")
        wt = torch.randn(9, cout, cin, device="cuda")
        s = torch.randn(B, cin, device="cuda")
        dm = torch.rand(B, cout, device="cuda")
        kf = torch.rand(4, 4, device="cuda")
        ro = 2 * res if up else res
        nz = torch.randn(1, 1, ro, ro, device="cuda")
        nw = torch.full((1,), 0.1, device="cuda")
        bias = torch.randn(cout, device="cuda")
        if up:
            wp = ops.split_conv_weights(wt) if __import__('os').environ.get('GX_NO_PLANES') is None else None
            fn = lambda: ops.modconv_up(x, wt, s, dm, kf, nz, nw, bias, True, wt_planes=wp)
        else:
            wp = ops.split_conv_weights(wt) if __import__('os').environ.get('GX_NO_PLANES') is None else None
            fn = lambda: ops.modconv(x, wt, 3, s, dm, nz, nw, bias, True, wt_planes=wp)
        ms = timeit(fn)
        fl = 2.0 * B * res * res * cin * cout * 9
        print(f"conv{'_up' if up else '   '} res={res:4d} {cin:4d}->{cout:4d}  {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


def bench_gemm():
    for (m, n, k, ta, tb) in [(320000, 512, 5376, False, True), (1 << 20, 512, 256, False, True),
                              (262144, 512, 512, False, True), (65536, 512, 1024, False, True),
                              (512, 256, 1 << 20, True, False), (512, 1024, 65536, True, False)]:
        a = torch.randn((k, m) if ta else (m, k), device="cuda")
        b = torch.randn((n, k) if tb else (k, n), device="cuda")
        out = torch.empty(m, n, device="cuda")
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        sk = int(max(1, min(512 // tiles, k // 256))) if ta else 1
        ms = timeit(lambda: ops.gemm(a, b, ta, tb, out=out, split_k=sk))
        print(f"gemm m={m} n={n} k={k} ta={ta} tb={tb} splitk={sk}  {ms:8.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TFLOP/s",
              flush=True)


def bench_blur_torgb(B=16):
    for res, c in [(256, 128), (128, 256)]:
        x = torch.randn(B, res, res, c, device="cuda")
        w3 = torch.randn(3, c, device="cuda")
        s = torch.randn(B, c, device="cuda")
        b3 = torch.randn(3, device="cuda")
        prev = torch.randn(B, 3, res // 2, res // 2, device="cuda")
        kf = torch.rand(4, 4, device="cuda")
        ms = timeit(lambda: ops.torgb(x, w3, s, 0.1, b3, prev, kf))
        print(f"torgb res={res} c={c}  {ms:8.3f} ms  {x.numel() * 4 / ms / 1e6:8.1f} GB/s (tap read)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("conv", "all"):
        bench_conv()
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("torgb", "all"):
        bench_blur_torgb()
