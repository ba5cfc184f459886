#!/usr/bin/env python3
"""Numerics of operand-splitting fp32 emulations on half-precision matrix cores, emulated on the CPU (exact products
in fp64, fp32 accumulation order ignored): relative error of C = A B^T against fp64 for
  fp32      every product and the sum rounded as an fp32 dot product (numpy float32 matmul),
  bf16x6    three bf16 pieces per operand (exact 24 bits), the six products above 2^-24       (this build's default engine),
  fp16x3    two fp16 pieces per operand (11 + 11 bits), a0 b0 + a0 b1 + a1 b0                 (DESIGN 10, item 0),
  tf32      one 11-bit piece per operand                                                      (the reference's cuDNN default).
usage: python tools/split_accuracy.py [K ...]"""
import sys
import numpy as np


def pieces_bf16(x):
    out, r = [], x.astype(np.float64)
    for _ in range(3):
        p = (r.astype(np.float32).view(np.uint32) + 0x7FFF + ((r.astype(np.float32).view(np.uint32) >> 16) & 1)) & 0xFFFF0000
        p = p.astype(np.uint32).view(np.float32).astype(np.float64)
        out.append(p); r = r - p
    return out


def pieces_fp16(x):
    a0 = x.astype(np.float16).astype(np.float64)
    a1 = ((x.astype(np.float64) - a0) * 2048.0).astype(np.float16).astype(np.float64) / 2048.0
    return a0, a1


def tf32(x):
    u = x.astype(np.float32).view(np.uint32)
    u = (u + 0xFFF + ((u >> 13) & 1)) & 0xFFFFE000
    return u.view(np.float32).astype(np.float64)


def main():
    ks = [int(a) for a in sys.argv[1:]] or [256, 512, 4608]
    rng = np.random.default_rng(0)
    print(f"{'K':>6} {'fp32':>10} {'bf16x6':>10} {'fp16x3':>10} {'tf32':>10}   (max |err| / max |C| over a 256 x 256 block)")
    for k in ks:
        a = rng.standard_normal((256, k)).astype(np.float32)
        b = (rng.standard_normal((256, k)) / np.sqrt(k)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64).T
        sc = np.abs(ref).max()
        e32 = np.abs((a @ b.T).astype(np.float64) - ref).max() / sc
        ah, am, al = pieces_bf16(a); bh, bm, bl = pieces_bf16(b)
        x6 = (al @ bh.T + ah @ bl.T + am @ bm.T) + (am @ bh.T + ah @ bm.T) + ah @ bh.T
        e6 = np.abs(x6.astype(np.float32).astype(np.float64) - ref).max() / sc
        a0, a1 = pieces_fp16(a); b0, b1 = pieces_fp16(b)
        x3 = a0 @ b0.T + (a0 @ b1.T + a1 @ b0.T)
        e3 = np.abs(x3.astype(np.float32).astype(np.float64) - ref).max() / sc
        et = np.abs(tf32(a) @ tf32(b).T - ref).max() / sc
        print(f"{k:6d} {e32:10.2e} {e6:10.2e} {e3:10.2e} {et:10.2e}")


if __name__ == "__main__":
    main()
