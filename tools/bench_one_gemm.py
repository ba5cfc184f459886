#!/usr/bin/env python3
"""One GEMM shape in a loop (for PMC passes).  usage: bench_one_gemm.py [nt|tn] [reps]"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ganecdotes_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
if kind == "nt":
    a = torch.randn(320000, 5376, device="cuda")
    b = torch.randn(512, 5376, device="cuda")
    out = torch.empty(320000, 512, device="cuda")
    f = lambda: ops.gemm(a, b, False, True, out=out)
else:
    a = torch.randn(1048576, 512, device="cuda")
    b = torch.randn(1048576, 256, device="cuda")
    out = torch.empty(512, 256, device="cuda")
    f = lambda: ops.gemm(a, b, True, False, out=out, split_k=64)
for _ in range(reps):
    f()
torch.cuda.synchronize()
