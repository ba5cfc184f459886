"""times ops.proto_raw_tables at the bench shape (the 128^2 level of 16 images: 262144 rows x 512, K = 8)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from ganecdotes_amd import ops

rows, d, k = 16 * 128 * 128, 512, 8
z = torch.randn(rows, d, device="cuda")
wc = torch.nn.functional.normalize(torch.randn(k, d, device="cuda"), dim=1)
yt = ops.fineq_ytable(rows, z.device)
for _ in range(3):
    ops.proto_raw_tables(z, wc, ytable=yt)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(50):
    ops.proto_raw_tables(z, wc, ytable=yt)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 50
print(f"proto_raw_tables {rows} x {d}, K = {k}: {ms * 1e3:.1f} us = {rows * d * 4 / ms / 1e9:.2f} TB/s")
