"""The mapping network on the bench step's 48 latents: one launch per layer (gx_equal_linear_skinny) against the split-K
GEMM + slab reduce + bias / activation route, interleaved in one process."""
import sys
import torch
sys.path.insert(0, ".")
from ganecdotes_amd import generator as Gm

net = Gm.MappingNetwork(512, 8, 0.01).cuda()
z = torch.randn(48, 512, device="cuda")


def timeit(n=200):
    for _ in range(20):
        net(z)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        net(z)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for _ in range(3):
    for rows in (256, 0):
        Gm.SKINNY_ROWS = rows
        print(f"{'skinny' if rows else 'gemm  '}: {timeit():7.1f} us per forward", flush=True)
