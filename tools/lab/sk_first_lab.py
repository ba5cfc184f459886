"""score GEMM 320000 x 5000 x 512 with / without the fused first Sinkhorn pass, and what the merge costs"""
import sys, torch
sys.path.insert(0, "/root/repo")
from ganecdotes_amd import ops
dev = "cuda"
torch.manual_seed(0)
m, n, k = 320000, 5000, 512
a = torch.nn.functional.normalize(torch.randn(m, k, device=dev), dim=1)
wc = torch.nn.functional.normalize(torch.randn(n, k, device=dev), dim=1)
wpl = ops.split_gemm_weights(wc)
out = torch.empty(m, n, device=dev)
def timeit(f, r=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(r): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / r
for _ in range(2):
    print("colmax only        %.3f ms" % timeit(lambda: ops.gemm_nt_planes(a, None, wpl, out, want_colmax=True)))
    print("colmax + first pass %.3f ms" % timeit(lambda: ops.gemm_nt_planes(a, None, wpl, out, want_colmax=True, sk_eps=0.005)))
sk = ops.Sinkhorn(out, 0.005, 10)
print("one sweep           %.3f ms" % timeit(lambda: sk.sweep(False)))
