"""one shape through gemm_nt_planes (and the old kernel) a few times, for PMC passes: one_xw.py sc|f1 [old]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from ganecdotes_amd import ops
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "sc"
old = len(sys.argv) > 2
torch.manual_seed(0)
if which == "sc":
    m, n, k = 320000, 5000, 512
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
    bp = ops.split_gemm_weights(b)
    f = (lambda: ops.gemm(a, b, False, True, out=out)) if old else (lambda: ops.gemm_nt_planes(a, None, bp, out))
else:
    m, n, k1, k2 = 1 << 21, 512, 128, 128
    a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev); b = torch.randn(n, k1 + k2, device=dev)
    out = torch.empty(m, n, device=dev); co = torch.randn(m // 4, n, device=dev)
    bp = ops.split_gemm_weights(b)
    f = (lambda: ops.gemm_upadd(a1, a2, b, out, co, 256)) if old else (lambda: ops.gemm_nt_planes(a1, a2, bp, out, coarse=co, fine_res=256))
for _ in range(3):
    f()
torch.cuda.synchronize()
