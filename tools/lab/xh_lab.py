"""fp16x3 (gemm_xh_kernel) against bf16x6 (gemm_xw_kernel): time on the pyramid / score shapes and error against fp64"""
import ctypes, sys, torch
sys.path.insert(0, ".")
from ganecdotes_amd import ops, _lib
from ganecdotes_amd.ops import _p, _stream, lib
L = lib()
def split_h3(b):
    n, k = b.shape
    pl = torch.empty(L.gx_gemm_weight_planes_h3_bytes(n, k) // 2, dtype=torch.int16, device=b.device)
    _lib.check(L.gx_split_gemm_weights_h3(_p(b), n, k, b.stride(0), _p(pl), _stream()), "split_h3")
    return pl
def gemm_h3(a1, a2, bp, out, row_scale=None, col_bias=None, coarse=None, fine_res=0):
    m, k1 = a1.shape; k2 = 0 if a2 is None else a2.shape[1]; n = out.shape[1]
    _lib.check(L.gx_gemm_f32_nt_planes_h3(m, n, k1, _p(a1), a1.stride(0), k2, _p(a2), 0 if a2 is None else a2.stride(0), _p(bp),
                                          _p(out), out.stride(0), 1.0, 0.0, _p(row_scale), _p(col_bias), _p(coarse), fine_res,
                                          _stream()), "h3")
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
dev = "cuda"; torch.manual_seed(0)
# accuracy vs fp64 (and the exact-fp32 MFMA engine) on a 2048-row sample
for k in (256, 512, 4608):
    a = torch.randn(2048, k, device=dev); b = torch.randn(512, k, device=dev) / k ** 0.5
    ref = a.double() @ b.double().t(); sc = ref.abs().max()
    o6 = torch.empty(2048, 512, device=dev); o3 = torch.empty_like(o6)
    ops.gemm_nt_planes(a, None, ops.split_gemm_weights(b), o6)
    gemm_h3(a, None, split_h3(b), o3)
    prev = ops.set_gemm_mode(0); o32 = ops.gemm(a, b, False, True); ops.set_gemm_mode(prev)
    e = lambda o: ((o.double() - ref).abs().max() / sc).item()
    print(f"K={k:5d} max|err|/max|C| vs fp64: exact-fp32 MFMA {e(o32):.2e} | bf16x6 {e(o6):.2e} | fp16x3 {e(o3):.2e}", flush=True)
for name, m, n, k1, k2, res in [("f1", 1 << 21, 512, 128, 128, 256), ("f2", 1 << 19, 512, 256, 256, 128), ("f3", 1 << 17, 512, 512, 512, 64)]:
    a1 = torch.randn(m, k1, device=dev); a2 = torch.randn(m, k2, device=dev); b = torch.randn(n, k1 + k2, device=dev)
    out = torch.empty(m, n, device=dev); co = torch.randn(m // 4, n, device=dev)
    bp6, bp3 = ops.split_gemm_weights(b), split_h3(b); fl = 2.0 * m * n * (k1 + k2)
    t6 = timeit(lambda: ops.gemm_nt_planes(a1, a2, bp6, out, coarse=co, fine_res=res))
    t3 = timeit(lambda: gemm_h3(a1, a2, bp3, out, coarse=co, fine_res=res))
    print(f"{name}: bf16x6 {t6:.3f} ms {fl / t6 / 1e9:.0f} TF/s | fp16x3 {t3:.3f} ms {fl / t3 / 1e9:.0f} TF/s", flush=True)
    del a1, a2, out, co
m, n, k = 320000, 5000, 512
a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
bp6, bp3 = ops.split_gemm_weights(b), split_h3(b); fl = 2.0 * m * n * k
t6 = timeit(lambda: ops.gemm_nt_planes(a, None, bp6, out)); t3 = timeit(lambda: gemm_h3(a, None, bp3, out))
print(f"scores 320000x5000x512: bf16x6 {t6:.3f} ms {fl / t6 / 1e9:.0f} TF/s | fp16x3 {t3:.3f} ms {fl / t3 / 1e9:.0f} TF/s", flush=True)
del a, out
m, n, k = 320000, 512, 5008
a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); out = torch.empty(m, n, device=dev)
bp6, bp3 = ops.split_gemm_weights(b), split_h3(b); fl = 2.0 * m * n * k
t6 = timeit(lambda: ops.gemm_nt_planes(a, None, bp6, out)); t3 = timeit(lambda: gemm_h3(a, None, bp3, out))
print(f"long-k 320000x512x5008: bf16x6 {t6:.3f} ms {fl / t6 / 1e9:.0f} TF/s | fp16x3 {t3:.3f} ms {fl / t3 / 1e9:.0f} TF/s", flush=True)
