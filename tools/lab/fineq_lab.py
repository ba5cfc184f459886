"""The pieces of the finest pyramid level at the bench shape (2B = 32 images of 256^2, CF = 128 + 128, D = 512, K = 8):
quadratic-form route (csrc/gx_fineq.hip) against the projected-table route, kernel by kernel.
    python tools/lab/fineq_lab.py [piece ...]      (no argument: everything)"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from ganecdotes_amd import ops

dev = "cuda"
which = set(sys.argv[1:])
B2, R, D, K, CF = 32, 256, 512, 8, 256
rows, rc = B2 * R * R, B2 * R * R // 4
torch.manual_seed(0)
t1 = torch.randn(rows, 128, device=dev)
t2 = torch.randn(rows, 128, device=dev)
zc = torch.randn(B2, R // 2, R // 2, D, device=dev)
zc2 = zc.reshape(-1, D)
wp = torch.randn(D, CF, device=dev) / 16
wc = torch.nn.functional.normalize(torch.randn(K, D, device=dev), dim=1)
bias = torch.zeros(K, device=dev)
ls = ops.lib().gx_lowrank_stride(K)
coeffs = torch.randn(rows, ls, device=dev) * 1e-3


def timeit(f, n=5, warm=10):
    for _ in range(warm):          # the first launches after an idle spell run 10-20 % slower (clock ramp): not the kernel
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def show(name, ms, flops=0.0, nbytes=0.0):
    print(f"{name:44s} {ms:7.3f} ms" + (f"  {flops / ms / 1e9:7.1f} TFLOP/s" if flops else "") +
          (f"  {nbytes / ms / 1e6:7.1f} GB/s" if nbytes else ""), flush=True)


def want(name):
    return not which or name in which


mg = ops.gemm(wp, wp, True, False)
wcf = ops.gemm(wc, wp, False, False)
yt = ops.fineq_ytable(rc, dev)
y2 = yt[:, :CF]
wft_planes = ops.split_gemm_weights(wp.t().contiguous())
wf_planes = ops.split_gemm_weights(wp)
if want("fwd"):
    show("y2 = 2 Zc Wf (0.5M x 512 -> 256)", timeit(lambda: ops.gemm_nt_planes(zc2, None, wft_planes, y2, alpha=2.0)), 2.0 * rc * D * CF)
    if "y2var" in which:       # where the 154 TFLOP/s of the Y2 GEMM come from: output stride, alpha, N (three interleaved rounds)
        yc = torch.empty(rc, CF, device=dev)
        w512 = ops.split_gemm_weights(torch.randn(512, D, device=dev) / 16)
        y512 = torch.empty(rc, 512, device=dev)
        for _ in range(3):
            show("  y2 as the step runs it (ld 288, alpha 2)", timeit(lambda: ops.gemm_nt_planes(zc2, None, wft_planes, y2, alpha=2.0)), 2.0 * rc * D * CF)
            show("  y2 into a contiguous (rows, 256) table", timeit(lambda: ops.gemm_nt_planes(zc2, None, wft_planes, yc, alpha=2.0)), 2.0 * rc * D * CF)
            show("  y2, alpha = 1", timeit(lambda: ops.gemm_nt_planes(zc2, None, wft_planes, y2, alpha=1.0)), 2.0 * rc * D * CF)
            show("  same A, N = 512", timeit(lambda: ops.gemm_nt_planes(zc2, None, w512, y512)), 2.0 * rc * D * 512)
    show("raw tables of Zc", timeit(lambda: ops.proto_raw_tables(zc2, wc)), 0, 4.0 * rc * (D + K + 1))
    ops.proto_raw_tables(zc2, wc, ytable=yt)
    pl = ops.split_gemm_weights(ops.fineq_pack_weights(mg, wcf))
    nblk = sum(8 - (t >> 1) for t in range(16)) + 16
    show("fineq_fwd", timeit(lambda: ops.fineq_fwd(t1, t2, pl, yt, bias, K, R)), 2.0 * rows * 512 * nblk)
    if ops.GEMM_H2:
        pl3 = ops.split_gemm_weights(ops.fineq_pack_weights(mg, wcf), h2=True)
        show("fineq_fwd f16x3", timeit(lambda: ops.fineq_fwd(t1, t2, pl3, yt, bias, K, R, a_range=ops.FEATURE_RANGE)), 2.0 * rows * 512 * nblk)
    fo = ops.fineq_fwd(t1, t2, pl, yt, bias, K, R)
    print("   fineq_fwd checksums (diff build to build):", " ".join(f"{float(x.double().sum()):.12e}" for x in fo), flush=True)
    # the projected-table route
    z = torch.empty(rows, D, device=dev)
    wpl = ops.split_gemm_weights(torch.randn(D, CF, device=dev) / 16)
    show("old: finest projection GEMM (+ upadd)", timeit(lambda: ops.gemm_nt_planes(t1, t2, wpl, z, coarse=zc2, fine_res=R)), 2.0 * rows * D * CF)
    z4 = z.reshape(B2, R, R, D)
    show("old: score tables pass", timeit(lambda: ops.upsample2_add_scores_(None, z4, wc, bias)), 0, 4.0 * rows * (D + K + 1))
if want("bwd"):
    show("fineq_bwd (S, P, Tt, Cpool in one pass)", timeit(lambda: ops.fineq_bwd(coeffs, t1, t2, K, R)), 2.0 * rows * 64 * 64 * 12, 4.0 * rows * CF * 1.25)
    s_, pm, tt, cpool = ops.fineq_bwd(coeffs, t1, t2, K, R)
    tam = ops.absmax(torch.cat([t1.reshape(-1), t2.reshape(-1)]))
    for _ in range(2):     # the f16x3 form (incl. its coefficient-maxima pass) beside the bf16x6 one, twice
        show("fineq_bwd f16x3 (+ coefficient maxima)", timeit(lambda: ops.fineq_bwd(coeffs, t1, t2, K, R, t_amax=tam)), 2.0 * rows * 64 * 64 * 12, 4.0 * rows * CF * 1.25)
        show("fineq_bwd bf16x6", timeit(lambda: ops.fineq_bwd(coeffs, t1, t2, K, R)), 2.0 * rows * 64 * 64 * 12, 4.0 * rows * CF * 1.25)
    show("coefficient maxima alone", timeit(lambda: ops.lowrank_coeff_amax(coeffs, K)), 0, 4.0 * rows * ls)
    sh = ops.fineq_bwd(coeffs, t1, t2, K, R, t_amax=tam)[0]
    print(f"   S f16x3 vs bf16x6: {float((sh - s_).abs().max() / s_.abs().max()):.2e}", flush=True)
    gc = torch.empty(rc, D, device=dev)
    show("Gc = Tt Wf^T (0.5M x 256 -> 512)", timeit(lambda: ops.gemm_nt_planes(tt, None, wf_planes, gc)), 2.0 * rc * D * CF)
    gwc = torch.empty(K, D, device=dev)
    show("finish at the coarse level (add_to, pool)", timeit(lambda: ops.proto_lowrank_finish(cpool, zc, wc, gwc, 0.0, want_pool=True, add_to=gc.reshape(zc.shape))),
         0, 4.0 * rc * D * 3.25)
    show("Gc GEMM with the finish in its epilogue", timeit(lambda: ops.gemm_nt_planes_finish(tt, wf_planes, cpool, zc, wc, gwc, 0.0, R // 2, True)),
         2.0 * rc * D * CF, 4.0 * rc * D * 2.25)
    gw = torch.empty(D, CF, device=dev)
    show("dWf: Zc^T Tt (512 x 256, k = 0.5M)", timeit(lambda: ops.gemm(zc2, tt, True, False, out=gw, split_k=64)), 2.0 * rc * D * CF)
    # the projected-table route
    z = torch.randn(B2, R, R, D, device=dev)
    show("old: dense finish at the fine level", timeit(lambda: ops.proto_lowrank_finish(coeffs, z, wc, gwc, 0.0, want_pool=True)), 0, 4.0 * rows * D * 2.25)
    g2 = z.reshape(rows, D)
    show("old: finest TN (512 x 256, k = 2M)", timeit(lambda: ops.gemm_tn_cat(g2, t1, t2, gw, split_k=64)), 2.0 * rows * D * CF)
