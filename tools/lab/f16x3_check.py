"""The two plane formats of the x-planes conv path side by side (ops.set_conv_planes): every tap of a full-size generator
pass against the exact-fp32 MFMA engine (GX_GEMM_MODE=0) — worst norm-wise and max-norm error per tap — and the time of a
pass under each.   usage: python tools/lab/f16x3_check.py [size=256] [batch=16]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from ganecdotes_amd import ops  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda")
g = bench.build_generator(size, dev)
for n, p in g.named_parameters():
    if n.endswith("noise.weight"):
        p.data.fill_(0.1)
g._invalidate()
torch.manual_seed(5)
z = torch.randn(batch, 512, device=dev)
ml = g.mean_latent(1024)


def run():
    w = g.style(z)
    img, feats = g([w], truncation=0.7, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
    return [img] + list(feats)


def timed(n=6):
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


prev = ops.set_gemm_mode(0)
ref = [t.double() for t in run()]
ops.set_gemm_mode(1)
for fmt, name in ((0, "bf16x6"), (1, "f16x3 ")):
    ops.set_conv_planes(fmt)
    out = run()
    nw = [float((o.double() - r).norm() / r.norm()) for o, r in zip(out, ref)]
    mx = [float((o.double() - r).abs().max() / r.abs().max()) for o, r in zip(out, ref)]
    bad = [bool(torch.isnan(o).any()) for o in out]
    print(f"{name}: pass {timed():7.3f} ms   norm-wise worst {max(nw):.2e} (image {nw[0]:.2e}, last tap {nw[-1]:.2e})   "
          f"max-norm worst {max(mx):.2e}   nan: {any(bad)}")
    print("        per tap norm-wise:", " ".join(f"{v:.1e}" for v in nw[1:]))
ops.set_gemm_mode(prev)
