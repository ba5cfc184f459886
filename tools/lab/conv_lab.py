"""The 3x3 modulated conv on the x-planes path (modconv_xp_kernel), layer by layer at the bench batch (32 views).
   usage: python tools/lab/conv_lab.py [views] [tall 0|1] [planes 0|1]"""
import math
import sys

import torch

sys.path.insert(0, ".")
from ganecdotes_amd import ops  # noqa: E402


def timeit(fn, n=8, warm=3):
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    if len(sys.argv) > 2:                                  # f16x3 tile: 1 = 256 x 128 on the large layers (default), 0 = 128 x 128
        ops.lib().gx_set_conv_tall(int(sys.argv[2]))
    if len(sys.argv) > 3:                                  # plane format: 1 = f16x3 (default), 0 = bf16x6
        ops.set_conv_planes(int(sys.argv[3]))
    dev = torch.device("cuda")
    torch.manual_seed(0)
    tot = 0.0
    for h, c in ((8, 512), (16, 512), (32, 512), (64, 512), (128, 256), (256, 128)):
        x = torch.randn(B, h, h, c, device=dev)
        wt = torch.randn(9, c, c, device=dev) / math.sqrt(9 * c)
        s = torch.rand(B, c, device=dev) + 0.5
        dm = torch.rand(B, c, device=dev) + 0.5
        wtp = ops.split_conv_weights(wt)
        xp = ops.modsplit(x, s)
        noise = torch.randn(1, 1, h, h, device=dev)
        nw = torch.tensor([0.1], device=dev)
        bias = torch.randn(c, device=dev)
        sn = torch.rand(B, c, device=dev) + 0.5
        npl = ops.xplanes_buffer(B, h, h, c, dev, "lab")
        t = timeit(lambda: ops.modconv(x, wt, 3, s, dm, noise, nw, bias, wt_planes=wtp, x_planes=xp, next_style=sn, next_planes=npl))
        gf = 2.0 * B * h * h * c * c * 9 / 1e9
        tot += t
        print(f"H={h:3d} C={c}: {t:.3f} ms  {gf / t:.0f} TFLOP/s", flush=True)
    print(f"sum {tot:.3f} ms")


if __name__ == "__main__":
    main()
