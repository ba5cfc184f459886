#!/usr/bin/env python3
"""A/B of the conv engines on the generator's big layers at the bench batch (2 views x 16 images): the in-kernel
split (modconv_kernel<true,true>) against the x-planes path (gx_modsplit + modconv_xp_kernel), interleaved rounds
in ONE process (HIP events).  Usage: python tools/bench_conv_xp.py [rounds]"""
import math
import sys

import torch

sys.path.insert(0, ".")
from ganecdotes_amd import ops  # noqa: E402

LAYERS = [  # (H in, Cin, Cout, upsample)
    (16, 512, 512, False), (16, 512, 512, True), (32, 512, 512, False), (32, 512, 512, True), (64, 512, 512, False),
    (64, 512, 256, True), (128, 256, 256, False), (128, 256, 128, True), (256, 128, 128, False)]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = 32
    ops.set_gemm_mode(1)
    dev = torch.device("cuda")
    kf = torch.tensor([[1., 3, 3, 1]], device=dev)
    kf = (kf.t() @ kf) / 16.0
    tot = {"old": 0.0, "xp": 0.0, "split": 0.0}
    for h, cin, cout, up in LAYERS:
        torch.manual_seed(h + cin)
        x = torch.randn(B, h, h, cin, device=dev)
        wt = torch.randn(9, cout, cin, device=dev) / math.sqrt(9 * cin)
        s = torch.rand(B, cin, device=dev) + 0.5
        dm = torch.rand(B, cout, device=dev) + 0.5
        wtp = ops.split_conv_weights(wt)
        oh = 2 * h if up else h
        noise, nw, bias = torch.randn(1, 1, oh, oh, device=dev), torch.tensor([0.1], device=dev), torch.randn(cout, device=dev)
        xp = ops.modsplit(x, s)

        def old():
            if up:
                ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True, wt_planes=wtp)
            else:
                ops.modconv(x, wt, 3, s, dm, noise, nw, bias, True, wt_planes=wtp)

        def new():
            if up:
                ops.modconv_up(x, wt, s, dm, kf, noise, nw, bias, True, wt_planes=wtp, x_planes=xp)
            else:
                ops.modconv(x, wt, 3, s, dm, noise, nw, bias, True, wt_planes=wtp, x_planes=xp)

        def split():
            ops.modsplit(x, s)
        t = {}
        for name, fn in (("old", old), ("xp", new), ("split", split)):
            fn()
        torch.cuda.synchronize()
        for name, fn in (("old", old), ("xp", new), ("split", split)):
            t[name] = []
        for _ in range(rounds):
            for name, fn in (("old", old), ("xp", new), ("split", split)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    fn()
                b.record()
                torch.cuda.synchronize()
                t[name].append(a.elapsed_time(b) / 3)
        gf = 2.0 * B * h * h * cin * cout * 9 / 1e9
        med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        for k in tot:
            tot[k] += med[k]
        print(f"H={h:3d} {cin:3d}->{cout:3d} {'up ' if up else '   '} {gf:7.1f} GF | old {med['old']:.3f} ms "
              f"{gf / med['old']:.0f} TF/s | xp {med['xp']:.3f} ms {gf / med['xp']:.0f} TF/s | split {med['split']:.3f} ms",
              flush=True)
    # per-tile overhead probe: 1x1 convs (one tap: ncc k-tiles per output tile) next to the 3x3 rows above
    for h, cin, cout in ((128, 256, 128), (128, 256, 256), (64, 512, 256), (256, 128, 128)):
        x = torch.randn(B, h, h, cin, device=dev)
        wt = torch.randn(1, cout, cin, device=dev) / math.sqrt(cin)
        s = torch.rand(B, cin, device=dev) + 0.5
        wtp = ops.split_conv_weights(wt)
        xp = ops.modsplit(x, s)
        ts = []
        for _ in range(rounds + 1):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                ops.modconv(x, wt, 1, s, None, None, None, None, False, wt_planes=wtp, x_planes=xp)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 3)
        t1 = sorted(ts[1:])[len(ts[1:]) // 2]
        gf = 2.0 * B * h * h * cin * cout / 1e9
        tiles = ((B * h * h + 127) // 128) * ((cout + 127) // 128)
        print(f"1x1 H={h:3d} {cin:3d}->{cout:3d}: {t1:.3f} ms  {gf / t1:.0f} TF/s  {tiles} tiles of {cin // 16} k-tiles")
    print(f"total old {tot['old']:.2f} ms, xp {tot['xp']:.2f} ms (+ split {tot['split']:.2f} ms)")


if __name__ == "__main__":
    main()
