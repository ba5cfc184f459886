#!/usr/bin/env python3
"""The pixel-major swapped CE on the per-pixel score tables (gx_swapped_ce_pixels_tab) at the bench shape: 16 images x 65 536
pixels, K prototypes, 5 patches of 20 000 samples per image, table rows ordered by membership (what a many-prototype step hands
it after the table Sinkhorn).  Prints the time and a few checksums to compare builds.
   usage: python tools/lab/ce_tab_lab.py [K] [P] [images] [mask|pixel] [global|image]"""
import sys

import torch

sys.path.insert(0, ".")
from ganecdotes_amd import ops  # noqa: E402


def timeit(fn, n=5, warm=2):
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    by_mask = (sys.argv[4] if len(sys.argv) > 4 else "mask") == "mask"
    per_image = (sys.argv[5] if len(sys.argv) > 5 else "global") == "image"
    hw, N = 65536, 20000
    dev = torch.device("cuda")
    torch.manual_seed(0)
    npix, rows = B * hw, B * N
    base = (torch.arange(B, device=dev) * hw).unsqueeze(1)
    flat = [(torch.stack([torch.randperm(hw, device=dev)[:N] for _ in range(B)]) + base).reshape(-1) for _ in range(P)]
    mkey = torch.zeros(npix, dtype=torch.int64, device=dev)
    for p_, f in enumerate(flat):
        mkey[f] += 1 << p_
    img = torch.arange(npix, device=dev) // hw
    if by_mask:
        key = torch.where(mkey > 0, (img << 14 if per_image else 0) + mkey, torch.full_like(mkey, 1 << 30))
    else:
        key = (mkey == 0).to(torch.int64)
    order = torch.argsort(key, stable=True)
    live = (mkey > 0).sum(dtype=torch.int32).reshape(1)
    cid = torch.empty(npix, dtype=torch.int64, device=dev)
    cid[order] = torch.arange(npix, device=dev)
    inv = torch.full((P, npix), -1, dtype=torch.int32, device=dev)
    for p in range(P):
        rm = torch.sort(cid[flat[p]]).values
        inv[p].scatter_(0, rm, torch.arange(rows, dtype=torch.int32, device=dev))
    row_group = img[order].to(torch.int32).contiguous() if per_image else None
    ngroups = B if per_image else 1
    tabs = [torch.randn(npix, K, device=dev) * 0.05 for _ in range(2)]
    la = torch.randn(2 * P * ngroups, K, device=dev) * 0.3

    class _Sk:          # what swapped_ce_pixels_tab needs of a solved batch
        eps = 0.005
    sk = _Sk()
    sk.la = la
    kpad = (K + 15) // 16 * 16
    ds, dt = torch.full((npix, kpad), 7.0, device=dev), torch.full((npix, kpad), 7.0, device=dev)
    loss, gb = torch.zeros(1, device=dev), torch.zeros(K, device=dev)

    def run():
        ops.swapped_ce_pixels_tab(tabs[0], tabs[1], sk, ngroups, inv, 0.01, 1e-3, loss, 1e-6, 0.0, ds, dt, kpad, gb, 0.0, live=live,
                                  live_frac=1.0, row_group=row_group)
    t = timeit(run)
    lv = int(live.item())
    gbytes = (8.0 * lv * K + 8.0 * lv * kpad) / 1e9
    print(f"K={K} P={P} images={B} rows by {'mask' if by_mask else 'pixel'}, {'per-image' if per_image else 'global'} problems: "
          f"{t:7.2f} ms  {gbytes / t:5.2f} TB/s of table bytes ({gbytes:.0f} GB)")
    print(f"   loss {float(loss):.9e}  gb {float(gb.double().abs().sum()):.9e}  ds {float(ds[:lv].double().abs().sum()):.9e}  "
          f"dt {float(dt[:lv].double().abs().sum()):.9e}  pad {float(ds[:lv, K:].abs().sum()):.1f}  past live {float(ds[lv:].mean()) if lv < npix else 7.0:.1f}")


if __name__ == "__main__":
    main()
