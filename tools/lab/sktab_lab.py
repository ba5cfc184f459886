#!/usr/bin/env python3
"""The table Sinkhorn (csrc/gx_sktab.hip) at the bench shape, pass by pass: 16 images x 65 536 pixels, K prototypes, 5 patches
of 20 000 samples per image, drawn pixels first.  Times gx_sinkhorn_table_first / _sweep against the per-problem sweeps they
replace (gx_sinkhorn_rows_batch's kernels).   usage: python tools/lab/sktab_lab.py [K] [P] [images] [mask|pixel]"""
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
    hw, N = 65536, 20000
    dev = torch.device("cuda")
    torch.manual_seed(0)
    npix, rows = B * hw, B * N
    base = (torch.arange(B, device=dev) * hw).unsqueeze(1)
    flat = [(torch.stack([torch.randperm(hw, device=dev)[:N] for _ in range(B)]) + base).reshape(-1) for _ in range(P)]
    drawn = torch.zeros(npix, dtype=torch.bool, device=dev)
    for f in flat:
        drawn[f] = True
    by_mask = (sys.argv[4] if len(sys.argv) > 4 else "mask") == "mask"
    if by_mask:        # table rows grouped by which patches drew them (what _score_tables does for the table solver)
        mkey = torch.zeros(npix, dtype=torch.int16, device=dev)
        for p_, f in enumerate(flat):
            mkey[f] += 1 << p_
        order = torch.argsort(torch.where(mkey > 0, mkey, torch.full_like(mkey, 1 << 14)), stable=True)
    else:
        order = torch.argsort((~drawn).to(torch.int8), stable=True)
    live = drawn.sum(dtype=torch.int32).reshape(1)
    cid = torch.empty(npix, dtype=torch.int64, device=dev)
    cid[order] = torch.arange(npix, device=dev)
    rowmaps = [torch.sort(cid[f]).values.to(torch.int32) for f in flat]
    inv = torch.full((P, npix), -1, dtype=torch.int32, device=dev)
    for p in range(P):
        inv[p].scatter_(0, rowmaps[p].long(), torch.arange(rows, dtype=torch.int32, device=dev))
    tabs = [torch.randn(npix, K, device=dev) * 0.05 for _ in range(2)]
    lv = int(live.item())
    print(f"table rows ordered by {'membership mask' if by_mask else 'pixel'}")
    print(f"K={K} P={P} images={B}: {npix} table rows, {lv} live ({lv / npix:.3f}), {rows} rows per problem")
    st = ops.SinkhornTable(tabs[0], tabs[1], inv, live, 0.005, 10, 1.0 / rows, rows)
    st.run()
    gb_tab = 2.0 * lv * K * 4 / 1e9
    gb_prob = 2.0 * P * rows * K * 4 / 1e9
    t_first = timeit(st.first)
    t_sweep = timeit(st.sweep)
    st.update(False)
    t_shift = timeit(lambda: st.sweep(shifted=True))
    t_all = timeit(lambda: st.run(), n=2, warm=1)
    print(f"table first : {t_first:7.2f} ms  ({gb_tab / t_first * 1e3:6.0f} GB/s of live table rows; {gb_prob / t_first * 1e3:6.0f} GB/s of problem rows)")
    print(f"table sweep : {t_sweep:7.2f} ms  ({gb_tab / t_sweep * 1e3:6.0f} GB/s of live table rows; {gb_prob / t_sweep * 1e3:6.0f} GB/s of problem rows)")
    print(f"shifted     : {t_shift:7.2f} ms  ({gb_tab / t_shift * 1e3:6.0f} GB/s of live table rows; {gb_prob / t_shift * 1e3:6.0f} GB/s of problem rows)   vouched for: {st.stats[:, 1].view(torch.int32).tolist()}")
    print(f"table solve : {t_all:7.2f} ms (10 iterations, both views, {2 * P} problems)")
    # the per-problem path it replaces: the patches of a view side by side, one view after the other
    def old():
        for v in range(2):
            skb = ops.SinkhornBatch.from_rows([tabs[v]] * P, rowmaps, 0.005, 10, 1.0 / rows, sorted_rows=True)
            for sk_ in skb.probs:
                sk_.cmax_preset, sk_.u_first_preset = ops.sinkhorn_first_rows(sk_.S, 0.005, sk_.rowmap)
            skb.run()
        return skb
    skb = old()
    t_old = timeit(old, n=2, warm=1)
    print(f"per-problem : {t_old:7.2f} ms (10 iterations, both views)   la diff vs table: {float((skb.la - st.la[1::2]).abs().max()):.2e}")


if __name__ == "__main__":
    main()
