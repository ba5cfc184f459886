"""The pixel-major swapped CE (gx_swapped_ce_pixels) at the bench shape of the reference's own K: 5 patches x 2 views x 320 000
rows x 5000 scores (64 GB), d(scores) tables of 2 x 1 048 576 pixels x 5008 (42 GB); against the per-pair fused CE."""
import sys
import torch
sys.path.insert(0, ".")
from ganecdotes_amd import ops

dev = "cuda"
P, B, N, K, HW = 5, 16, 20000, 5000, 65536
rows, npix, Kp = B * N, B * HW, 5008
torch.manual_seed(0)
S = torch.empty(P, 2, rows, K, device=dev)
for p in range(P):
    for v in range(2):
        S[p, v].normal_(0, 0.05)           # cosine scores of 512-dim unit vectors against 5000 prototypes
skb = ops.SinkhornBatch(S.reshape(P * 2, rows, K), 0.05, 10)
skb.run()                                  # the la vectors a step would hand over
inv = torch.full((P, npix), -1, dtype=torch.int32, device=dev)
nrow = torch.arange(rows, dtype=torch.int32, device=dev)
base = (torch.arange(B, device=dev) * HW).unsqueeze(1)
for p in range(P):
    picks = torch.stack([torch.randperm(HW, device=dev)[:N] for _ in range(B)])
    inv[p].scatter_(0, (picks + base).reshape(-1), nrow)
ds, dt = torch.empty(npix, Kp, device=dev), torch.empty(npix, Kp, device=dev)
loss, gb = torch.zeros(1, device=dev), torch.zeros(K, device=dev)


def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t = timeit(lambda: ops.swapped_ce_pixels(S, skb, 1, inv, 0.01, 1e-6, loss, 1e-6, 0.0, ds, dt, Kp, gb, 0.0))
gbytes = (8.0 * P * rows * K + 8.0 * npix * Kp) / 1e9
print(f"pixel-major CE, all patches: {t:7.2f} ms  {gbytes / t:6.2f} TB/s ({gbytes:.0f} GB)")
d_s, d_t = torch.empty(rows, Kp, device=dev), torch.empty(rows, Kp, device=dev)
t2 = timeit(lambda: [ops.swapped_ce_fused(skb[2 * p], skb[2 * p + 1], 0.01, 1e-6, loss, 1e-6, 0.0, d_s[:, :K], d_t[:, :K], gb, 0.0)
                     for p in range(P)])
print(f"per-pair fused CE x {P}:       {t2:7.2f} ms  {16.0 * P * rows * K / 1e9 / t2:6.2f} TB/s")
