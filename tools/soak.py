#!/usr/bin/env python3
"""Soak run: many optimizer steps queued back to back with no host sync (the host runs far ahead of the GPU),
then one read-back.  usage: soak.py [steps] [prototypes] [batch] [scope] [source_pdf] [overlap: on|off]
overlap on (default): the views of batch i + 1 queued on the view stream before step i, as pretrain() and bench.py do."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                    # noqa: E402
from ganecdotes_amd.swav_clustering import SwAVClustering      # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
scope = sys.argv[4] if len(sys.argv) > 4 else "global"
pdf = sys.argv[5] if len(sys.argv) > 5 else "uniform"
overlap = (sys.argv[6] if len(sys.argv) > 6 else "on") == "on"
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
wl = bench.workload(batch, k, scope=scope)
wl["sinkhorn_args"]["source_pdf"] = pdf
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
torch.manual_seed(42)
np.random.seed(42)
sw.build_heads()
sw._sk_probe = [] if k > 64 else None            # many prototypes: the table Sinkhorn's per-problem step statistics of every step
import time                                                     # noqa: E402
warm = min(2, steps - 1)                                        # the first steps allocate; timed from the third on
stage = sw.prefetch_views if overlap else sw.prepare_batch


def run(n):
    out, nxt = [], stage(sw.draw_batch(batch))
    for i in range(n):
        cur = nxt
        nxt = stage(sw.draw_batch(batch)) if i + 1 < n else None
        out.append(sw.train_step(cur))
    return out


losses = run(warm)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses += run(steps - warm)
l = torch.cat(losses).cpu().numpy()
ms = (time.perf_counter() - t0) / (steps - warm) * 1e3
w = sw.projection[0].weight.data
print(f"K={k} B={batch} scope={scope} pdf={pdf} overlap={'on' if overlap else 'off'}: finite={bool(np.isfinite(l).all())} loss {l[0]:.4f} -> {l[-1]:.4f} "
      f"(min {l.min():.4f}); Wp finite={bool(torch.isfinite(w).all())}; "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; {ms:.1f} ms/step")
if sw._sk_probe:
    st = torch.stack(sw._sk_probe).cpu()
    ok = st[..., 1].contiguous().view(torch.int32)
    print(f"  table Sinkhorn: {st.shape[0]} steps x {st.shape[1]} problems; last update vouched for the shifted form in "
          f"{100.0 * ok.float().mean():.1f} % of them; largest last step of la {st[..., 0].abs().max():.3f} log2 units")
