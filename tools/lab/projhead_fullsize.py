"""A few full-size steps (16 images, 5 patches of 20 000 pixels, D = 512) with a non-linear projection network:
time per step, loss and peak memory.      python tools/lab/projhead_fullsize.py ['1-layer'|'2-layer'] [prototypes]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                    # noqa: E402
from ganecdotes_amd.swav_clustering import SwAVClustering      # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "2-layer"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
wl = bench.workload(16, k)
wl["swav_args"]["projn_nw"] = kind
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
torch.manual_seed(42)
np.random.seed(42)
sw.build_heads()
losses = [sw.train_step(sw.draw_batch(16)) for _ in range(2)]
torch.cuda.synchronize()
t0 = time.time()
losses += [sw.train_step(sw.draw_batch(16)) for _ in range(4)]
torch.cuda.synchronize()
dt = (time.time() - t0) / 4
l = torch.cat(losses).cpu().numpy()
print(f"projn_nw={kind} K={k}: {dt * 1e3:.1f} ms/step = {16 / dt:.1f} images/s; loss {l[0]:.4f} -> {l[-1]:.4f} "
      f"finite={bool(np.isfinite(l).all())}; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
