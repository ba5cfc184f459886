import sys, torch, numpy as np
sys.path.insert(0, ".")
import bench
from ganecdotes_amd.swav_clustering import SwAVClustering
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
wl = bench.workload(16, 8)
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
torch.manual_seed(42); np.random.seed(42)
sw.build_heads()
losses = []
for i in range(150):
    losses.append(sw.train_step(sw.draw_batch(16)))
l = torch.cat(losses).cpu().numpy()
print("finite", np.isfinite(l).all(), "first", l[:5], "last", l[-5:], "min", l.min(), "max", l.max())
w = sw.projection[0].weight.data
print("Wp finite", bool(torch.isfinite(w).all()), float(w.abs().max()))
