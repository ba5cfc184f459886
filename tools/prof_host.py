import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from ganecdotes_amd.swav_clustering import SwAVClustering
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
wl = bench.workload(16, 8)
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
sw.build_heads()
sw.train_step(sw.draw_batch(16)); torch.cuda.synchronize()
batches = [sw.draw_batch(16) for _ in range(3)]
t = time.time()
pr = cProfile.Profile(); pr.enable()
for b in batches:
    sw.train_step(b)
torch.cuda.synchronize()
pr.disable()
print("3 steps", time.time() - t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
