import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from ganecdotes_amd.swav_clustering import SwAVClustering
print("threads", torch.get_num_threads(), "cpus", os.cpu_count())
t = time.time(); [torch.randperm(65536) for _ in range(20)]; print("randperm", (time.time() - t) / 20)
t = time.time(); [torch.randn(1, 512) for _ in range(100)]; print("randn", (time.time() - t) / 100)
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
wl = bench.workload(16, 8)
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
sw.build_heads()
for i in range(3):
    t = time.time(); b = sw.draw_batch(16); t1 = time.time(); pb = sw.prepare_batch(b); t2 = time.time()
    sw.train_step(pb); t3 = time.time(); torch.cuda.synchronize(); t4 = time.time()
    print(f"draw {t1-t:.3f} prepare {t2-t1:.3f} launch {t3-t2:.3f} drain {t4-t3:.3f}")
