import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import bench
from ganecdotes_amd.swav_clustering import SwAVClustering
dev = torch.device("cuda", 0)
g = bench.build_generator(512, dev)
wl = bench.workload(4, 16, size=512)
wl["perturb_args"] = dict(truncation=0.7, n_layers=7, n_samples=1, layer_no=None, perturb_std=[1.0] * 7)
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
torch.manual_seed(42); np.random.seed(42)
sw.build_heads()
for i in range(3):
    l = sw.train_step(sw.draw_batch(4))
torch.cuda.synchronize(); t0 = time.time()
ls = [sw.train_step(sw.draw_batch(4)) for _ in range(10)]
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f"car-512 shape: B=4/GPU, K=16: {1e3*dt:.1f} ms/step = {4/dt:.1f} img/s; loss {float(ls[0]):.3f} -> {float(ls[-1]):.3f}; peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
