"""The car-512 shape (BASELINE.json config 4: true 512^2 generator, 4 images per GPU, hlen = 5376 of 5504 channels): a few timed
steps.   usage: python tools/car512_check.py [prototypes = 16] [eps = 0.005] [uniform|image] [steps = 10] [tables|samples]
hfc_with_swav_car_config.py:52,65 sets 4000 prototypes and eps = 0.01"""
import sys, time, torch, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from ganecdotes_amd.swav_clustering import SwAVClustering
dev = torch.device("cuda", 0)
g = bench.build_generator(512, dev)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
pdf = sys.argv[3] if len(sys.argv) > 3 else "uniform"
nst = int(sys.argv[4]) if len(sys.argv) > 4 else 10
wl = bench.workload(4, K, size=512)
wl["sinkhorn_args"].update(eps=eps, source_pdf=pdf)
if len(sys.argv) > 5:                                   # "tables" / "samples": force the per-pixel tables of DESIGN par. 4.15 on / off
    wl["swav_args"]["ds_table"] = sys.argv[5] == "tables"
wl["perturb_args"] = dict(truncation=0.7, n_layers=7, n_samples=1, layer_no=None, perturb_std=[1.0] * 7)
sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
torch.manual_seed(42); np.random.seed(42)
sw.build_heads()
for i in range(3):
    l = sw.train_step(sw.draw_batch(4))
torch.cuda.synchronize(); t0 = time.time()
ls = [sw.train_step(sw.draw_batch(4)) for _ in range(nst)]
torch.cuda.synchronize(); dt = (time.time() - t0) / nst
print(f"car-512 shape: B=4/GPU, K={K}, eps={eps}, {pdf}: {1e3*dt:.1f} ms/step = {4/dt:.1f} img/s; loss {float(ls[0]):.3f} -> {float(ls[-1]):.3f}; peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
