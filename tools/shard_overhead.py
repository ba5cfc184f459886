import sys, time, torch, numpy as np
sys.path.insert(0, ".")
import bench
from ganecdotes_amd import sharded
from ganecdotes_amd.swav_clustering import SwAVClustering
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
def run(fake):
    sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **bench.workload(16, 8))
    torch.manual_seed(42); np.random.seed(42)
    sw.build_heads()
    if fake:
        sw.world, sw.rank = 2, 0
        sw._reduce_fn = lambda t, op: None
        sharded.allreduce_flat = lambda ts, grp: ts
    for _ in range(3):
        sw.train_step(sw.draw_batch(16))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        sw.train_step(sw.draw_batch(16))
    torch.cuda.synchronize()
    return (time.time() - t0) / 10
a = run(False); b = run(True)
print(f"1-rank path {1e3*a:.2f} ms/step; sharded code path (collectives stubbed) {1e3*b:.2f} ms/step: +{1e3*(b-a):.2f} ms")
