"""Run-to-run: three steps twice from the same state and draws, route by route — bitwise equal or not, and by how much."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from ganecdotes_amd.swav_clustering import SwAVClustering
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
def run(k, batch, scope, pdf, extra=None):
    wl = bench.workload(batch, k, scope=scope)
    wl["sinkhorn_args"]["source_pdf"] = pdf
    wl["swav_args"].update(extra or {})
    torch.manual_seed(41); np.random.seed(41)          # (the constructor draws the mean latent: seed BEFORE it)
    sw = SwAVClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, **wl)
    torch.manual_seed(42); np.random.seed(42)
    sw.build_heads()
    ls = [sw.train_step(sw.draw_batch(batch)) for _ in range(3)]
    return torch.cat(ls).cpu(), [x.clone() for x in sw._grads], [p.clone() for p in sw._params()], sw.last_route.name
for cfg in ((5000, 8, "global", "uniform", None), (5000, 4, "image", "uniform", None), (4001, 4, "image", "uniform", None),
            (200, 8, "global", "image", None), (8000, 4, "global", "uniform", None), (5000, 4, "global", "uniform", {"sinkhorn_table": False}),
            (5000, 4, "global", "uniform", {"score_table": False}), (40, 8, "global", "uniform", None),
            (8, 8, "global", "uniform", {"projection_mode": "panel"}), (8, 8, "global", "uniform", {"lowrank_proto": False})):
    a, b = run(*cfg), run(*cfg)
    same = torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1] + a[2], b[1] + b[2]))
    worst = max(float((x - y).abs().max() / (y.abs().max() + 1e-30)) for x, y in zip(a[1], b[1]))
    print(cfg, a[3], "bitwise equal:", same, "worst grad rel diff", worst, flush=True)
