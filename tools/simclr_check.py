"""BASELINE.json config 5 (ffhq-256 hfc_with_simclr: 20 picked pixel pairs per iteration, T = 1.0, one image per iteration): a few
hundred timed iterations.   usage: python tools/simclr_check.py [iterations = 300]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                           # noqa: E402
from ganecdotes_amd.simclr_clustering import SimCLRClustering          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
if len(sys.argv) > 2:
    torch.set_num_threads(int(sys.argv[2]))
print("torch threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))
dev = torch.device("cuda", 0)
g = bench.build_generator(256, dev)
args = dict(num_iters=1, batch_size=20, patch_size=20000, hf_interp="nearest", trust_coeff=0.01,
            train_args=dict(lr=0.01, momentum=0.9), temperature=1.0, nclasses=512, hlen=5376,
            epoch_print_freq=1, max_masks=4, use_larc=True, transforms=True)
sim = SimCLRClustering(model=g, model_config=bench.ModelCfg, out_dir=None, device=dev, simclr_args=args,
                       perturb_args=dict(truncation=0.7, n_layers=6, n_samples=1, layer_no=None, perturb_std=[1.0] * 6))
torch.manual_seed(42)
np.random.seed(42)
sim.build_heads()
for _ in range(20):
    sim.train_step(sim.draw_iteration())
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = [sim.train_step(sim.draw_iteration()) for _ in range(n)]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
l = torch.cat([x.reshape(1) for x in losses]).cpu().numpy()
print(f"hfc_with_simclr ffhq-256: {1e3 * dt:.3f} ms per iteration = {1.0 / dt:.1f} images/s; loss {l[0]:.4f} -> {l[-1]:.4f}; finite={bool(np.isfinite(l).all())}")
# where an iteration goes: the host-side draw alone, the step alone (queued back to back), and the GPU time of a step
t0 = time.perf_counter()
draws = [sim.draw_iteration() for _ in range(50)]
td = (time.perf_counter() - t0) / 50
torch.cuda.synchronize()
t0 = time.perf_counter()
for d in draws:
    sim.train_step(d)
th = (time.perf_counter() - t0) / 50                     # host time to queue a step
torch.cuda.synchronize()
tq = (time.perf_counter() - t0) / 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for d in draws:
    sim.train_step(d)
e1.record()
torch.cuda.synchronize()
print(f"  draw_iteration (host) {1e3 * td:.3f} ms; train_step: host {1e3 * th:.3f} ms to queue, {1e3 * tq:.3f} ms to finish; GPU span {e0.elapsed_time(e1) / 50:.3f} ms per step")
