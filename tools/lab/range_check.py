"""How far the modulated activations of the random-init generators sit below the f16x3 convs' range contract (|x * style| < 32752):
max |tap| per tap, max |style|, their product as a bound on |x * style|.   usage: python tools/lab/range_check.py"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ganecdotes_amd import ops
dev = torch.device("cuda")
for size in (256, 512):
    g = bench.build_generator(size, dev)
    torch.manual_seed(3)
    z = torch.randn(8, 512, device=dev)
    ml = g.mean_latent(1024)
    for trunc in (0.7, 1.0):
        w = g.style(z)
        img, feats = g([w], truncation=trunc, truncation_latent=ml, input_is_latent=True, randomize_noise=False)
        d = g._prepare()
        lat = (ml + trunc * (w - ml)).unsqueeze(1).repeat(1, g.n_latent, 1)
        sp, dm = g._styles(lat, d)
        print(size, trunc, "max |tap| per tap:", [round(float(f.abs().max()), 1) for f in feats], " max |style|:", round(float(sp.abs().max()), 2),
              " worst product bound:", round(max(float(f.abs().max()) for f in feats) * float(sp.abs().max()), 1))
