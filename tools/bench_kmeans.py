#!/usr/bin/env python3
"""hfc_kmeans (SURVEY.md par. 8f rank 4) on the ffhq-256 configuration: per-layer fit times of the HIP path, and
scikit-learn (the library the reference calls, 1.7.2 in this image) on the same panels on the host cores.

    python tools/bench_kmeans.py [--n_init 10] [--cpu_layers 0,1,2] [--cpu_n_init 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n_init", type=int, default=10)
    ap.add_argument("--cpu_layers", default="0,1,2,3,4")
    ap.add_argument("--cpu_n_init", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--layers", default="0,1,2,3,4")
    args = ap.parse_args()
    from ganecdotes_amd.config_defaults import kmeans_segmentor
    from ganecdotes_amd.generator import Generator
    from ganecdotes_amd.kmeans import KMeans, _pair
    dev = torch.device("cuda")
    cfg = kmeans_segmentor()
    torch.manual_seed(42)
    g = Generator(256, 512, 8).eval().to(dev)
    pc = cfg["hfc_prep_args"]["perturb_args"]
    nl, ns = pc["n_layers"], pc["n_samples"]
    mean_latent = g.mean_latent(4096)
    torch.cuda.synchronize()
    t0 = time.time()
    lat = g.style(torch.randn(nl * ns, 512, device=dev)).unsqueeze(1).repeat(1, g.n_latent, 1)
    _, feats = g([lat], truncation=0.7, truncation_latent=mean_latent, input_is_latent=True, randomize_noise=False)
    torch.cuda.synchronize()
    t_feat = time.time() - t0
    out = dict(features_s=t_feat, layers=[])
    total_gpu = 0.0
    for k in [int(v) for v in args.layers.split(",")]:
        pair = tuple(feats[2 * k + j][k * ns:(k + 1) * ns] for j in (1, 2))
        t1, t2, (b, h, w) = _pair(pair)
        K = cfg["clusters_per_layer"][k]
        KMeans(K, n_init=1, random_state=args.seed, max_iter=3).fit_taps(t1, t2)          # warm the allocator / kernels
        torch.cuda.synchronize()
        t0 = time.time()
        km = KMeans(K, n_init=args.n_init, random_state=args.seed).fit_taps(t1, t2)
        torch.cuda.synchronize()
        dt = time.time() - t0
        total_gpu += dt
        row = dict(layer=k, rows=t1.shape[0], channels=t1.shape[1] + t2.shape[1], clusters=K, hip_s=dt, n_init=args.n_init,
                   n_iter_best=km.n_iter_, inertia=km.inertia_)
        if str(k) in args.cpu_layers.split(","):
            from sklearn.cluster import KMeans as SkKMeans
            x = torch.cat([t1, t2], 1).cpu().numpy()
            t0 = time.time()
            sk = SkKMeans(n_clusters=K, n_init=args.cpu_n_init, random_state=args.seed).fit(x)
            row.update(sklearn_s=time.time() - t0, sklearn_n_init=args.cpu_n_init, sklearn_n_iter=int(sk.n_iter_),
                       sklearn_inertia=float(sk.inertia_), cores=os.cpu_count())
            if args.cpu_n_init == 1 and args.n_init == 1:
                row["labels_agree"] = float((sk.labels_ == km.labels_).mean())
        out["layers"].append(row)
        print(json.dumps(row), flush=True)
    out["hip_fit_total_s"] = total_gpu
    print(json.dumps(out))


if __name__ == "__main__":
    main()
