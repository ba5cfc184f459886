#!/usr/bin/env python3
"""GV15: golden vectors for the hfc_kmeans baseline.

Runs only in the build container; imports scikit-learn (1.7.2 here; the reference pins 1.0.2,
environment.yaml:351) and the reference's baseline/hfc_kmeans modules at run time and writes plain arrays.
Captured:
  A. sklearn.cluster.KMeans itself on a small float32 blob set (the third-party algorithm the path calls):
     explicit init (converged, max_iter-limited, and an init with an unreachable centre -> empty-cluster
     relocation), seeded k-means++ (n_init 1 and 3) and sklearn.cluster.kmeans_plusplus' chosen indices;
  B. the reference's HFCPreprocessor.train_hfc_model(return_aug=True) on a mini generator built from the
     reference's own classes (3 hfc layers, clusters [3, 4, 6], 3 samples per layer, kmeans_args
     random_state=0 n_init=2): the hidden-feature panels, the perturbed latents, the per-layer cluster
     centres / labels / inertia and means / stds; then predict_hfc_vectors on a second latent.
torchvision is absent from the image: `transforms.Resize(size, interpolation=NEAREST)` (used on tensors at
hfc_kmeans_clustering.py:203-206) is given as torch.nn.functional.interpolate(mode='nearest'), the call
torchvision's tensor path dispatches to — that one step is therefore pinned to torch, not to torchvision.
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_golden import REF, _Logger, _install_placeholders, _np, out_dir_arg   # noqa: E402


def blobs(seed, n=700, c=24, k=5):
    rng = np.random.RandomState(seed)
    centres = rng.randn(k, c).astype(np.float32) * 2.0 + 3.0
    x = centres[rng.randint(0, k, n)] + rng.randn(n, c).astype(np.float32)
    return np.ascontiguousarray(x, np.float32)


def main():
    out_dir = out_dir_arg(__doc__.splitlines()[0])
    _install_placeholders()
    # one thread everywhere: sklearn's threaded chunk sums (and torch's) round differently from run to run otherwise
    # (A_*_inertia +-0.004, centres +-5e-7 between two captures), and the fixture must regenerate bit for bit
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(1)
    import sklearn
    from sklearn.cluster import KMeans, kmeans_plusplus
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)

    arrs = dict(sklearn_version=np.asarray(sklearn.__version__))
    # ------------------------------------------------------------------ A: sklearn itself
    x = blobs(15)
    arrs["A_x"] = x

    def keep(tag, km):
        arrs[f"A_{tag}_centers"] = km.cluster_centers_.astype(np.float32)
        arrs[f"A_{tag}_labels"] = km.labels_.astype(np.int32)
        arrs[f"A_{tag}_inertia"] = np.float64(km.inertia_)
        arrs[f"A_{tag}_n_iter"] = np.int64(km.n_iter_)

    init = x[[3, 99, 250, 404, 600]].copy()
    arrs["A_init"] = init
    keep("explicit", KMeans(n_clusters=5, init=init, n_init=1).fit(x))
    keep("maxiter2", KMeans(n_clusters=5, init=init, n_init=1, max_iter=2).fit(x))
    far = init.copy()
    far[2] = 1000.0                                  # no sample is nearest to it -> relocated in iteration 1
    arrs["A_init_far"] = far
    keep("far", KMeans(n_clusters=5, init=far, n_init=1).fit(x))
    keep("pp1", KMeans(n_clusters=5, n_init=1, random_state=3).fit(x))
    keep("pp3", KMeans(n_clusters=7, n_init=3, random_state=11).fit(x))
    xc = x - x.mean(axis=0)
    _, idx = kmeans_plusplus(xc, 6, random_state=np.random.RandomState(5))
    arrs["A_pp_indices"] = idx.astype(np.int64)
    # fewer distinct points than clusters: relocation is pointless, _average_centers parks the empty clusters
    dup = np.repeat(x[:3], 5, axis=0)
    arrs["A_dup_x"] = dup
    for i, init_d in enumerate((np.concatenate([x[:3], x[:2] + 0.25]), np.concatenate([x[:2] + 0.25, x[:3]]))):
        arrs[f"A_dup{i}_init"] = init_d.astype(np.float32)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            keep(f"dup{i}", KMeans(n_clusters=5, init=init_d, n_init=1).fit(dup))
    arrs["A_pred_labels"] = KMeans(n_clusters=5, init=init, n_init=1).fit(x).predict(blobs(16)).astype(np.int32)

    # ------------------------------------------------------------------ B: the reference's own classes
    sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("agg")
    import types
    tvt = sys.modules["torchvision.transforms"]

    class _Resize:                                   # see the module docstring
        def __init__(self, size, interpolation=None):
            self.size = size if isinstance(size, (tuple, list)) else (size, size)

        def __call__(self, t):
            if not t.is_floating_point():        # torchvision's tensor path casts integer images to float32 and back
                return F.interpolate(t.float(), size=tuple(self.size), mode="nearest").round().to(t.dtype)
            return F.interpolate(t, size=tuple(self.size), mode="nearest")
    tvt.Resize = _Resize
    from models.stylegan2 import model as M
    from baseline.hfc_kmeans import segmentor as SEG

    channels = {4: 16, 8: 8, 16: 8, 32: 8}      # equal panel widths: the reference's np.savez of means/stds needs a
    style_dim = 16                              # rectangular list under numpy 2 (hfc_kmeans_clustering.py:89-91)
    torch.manual_seed(1500)
    g = M.Generator(32, style_dim, 2)
    g.channels = channels
    g.input = M.ConstantInput(channels[4])
    g.conv1 = M.StyledConv(channels[4], channels[4], 3, style_dim)
    g.to_rgb1 = M.ToRGB(channels[4], style_dim, upsample=False)
    g.convs, g.to_rgbs = torch.nn.ModuleList(), torch.nn.ModuleList()
    cin = channels[4]
    for i in range(3, g.log_size + 1):
        cout = channels[2 ** i]
        g.convs.append(M.StyledConv(cin, cout, 3, style_dim, upsample=True))
        g.convs.append(M.StyledConv(cout, cout, 3, style_dim))
        g.to_rgbs.append(M.ToRGB(cout, style_dim))
        cin = cout
    for m in g.modules():
        if isinstance(m, M.NoiseInjection):
            m.weight.data.fill_(0.1)
        if isinstance(m, M.FusedLeakyReLU):
            m.bias.data.normal_(0, 0.2)
    for m in g.style:
        if isinstance(m, M.EqualLinear):
            m.bias.data.normal_(0, 10.0)
    g.eval()
    for k_, v in g.state_dict().items():
        arrs[f"sd_{k_}"] = _np(v)

    n_layers, clusters, n_samples, out_size = 3, [3, 4, 6], 3, 32
    cfg = types.SimpleNamespace(num_latents_for_mean=64)
    tmp = tempfile.mkdtemp()
    prep = SEG.HFCPreprocessor(
        model=g, model_config=cfg,
        perturb_args=dict(truncation=0.7, n_layers=n_layers, n_samples=n_samples, perturb_std=[1.0, 0.6, 1.0]),
        hfc_args=dict(kmeans_args=dict(random_state=0, n_init=2, verbose=0),
                      base_args=dict(out_dir=None, n_layers=n_layers, clusters_per_layer=clusters, out_size=out_size,
                                     presaved=False)),
        hfc_algo='hfc_kmeans', hier_encode=False, train=True, out_dir=tmp, logger=_Logger())
    torch.manual_seed(1501)
    z = torch.randn(2, style_dim)
    with torch.no_grad():
        w = g.style(z)
    arrs.update(B_w=_np(w), B_clusters=np.asarray(clusters, np.int64), B_n_samples=np.int64(n_samples),
                B_out_size=np.int64(out_size), B_perturb_std=np.asarray([1.0, 0.6, 1.0]),
                B_seed_train=np.int64(1502), B_seed_predict=np.int64(1503))
    torch.manual_seed(1502)
    feats, lat_list = prep.train_hfc_model(w[0:1], return_aug=True)
    for n in range(n_layers):
        arrs[f"B_feat_{n}"] = _np(feats[n])
        km = prep.hfc_model.clusterers[n]
        arrs[f"B_centers_{n}"] = km.cluster_centers_.astype(np.float32)
        arrs[f"B_labels_{n}"] = km.labels_.astype(np.int32)
        arrs[f"B_inertia_{n}"] = np.float64(km.inertia_)
        arrs[f"B_mean_{n}"] = np.asarray(prep.hfc_model.means[n], np.float32)
        arrs[f"B_std_{n}"] = np.asarray(prep.hfc_model.stds[n], np.float32)
    arrs["B_new_latents"] = np.stack([_np(t) for t in lat_list])
    torch.manual_seed(1503)
    preds, labels = prep.predict_hfc_vectors(w[1:2])
    arrs["B_pred_maps"] = _np(preds)
    for n in range(n_layers):
        arrs[f"B_pred_labels_{n}"] = _np(labels[n]).astype(np.int64)
    stats = np.load(os.path.join(tmp, "model_stats.npz"), allow_pickle=True)
    arrs["B_saved_files"] = np.asarray(sorted(os.listdir(tmp)))
    arrs["B_stats_keys"] = np.asarray(sorted(stats.files))

    # ------------------------------------------------------------------ C: hfc_algo='hfc_kmeans_hier'
    # HierarchicalKMeansHFC.hierarchical_fit (hfc_kmeans_clustering.py:212-306, :341-390): layer n + 1 starts from the
    # centres propagated from layer n's labels (ONE scalar mean per label, every channel equal, each row repeated twice
    # — so clusters_per_layer must double from layer to layer), k-means++ only for layer 0.
    import warnings
    clusters_h = [3, 6, 12]
    tmp_h = tempfile.mkdtemp()
    prep_h = SEG.HFCPreprocessor(
        model=g, model_config=cfg,
        perturb_args=dict(truncation=0.7, n_layers=n_layers, n_samples=n_samples, perturb_std=[1.0, 0.6, 1.0]),
        hfc_args=dict(kmeans_args=dict(random_state=0, n_init=2, verbose=0),
                      base_args=dict(out_dir=None, n_layers=n_layers, clusters_per_layer=clusters_h, out_size=out_size,
                                     presaved=False)),
        hfc_algo='hfc_kmeans_hier', hier_encode=False, train=True, out_dir=tmp_h, logger=_Logger())
    arrs.update(C_clusters=np.asarray(clusters_h, np.int64), C_seed_train=np.int64(1504), C_seed_predict=np.int64(1505))
    torch.manual_seed(1504)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")              # "Explicit initial center position passed: performing only one init"
        feats_h, _ = prep_h.train_hfc_model(w[0:1], return_aug=True)
    for n in range(n_layers):
        km = prep_h.hfc_model.clusterers[n]
        arrs[f"C_feat_{n}"] = _np(feats_h[n])
        arrs[f"C_centers_{n}"] = km.cluster_centers_.astype(np.float32)
        arrs[f"C_labels_{n}"] = km.labels_.astype(np.int32)
        arrs[f"C_inertia_{n}"] = np.float64(km.inertia_)
        arrs[f"C_n_iter_{n}"] = np.int64(km.n_iter_)
        if n:
            arrs[f"C_init_{n}"] = np.asarray(km.init, np.float32)          # the propagated centres that seeded this layer
    torch.manual_seed(1505)
    preds_h, labels_h = prep_h.predict_hfc_vectors(w[1:2])
    arrs["C_pred_maps"] = _np(preds_h)
    for n in range(n_layers):
        arrs[f"C_pred_labels_{n}"] = _np(labels_h[n]).astype(np.int64)
    arrs["C_saved_files"] = np.asarray(sorted(os.listdir(tmp_h)))
    # ------------------------------------------------------------------ D: hier_encode=True in the reference itself
    # multi_sample_hierarchical_encoding (hfc_kmeans_clustering.py:484-546) draws torch.randn(1, 512) (style_dim fixed at
    # 512) and builds its features WITHOUT skip_const, so layer n's clusterer is handed the features of block n - 1 and
    # layer 0 the constant-input conv's, whose width differs from a block pair's in every StyleGAN2 config (ffhq-256: 512
    # against 1024).  A style_dim-512 mini generator with that property (12 against 8 + 8) records what the reference
    # does with it.
    torch.manual_seed(1506)
    g5 = M.Generator(32, 512, 2)
    ch5 = {4: 12, 8: 8, 16: 8, 32: 8}
    g5.channels = ch5
    g5.input = M.ConstantInput(ch5[4])
    g5.conv1 = M.StyledConv(ch5[4], ch5[4], 3, 512)
    g5.to_rgb1 = M.ToRGB(ch5[4], 512, upsample=False)
    g5.convs, g5.to_rgbs = torch.nn.ModuleList(), torch.nn.ModuleList()
    cin = ch5[4]
    for i in range(3, g5.log_size + 1):
        cout = ch5[2 ** i]
        g5.convs.append(M.StyledConv(cin, cout, 3, 512, upsample=True))
        g5.convs.append(M.StyledConv(cout, cout, 3, 512))
        g5.to_rgbs.append(M.ToRGB(cout, 512))
        cin = cout
    g5.eval()
    prep_e = SEG.HFCPreprocessor(
        model=g5, model_config=cfg,
        perturb_args=dict(truncation=0.7, n_layers=n_layers, n_samples=n_samples, perturb_std=[1.0, 0.6, 1.0]),
        hfc_args=dict(kmeans_args=dict(random_state=0, n_init=2, verbose=0),
                      base_args=dict(out_dir=None, n_layers=n_layers, clusters_per_layer=clusters, out_size=out_size,
                                     presaved=False)),
        hfc_algo='hfc_kmeans', hier_encode=True, hle_samples=2, train=True, out_dir=tempfile.mkdtemp(), logger=_Logger())
    with torch.no_grad():
        w5 = g5.style(torch.randn(1, 512))
    try:
        prep_e.train_hfc_model(w5)
        outcome = "ran"
    except Exception as e:                                                    # noqa: BLE001
        outcome = f"{type(e).__name__}: {e}"
    arrs["D_hier_encode_outcome"] = np.asarray(outcome)
    print("hier_encode=True in the reference:", outcome)

    path = os.path.join(out_dir, "gv15_kmeans.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, len(arrs), "arrays;", os.path.getsize(path), "bytes")
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in arrs.items() if not k.startswith("sd_")})


if __name__ == "__main__":
    main()
