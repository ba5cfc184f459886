"""TEST INFRASTRUCTURE ONLY — CPU restatement of the hfc_kmeans baseline.

Two layers are restated:
  * the reference's own code: `FlatKMeansHFC` (baseline/hfc_kmeans/hfc_kmeans_clustering.py:127-209:
    fit one KMeans per generator layer on the (pixels, channels) panel of that layer, predict = nearest
    centre -> one-hot label maps resized NEAREST to out_size) and `HFCPreprocessor`
    (baseline/hfc_kmeans/segmentor.py:66-164 train_hfc_model, :168-232 predict_hfc_vectors; feature pairing
    lib/oneshot/image_augmentor.py:78-86 with skip_const=True);
  * the third-party algorithm it calls: `sklearn.cluster.KMeans(n_clusters=K, **kmeans_args)`.  scikit-learn
    is NOT part of /root/reference; the reference pins scikit-learn 1.0.2 (environment.yaml:351) and this
    container holds 1.7.2.  What is restated is 1.7.2's dense float32 path — `KMeans.fit`
    (sklearn/cluster/_kmeans.py:1427-1554), greedy k-means++ (`_kmeans_plusplus`, :174-276),
    `_kmeans_single_lloyd` (:624-752), `lloyd_iter_chunked_dense` / `_update_chunk_dense`
    (_k_means_lloyd.pyx), `_relocate_empty_clusters_dense`, `_average_centers`, `_center_shift`,
    `_inertia_dense` (_k_means_common.pyx), `_tolerance` (:279-288), `_is_same_clustering`.
    Deltas of the pinned 1.0.2 against this (same published algorithm): n_init defaults to 10 (1.7.2:
    'auto' = 1 for k-means++), the default solver is Elkan's (an exact acceleration of the same Lloyd
    iterates), and the first k-means++ centre is drawn with randint instead of choice — the reference passes
    no random_state, so its RNG stream is not part of the contract.
Pinned by tests/golden/gv15_kmeans.npz (tools/capture_golden_kmeans.py: sklearn 1.7.2 itself with explicit
init and with seeded k-means++, and the reference's own FlatKMeansHFC / HFCPreprocessor on a mini generator).
Only tests/, __graft_entry__.smoke() and bench legs may import this module.
"""
import numpy as np

CHUNK = 256          # CHUNK_SIZE of _k_means_common.pxd: samples per E/M chunk


def _row_sumsq(a):
    return np.einsum("ij,ij->i", a, a)          # sklearn.utils.extmath.row_norms(squared=True)


def euclidean_sq(Y, X, xsq):
    """sklearn.metrics.pairwise._euclidean_distances(Y, X, Y_norm_squared=xsq, squared=True) for float32
    input: the cross term is accumulated in float64 (the `_euclidean_distances_upcast` branch), clipped at
    0 and returned in float32."""
    Y64, X64 = Y.astype(np.float64), X.astype(np.float64)
    d = -2.0 * (Y64 @ X64.T)
    d += _row_sumsq(Y64)[:, None]
    d += xsq.astype(np.float64)[None, :]
    np.maximum(d, 0, out=d)
    return d.astype(np.float32)


def kmeans_plusplus(X, k, xsq, rs, n_local_trials=None):
    """greedy k-means++ (_kmeans.py:174-276) with unit sample weights; rs = numpy RandomState"""
    n = X.shape[0]
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(k))
    w = np.ones(n, X.dtype)
    ids = np.full(k, -1, dtype=np.int64)
    ids[0] = rs.choice(n, p=w / w.sum())
    closest = euclidean_sq(X[ids[0]][None], X, xsq)
    pot = closest @ w
    for c in range(1, k):
        rand_vals = rs.uniform(size=n_local_trials) * pot
        cand = np.searchsorted(np.cumsum(w * closest, dtype=np.float64).ravel(), rand_vals.ravel())
        np.clip(cand, None, n - 1, out=cand)
        d = euclidean_sq(X[cand], X, xsq)
        np.minimum(closest, d, out=d)
        pots = d @ w.reshape(-1, 1)
        best = int(np.argmin(pots))
        pot = pots[best]
        closest = d[best][None]
        ids[c] = cand[best]
    return X[ids].copy(), ids


def _e_step(X, centers):
    """labels of _update_chunk_dense: argmin_j (|c_j|^2 - 2 x.c_j), first minimum wins, float32 GEMM per
    256-sample chunk"""
    csq = _row_sumsq(centers)
    labels = np.empty(X.shape[0], np.int32)
    for s in range(0, X.shape[0], CHUNK):
        pd = csq[None, :] - 2.0 * (X[s:s + CHUNK] @ centers.T)
        labels[s:s + CHUNK] = np.argmin(pd, axis=1)
    return labels


def _relocate_empty(X, centers_old, sums, counts, labels):
    empty = np.where(counts == 0)[0]
    if empty.size == 0:
        return
    dist = ((X - centers_old[labels]) ** 2).sum(axis=1)
    far = np.argpartition(dist, -empty.size)[:-empty.size - 1:-1]
    if dist.max() == 0:
        return
    for idx, new in enumerate(empty):
        old = labels[far[idx]]
        sums[old] -= X[far[idx]]
        sums[new] = X[far[idx]]
        counts[new] = 1
        counts[old] -= 1


def lloyd_iter(X, centers):
    """one lloyd_iter_chunked_dense with update_centers=True -> (labels, new centres, shift per centre)"""
    labels = _e_step(X, centers)
    k = centers.shape[0]
    sums = np.zeros_like(centers)
    np.add.at(sums, labels, X)
    counts = np.bincount(labels, minlength=k).astype(X.dtype)
    _relocate_empty(X, centers, sums, counts, labels)
    new = sums.copy()                                   # _average_centers, in place and in index order: an empty
    big = int(np.argmax(counts))                        # cluster (only left when relocation was pointless) copies
    for j in range(k):                                  # the biggest cluster's row AS IT IS at that moment — the
        if counts[j] > 0:                               # raw sum when big > j, the mean when big < j
            new[j] *= X.dtype.type(1.0) / counts[j]
        else:
            new[j] = new[big]
    shift = np.sqrt(((new - centers) ** 2).sum(axis=1))
    return labels, new, shift


def inertia(X, centers, labels):
    return float(((X - centers[labels]) ** 2).sum(axis=1, dtype=X.dtype).sum())


def lloyd(X, centers_init, max_iter=300, tol=0.0):
    """_kmeans_single_lloyd (:624-752) -> labels, inertia, centres, n_iter"""
    centers = centers_init.copy()
    labels_old = np.full(X.shape[0], -1, np.int32)
    strict = False
    for i in range(max_iter):
        labels, centers, shift = lloyd_iter(X, centers)
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol:
            break
        labels_old = labels
    if not strict:
        labels = _e_step(X, centers)
    return labels, inertia(X, centers, labels), centers, i + 1


def is_same_clustering(l1, l2, k):
    """_k_means_common.pyx _is_same_clustering: equal up to a permutation of the labels"""
    mapping = np.full(k, -1, np.int64)
    for a, b in zip(l1, l2):
        if mapping[a] == -1:
            mapping[a] = b
        elif mapping[a] != b:
            return False
    return True


def tolerance(X, tol):
    return np.mean(np.var(X, axis=0)) * tol


def kmeans_fit(X, k, n_init=1, max_iter=300, tol=1e-4, random_state=None, init=None):
    """KMeans(n_clusters=k, n_init, max_iter, tol, random_state, init).fit(X) for float32 X (n, c).
    init: None = 'k-means++', or an explicit (k, c) array.  Returns dict(cluster_centers_, labels_,
    inertia_, n_iter_, init_indices (last run, k-means++ only))."""
    X = np.array(X, dtype=np.float32, order="C")
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    tol_ = tolerance(X, tol)
    mean = X.mean(axis=0)
    X = X - mean
    if init is not None:
        init = np.array(init, dtype=np.float32) - mean
    xsq = _row_sumsq(X)
    best = None
    ids = None
    for _ in range(n_init):
        if init is None:
            c0, ids = kmeans_plusplus(X, k, xsq, rs)
        else:
            c0 = init.copy()
        labels, inert, centers, n_iter = lloyd(X, c0, max_iter, tol_)
        if best is None or (inert < best[1] and not is_same_clustering(labels, best[0], k)):
            best = (labels, inert, centers, n_iter)
    labels, inert, centers, n_iter = best
    return dict(cluster_centers_=centers + mean, labels_=labels, inertia_=inert, n_iter_=n_iter, init_indices=ids)


def kmeans_predict(X, centers):
    """KMeans.predict (:1066-1097): the E step against cluster_centers_ on the raw (uncentred) data"""
    return _e_step(np.ascontiguousarray(X, dtype=np.float32), np.ascontiguousarray(centers, dtype=np.float32))


# ------------------------------------------------------------------------------ the reference's own layer
def panel(feat):
    """hfeat of hfc_kmeans_clustering.py:157 / :178: (b, c, h, w) -> (b*h*w, c), rows ordered (b, y, x)"""
    b, c, h, w = feat.shape
    return np.ascontiguousarray(np.transpose(feat, (1, 0, 2, 3)).reshape(c, -1).T)


def nearest_resize_index(n_in, n_out):
    """source index of torch's 'nearest' interpolation (transforms.Resize(..., NEAREST) on a tensor,
    hfc_kmeans_clustering.py:203-206): floor(dst * in/out) with the scale in float32"""
    scale = np.float32(n_in) / np.float32(n_out)
    return np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64), n_in - 1)


def flat_fit(hidden_feat, clusters_per_layer, **kmeans_args):
    """FlatKMeansHFC.fit (:66-91, :146-166) -> per layer dict of kmeans_fit + means / stds (population std)"""
    kmeans_args.pop("verbose", None)
    out = []
    for n, feat in enumerate(hidden_feat):
        x = panel(np.asarray(feat, np.float32))
        m = kmeans_fit(x, clusters_per_layer[n], **kmeans_args)
        m["mean"], m["std"] = x.mean(axis=0), x.std(axis=0)
        out.append(m)
    return out


def propagated_centers(feat_new, labels_old, old_shape, k_old):
    """HierarchicalKMeansHFC.calculate_cluster_centers (:341-390): the old layer's labels, nearest-resized to the new
    layer's resolution; per label ONE scalar — the mean over its pixels and ALL channels of the new features
    (`lbl_vect.mean()`, :380) — in every channel of that label's row (labels that do not occur keep a zero row); every
    row repeated twice (:384-386).  -> (2 k_old, channels) float32"""
    feat_new = np.asarray(feat_new, np.float32)
    b, h, w = old_shape
    bn, cn, hn, wn = feat_new.shape
    lab = np.asarray(labels_old).reshape(b, h, w)
    lab = lab[:, nearest_resize_index(h, hn)][:, :, nearest_resize_index(w, wn)].reshape(-1)
    ff = np.transpose(feat_new, (1, 0, 2, 3)).reshape(cn, -1)
    centers = np.zeros((k_old, cn), np.float32)
    for lbl in np.unique(lab):
        centers[int(lbl), :] = ff[:, lab == lbl].mean(dtype=np.float32)
    return np.repeat(centers, 2, axis=0)


def hier_fit(hidden_feat, clusters_per_layer, **kmeans_args):
    """HierarchicalKMeansHFC.hierarchical_fit (:235-306): k-means++ for layer 0 only; layer n + 1 starts from the centres
    propagated from layer n's labels (one run: an explicit init overrides n_init, as scikit-learn does)"""
    kmeans_args.pop("verbose", None)
    out, init = [], None
    for n, feat in enumerate(hidden_feat):
        feat = np.asarray(feat, np.float32)
        kw = dict(kmeans_args)
        if init is not None:
            kw.update(init=init, n_init=1)
        m = kmeans_fit(panel(feat), clusters_per_layer[n], **kw)
        m["init"] = init
        out.append(m)
        if n + 1 < len(hidden_feat):
            b, _, h, w = feat.shape
            init = propagated_centers(hidden_feat[n + 1], m["labels_"], (b, h, w), clusters_per_layer[n])
    return out


def flat_predict(hidden_feat, centers_per_layer, out_size):
    """FlatKMeansHFC.predict (:95-110, :169-208) -> (cluster_maps (b, sum K, out, out) in {0,1},
    [labels (b, 1, h, w) per layer])"""
    maps, labels = [], []
    for feat, centers in zip(hidden_feat, centers_per_layer):
        feat = np.asarray(feat, np.float32)
        b, c, h, w = feat.shape
        lab = kmeans_predict(panel(feat), centers).reshape(b, 1, h, w)
        k = centers.shape[0]
        onehot = (lab == np.arange(k).reshape(1, k, 1, 1)).astype(np.float32)
        iy, ix = nearest_resize_index(h, out_size), nearest_resize_index(w, out_size)
        maps.append(onehot[:, :, iy][:, :, :, ix])
        labels.append(lab.astype(np.int64))
    return np.concatenate(maps, axis=1), labels


def layer_features(taps, skip_const=True):
    """image_augmentor.py:78-86: pair the two convs of every synthesis block along the channels"""
    n_layers = len(taps) // 2
    pairs = [np.concatenate([taps[2 * n + 1], taps[2 * n + 2]], axis=1) for n in range(n_layers)]
    return pairs if skip_const else [taps[0]] + pairs


# ------------------------------------------------------------------------------ HFCPreprocessor (torch, CPU)
def _burn_forward_noise(sd):
    """`model([w], return_latents=True, ...)` runs with randomize_noise=True (model.py:565 default): every
    NoiseInjection draws image.new_empty(b, 1, h, w).normal_() — consumed from the stream, result unused."""
    import torch
    from . import stylegan2_ref as G
    for i in range(G.num_layers_of(sd)):
        r = 2 ** ((i + 5) // 2)
        torch.empty(1, 1, r, r).normal_()


def train_features(sd, w, style_dim, *, num_latents_for_mean, truncation, n_layers, n_samples, perturb_std):
    """HFCPreprocessor.train_hfc_model up to the k-means fit (segmentor.py:82-148), drawing from torch's
    global CPU generator in the reference's order.  w (1, S) mapped latent.
    Returns (hidden_features [n_layers x (n_samples, C_k, r_k, r_k)], new_latent_list, draws) where draws =
    dict(mean_z, noise[k][n]) are the normal draws a HIP run is fed to reproduce the features."""
    import torch
    from . import stylegan2_ref as G
    draws = {}
    draws["mean_z"] = torch.randn(num_latents_for_mean, style_dim)
    mean_latent = G.mapping(draws["mean_z"], sd).mean(0, keepdim=True)
    _, _, lat = G.generator_forward(w, sd, truncation=truncation, truncation_latent=mean_latent, input_is_latent=True)
    _burn_forward_noise(sd)
    feats, new_list, draws["noise"] = [], [], []
    for k in range(n_layers):
        stds = [0.0] * (2 * n_layers)
        stds[2 * k] = stds[2 * k + 1] = perturb_std[k]
        pert, row = [], []
        for n in range(2 * n_layers):                       # image_augmentor.py:40-53
            z = torch.randn(n_samples, style_dim)
            row.append(z)
            base = lat[0, n].repeat(n_samples, 1)
            pert.append((1 - stds[n]) * base + stds[n] * G.mapping(z, sd))
        draws["noise"].append(row)
        new = lat.repeat(n_samples, 1, 1).clone()
        new[:, 2 * k] = pert[2 * k]
        new[:, 2 * k + 1] = pert[2 * k + 1]
        new_list += [pert[2 * k], pert[2 * k + 1]]
        _, taps, _ = G.generator_forward(new, sd, truncation=truncation, truncation_latent=mean_latent,
                                         input_is_latent=True)
        feats.append(layer_features([t.numpy() for t in taps])[k])
    return feats, new_list, draws, mean_latent


def predict_features(sd, w, style_dim, *, num_latents_for_mean, truncation, n_layers):
    """the feature half of HFCPreprocessor.predict_hfc_vectors (segmentor.py:180-205)"""
    import torch
    from . import stylegan2_ref as G
    mean_z = torch.randn(num_latents_for_mean, style_dim)
    mean_latent = G.mapping(mean_z, sd).mean(0, keepdim=True)
    _, _, lat = G.generator_forward(w, sd, truncation=truncation, truncation_latent=mean_latent, input_is_latent=True)
    _, taps, _ = G.generator_forward(lat, sd, truncation=0.7, truncation_latent=mean_latent, input_is_latent=True)
    return layer_features([t.numpy() for t in taps])[:n_layers], mean_z
