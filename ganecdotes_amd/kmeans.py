"""hfc_kmeans baseline on HIP (SURVEY.md par. 8f rank 4).

Mirrors, with the reference's names and argument meaning:
  * `KMeans`              the estimator FlatKMeansHFC builds per layer (`sklearn.cluster.KMeans(n_clusters=K,
                          **kmeans_args)`, hfc_kmeans_clustering.py:154-155): centring, tolerance scaled by the mean
                          feature variance, greedy k-means++ driven by a numpy RandomState (the same stream as
                          sklearn's when `random_state` is given), Lloyd iterations with sklearn's two stopping
                          rules, empty-cluster relocation, best-of-n_init.  Attributes `cluster_centers_`,
                          `labels_`, `inertia_`, `n_iter_`; `predict`.
  * `FlatKMeansHFC`       (hfc_kmeans_clustering.py:11-209) fit / predict over the generator layers,
                          `clusterer_layer_{n}.sav` + `model_stats.npz` on disk.
  * `HFCPreprocessor`     (baseline/hfc_kmeans/segmentor.py:11-232) train_hfc_model / predict_hfc_vectors.
n_init defaults to 10, the default of the scikit-learn the reference pins (1.0.2, environment.yaml:351).
A layer's panel is never concatenated or moved to the host: the generator's NHWC taps already are (pixels, channels)
matrices, the fit runs on them in HBM, and the label maps of all layers are written by one kernel in the NHWC layout
the one-shot head consumes.  No CPU fallback: every entry point raises without a HIP device.
  * `HierarchicalKMeansHFC` (hfc_kmeans_clustering.py:212-390, `hfc_algo='hfc_kmeans_hier'`) each layer seeded from the
                          layer before.
`hier_encode=True` raises: it raises in the reference itself (see HFCPreprocessor.__init__; the shipped config sets
it False, configs/segmentors/hfc_kmeans_config.py:22-36).
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from . import _lib, ops
from .ops import _p, _stream, lib, workspace


def _pair(feat):
    """a layer's features -> (tap1 (rows, C1), tap2 (rows, C2) or None, (b, h, w)).  Accepts the reference's
    (b, C, h, w) tensor or a (tap1, tap2) pair of them (the two convs of a synthesis block, which
    image_augmentor.py:79-81 concatenates); NCHW views of NHWC storage (what Generator returns) are used in place."""
    taps = list(feat) if isinstance(feat, (tuple, list)) else [feat]
    if len(taps) not in (1, 2):
        raise RuntimeError("a layer is one feature tensor or a pair of them")
    flat, shape = [], None
    for t in taps:
        if not t.is_cuda:
            raise RuntimeError("ganecdotes_amd.kmeans runs on HIP devices only (no CPU fallback)")
        b, c, h, w = t.shape
        if shape is not None and shape != (b, h, w):
            raise RuntimeError("the two taps of a layer differ in batch or resolution")
        shape = (b, h, w)
        flat.append(t.permute(0, 2, 3, 1).contiguous().float().reshape(b * h * w, c))
    return flat[0], (flat[1] if len(flat) == 2 else None), shape


class _Reader:
    """small device -> pinned host reads (one stream sync each)"""

    def __init__(self):
        self.buf = {}

    def __call__(self, t):
        key = (t.dtype, t.numel())
        h = self.buf.get(key)
        if h is None:
            h = self.buf[key] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
        h.copy_(t.reshape(-1), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return h.numpy()


class KMeans:
    def __init__(self, n_clusters=8, *, init='k-means++', n_init=10, max_iter=300, tol=1e-4, verbose=0,
                 random_state=None, copy_x=True, algorithm='auto'):
        if not (isinstance(init, str) and init == 'k-means++') and not hasattr(init, 'shape'):
            raise NotImplementedError("init must be 'k-means++' or an explicit (n_clusters, n_features) array")
        if algorithm not in ('auto', 'lloyd', 'full', 'elkan'):
            raise ValueError(f"unknown algorithm {algorithm!r}")
        # 'elkan' (what 'auto' means in scikit-learn 1.0.2) is an exact acceleration of the same iterates: Lloyd's here
        self.n_clusters, self.init, self.n_init, self.max_iter, self.tol = int(n_clusters), init, n_init, int(max_iter), tol
        self.verbose, self.random_state, self.copy_x, self.algorithm = verbose, random_state, copy_x, algorithm

    # ------------------------------------------------------------------ device pieces
    def _scores(self, X, centers, out):
        return ops.gemm(X, centers, False, True, out=out)

    def _assign(self, S, csq, prev, labels, onehot=None, counts=None, nchanged=None):
        _lib.check(lib().gx_kmeans_assign(_p(S), S.shape[0], S.shape[1], S.stride(0), _p(csq), _p(prev), _p(labels),
                                          _p(onehot), onehot.stride(0) if onehot is not None else 0, _p(counts),
                                          _p(nchanged), _stream()), "gx_kmeans_assign")

    def _row_dist(self, X, centers, labels, dist, total=None):
        _lib.check(lib().gx_kmeans_row_dist(_p(X), X.shape[0], X.shape[1], X.stride(0), _p(centers), centers.shape[0],
                                            _p(labels), _p(dist), _p(total), _stream()), "gx_kmeans_row_dist")

    def _kmeans_plusplus(self, X, xsq, rs, read):
        """greedy k-means++ (sklearn _kmeans_plusplus): 2 + log(K) candidates per step drawn proportionally to the
        squared distance to the closest chosen centre; the one that lowers the potential most is kept."""
        rows, k, dev = X.shape[0], self.n_clusters, X.device
        T = 2 + int(np.log(k))
        L = lib()
        wsb = L.gx_kmeans_pp_workspace_bytes(T)
        ws = workspace("kmeans_pp", wsb, dev)
        w = np.ones(rows, np.float32)
        ids = [int(rs.choice(rows, p=w / w.sum()))]
        closest = torch.empty(rows, dtype=torch.float32, device=dev)
        G = torch.empty(rows, T, dtype=torch.float32, device=dev)
        pots = torch.empty(T, dtype=torch.float64, device=dev)
        cand = torch.tensor(ids, dtype=torch.int32, device=dev)
        ops.gemm(X, X.index_select(0, cand.long()), False, True, out=G[:, :1])
        _lib.check(L.gx_kmeans_pp_commit(_p(G), rows, 1, G.stride(0), _p(xsq), _p(cand), 0, 1, _p(closest), _stream()),
                   "gx_kmeans_pp_commit")
        _lib.check(L.gx_kmeans_pp_pots(_p(G), rows, 1, G.stride(0), _p(xsq), _p(cand), None, _p(pots), _p(ws), wsb,
                                       _stream()), "gx_kmeans_pp_pots")
        pot = float(read(pots[:1])[0])
        cand = torch.empty(T, dtype=torch.int32, device=dev)
        for _ in range(1, k):
            targets = (rs.uniform(size=T) * np.float32(pot)).astype(np.float64)
            _lib.check(L.gx_kmeans_pp_sample(_p(closest), rows, targets.ctypes.data_as(ctypes.c_void_p), T, _p(cand),
                                             _p(ws), wsb, _stream()), "gx_kmeans_pp_sample")
            ops.gemm(X, X.index_select(0, cand.long()), False, True, out=G)
            _lib.check(L.gx_kmeans_pp_pots(_p(G), rows, T, G.stride(0), _p(xsq), _p(cand), _p(closest), _p(pots), _p(ws),
                                           wsb, _stream()), "gx_kmeans_pp_pots")
            hp = read(pots).copy()
            hc = read(cand)
            best = int(np.argmin(hp.astype(np.float32)))          # candidates_pot is float32 in sklearn
            pot = float(hp[best])
            ids.append(int(hc[best]))
            _lib.check(L.gx_kmeans_pp_commit(_p(G), rows, T, G.stride(0), _p(xsq), _p(cand), best, 0, _p(closest),
                                             _stream()), "gx_kmeans_pp_commit")
        idx = torch.tensor(ids, dtype=torch.long, device=dev)
        return X.index_select(0, idx).contiguous(), np.asarray(ids, np.int64)

    def _relocate(self, X, centers_old, labels, sums, counts, hcounts):
        """_relocate_empty_clusters_dense: every empty cluster takes one of the samples farthest from their
        centres (farthest first), which leaves its old cluster."""
        empty = np.where(hcounts == 0)[0]
        dist = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
        self._row_dist(X, centers_old, labels, dist)
        far = torch.topk(dist, len(empty)).indices
        if float(dist[far[0]]) == 0.0:
            return
        for new, fi in zip(empty.tolist(), far.tolist()):
            old = int(labels[fi])
            sums[old] -= X[fi]
            sums[new] = X[fi]
            counts[new] = 1
            counts[old] -= 1

    def _lloyd(self, X, c0, tol, read):
        """_kmeans_single_lloyd -> (labels int32 (rows,), inertia, centers (K, C), n_iter), all but inertia on the device"""
        rows, C = X.shape
        k, dev = self.n_clusters, X.device
        L = lib()
        cen = [c0.contiguous().clone(), torch.empty_like(c0)]
        lab = [torch.empty(rows, dtype=torch.int32, device=dev) for _ in range(2)]
        S = torch.empty(rows, k, dtype=torch.float32, device=dev)
        direct = bool(L.gx_kmeans_accumulate_supported(k))       # else: sums = onehot^T . X on the GEMM engine
        onehot = None if direct else torch.empty(rows, k, dtype=torch.float32, device=dev)
        if direct:
            wsb = L.gx_kmeans_accumulate_workspace_bytes(rows, k, C)
            ws = workspace("kmeans_acc", wsb, dev)
        sums = torch.empty(k, C, dtype=torch.float32, device=dev)
        stat = torch.zeros(2 * k + 1, dtype=torch.int32, device=dev)     # label histogram | rows that changed | shift^2
        counts, nchanged, shift2 = stat[:k], stat[k:k + 1], stat[k + 1:].view(torch.float32)
        csq = ops.row_sumsq(cen[0])
        split = int(max(1, min(64, rows // 4096)))
        prev, strict, n_iter = None, False, 0
        for i in range(self.max_iter):
            n_iter = i + 1
            cur = lab[i & 1]
            self._scores(X, cen[0], S)
            self._assign(S, csq, prev, cur, onehot, counts, nchanged)
            if direct:
                _lib.check(L.gx_kmeans_accumulate(_p(X), rows, C, X.stride(0), _p(cur), k, _p(sums), _p(ws), wsb, _stream()),
                           "gx_kmeans_accumulate")
            else:
                ops.gemm(onehot, X, True, False, out=sums, split_k=split)
            for attempt in range(2):
                _lib.check(L.gx_kmeans_update(_p(sums), _p(counts), k, C, _p(cen[0]), _p(cen[1]), _p(csq), _p(shift2),
                                              _stream()), "gx_kmeans_update")
                hs = read(stat).copy()                               # the iteration's only host sync
                if attempt or not (hs[:k] == 0).any():
                    break
                self._relocate(X, cen[0], cur, sums, counts, hs[:k])
            cen.reverse()
            if hs[k] == 0:
                strict = True
                break
            if float(hs[k + 1:].view(np.float32).astype(np.float64).sum()) <= tol:
                break
            prev = cur
        labels = lab[(n_iter - 1) & 1]
        if not strict:
            self._scores(X, cen[0], S)
            self._assign(S, csq, None, labels)
        dist = torch.empty(rows, dtype=torch.float32, device=dev)
        total = torch.empty(1, dtype=torch.float64, device=dev)
        self._row_dist(X, cen[0], labels, dist, total)
        return labels, float(read(total)[0]), cen[0], n_iter

    @staticmethod
    def _same_clustering(a, b):
        """_is_same_clustering: equal up to a relabelling"""
        pairs = np.unique(a.astype(np.int64) * (int(b.max()) + 1) + b)
        return len(pairs) == len(np.unique(a))

    # ------------------------------------------------------------------ estimator API
    def fit_taps(self, tap1, tap2=None):
        """fit on the panel [tap1 | tap2] ((rows, C1), (rows, C2) float32 HIP tensors) without forming it first"""
        if not tap1.is_cuda:
            raise RuntimeError("ganecdotes_amd.kmeans runs on HIP devices only (no CPU fallback)")
        read = _Reader()
        rows, dev = tap1.shape[0], tap1.device
        if rows < self.n_clusters:
            raise ValueError(f"n_samples={rows} should be >= n_clusters={self.n_clusters}.")
        stats = [ops.col_stats(t.contiguous()) for t in (tap1, tap2) if t is not None]
        mean = torch.cat([s[0] for s in stats])
        var = torch.cat([s[1] for s in stats])
        self.means_, self.stds_ = mean, var.sqrt()
        tol = float(var.mean()) * self.tol
        C = mean.numel()
        X = torch.empty(rows, C, dtype=torch.float32, device=dev)
        xsq = torch.empty(rows, dtype=torch.float32, device=dev)
        _lib.check(lib().gx_kmeans_center_panel(_p(tap1), tap1.shape[1], _p(tap2), 0 if tap2 is None else tap2.shape[1],
                                                rows, _p(mean), _p(X), _p(xsq), _stream()), "gx_kmeans_center_panel")
        # sklearn.utils.check_random_state: None -> numpy's GLOBAL RandomState (which the reference seeds with 42 at
        # import, lib/util/util.py:21-24, and consumes in order: its fits are reproducible), an int -> a fresh
        # RandomState(seed), a RandomState -> itself.  (The pinned scikit-learn 1.0.2 draws the first k-means++
        # centre with randint where 1.7.2, which GV15 pins, uses choice(p=...): same stream, different draw.)
        if self.random_state is None:
            rs = np.random.mtrand._rand
        elif isinstance(self.random_state, np.random.RandomState):
            rs = self.random_state
        else:
            rs = np.random.RandomState(self.random_state)
        explicit = hasattr(self.init, 'shape')
        if explicit:
            init = torch.as_tensor(np.asarray(self.init, np.float32)).to(dev) - mean
            if tuple(init.shape) != (self.n_clusters, C):
                raise ValueError(f"The shape of the initial centers {tuple(init.shape)} does not match "
                                 f"({self.n_clusters}, {C})")
        n_init = 1 if self.n_init == 'auto' else int(self.n_init)      # 'auto' (scikit-learn >= 1.2): 1 run for k-means++
        best = None
        for _ in range(1 if explicit else n_init):
            if explicit:
                c0, self.init_indices_ = init, None
            else:
                c0, self.init_indices_ = self._kmeans_plusplus(X, xsq, rs, read)
            labels, inertia, centers, n_iter = self._lloyd(X, c0, tol, read)
            hl = read(labels).copy()
            if best is None or (inertia < best[1] and not self._same_clustering(hl, best[0])):
                best = (hl, inertia, centers + mean, n_iter)
        self.labels_, self.inertia_, self.centers_device_, self.n_iter_ = best
        self.cluster_centers_ = self.centers_device_.cpu().numpy()
        self.n_features_in_ = C
        return self

    def fit(self, X, y=None, sample_weight=None):
        if sample_weight is not None:
            raise NotImplementedError("sample_weight is not used by the hfc_kmeans path")
        X = torch.as_tensor(X)
        if not X.is_cuda:
            raise RuntimeError("ganecdotes_amd.kmeans runs on HIP devices only (no CPU fallback): pass a HIP tensor")
        return self.fit_taps(X.float().contiguous())

    def set_centers(self, centers, device):
        """adopt fitted centres (a loaded sklearn / saved model): (K, C) array"""
        self.cluster_centers_ = np.ascontiguousarray(centers, np.float32)
        self.centers_device_ = torch.from_numpy(self.cluster_centers_).to(device)
        self.n_clusters = self.cluster_centers_.shape[0]
        return self

    def predict_taps(self, tap1, tap2=None):
        """KMeans.predict on the raw panel [tap1 | tap2]: int32 HIP labels (rows,)"""
        cen = self.centers_device_
        rows, dev = tap1.shape[0], tap1.device
        S = torch.empty(rows, cen.shape[0], dtype=torch.float32, device=dev)
        if tap2 is None:
            ops.gemm(tap1, cen, False, True, out=S)
        elif tap1.shape[1] % 32 == 0:
            ops.gemm_cat(tap1, tap2, cen, S)
        else:
            X = torch.empty(rows, cen.shape[1], dtype=torch.float32, device=dev)
            _lib.check(lib().gx_kmeans_center_panel(_p(tap1), tap1.shape[1], _p(tap2), tap2.shape[1], rows, None, _p(X), None,
                                                    _stream()), "gx_kmeans_center_panel")
            ops.gemm(X, cen, False, True, out=S)
        labels = torch.empty(rows, dtype=torch.int32, device=dev)
        self._assign(S, ops.row_sumsq(cen.contiguous()), None, labels)
        return labels

    def predict(self, X):
        X = torch.as_tensor(X)
        if not X.is_cuda:
            raise RuntimeError("ganecdotes_amd.kmeans runs on HIP devices only (no CPU fallback): pass a HIP tensor")
        return self.predict_taps(X.float().contiguous())


def label_maps(labels, res, ks, batch, out_size, hi=1.0, lo=-1.0):
    """one-hot maps of all layers, nearest-resized to out_size: NHWC (B, out, out, sum K) with values hi / lo"""
    dev = labels[0].device
    out = torch.empty(batch, out_size, out_size, int(sum(ks)), dtype=torch.float32, device=dev)
    n = len(labels)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in labels])
    _lib.check(lib().gx_kmeans_label_maps(n, ptrs, (ctypes.c_int * n)(*res), (ctypes.c_int * n)(*ks), batch, out_size,
                                          hi, lo, _p(out), _stream()), "gx_kmeans_label_maps")
    return out


class BaseHFCModel(object):
    """hfc_kmeans_clustering.py:11-124: per-layer clusterers, `clusterer_layer_{n}.sav`, `model_stats.npz`."""

    def __init__(self, out_dir, n_layers=6, clusters_per_layer=[], out_size=128, presaved=False, logger=None,
                 device='cuda'):
        self.out_dir = out_dir
        os.makedirs(self.out_dir, exist_ok=True)
        self.n_layer, self.clusters_per_layer, self.out_size = n_layers, list(clusters_per_layer), out_size
        self.presaved, self.logger, self.device = presaved, logger, torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("ganecdotes_amd.kmeans runs on HIP devices only (no CPU fallback)")
        lib()
        self.model_fpaths = [os.path.join(self.out_dir, f"clusterer_layer_{n}.sav") for n in range(self.n_layer)]
        self.stats_file = os.path.join(self.out_dir, 'model_stats.npz')
        if self.presaved and os.path.exists(self.stats_file):
            stats = np.load(self.stats_file, allow_pickle=True)
            self.means, self.stds = stats['means'], stats['stds']
        else:
            self.stds = self.means = [None] * len(self.clusters_per_layer)       # one list, as in the reference (:54)
        if self.presaved:
            self.load()

    def load(self):
        if not all(os.path.exists(fp) for fp in self.model_fpaths):
            raise FileNotFoundError('Models not found - use BaseHFCModel.fit() to create model first!')
        self.clusterers = []
        for fp in self.model_fpaths:                         # a pickled sklearn KMeans (the reference's files) or ours
            with open(fp, 'rb') as f:
                m = pickle.load(f)
            self.clusterers.append(KMeans(m.cluster_centers_.shape[0]).set_centers(m.cluster_centers_, self.device))

    def fit(self, hidden_feat):
        assert len(hidden_feat) == self.n_layer
        self.clusterers = []
        for n in range(self.n_layer):
            model = self._layerwise_fit(hidden_feat[n], n)
            self.clusterers.append(model)
            self.save_model(model, n)
            msg = f"Fitted model for Layer {n}"
            print(msg) if self.logger is None else self.logger.info(msg)
        stats = dict(means=np.empty(len(self.means), object), stds=np.empty(len(self.stds), object))
        for i in range(len(self.means)):                     # ragged across layers: object arrays (numpy >= 1.24)
            stats['means'][i], stats['stds'][i] = self.means[i], self.stds[i]
        np.savez_compressed(self.stats_file, **stats)

    def predict(self, hidden_feat):
        """-> (cluster_maps (B, sum K, out, out) in {0, 1}, an NCHW view of the NHWC maps; [labels (B, 1, h, w)])"""
        maps, labels = self.predict_nhwc(hidden_feat, hi=1.0, lo=0.0)
        return maps.permute(0, 3, 1, 2), labels

    def predict_nhwc(self, hidden_feat, hi=1.0, lo=-1.0):
        assert len(hidden_feat) == self.n_layer
        if not hasattr(self, 'clusterers'):      # evaluate.py's order (predict without fit in this process): the
            self.load()                          # reference raises AttributeError there; the saved models are loaded
        labs, res, batch = [], [], None
        for n in range(self.n_layer):
            t1, t2, (b, h, w) = _pair(hidden_feat[n])
            if h != w:
                raise RuntimeError("square feature maps expected")
            batch = b
            labs.append(self.clusterers[n].predict_taps(t1, t2))
            res.append(h)
        ks = [c.n_clusters for c in self.clusterers]
        maps = label_maps(labs, res, ks, batch, self.out_size, hi, lo)
        return maps, [l.reshape(batch, 1, r, r).long() for l, r in zip(labs, res)]

    def _layerwise_fit(self, feat, n):
        return

    def save_model(self, model, n):
        """a fitted sklearn.cluster.KMeans when scikit-learn is importable (the reference's on-disk format:
        evaluate.py --method hfc_kmeans unpickles it), else a plain attribute holder"""
        try:
            from sklearn.cluster import KMeans as SkKMeans
            m = SkKMeans(n_clusters=model.n_clusters)
            m._n_threads = 1
        except ImportError:
            m = SavedKMeans()
        m.cluster_centers_ = np.asarray(model.cluster_centers_, np.float32)
        m.labels_, m.inertia_, m.n_iter_ = model.labels_, model.inertia_, model.n_iter_
        m.n_features_in_ = m.cluster_centers_.shape[1]
        m._n_features_out = m.cluster_centers_.shape[0]
        with open(self.model_fpaths[n], 'wb') as f:
            pickle.dump(m, f)


class SavedKMeans(object):
    pass


class FlatKMeansHFC(BaseHFCModel):
    def __init__(self, kmeans_args, base_args):
        self.kmeans_args = kmeans_args.copy()
        BaseHFCModel.__init__(self, **base_args)

    def _layerwise_fit(self, feat, n):
        t1, t2, _ = _pair(feat)
        clusterer = KMeans(n_clusters=self.clusters_per_layer[n], **self.kmeans_args).fit_taps(t1, t2)
        # the reference assigns `self.means[n], self.stds[n] = mean, std` on ONE aliased list (:54, :159-160):
        # both entries end up holding the std
        self.means[n] = clusterer.means_.cpu().numpy()
        self.stds[n] = clusterer.stds_.cpu().numpy()
        return clusterer


class HierarchicalKMeansHFC(BaseHFCModel):
    """hfc_kmeans_clustering.py:212-390 (`hfc_algo='hfc_kmeans_hier'`): layer 0 is fitted like FlatKMeansHFC's, every
    later layer starts Lloyd's iterations from centres propagated from the layer before — the labels of layer n,
    nearest-resized to layer n + 1's resolution, give ONE scalar per label (the mean over that label's pixels AND all
    channels of layer n + 1, `lbl_vect.mean()` at :380), written to every channel of the label's row and each row
    repeated twice (:384-386) — so clusters_per_layer has to double from layer to layer, and the duplicated rows are told
    apart by the empty-cluster relocation of the first iteration.  No `model_stats.npz` is written (only fit() does)."""

    def __init__(self, kmeans_args, base_args):
        self.kmeans_args = kmeans_args.copy()
        BaseHFCModel.__init__(self, **base_args)

    def hierarchical_fit(self, hidden_feat):
        assert len(hidden_feat) == self.n_layer
        self.clusterers = []
        self._cluster_centers = None
        for n in range(self.n_layer):
            model = self._layerwise_fit(hidden_feat[n], n)
            if n != self.n_layer - 1:
                self._cluster_centers = self.calculate_cluster_centers(hidden_feat[n], hidden_feat[n + 1], model.labels_, n + 1)
            self.clusterers.append(model)
            self.save_model(model, n)
            msg = f"Fitted model for Layer {n}"
            print(msg) if self.logger is None else self.logger.info(msg)

    def _layerwise_fit(self, feat, n):
        t1, t2, _ = _pair(feat)
        kw = dict(self.kmeans_args)
        if getattr(self, '_cluster_centers', None) is not None:
            kw['init'] = self._cluster_centers            # an explicit init: one run, whatever n_init says (as in scikit-learn)
        return KMeans(n_clusters=self.clusters_per_layer[n], **kw).fit_taps(t1, t2)

    def calculate_cluster_centers(self, feat_old, feat_new, labels, n):
        """-> (2 * clusters_per_layer[n - 1], channels of layer n) float32 numpy: the init of layer n's k-means"""
        _, _, (b, h, w) = _pair(feat_old)
        t1, t2, (bn, hn, wn) = _pair(feat_new)
        dev = t1.device
        lab = torch.as_tensor(np.asarray(labels)).to(dev).reshape(b, h, w).long()
        iy = (torch.arange(hn, device=dev) * h) // hn                 # Resize(NEAREST): source = floor(dest * in / out)
        ix = (torch.arange(wn, device=dev) * w) // wn
        lab = lab[:, iy][:, :, ix].reshape(-1)
        k_old = self.clusters_per_layer[n - 1]
        rowsum = t1.sum(1, dtype=torch.float64)
        cn = t1.shape[1]
        if t2 is not None:
            rowsum = rowsum + t2.sum(1, dtype=torch.float64)
            cn += t2.shape[1]
        tot = torch.zeros(k_old, dtype=torch.float64, device=dev).index_add_(0, lab, rowsum)
        cnt = torch.bincount(lab, minlength=k_old).double()
        val = torch.where(cnt > 0, tot / (cnt.clamp(min=1) * cn), torch.zeros_like(tot))      # absent labels keep a zero row
        centers = val.float().cpu().numpy()[:, None].repeat(cn, axis=1)
        return np.repeat(centers, 2, axis=0)


class HFCPreprocessor(object):
    """baseline/hfc_kmeans/segmentor.py:11-232"""

    def __init__(self, model, model_config, perturb_args, hfc_args, hfc_algo='hfc_kmeans', hier_encode=True,
                 hle_samples=500, train=True, out_dir=None, logger=None, presaved=None):
        self.model, self.perturb_config, self.hfc_args = model, perturb_args, hfc_args
        self.hier_encode, self.hfc_algo, self.out_dir, self.train = hier_encode, hfc_algo, out_dir, train
        self.logger, self.model_config, self.hle_samples = logger, model_config, hle_samples
        assert self.hfc_algo in ['hfc_kmeans', 'hfc_kmeans_hier']
        if self.hier_encode:
            # the reference's own hier_encode=True cannot run: multi_sample_hierarchical_encoding
            # (hfc_kmeans_clustering.py:484-546) builds its features without skip_const, so layer 0's clusterer is handed
            # the constant-input conv's features and scikit-learn raises "X has 512 features, but KMeans is expecting 1024"
            # in train_hfc_model (GV15 `D_hier_encode_outcome` holds the reference's exception on a mini generator)
            raise NotImplementedError("hier_encode=True raises in the reference itself (feature-count mismatch in "
                                      "multi_sample_hierarchical_encoding); the shipped hfc_kmeans config sets it False")
        self.device = next(model.parameters()).device
        if self.device.type != 'cuda':
            raise RuntimeError("ganecdotes_amd.kmeans runs on HIP devices only (no CPU fallback)")
        self.hfc_args['base_args']['out_dir'] = self.out_dir
        self.hfc_args['base_args']['logger'] = self.logger
        if presaved is not None:                             # pretrain.py / evaluate.py set hfc_prep_args['presaved']
            self.hfc_args['base_args']['presaved'] = presaved
        cls = FlatKMeansHFC if self.hfc_algo == 'hfc_kmeans' else HierarchicalKMeansHFC
        self.hfc_model = cls(hfc_args['kmeans_args'], dict(hfc_args['base_args'], device=self.device))
        self.trained_beliefs = None

    # the normal draws of the path, in the reference's order (tests feed recorded draws through this hook)
    def _randn(self, *shape):
        return torch.randn(*shape, device=self.device)

    def _mean_latent(self):
        z = self._randn(self.model_config.num_latents_for_mean, self.model.style_dim)
        return self.model.style(z).mean(0, keepdim=True)

    def _features(self, latents, truncation, mean_latent):
        """create_images_and_features_from_perturbed_latents(..., skip_const=True) (image_augmentor.py:56-86):
        the two convs of every synthesis block, as (tap, tap) pairs"""
        _, feats = self.model([latents], truncation=truncation, truncation_latent=mean_latent, input_is_latent=True,
                              randomize_noise=False)
        return [(feats[2 * n + 1], feats[2 * n + 2]) for n in range(len(feats) // 2)]

    def train_hfc_model(self, input_latent, return_aug=False):
        pc = self.perturb_config
        nl, ns = pc['n_layers'], pc['n_samples']
        mean_latent = self._mean_latent()
        truncation = pc['truncation']
        if input_latent.dim() == 1:
            input_latent = input_latent.unsqueeze(0)
        elif input_latent.dim() == 2 and input_latent.shape[0] > 1:
            input_latent = input_latent.unsqueeze(0)
        _, w_latents = self.model([input_latent.to(self.device)], return_latents=True, truncation_latent=mean_latent,
                                  truncation=truncation, input_is_latent=True, randomize_noise=False)
        # every layer's support set in ONE generator batch: layer k perturbs latents 2k and 2k+1 of n_samples copies
        noise = torch.stack([torch.stack([self._randn(ns, self.model.style_dim) for _ in range(2 * nl)])
                             for _ in range(nl)])                                   # (nl, 2 nl, ns, S)
        mapped = self.model.style(noise.reshape(-1, self.model.style_dim)).reshape(noise.shape)
        new_latents = w_latents.repeat(nl * ns, 1, 1).reshape(nl, ns, -1, self.model.style_dim).clone()
        new_latent_list = []
        for k in range(nl):
            std = float(pc['perturb_std'][k])
            for j in (2 * k, 2 * k + 1):
                pert = (1 - std) * w_latents[0, j].unsqueeze(0) + std * mapped[k, j]
                new_latents[k, :, j] = pert
                new_latent_list.append(pert)
        pairs = self._features(new_latents.reshape(nl * ns, -1, self.model.style_dim), truncation, mean_latent)
        hidden_features = [tuple(t[k * ns:(k + 1) * ns] for t in pairs[k]) for k in range(nl)]
        for k in range(nl):
            self.logger.info(f"Generated features for Layer: {k}") if self.logger else None
        if self.hfc_algo == 'hfc_kmeans_hier':
            self.hfc_model.hierarchical_fit(hidden_features)
        else:
            self.hfc_model.fit(hidden_features)
        if return_aug:
            return [torch.cat(p, 1) for p in hidden_features], new_latent_list

    def predict_hfc_nhwc(self, input_latent):
        """predict_hfc_vectors in the head's layout: (B, out, out, sum K) with values +-1, [labels]"""
        mean_latent = self._mean_latent()
        truncation = self.perturb_config['truncation']
        _, w_latents = self.model([input_latent.to(self.device)], return_latents=True, truncation_latent=mean_latent,
                                  truncation=truncation, input_is_latent=True, randomize_noise=False)
        pairs = self._features(w_latents, 0.7, mean_latent)          # the reference hard-codes 0.7 here (:195)
        return self.hfc_model.predict_nhwc(pairs[:self.perturb_config['n_layers']])

    def predict_hfc_vectors(self, input_latent):
        maps, labels = self.predict_hfc_nhwc(input_latent)
        return maps.permute(0, 3, 1, 2), labels
