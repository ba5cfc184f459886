"""hfc_kmeans baseline on HIP (SURVEY.md par. 8f rank 4) against the golden vectors captured from
scikit-learn and from the reference's FlatKMeansHFC / HFCPreprocessor (GV15) and against the CPU oracle.
Runs on a real MI355X only."""
import ctypes
import os
import pickle
import types

import numpy as np
import pytest
import torch

from conftest import golden_sd, load_golden

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ganecdotes_amd import ops
    ops.lib()
    return torch.device("cuda")


@pytest.fixture(params=[0, 1], ids=["f32mfma", "bf16x6"])
def engine(request, gpu):
    from ganecdotes_amd import ops
    prev = ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(prev)


def blobs(seed, n=700, c=24, k=5, spread=2.0):
    rng = np.random.RandomState(seed)
    centres = rng.randn(k, c).astype(np.float32) * spread + 3.0
    return np.ascontiguousarray(centres[rng.randint(0, k, n)] + rng.randn(n, c).astype(np.float32), np.float32)


# ------------------------------------------------------------------------------------- kernels
def test_assign_ties_nan_histogram(gpu):
    from ganecdotes_amd.kmeans import KMeans
    rng = np.random.RandomState(0)
    for rows, k in ((1000, 5), (333, 64), (4100, 130), (17, 3)):
        S = rng.randn(rows, k).astype(np.float32)
        csq = rng.rand(k).astype(np.float32)
        S[3, :] = 0.25
        csq_t = csq.copy()
        if k >= 5:
            csq_t[1] = csq_t[4] = csq_t.min() - 1.0             # an exact tie in row 3: the first index wins
        S[5, :] = np.nan
        want = np.argmin(np.where(np.isnan(S), np.inf, csq_t[None] - 2 * S), axis=1)
        want[5] = 0
        prev = want.copy().astype(np.int32)
        prev[::7] = (prev[::7] + 1) % k
        labels = torch.empty(rows, dtype=torch.int32, device=gpu)
        onehot = torch.full((rows, k), 7.0, device=gpu)
        counts = torch.full((k,), -1, dtype=torch.int32, device=gpu)
        nch = torch.full((1,), -1, dtype=torch.int32, device=gpu)
        KMeans(k)._assign(T(S).cuda(), T(csq_t).cuda(), T(prev).cuda(), labels, onehot, counts, nch)
        np.testing.assert_array_equal(labels.cpu().numpy(), want)
        np.testing.assert_array_equal(onehot.cpu().numpy(), np.eye(k, dtype=np.float32)[want])
        np.testing.assert_array_equal(counts.cpu().numpy(), np.bincount(want, minlength=k))
        assert int(nch) == int((prev != want).sum())
        KMeans(k)._assign(T(S).cuda(), T(csq_t).cuda(), None, labels, None, None, nch)
        assert int(nch) == rows


def test_accumulate_sums(gpu):
    """per-cluster sums in one pass over X: LDS accumulators, fixed-order chunk reduction"""
    from ganecdotes_amd import _lib
    from ganecdotes_amd.ops import _p, _stream, lib, workspace
    rng = np.random.RandomState(4)
    L = lib()
    assert L.gx_kmeans_accumulate_supported(256) and not L.gx_kmeans_accumulate_supported(257)
    for rows, c, k in ((70000, 96, 64), (513, 300, 5), (4099, 1024, 130), (65, 8, 256)):
        x = rng.randn(rows, c).astype(np.float32)
        lab = rng.randint(0, k, rows).astype(np.int32)
        lab[: rows // 3] = 1                                   # long runs of one label
        dx, dl = T(x).cuda(), T(lab).cuda()
        sums = torch.empty(k, c, device=gpu)
        wsb = L.gx_kmeans_accumulate_workspace_bytes(rows, k, c)
        ws = workspace("kmeans_acc", wsb, gpu)
        for _ in range(2):
            _lib.check(L.gx_kmeans_accumulate(_p(dx), rows, c, c, _p(dl), k, _p(sums), _p(ws), wsb, _stream()), "acc")
            want = np.zeros((k, c), np.float64)
            np.add.at(want, lab, x.astype(np.float64))
            got = sums.cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-5 * np.abs(x).max() * max(1, rows // k))
        first = got.copy()
        _lib.check(L.gx_kmeans_accumulate(_p(dx), rows, c, c, _p(dl), k, _p(sums), _p(ws), wsb, _stream()), "acc")
        np.testing.assert_array_equal(sums.cpu().numpy(), first)               # reproducible bit for bit


def test_many_clusters_onehot_path(gpu, engine):
    """more than 256 clusters: the sums come from onehot^T . X on the GEMM engine"""
    from oracle import kmeans_ref as K
    from ganecdotes_amd.kmeans import KMeans
    x = blobs(31, n=6000, c=32, k=40, spread=1.5)
    init = x[np.random.RandomState(0).choice(6000, 300, replace=False)]
    ref = K.kmeans_fit(x, 300, n_init=1, init=init, max_iter=15)
    km = KMeans(300, init=init, n_init=1, max_iter=15).fit(T(x).cuda())
    assert (km.labels_ == ref["labels_"]).mean() > 0.995
    assert abs(km.inertia_ - ref["inertia_"]) <= 1e-3 * ref["inertia_"]


def test_random_state_none_uses_the_seeded_global_stream(gpu):
    """random_state=None (the shipped kmeans_args) = sklearn's check_random_state(None): numpy's GLOBAL
    RandomState, which the reference seeds with 42 at import (lib/util/util.py:21-24) — the fit is
    reproducible under that seed, equals the fit driven by an explicit RandomState(42), and advances
    the global stream."""
    from ganecdotes_amd.kmeans import KMeans
    rng = np.random.RandomState(5)
    x = (rng.randn(400, 12) + 3 * rng.randint(0, 4, (400, 1))).astype(np.float32)
    np.random.seed(42)
    a = KMeans(5, n_init=2).fit(T(x).cuda())
    after = np.random.rand()
    np.random.seed(42)
    b = KMeans(5, n_init=2).fit(T(x).cuda())
    c = KMeans(5, n_init=2, random_state=np.random.RandomState(42)).fit(T(x).cuda())
    np.testing.assert_array_equal(a.labels_, b.labels_)
    np.testing.assert_array_equal(a.labels_, c.labels_)
    np.random.seed(42)
    assert after != np.random.rand()                      # the global stream was consumed by the fit


def test_edge_cases_vs_oracle(gpu):
    """one cluster, as many clusters as samples, odd channel counts, duplicated samples (fewer distinct points
    than clusters: relocation is pointless and the degenerate centres stay put), too few samples"""
    from oracle import kmeans_ref as K
    from ganecdotes_amd.kmeans import KMeans
    rng = np.random.RandomState(41)
    x = rng.randn(50, 7).astype(np.float32) + 1.5
    for k in (1, 3):
        ref = K.kmeans_fit(x, k, n_init=1, random_state=2)
        km = KMeans(k, n_init=1, random_state=2).fit(T(x).cuda())
        np.testing.assert_array_equal(km.labels_, ref["labels_"])
        np.testing.assert_allclose(km.cluster_centers_, ref["cluster_centers_"], atol=2e-5)
        assert km.n_iter_ == ref["n_iter_"]
    small = x[:6]
    ref = K.kmeans_fit(small, 6, n_init=1, init=small[::-1].copy())
    km = KMeans(6, init=small[::-1].copy(), n_init=1).fit(T(small).cuda())
    np.testing.assert_array_equal(km.labels_, ref["labels_"])
    assert km.inertia_ <= 1e-8 and ref["inertia_"] <= 1e-8
    dup = np.repeat(x[:3], 5, axis=0)                      # 3 distinct points, 5 clusters
    init = np.concatenate([x[:3], x[:2] + 0.25])
    ref = K.kmeans_fit(dup, 5, n_init=1, init=init)
    km = KMeans(5, init=init, n_init=1).fit(T(dup).cuda())
    np.testing.assert_array_equal(km.labels_, ref["labels_"])
    assert len(set(km.labels_.tolist())) == 3
    g = load_golden("gv15_kmeans")                         # the same situation as sklearn itself resolves it
    for i in (0, 1):
        km = KMeans(5, init=g[f"A_dup{i}_init"], n_init=1).fit(T(g["A_dup_x"]).cuda())
        np.testing.assert_array_equal(km.labels_, g[f"A_dup{i}_labels"])
        assert km.n_iter_ == int(g[f"A_dup{i}_n_iter"])
        used = sorted(set(km.labels_.tolist()))
        np.testing.assert_allclose(km.cluster_centers_[used], g[f"A_dup{i}_centers"][used], rtol=0, atol=2e-5)
        # the second iteration's means differ from the points by rounding residue, so sklearn relocates the two empty
        # clusters onto whichever duplicates carry the largest residue — a choice made by summation order; here as
        # there they land on data points
        pts = g["A_dup_x"][::5]
        for j in set(range(5)) - set(used):
            assert np.abs(pts - km.cluster_centers_[j]).max(axis=1).min() < 2e-5
            assert np.abs(pts - g[f"A_dup{i}_centers"][j]).max(axis=1).min() < 2e-5
    with pytest.raises(ValueError):
        KMeans(8, n_init=1).fit(T(x[:5]).cuda())
    with pytest.raises(RuntimeError):                      # no CPU fallback
        KMeans(2).fit(T(x))


def test_update_rowdist_center_panel(gpu):
    from ganecdotes_amd import _lib
    from ganecdotes_amd.kmeans import KMeans
    from ganecdotes_amd.ops import _p, _stream, lib
    rng = np.random.RandomState(1)
    k, c, rows = 7, 100, 901
    sums, old = rng.randn(k, c).astype(np.float32), rng.randn(k, c).astype(np.float32)
    counts = np.array([3, 1, 0, 20, 5, 9, 2], np.int32)
    new = torch.empty(k, c, device=gpu)
    csq, sh = torch.empty(k, device=gpu), torch.empty(k, device=gpu)
    dsums, dcounts, dold = T(sums).cuda(), T(counts).cuda(), T(old).cuda()
    _lib.check(lib().gx_kmeans_update(_p(dsums), _p(dcounts), k, c, _p(dold), _p(new), _p(csq), _p(sh), _stream()), "update")
    want = sums * (np.float32(1) / np.maximum(counts, 1).astype(np.float32))[:, None]
    want[2] = sums[3]              # empty cluster below the biggest one (k = 3): sklearn's in-place loop copies the raw sum
    np.testing.assert_allclose(new.cpu().numpy(), want, rtol=1e-6)
    np.testing.assert_allclose(sh.cpu().numpy(), ((want - old) ** 2).sum(1), rtol=1e-5)
    np.testing.assert_allclose(csq.cpu().numpy(), (want ** 2).sum(1), rtol=1e-5)
    counts2 = np.array([30, 1, 0, 20, 5, 9, 2], np.int32)      # ... above it (k = 0): the mean
    dc2 = T(counts2).cuda()
    _lib.check(lib().gx_kmeans_update(_p(dsums), _p(dc2), k, c, _p(dold), _p(new), _p(csq), _p(sh), _stream()), "update")
    np.testing.assert_allclose(new.cpu().numpy()[2], sums[0] * (np.float32(1) / np.float32(30)), rtol=1e-6)
    x = rng.randn(rows, c).astype(np.float32)
    lab = rng.randint(0, k, rows).astype(np.int32)
    dist = torch.empty(rows, device=gpu)
    tot = torch.empty(1, dtype=torch.float64, device=gpu)
    KMeans(k)._row_dist(T(x).cuda(), T(old).cuda(), T(lab).cuda(), dist, tot)
    wd = ((x.astype(np.float64) - old[lab]) ** 2).sum(1)
    np.testing.assert_allclose(dist.cpu().numpy(), wd, rtol=1e-5)
    np.testing.assert_allclose(float(tot), wd.sum(), rtol=1e-6)
    t1, t2, mean = x[:, :36].copy(), x[:, 36:].copy(), rng.randn(c).astype(np.float32)
    X, xsq = torch.empty(rows, c, device=gpu), torch.empty(rows, device=gpu)
    d1, d2, dmean = T(t1).cuda(), T(t2).cuda(), T(mean).cuda()
    _lib.check(lib().gx_kmeans_center_panel(_p(d1), 36, _p(d2), c - 36, rows, _p(dmean), _p(X), _p(xsq), _stream()), "panel")
    np.testing.assert_array_equal(X.cpu().numpy(), x - mean)
    np.testing.assert_allclose(xsq.cpu().numpy(), ((x - mean).astype(np.float64) ** 2).sum(1), rtol=1e-5)


def test_kmeanspp_sample_pots_commit(gpu):
    from ganecdotes_amd import _lib, ops
    from ganecdotes_amd.ops import _p, _stream, lib, workspace
    rng = np.random.RandomState(2)
    for rows in (50, 1024, 70001):
        closest = (rng.rand(rows).astype(np.float32) ** 4) * 3
        closest[rng.randint(0, rows, 5)] = 0.0
        cum = np.cumsum(closest, dtype=np.float64)
        targets = np.concatenate([rng.rand(4) * cum[-1], [0.0, cum[-1] * (1 + 1e-9), cum[10], cum[-1]]])
        want = np.minimum(np.searchsorted(cum, targets), rows - 1)
        cand = torch.empty(len(targets), dtype=torch.int32, device=gpu)
        dc = T(closest).cuda()
        wsb = lib().gx_kmeans_pp_workspace_bytes(len(targets))
        ws = workspace("kmeans_pp", wsb, gpu)
        _lib.check(lib().gx_kmeans_pp_sample(_p(dc), rows, targets.ctypes.data_as(ctypes.c_void_p), len(targets), _p(cand),
                                             _p(ws), wsb, _stream()), "sample")
        got = cand.cpu().numpy()
        # a knife-edge target (== a cumsum entry) may resolve one slot apart under a different summation order
        assert np.all(np.abs(got - want) <= 1), (rows, got, want)
        assert np.array_equal(got[:5], want[:5])
    rows, c, t = 5000, 40, 5
    x = rng.randn(rows, c).astype(np.float32)
    xsq = (x ** 2).sum(1).astype(np.float32)
    ids = rng.randint(0, rows, t).astype(np.int32)
    closest = (rng.rand(rows) * 60).astype(np.float32)
    dx, dcand = T(x).cuda(), T(ids).cuda()
    G = ops.gemm(dx, dx[dcand.long()].contiguous(), False, True)
    D = np.maximum(xsq[:, None] + xsq[ids][None] - 2 * (x.astype(np.float64) @ x[ids].T.astype(np.float64)), 0)
    wsb = lib().gx_kmeans_pp_workspace_bytes(t)
    ws = workspace("kmeans_pp", wsb, gpu)
    pots = torch.empty(t, dtype=torch.float64, device=gpu)
    dxsq, dcl = T(xsq).cuda(), T(closest).cuda()
    for cl in (None, closest):
        _lib.check(lib().gx_kmeans_pp_pots(_p(G), rows, t, G.stride(0), _p(dxsq), _p(dcand),
                                           None if cl is None else _p(dcl), _p(pots), _p(ws), wsb, _stream()), "pots")
        want = (D if cl is None else np.minimum(D, cl[:, None])).sum(0)
        np.testing.assert_allclose(pots.cpu().numpy(), want, rtol=1e-5)
    _lib.check(lib().gx_kmeans_pp_commit(_p(G), rows, t, G.stride(0), _p(dxsq), _p(dcand), 3, 0, _p(dcl), _stream()),
               "commit")
    np.testing.assert_allclose(dcl.cpu().numpy(), np.minimum(closest, D[:, 3]), rtol=1e-4, atol=1e-3)


def test_label_maps_match_oracle(gpu):
    from oracle import kmeans_ref as K
    from ganecdotes_amd.kmeans import label_maps
    rng = np.random.RandomState(3)
    for out, res, ks in ((32, (8, 16, 32), (3, 4, 6)), (24, (8, 12, 48), (5, 70, 2))):
        labs = [rng.randint(0, k, (2, r, r)).astype(np.int32) for r, k in zip(res, ks)]
        got = label_maps([T(l).cuda() for l in labs], res, ks, 2, out).cpu().numpy()
        want = []
        for l, r, k in zip(labs, res, ks):
            onehot = (l[:, None] == np.arange(k).reshape(1, k, 1, 1)).astype(np.float32)
            iy = K.nearest_resize_index(r, out)
            want.append(onehot[:, :, iy][:, :, :, iy])
            ref = torch.nn.functional.interpolate(T(onehot), size=(out, out), mode="nearest").numpy()
            np.testing.assert_array_equal(want[-1], ref)        # the oracle's index rule is torch's 'nearest'
        np.testing.assert_array_equal(got, np.concatenate(want, 1).transpose(0, 2, 3, 1) * 2 - 1)


# ------------------------------------------------------------------------------------- estimator vs sklearn (GV15 A)
@pytest.mark.parametrize("tag,kw", [("explicit", {}), ("maxiter2", dict(max_iter=2)), ("far", {})])
def test_lloyd_golden_gv15(gpu, engine, tag, kw):
    """explicit init: sklearn's iterates, stopping rules, empty-cluster relocation ('far') and final E step"""
    from ganecdotes_amd.kmeans import KMeans
    g = load_golden("gv15_kmeans")
    init = g["A_init_far"] if tag == "far" else g["A_init"]
    km = KMeans(init.shape[0], init=init, n_init=1, **kw).fit(T(g["A_x"]).cuda())
    assert km.n_iter_ == int(g[f"A_{tag}_n_iter"])
    np.testing.assert_array_equal(km.labels_, g[f"A_{tag}_labels"])
    np.testing.assert_allclose(km.cluster_centers_, g[f"A_{tag}_centers"], rtol=0, atol=3e-5)
    assert abs(km.inertia_ - float(g[f"A_{tag}_inertia"])) <= 1e-5 * float(g[f"A_{tag}_inertia"])


def test_kmeans_plusplus_stream_golden_gv15(gpu, engine):
    """seeded k-means++: the RandomState stream reproduces sklearn's chosen indices and final clusterings"""
    from ganecdotes_amd.kmeans import KMeans, _Reader
    g = load_golden("gv15_kmeans")
    x = g["A_x"]
    xc = T(x - x.mean(axis=0)).cuda().contiguous()
    xsq = (xc * xc).sum(1)
    _, ids = KMeans(6)._kmeans_plusplus(xc, xsq, np.random.RandomState(5), _Reader())
    np.testing.assert_array_equal(ids, g["A_pp_indices"])
    for tag, k, n_init, seed in (("pp1", 5, 1, 3), ("pp3", 7, 3, 11)):
        km = KMeans(k, n_init=n_init, random_state=seed).fit(T(x).cuda())
        np.testing.assert_array_equal(km.labels_, g[f"A_{tag}_labels"])
        np.testing.assert_allclose(km.cluster_centers_, g[f"A_{tag}_centers"], rtol=0, atol=3e-5)
        assert km.n_iter_ == int(g[f"A_{tag}_n_iter"])
    km = KMeans(5).set_centers(g["A_explicit_centers"], gpu)
    np.testing.assert_array_equal(km.predict(T(blobs(16)).cuda()).cpu().numpy(), g["A_pred_labels"])


def test_fit_matches_oracle_larger(gpu, engine):
    """a panel of generator-layer size class (20000 x 96, 16 clusters, overlapping blobs): same seed -> the
    oracle's clustering up to boundary pixels"""
    from oracle import kmeans_ref as K
    from ganecdotes_amd.kmeans import KMeans
    x = blobs(21, n=20000, c=96, k=16, spread=0.6)
    ref = K.kmeans_fit(x, 16, n_init=1, random_state=4, max_iter=40)
    t1, t2 = T(x[:, :64].copy()).cuda(), T(x[:, 64:].copy()).cuda()
    km = KMeans(16, n_init=1, random_state=4, max_iter=40).fit_taps(t1, t2)
    np.testing.assert_array_equal(km.init_indices_, ref["init_indices"])
    assert (km.labels_ == ref["labels_"]).mean() > 0.999
    assert abs(km.inertia_ - ref["inertia_"]) <= 1e-4 * ref["inertia_"]
    np.testing.assert_allclose(km.cluster_centers_, ref["cluster_centers_"], rtol=0, atol=2e-3)
    assert abs(km.n_iter_ - ref["n_iter_"]) <= 2
    np.testing.assert_array_equal(km.predict_taps(t1, t2).cpu().numpy(), K.kmeans_predict(x, km.cluster_centers_))


# ------------------------------------------------------------------------------------- the reference's classes (GV15 B)
def _mini(g, gpu):
    from ganecdotes_amd.generator import Generator
    gen = Generator(32, 16, 2, channels={4: 16, 8: 8, 16: 8, 32: 8})
    gen.load_state_dict(golden_sd(g))
    return gen.eval().to(gpu)


class _Log:
    def info(self, msg):
        pass


def _prep(gen, g, tmp_path, **kmeans_args):
    from ganecdotes_amd.kmeans import HFCPreprocessor
    nl = len(g["B_clusters"])
    return HFCPreprocessor(
        model=gen, model_config=types.SimpleNamespace(num_latents_for_mean=64),
        perturb_args=dict(truncation=0.7, n_layers=nl, n_samples=int(g["B_n_samples"]),
                          perturb_std=g["B_perturb_std"].tolist()),
        hfc_args=dict(kmeans_args=dict(verbose=0, **kmeans_args),
                      base_args=dict(out_dir=None, n_layers=nl, clusters_per_layer=g["B_clusters"].tolist(),
                                     out_size=int(g["B_out_size"]), presaved=False)),
        hfc_algo='hfc_kmeans', hier_encode=False, train=True, out_dir=str(tmp_path), logger=_Log())


def test_hfc_preprocessor_golden_gv15(gpu, tmp_path):
    """train_hfc_model / predict_hfc_vectors against the reference's own run on its mini generator: the recorded
    normal draws are fed through the `_randn` hook, k-means uses the same RandomState stream (random_state=0)"""
    from oracle import kmeans_ref as K
    g = load_golden("gv15_kmeans")
    gen = _mini(g, gpu)
    prep = _prep(gen, g, tmp_path, random_state=0, n_init=2)
    sd, w, nl = golden_sd(g), T(g["B_w"]), len(g["B_clusters"])
    torch.manual_seed(int(g["B_seed_train"]))
    _, _, draws, _ = K.train_features(sd, w[0:1], 16, num_latents_for_mean=64, truncation=0.7, n_layers=nl,
                                      n_samples=int(g["B_n_samples"]), perturb_std=g["B_perturb_std"].tolist())
    queue = [draws["mean_z"]] + [z for row in draws["noise"] for z in row]
    prep._randn = lambda *shape: queue.pop(0).to(gpu)
    feats, lat_list = prep.train_hfc_model(w[0:1].cuda(), return_aug=True)
    assert not queue
    np.testing.assert_allclose(torch.stack(lat_list).cpu().numpy(), g["B_new_latents"], rtol=1e-4, atol=1e-5)
    for n in range(nl):
        np.testing.assert_allclose(feats[n].cpu().numpy(), g[f"B_feat_{n}"], rtol=1e-3, atol=2e-4)
        km = prep.hfc_model.clusterers[n]
        np.testing.assert_array_equal(km.labels_, g[f"B_labels_{n}"])
        # centres are means of the HIP generator's features: the feature tolerance above carries over
        np.testing.assert_allclose(km.cluster_centers_, g[f"B_centers_{n}"], rtol=1e-3, atol=1e-3)
        assert abs(km.inertia_ - float(g[f"B_inertia_{n}"])) <= 1e-3 * float(g[f"B_inertia_{n}"])
        np.testing.assert_allclose(prep.hfc_model.stds[n], g[f"B_std_{n}"], rtol=1e-4, atol=1e-5)
        assert prep.hfc_model.means[n] is prep.hfc_model.stds[n]            # the reference's aliased lists
    assert sorted(os.listdir(tmp_path)) == sorted(g["B_saved_files"].tolist())
    torch.manual_seed(int(g["B_seed_predict"]))
    _, mean_z = K.predict_features(sd, w[1:2], 16, num_latents_for_mean=64, truncation=0.7, n_layers=nl)
    prep._randn = lambda *shape: mean_z.to(gpu)
    preds, labels = prep.predict_hfc_vectors(w[1:2].cuda())
    for n in range(nl):
        np.testing.assert_array_equal(labels[n].cpu().numpy(), g[f"B_pred_labels_{n}"])
    np.testing.assert_array_equal(preds.cpu().numpy(), g["B_pred_maps"])
    assert preds.shape == g["B_pred_maps"].shape and preds.permute(0, 2, 3, 1).is_contiguous()


def test_hierarchical_kmeans_golden_gv15(gpu, tmp_path):
    """hfc_algo='hfc_kmeans_hier' (HierarchicalKMeansHFC, hfc_kmeans_clustering.py:212-390) against the reference's own
    run: the centres propagated from layer to layer (one scalar per label, rows repeated twice), the fits they seed —
    duplicated initial rows separated by the empty-cluster relocation —, the saved files and predict"""
    from oracle import kmeans_ref as K
    from ganecdotes_amd.kmeans import HFCPreprocessor, HierarchicalKMeansHFC
    g = load_golden("gv15_kmeans")
    gen = _mini(g, gpu)
    nl, clusters = len(g["C_clusters"]), g["C_clusters"].tolist()
    prep = HFCPreprocessor(
        model=gen, model_config=types.SimpleNamespace(num_latents_for_mean=64),
        perturb_args=dict(truncation=0.7, n_layers=nl, n_samples=int(g["B_n_samples"]), perturb_std=g["B_perturb_std"].tolist()),
        hfc_args=dict(kmeans_args=dict(verbose=0, random_state=0, n_init=2),
                      base_args=dict(out_dir=None, n_layers=nl, clusters_per_layer=clusters, out_size=int(g["B_out_size"]),
                                     presaved=False)),
        hfc_algo='hfc_kmeans_hier', hier_encode=False, train=True, out_dir=str(tmp_path), logger=_Log())
    assert isinstance(prep.hfc_model, HierarchicalKMeansHFC)
    sd, w = golden_sd(g), T(g["B_w"])
    torch.manual_seed(int(g["C_seed_train"]))
    _, _, draws, _ = K.train_features(sd, w[0:1], 16, num_latents_for_mean=64, truncation=0.7, n_layers=nl,
                                      n_samples=int(g["B_n_samples"]), perturb_std=g["B_perturb_std"].tolist())
    queue = [draws["mean_z"]] + [z for row in draws["noise"] for z in row]
    prep._randn = lambda *shape: queue.pop(0).to(gpu)
    feats, _ = prep.train_hfc_model(w[0:1].cuda(), return_aug=True)
    for n in range(nl):
        np.testing.assert_allclose(feats[n].cpu().numpy(), g[f"C_feat_{n}"], rtol=1e-3, atol=2e-4)
        km = prep.hfc_model.clusterers[n]
        if n:
            np.testing.assert_allclose(np.asarray(km.init), g[f"C_init_{n}"], rtol=1e-4, atol=1e-5)
        np.testing.assert_array_equal(km.labels_, g[f"C_labels_{n}"])
        np.testing.assert_allclose(km.cluster_centers_, g[f"C_centers_{n}"], rtol=1e-3, atol=1e-3)
        assert abs(km.inertia_ - float(g[f"C_inertia_{n}"])) <= 1e-3 * float(g[f"C_inertia_{n}"])
        assert km.n_iter_ == int(g[f"C_n_iter_{n}"])
    assert sorted(os.listdir(tmp_path)) == sorted(g["C_saved_files"].tolist())
    torch.manual_seed(int(g["C_seed_predict"]))
    _, mean_z = K.predict_features(sd, w[1:2], 16, num_latents_for_mean=64, truncation=0.7, n_layers=nl)
    prep._randn = lambda *shape: mean_z.to(gpu)
    preds, labels = prep.predict_hfc_vectors(w[1:2].cuda())
    for n in range(nl):
        np.testing.assert_array_equal(labels[n].cpu().numpy(), g[f"C_pred_labels_{n}"])
    np.testing.assert_array_equal(preds.cpu().numpy(), g["C_pred_maps"])
    # the flag the reference cannot run itself (GV15 D_hier_encode_outcome) is refused, with the reason
    with pytest.raises(NotImplementedError, match="raises in the reference itself"):
        HFCPreprocessor(model=gen, model_config=None, perturb_args={}, hfc_args={}, hier_encode=True, out_dir=str(tmp_path))


def test_saved_models_are_sklearn_pickles(gpu, tmp_path):
    """clusterer_layer_{n}.sav unpickles to a fitted sklearn KMeans whose CPU predict agrees with the HIP labels;
    a fresh (presaved) FlatKMeansHFC loads them back"""
    sklearn_cluster = pytest.importorskip("sklearn.cluster")
    from ganecdotes_amd.kmeans import FlatKMeansHFC
    rng = np.random.RandomState(8)
    feats = [(torch.randn(2, 24, r, r, device=gpu) + 2 * torch.randn(1, 24, 1, 1, device=gpu) * (torch.arange(r, device=gpu) > r // 2),
              torch.randn(2, 8, r, r, device=gpu)) for r in (8, 16)]
    base = dict(out_dir=str(tmp_path), n_layers=2, clusters_per_layer=[3, 5], out_size=16, presaved=False, device=gpu)
    hfc = FlatKMeansHFC(dict(random_state=1, n_init=2, verbose=0), dict(base))
    hfc.fit(feats)
    maps, labels = hfc.predict(feats)
    assert maps.shape == (2, 8, 16, 16) and set(np.unique(maps.cpu().numpy())) == {0.0, 1.0}
    assert torch.all(maps.sum(1) == 2)                                       # one cluster per layer and pixel
    for n in range(2):
        with open(os.path.join(tmp_path, f"clusterer_layer_{n}.sav"), "rb") as f:
            sk = pickle.load(f)
        assert isinstance(sk, sklearn_cluster.KMeans)
        panel = torch.cat(feats[n], 1).permute(0, 2, 3, 1).reshape(-1, 32).cpu().numpy()
        assert (sk.predict(panel) == labels[n].reshape(-1).cpu().numpy()).mean() > 0.999
    again = FlatKMeansHFC(dict(verbose=0), dict(base, presaved=True))
    maps2, _ = again.predict(feats)
    assert torch.equal(maps, maps2)
    stats = np.load(os.path.join(tmp_path, "model_stats.npz"), allow_pickle=True)
    assert sorted(stats.files) == ["means", "stds"] and stats["means"][0].shape == (32,)


# ------------------------------------------------------------------------------------- full-size properties
def test_fullsize_layer_properties(gpu):
    """the largest panel of the ffhq-256 config (4 x 128 x 128 pixels, 256 + 256 channels, 64 clusters): a strict
    fixed point has every centre equal to the mean of its members, every pixel at its nearest centre, and the
    inertia of the fit does not exceed that of its own initialisation"""
    from ganecdotes_amd.kmeans import KMeans
    torch.manual_seed(5)
    rows, c, k = 4 * 128 * 128, 256, 64
    proto = torch.randn(k, 2 * c, device=gpu)
    member = torch.randint(0, k, (rows,), device=gpu)
    x = proto[member] * 0.5 + torch.randn(rows, 2 * c, device=gpu)
    t1, t2 = x[:, :c].contiguous(), x[:, c:].contiguous()
    km = KMeans(k, n_init=1, random_state=0, max_iter=300, tol=0.0).fit_taps(t1, t2)
    assert km.n_iter_ < 300
    lab = torch.from_numpy(km.labels_).long().to(gpu)
    cen = km.centers_device_.double()
    counts = torch.bincount(lab, minlength=k)
    assert int(counts.min()) > 0
    means = torch.zeros(k, 2 * c, dtype=torch.float64, device=gpu).index_add_(0, lab, x.double()) / counts[:, None]
    assert float((means - cen).abs().max()) < 1e-4
    d = torch.cdist(x.double()[::16], cen)
    best = d.min(1).values
    mine = d.gather(1, lab[::16, None])[:, 0]
    assert float(((mine - best) / best).max()) < 1e-5
    inertia = float(((x.double() - cen[lab]) ** 2).sum())
    assert abs(inertia - km.inertia_) <= 1e-5 * inertia
    init_inertia = float((torch.cdist(x.double(), x.double()[torch.from_numpy(km.init_indices_).to(gpu)]).min(1).values ** 2).sum())
    assert km.inertia_ <= init_inertia
    pred = km.predict_taps(t1, t2)
    assert (pred.cpu().numpy() == km.labels_).mean() > 0.9999


def test_pipeline_hfc_kmeans_synthetic(gpu, tmp_path):
    """pretrain.py --method hfc_kmeans end to end on the ffhq-256 shapes: k-means per layer, label maps, the
    'S' head fine-tuned on them, the IoU loop; then evaluate.py's order (presaved models, no fit)"""
    from configs import mapper
    from src.one_shot_pipeline import OneShotPipeline
    pipe = OneShotPipeline(out_dir=str(tmp_path), model='ffhq-256', segmentor='hfc_kmeans', num_test_samples=2,
                           device='cuda:0', synthetic=True)
    pipe.preprocessor.hfc_model.kmeans_args.update(n_init=1, random_state=0, max_iter=30)
    pipe.trainer_config.num_epochs = 20
    pipe.run_pipeline()
    assert pipe.one_shot_features.shape == (1, 124, 256, 256)
    assert pipe.head_losses[-1] < pipe.head_losses[0]
    assert 0.0 <= pipe.results['mIoU'] <= 1.0
    assert os.path.exists(os.path.join(tmp_path, "clusterer_layer_4.sav"))
    first = pipe.one_shot_features.clone()
    pipe2 = OneShotPipeline(out_dir=str(tmp_path), model='ffhq-256', segmentor='hfc_kmeans', num_test_samples=2,
                            device='cuda:0', synthetic=True)
    pipe2.seg_config.train_hfc = False
    pipe2.trainer_config.num_epochs = 2
    pipe2.run_pipeline()
    # same generator (seed 42), same latent; mean_latent is redrawn per call like the reference's, so a few
    # boundary pixels may move
    assert (pipe2.one_shot_features == first).float().mean() > 0.98
